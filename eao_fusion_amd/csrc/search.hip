// search.hip -- the remaining guided searches of the reference matcher for MI355X (gfx950), host side.
//
// Stands behind (include/eao_fusion.h cites each entry point):
//   SearchByProjection(KeyFrame*, Scw, ...)            src/ORBmatcher.cc:290-403     eao_search_by_projection_sim3
//   SearchByProjection(Frame&, KeyFrame*, set, ...)     src/ORBmatcher.cc:1474-1601   eao_search_by_projection_kf
//   SearchByBoW (KF-Frame, KF-KF)                       src/ORBmatcher.cc:159-288, 522-655   eao_search_by_bow
//   SearchForTriangulation (+ CheckDistEpipolarLine)    src/ORBmatcher.cc:657-823, 140-157   eao_search_for_triangulation
//   SearchForInitialization                             src/ORBmatcher.cc:405-520     eao_search_for_initialization
//   Fuse (both overloads, search half)                  src/ORBmatcher.cc:825-975, 977-1100  eao_fuse_search
//   SearchBySim3                                        src/ORBmatcher.cc:1102-1326   eao_search_by_sim3
// Division of labour, as for the two per-frame searches in match.hip: the data-parallel part -- every candidate of every
// query inside its grid window (or vocabulary bucket) with its 256-bit Hamming distance, in upstream's visiting order --
// runs on the GPU (k_match_candidates / k_pair_distances); the per-query geometry (a dozen float operations that must
// round exactly like the cv::Mat expressions upstream) and the order-dependent selection loops are replayed on the host
// over those lists.  cv::Mat float semantics used throughout: A*x (+b) accumulates in double and rounds once; cv::norm
// and Mat::dot accumulate in double; scalar * matrix is an element-wise float product; PredictScale is float logf/ceilf.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "match_internal.h"
#include "chain_internal.h"
#include "search_internal.h"

using eao::match::Lists;
using eao::match::Query;

namespace {

constexpr int TH_HIGH = refc::TH_HIGH, TH_LOW = refc::TH_LOW, HISTO_LENGTH = refc::HISTO_LENGTH;

void affine3(const float* A, const float* x, const float* b, float alpha, float* y) {
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)A[r * 3 + k] * (double)x[k];
        y[r] = (float)((double)alpha * s + (b ? (double)b[r] : 0.0));
    }
}
float norm3(const float* v) { return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
double dot3(const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }
// MapPoint::PredictScale (src/MapPoint.cc:385-394): this fork does not clamp the level
int predict_scale(float maxDistance, float currentDist, float logScaleFactor) {
    const float ratio = maxDistance / currentDist;
    return (int)std::ceil(std::log(ratio) / logScaleFactor);
}
bool finite_n(const float* v, int n) {      // a pose / matrix argument: NaN or Inf would turn the window arithmetic (floor / ceil to int) into undefined behaviour
    for (int i = 0; i < n; i++) if (!std::isfinite(v[i])) return false;
    return true;
}
bool in_image(const eao_frame_view* K, float x, float y) {   // KeyFrame::IsInImage, src/KeyFrame.cc:649-652
    return x >= K->min_x && x < K->max_x && y >= K->min_y && y < K->max_y;
}
void split_pose(const float* T, float* R, float* t) {
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c]; t[r] = T[r * 4 + 3]; }
}
void camera_centre(const float* R, const float* t, float* O) {   // -R^T t
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)R[k * 3 + i] * (double)t[k];
        O[i] = (float)(-s);
    }
}
void decompose_sim3(const float* S, float* Rcw, float* tcw, float* Ow) {   // src/ORBmatcher.cc:298-303
    const float scw = (float)std::sqrt((double)S[0] * S[0] + (double)S[1] * S[1] + (double)S[2] * S[2]);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = S[r * 4 + c] / scw;
        tcw[r] = S[r * 4 + 3] / scw;
    }
    camera_centre(Rcw, tcw, Ow);
}

struct RotHist {    // rotation-consistency histogram + ComputeThreeMaxima (src/ORBmatcher.cc:1603-1644)
    std::vector<int> bin[HISTO_LENGTH];
    float factor;
    explicit RotHist(float f) : factor(f) {}
    void add(float a1, float a2, int payload) {
        float rot = a1 - a2;
        if (rot < 0.0) rot += 360.0f;
        int b = (int)std::round(rot * factor);
        if (b == HISTO_LENGTH) b = 0;
        bin[b].push_back(payload);
    }
    template <typename Fn>
    void reject_minor(Fn&& drop) const {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = (int)bin[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int v : bin[i]) drop(v);
    }
};

Query window(float u, float v, float r, int minLevel, int maxLevel) {
    Query q;
    q.x = u; q.y = v; q.r = r; q.minLevel = minLevel; q.maxLevel = maxLevel;
    q.urRef = 0.f; q.urTol = INFINITY; q.active = 1;
    return q;
}
Query inactive() { Query q = window(0, 0, 0, -1, -1); q.active = 0; return q; }

bool points_ok(const eao_map_points* P, bool needNormal) {
    return P && P->n >= 0 && (P->n == 0 || (P->active && P->Xw && P->min_dist_inv && P->max_dist_inv && P->max_dist && P->desc &&
                                            (!needNormal || P->normal)));
}
bool view_ok(const eao_frame_view* F) {
    return F && F->n >= 0 && (F->n == 0 || (F->kp_x && F->kp_y && F->kp_octave && F->kp_angle && F->u_right && F->descriptors)) &&
           F->scale_factors && F->nlevels > 0 && F->grid_cols > 0 && F->grid_rows > 0;
}

// projection of one map point with the range / viewing-angle tests shared by SearchByProjection(KF, Scw), both Fuse
// overloads: returns false where upstream `continue`s
struct Shot { float u, v, invz, dist; int level; };
bool shoot(const eao_frame_view* K, const float* Rcw, const float* tcw, const float* Ow, float fx, float fy, float cx, float cy,
           const eao_map_points* P, int i, bool invzInDouble, Shot& s) {
    const float* Xw = P->Xw + 3 * i;
    float pc[3];
    affine3(Rcw, Xw, tcw, 1.f, pc);
    if (pc[2] < 0.0f) return false;
    s.invz = invzInDouble ? (float)(1.0 / pc[2]) : 1 / pc[2];
    const float x = pc[0] * s.invz, y = pc[1] * s.invz;
    s.u = fx * x + cx; s.v = fy * y + cy;
    if (!in_image(K, s.u, s.v)) return false;
    const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
    s.dist = norm3(PO);
    if (s.dist < P->min_dist_inv[i] || s.dist > P->max_dist_inv[i]) return false;
    if (dot3(PO, P->normal + 3 * i) < 0.5 * s.dist) return false;
    s.level = predict_scale(P->max_dist[i], s.dist, K->log_scale_factor);
    return s.level >= 0 && s.level < K->nlevels;   // upstream would index mvScaleFactors out of range otherwise
}

// merge walk over two feature vectors (std::map iteration + lower_bound, src/ORBmatcher.cc:175-262)
template <typename Fn>
void for_common_nodes(const eao_feature_vector* f1, const eao_feature_vector* f2, Fn&& fn) {
    int a = 0, b = 0;
    while (a < f1->n_nodes && b < f2->n_nodes) {
        if (f1->node_id[a] == f2->node_id[b]) { fn(a, b); a++; b++; }
        else if (f1->node_id[a] < f2->node_id[b]) { while (a < f1->n_nodes && f1->node_id[a] < f2->node_id[b]) a++; }
        else { while (b < f2->n_nodes && f2->node_id[b] < f1->node_id[a]) b++; }
    }
}
bool fv_ok(const eao_feature_vector* f, int n) {
    if (!f || f->n_nodes < 0) return false;
    if (f->n_nodes == 0) return true;
    if (!f->node_id || !f->node_start || !f->index) return false;
    for (int k = 0; k < f->n_nodes; k++) {
        if (k && f->node_id[k] <= f->node_id[k - 1]) return false;
        if (f->node_start[k + 1] < f->node_start[k]) return false;
    }
    for (int p = f->node_start[0]; p < f->node_start[f->n_nodes]; p++)
        if ((int)f->index[p] >= n) return false;
    return f->node_start[0] == 0;
}

}  // namespace

bool eao::search::feature_vector_ok(const eao_feature_vector* f, int n) { return fv_ok(f, n); }

eao_status eao::search::projection_sim3(const eao_frame_view* KF, const match::Resident* res, const float* Scw, float fx, float fy, float cx, float cy,
                                        const eao_map_points* pts, int32_t th, int32_t* kp_match, int32_t* nmatches) {
    EAO_REQUIRE(view_ok(KF) && Scw && points_ok(pts, true) && kp_match && nmatches, "bad argument");
    EAO_REQUIRE(finite_n(Scw, 16), "Scw holds a NaN / Inf");
    float Rcw[9], tcw[3], Ow[3];
    decompose_sim3(Scw, Rcw, tcw, Ow);
    const int n = pts->n;
    std::vector<Query> q(n, inactive());
    std::vector<int> level(n, 0);
    for (int i = 0; i < n; i++) {
        if (!pts->active[i]) continue;
        Shot s;
        if (!shoot(KF, Rcw, tcw, Ow, fx, fy, cx, cy, pts, i, false, s)) continue;
        level[i] = s.level;
        q[i] = window(s.u, s.v, th * KF->scale_factors[s.level], -1, -1);
    }
    Lists L;
    eao_status st = eao::match::build_lists(KF, q, pts->desc, L, res);
    if (st) return st;
    std::vector<uint8_t> occ(KF->n, 0);
    if (KF->occupied) std::memcpy(occ.data(), KF->occupied, KF->n);
    for (int k = 0; k < KF->n; k++) kp_match[k] = -1;
    int nm = 0;
    for (int i = 0; i < n; i++) {
        if (!q[i].active || L.count[i] == 0) continue;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < L.count[i]; c++) {
            const unsigned it = L.items[L.start[i] + c];
            const int k = (int)(it & 0xFFFF), d = (int)(it >> 16);
            if (occ[k]) continue;
            const int kl = KF->kp_octave[k];
            if (kl < level[i] - 1 || kl > level[i]) continue;
            if (d < bestDist) { bestDist = d; bestIdx = k; }
        }
        if (bestDist <= TH_LOW) { kp_match[bestIdx] = i; occ[bestIdx] = 1; nm++; }
    }
    *nmatches = nm;
    return EAO_OK;
}

eao_status eao::search::projection_kf(const eao_frame_view* Cur, const match::Resident* res, const float* Tcw, float fx, float fy, float cx, float cy,
                                      const eao_map_points* pts, const float* kf_angle, float th, int32_t orb_dist,
                                      int32_t check_orientation, int32_t* cur_match, int32_t* nmatches) {
    EAO_REQUIRE(view_ok(Cur) && Tcw && points_ok(pts, false) && cur_match && nmatches && (pts->n == 0 || kf_angle), "bad argument");
    EAO_REQUIRE(finite_n(Tcw, 16), "Tcw holds a NaN / Inf");
    float Rcw[9], tcw[3], Ow[3];
    split_pose(Tcw, Rcw, tcw);
    camera_centre(Rcw, tcw, Ow);
    const int n = pts->n;
    std::vector<Query> q(n, inactive());
    for (int i = 0; i < n; i++) {
        if (!pts->active[i]) continue;
        const float* Xw = pts->Xw + 3 * i;
        float xc[3];
        affine3(Rcw, Xw, tcw, 1.f, xc);
        const float invzc = (float)(1.0 / xc[2]);
        const float u = fx * xc[0] * invzc + cx, v = fy * xc[1] * invzc + cy;
        if (u < Cur->min_x || u > Cur->max_x) continue;
        if (v < Cur->min_y || v > Cur->max_y) continue;
        const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < pts->min_dist_inv[i] || dist3D > pts->max_dist_inv[i]) continue;
        const int lvl = predict_scale(pts->max_dist[i], dist3D, Cur->log_scale_factor);
        if (lvl < 0 || lvl >= Cur->nlevels) continue;
        q[i] = window(u, v, th * Cur->scale_factors[lvl], lvl - 1, lvl + 1);
    }
    Lists L;
    eao_status st = eao::match::build_lists(Cur, q, pts->desc, L, res);
    if (st) return st;
    std::vector<uint8_t> occ(Cur->n, 0);
    if (Cur->occupied) std::memcpy(occ.data(), Cur->occupied, Cur->n);
    for (int k = 0; k < Cur->n; k++) cur_match[k] = -1;
    RotHist hist(1.0f / HISTO_LENGTH);
    int nm = 0;
    for (int i = 0; i < n; i++) {
        if (!q[i].active || L.count[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < L.count[i]; c++) {
            const unsigned it = L.items[L.start[i] + c];
            const int k = (int)(it & 0xFFFF), d = (int)(it >> 16);
            if (occ[k]) continue;
            if (d < bestDist) { bestDist = d; bestIdx2 = k; }
        }
        if (bestDist <= orb_dist) {
            cur_match[bestIdx2] = i; occ[bestIdx2] = 1; nm++;
            if (check_orientation) hist.add(kf_angle[i], Cur->kp_angle[bestIdx2], bestIdx2);
        }
    }
    if (check_orientation) hist.reject_minor([&](int k) { cur_match[k] = -1; nm--; });
    *nmatches = nm;
    return EAO_OK;
}

extern "C" {

eao_status eao_search_by_projection_sim3(const eao_frame_view* KF, const float* Scw, float fx, float fy, float cx, float cy,
                                         const eao_map_points* pts, int32_t th, int32_t* kp_match, int32_t* nmatches) {
    return eao::search::projection_sim3(KF, nullptr, Scw, fx, fy, cx, cy, pts, th, kp_match, nmatches);
}
eao_status eao_search_by_projection_kf(const eao_frame_view* Cur, const float* Tcw, float fx, float fy, float cx, float cy,
                                       const eao_map_points* pts, const float* kf_angle, float th, int32_t orb_dist,
                                       int32_t check_orientation, int32_t* cur_match, int32_t* nmatches) {
    return eao::search::projection_kf(Cur, nullptr, Tcw, fx, fy, cx, cy, pts, kf_angle, th, orb_dist, check_orientation, cur_match, nmatches);
}

eao_status eao_search_by_bow(int32_t mode, int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                             const eao_feature_vector* fv1, int32_t n2, const uint8_t* desc2, const float* angle2,
                             const uint8_t* valid2, const eao_feature_vector* fv2, float nnratio, int32_t check_orientation,
                             int32_t* match12, int32_t* nmatches) {
    EAO_REQUIRE((mode == 0 || mode == 1) && n1 >= 0 && n2 >= 0 && match12 && nmatches, "bad argument");
    EAO_REQUIRE(n1 == 0 || (desc1 && angle1 && valid1), "side 1 arrays missing");
    EAO_REQUIRE(n2 == 0 || (desc2 && angle2 && (mode == 0 || valid2)), "side 2 arrays missing");
    EAO_REQUIRE(fv_ok(fv1, n1) && fv_ok(fv2, n2), "malformed feature vector");
    // every (query, candidate) pair of the common nodes, in visiting order
    std::vector<int> ia, ib;
    for_common_nodes(fv1, fv2, [&](int a, int b) {
        for (int p = fv1->node_start[a]; p < fv1->node_start[a + 1]; p++) {
            const int idx1 = (int)fv1->index[p];
            if (!valid1[idx1]) continue;
            for (int qx = fv2->node_start[b]; qx < fv2->node_start[b + 1]; qx++) {
                const int idx2 = (int)fv2->index[qx];
                if (mode == 1 && !valid2[idx2]) continue;
                ia.push_back(idx1); ib.push_back(idx2);
            }
        }
    });
    std::vector<unsigned short> dist;
    eao_status st = eao::match::pair_distances(desc1, n1, desc2, n2, ia, ib, dist);
    if (st) return st;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> taken2(n2, 0);
    RotHist hist(1.0f / HISTO_LENGTH);
    int nm = 0;
    size_t cur = 0;
    for_common_nodes(fv1, fv2, [&](int a, int b) {
        for (int p = fv1->node_start[a]; p < fv1->node_start[a + 1]; p++) {
            const int idx1 = (int)fv1->index[p];
            if (!valid1[idx1]) continue;
            int best1 = 256, bestIdx2 = -1, best2 = 256;
            for (int qx = fv2->node_start[b]; qx < fv2->node_start[b + 1]; qx++) {
                const int idx2 = (int)fv2->index[qx];
                if (mode == 1 && !valid2[idx2]) continue;
                const int d = dist[cur++];
                if (taken2[idx2]) continue;
                if (d < best1) { best2 = best1; best1 = d; bestIdx2 = idx2; }
                else if (d < best2) { best2 = d; }
            }
            const bool close = mode == 0 ? best1 <= TH_LOW : best1 < TH_LOW;
            if (close && (float)best1 < nnratio * (float)best2) {
                match12[idx1] = bestIdx2;
                taken2[bestIdx2] = 1;
                if (check_orientation) hist.add(angle1[idx1], angle2[bestIdx2], idx1);
                nm++;
            }
        }
    });
    if (check_orientation) hist.reject_minor([&](int k) { match12[k] = -1; nm--; });
    *nmatches = nm;
    return EAO_OK;
}

}  // extern "C"

namespace {
// SearchForTriangulation (src/ORBmatcher.cc:657-823) in two halves: every (keypoint of K1, keypoint of K2) pair of the common vocabulary nodes in visiting
// order (ib shifted by `base2`: where K2's descriptors start in a concatenated array) ...
void tri_pairs(const eao_frame_view* K1, const eao_feature_vector* fv1, const eao_frame_view* K2, const eao_feature_vector* fv2, int only_stereo, int base2,
               std::vector<int>& ia, std::vector<int>& ib) {
    auto skip1 = [&](int i) { return (K1->occupied && K1->occupied[i]) || (only_stereo && !(K1->u_right[i] >= 0)); };
    auto skip2 = [&](int j) { return (K2->occupied && K2->occupied[j]) || (only_stereo && !(K2->u_right[j] >= 0)); };
    for_common_nodes(fv1, fv2, [&](int a, int b) {
        for (int p = fv1->node_start[a]; p < fv1->node_start[a + 1]; p++) {
            const int idx1 = (int)fv1->index[p];
            if (skip1(idx1)) continue;
            for (int qx = fv2->node_start[b]; qx < fv2->node_start[b + 1]; qx++) {
                const int idx2 = (int)fv2->index[qx];
                if (skip2(idx2)) continue;
                ia.push_back(idx1); ib.push_back(base2 + idx2);
            }
        }
    });
}
// ... and the selection over their distances (`dist` points at this neighbour's first pair; returns how many it consumed)
size_t tri_replay(const eao_frame_view* K1, const eao_feature_vector* fv1, const eao_frame_view* K2, const eao_feature_vector* fv2, const float* F12, float ex,
                  float ey, int only_stereo, int check_orientation, const unsigned short* dist, int32_t* match12, int32_t* nmatches) {
    auto skip1 = [&](int i) { return (K1->occupied && K1->occupied[i]) || (only_stereo && !(K1->u_right[i] >= 0)); };
    auto skip2 = [&](int j) { return (K2->occupied && K2->occupied[j]) || (only_stereo && !(K2->u_right[j] >= 0)); };
    for (int i = 0; i < K1->n; i++) match12[i] = -1;
    RotHist hist(1.0f / HISTO_LENGTH);
    int nm = 0;
    size_t cur = 0;
    for_common_nodes(fv1, fv2, [&](int a, int b) {
        for (int p = fv1->node_start[a]; p < fv1->node_start[a + 1]; p++) {
            const int idx1 = (int)fv1->index[p];
            if (skip1(idx1)) continue;
            const bool stereo1 = K1->u_right[idx1] >= 0;
            const float x1 = K1->kp_x[idx1], y1 = K1->kp_y[idx1];
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int qx = fv2->node_start[b]; qx < fv2->node_start[b + 1]; qx++) {
                const int idx2 = (int)fv2->index[qx];
                if (skip2(idx2)) continue;          // upstream never sets vbMatched2, so side 2 is never consumed
                const int d = dist[cur++];
                if (d > TH_LOW || d > bestDist) continue;
                const bool stereo2 = K2->u_right[idx2] >= 0;
                const float x2 = K2->kp_x[idx2], y2 = K2->kp_y[idx2];
                const int oct2 = K2->kp_octave[idx2];
                if (!stereo1 && !stereo2) {
                    const float dex = ex - x2, dey = ey - y2;
                    if (dex * dex + dey * dey < 100 * K2->scale_factors[oct2]) continue;
                }
                // CheckDistEpipolarLine (src/ORBmatcher.cc:140-157)
                const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
                const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
                const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
                const float num = la * x2 + lb * y2 + lc;
                const float den = la * la + lb * lb;
                if (den == 0) continue;
                const float dsqr = num * num / den;
                if (dsqr < refc::EPIPOLAR_CHI2 * K2->level_sigma2[oct2]) { bestIdx2 = idx2; bestDist = d; }
            }
            if (bestIdx2 >= 0) {
                match12[idx1] = bestIdx2;
                nm++;
                if (check_orientation) hist.add(K1->kp_angle[idx1], K2->kp_angle[bestIdx2], idx1);
            }
        }
    });
    if (check_orientation) hist.reject_minor([&](int k) { match12[k] = -1; nm--; });
    *nmatches = nm;
    return cur;
}
bool octaves_ok(const eao_frame_view* K) {      // the replay indexes scale_factors / level_sigma2 with them
    for (int i = 0; i < K->n; i++) if (K->kp_octave[i] < 0 || K->kp_octave[i] >= K->nlevels) return false;
    return true;
}
}  // namespace

extern "C" {

eao_status eao_search_for_triangulation(const eao_frame_view* K1, const eao_feature_vector* fv1, const eao_frame_view* K2,
                                        const eao_feature_vector* fv2, const float* F12, float ex, float ey,
                                        int32_t only_stereo, int32_t check_orientation, int32_t* match12, int32_t* nmatches) {
    return eao_search_for_triangulation_batch(K1, fv1, 1, &K2, &fv2, F12, &ex, &ey, only_stereo, check_orientation, match12, nmatches);
}

/* LocalMapping::CreateNewMapPoints calls SearchForTriangulation once per neighbour keyframe (src/LocalMapping.cc:211-290: 10 or 20 of them per new
 * keyframe): here ALL neighbours in one call -- the current keyframe's descriptors and every neighbour's travel to the device once, one launch computes
 * the distances of every (neighbour, vocabulary-node) pair, one copy brings them back; the selection is replayed per neighbour exactly as in a single call. */
eao_status eao_search_for_triangulation_batch(const eao_frame_view* K1, const eao_feature_vector* fv1, int32_t n_nb, const eao_frame_view* const* K2s,
                                              const eao_feature_vector* const* fv2s, const float* F12s, const float* exs, const float* eys,
                                              int32_t only_stereo, int32_t check_orientation, int32_t* match12, int32_t* nmatches) {
    EAO_REQUIRE(view_ok(K1) && n_nb >= 0 && (n_nb == 0 || (K2s && fv2s && F12s && exs && eys && match12 && nmatches)), "bad argument");
    EAO_REQUIRE(fv_ok(fv1, K1->n), "malformed feature vector");
    if (n_nb == 0) return EAO_OK;
    std::vector<int> ia, ib;
    std::vector<size_t> first(n_nb + 1, 0);
    std::vector<uint8_t> descB;
    int base2 = 0;
    for (int k = 0; k < n_nb; k++) {
        const eao_frame_view* K2 = K2s[k];
        EAO_REQUIRE(view_ok(K2) && K2->level_sigma2 && fv_ok(fv2s[k], K2->n), "neighbour %d: bad frame view or feature vector", k);
        EAO_REQUIRE(octaves_ok(K2), "neighbour %d: a keypoint's octave lies outside its %d levels", k, K2->nlevels);
        for (int q = 0; q < 9; q++) EAO_REQUIRE(std::isfinite(F12s[9 * k + q]), "neighbour %d: F12 holds a NaN / Inf", k);
        first[k] = ia.size();
        tri_pairs(K1, fv1, K2, fv2s[k], only_stereo, base2, ia, ib);
        descB.insert(descB.end(), K2->descriptors, K2->descriptors + 32 * (size_t)K2->n);
        base2 += K2->n;
    }
    first[n_nb] = ia.size();
    std::vector<unsigned short> dist;
    eao_status st = eao::match::pair_distances(K1->descriptors, K1->n, descB.data(), base2, ia, ib, dist);
    if (st) return st;
    for (int k = 0; k < n_nb; k++) {
        const size_t used = tri_replay(K1, fv1, K2s[k], fv2s[k], F12s + 9 * k, exs[k], eys[k], only_stereo, check_orientation, dist.data() + first[k],
                                       match12 + (size_t)k * K1->n, nmatches + k);
        if (used != first[k + 1] - first[k]) { eao::set_error("internal: neighbour %d consumed %zu of %zu pair distances", k, used, first[k + 1] - first[k]); return EAO_ERR_INTERNAL; }
    }
    return EAO_OK;
}

eao_status eao_search_for_initialization(int32_t n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1,
                                         const eao_frame_view* F2, float* prev_matched, int32_t window_size, float nnratio,
                                         int32_t check_orientation, int32_t* match12, int32_t* nmatches) {
    return eao::search::initialization(n1, octave1, angle1, desc1, F2, nullptr, prev_matched, window_size, nnratio, check_orientation, match12, nmatches);
}
}  // extern "C"

eao_status eao::search::initialization(int32_t n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1, const eao_frame_view* F2,
                                       const match::Resident* res, float* prev_matched, int32_t window_size, float nnratio, int32_t check_orientation,
                                       int32_t* match12, int32_t* nmatches) {
    EAO_REQUIRE(n1 >= 0 && view_ok(F2) && match12 && nmatches && (n1 == 0 || (octave1 && angle1 && desc1 && prev_matched)), "bad argument");
    std::vector<Query> q(n1, inactive());
    for (int i = 0; i < n1; i++) {
        if (octave1[i] > 0) continue;
        q[i] = window(prev_matched[2 * i], prev_matched[2 * i + 1], (float)window_size, octave1[i], octave1[i]);
    }
    Lists L;
    eao_status st = eao::match::build_lists(F2, q, desc1, L, res);
    if (st) return st;
    int nm = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    RotHist hist(1.0f / HISTO_LENGTH);
    std::vector<int> matchedDistance(F2->n, INT_MAX), matches21(F2->n, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        if (!q[i1].active || L.count[i1] == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < L.count[i1]; c++) {
            const unsigned it = L.items[L.start[i1] + c];
            const int i2 = (int)(it & 0xFFFF), d = (int)(it >> 16);
            if (matchedDistance[i2] <= d) continue;
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) { bestDist2 = d; }
        }
        if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { match12[matches21[bestIdx2]] = -1; nm--; }
            match12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) hist.add(angle1[i1], F2->kp_angle[bestIdx2], i1);
        }
    }
    if (check_orientation) hist.reject_minor([&](int k) { if (match12[k] >= 0) { match12[k] = -1; nm--; } });
    for (int i1 = 0; i1 < n1; i1++)
        if (match12[i1] >= 0) { prev_matched[2 * i1] = F2->kp_x[match12[i1]]; prev_matched[2 * i1 + 1] = F2->kp_y[match12[i1]]; }
    *nmatches = nm;
    return EAO_OK;
}

extern "C" {

eao_status eao_fuse_search(const eao_frame_view* KF, int32_t use_sim3, const float* pose, float fx, float fy, float cx, float cy,
                           float bf, const eao_map_points* pts, float th, int32_t* best_kp, int32_t* nfused) {
    return eao_fuse_search_batch(1, &KF, use_sim3, pose, fx, fy, cx, cy, bf, pts, th, best_kp, nfused);
}

/* LocalMapping::SearchInNeighbors fuses the current keyframe's map points into every target keyframe (src/LocalMapping.cc:458-520: Fuse(pKFi, vpMapPointMatches)
 * per target): here the search half for ALL targets in one call -- the points' descriptors are staged once, every target's frame and windows go up in the same
 * copy, one launch per target back to back, one synchronisation.  poses: 15 floats (use_sim3 = 0) or 16 (use_sim3 = 1) per target.  best_kp: n_kf x pts->n.
 * As for the single call, replacing / adding observations is the caller's part, target by target in order; because an earlier target's fusions change the
 * map (a point replaced, or merged into one that the next target already observes), the caller re-checks isBad() AND IsInKeyFrame(pKFi) before it uses a
 * later target's candidate -- upstream's own two `continue`s at the head of Fuse's loop (src/ORBmatcher.cc:851-861). */
eao_status eao_fuse_search_batch(int32_t n_kf, const eao_frame_view* const* KFs, int32_t use_sim3, const float* poses, float fx, float fy, float cx, float cy,
                                 float bf, const eao_map_points* pts, float th, int32_t* best_kp, int32_t* nfused) {
    EAO_REQUIRE(n_kf >= 0 && (n_kf == 0 || (KFs && poses && best_kp && nfused)) && points_ok(pts, true), "bad argument");
    if (n_kf == 0) return EAO_OK;
    const int n = pts->n, plen = use_sim3 ? 16 : 15;
    std::vector<std::vector<Query>> q(n_kf, std::vector<Query>(n, inactive()));
    std::vector<std::vector<Shot>> shot(n_kf, std::vector<Shot>(n));
    std::vector<const uint8_t*> qd(n_kf, pts->desc);
    for (int f = 0; f < n_kf; f++) {
        const eao_frame_view* KF = KFs[f];
        EAO_REQUIRE(view_ok(KF) && (use_sim3 || KF->inv_level_sigma2), "target %d: bad frame view", f);
        EAO_REQUIRE(octaves_ok(KF), "target %d: a keypoint's octave lies outside its %d levels", f, KF->nlevels);
        const float* pose = poses + (size_t)plen * f;
        for (int k = 0; k < plen; k++) EAO_REQUIRE(std::isfinite(pose[k]), "target %d: the pose holds a NaN / Inf", f);
        float Rcw[9], tcw[3], Ow[3];
        if (use_sim3) decompose_sim3(pose, Rcw, tcw, Ow);
        else { std::memcpy(Rcw, pose, 36); std::memcpy(tcw, pose + 9, 12); std::memcpy(Ow, pose + 12, 12); }
        for (int i = 0; i < n; i++) {
            best_kp[(size_t)f * n + i] = -1;
            if (!pts->active[i]) continue;
            if (!shoot(KF, Rcw, tcw, Ow, fx, fy, cx, cy, pts, i, use_sim3 != 0, shot[f][i])) continue;
            q[f][i] = window(shot[f][i].u, shot[f][i].v, th * KF->scale_factors[shot[f][i].level], -1, -1);
        }
    }
    std::vector<Lists> L(n_kf);
    eao_status st = eao::match::build_lists_multi(n_kf, KFs, q.data(), qd.data(), L.data());
    if (st) return st;
    for (int f = 0; f < n_kf; f++) {
        const eao_frame_view* KF = KFs[f];
        int nf = 0;
        for (int i = 0; i < n; i++) {
            if (!q[f][i].active || L[f].count[i] == 0) continue;
            const Shot& s = shot[f][i];
            const float ur = s.u - bf * s.invz;
            int bestDist = use_sim3 ? INT_MAX : 256, bestIdx = -1;
            for (int c = 0; c < L[f].count[i]; c++) {
                const unsigned it = L[f].items[L[f].start[i] + c];
                const int k = (int)(it & 0xFFFF), d = (int)(it >> 16);
                const int kl = KF->kp_octave[k];
                if (kl < s.level - 1 || kl > s.level) continue;
                if (!use_sim3) {   // reprojection gates of the pose overload (src/ORBmatcher.cc:915-941)
                    const float exx = s.u - KF->kp_x[k], eyy = s.v - KF->kp_y[k];
                    if (KF->u_right[k] >= 0) {
                        const float er = ur - KF->u_right[k];
                        const float e2 = exx * exx + eyy * eyy + er * er;
                        if (e2 * KF->inv_level_sigma2[kl] > refc::FUSE_CHI2_STEREO) continue;
                    } else {
                        const float e2 = exx * exx + eyy * eyy;
                        if (e2 * KF->inv_level_sigma2[kl] > refc::FUSE_CHI2_MONO) continue;
                    }
                }
                if (d < bestDist) { bestDist = d; bestIdx = k; }
            }
            if (bestDist <= TH_LOW) { best_kp[(size_t)f * n + i] = bestIdx; nf++; }
        }
        nfused[f] = nf;
    }
    return EAO_OK;
}

eao_status eao_search_by_sim3(const eao_frame_view* K1, const float* T1w, const eao_map_points* pts1, const eao_frame_view* K2,
                              const float* T2w, const eao_map_points* pts2, float fx, float fy, float cx, float cy, float s12,
                              const float* R12, const float* t12, float th, int32_t* match12, int32_t* nfound) {
    return eao::search::by_sim3(K1, nullptr, T1w, pts1, K2, nullptr, T2w, pts2, fx, fy, cx, cy, s12, R12, t12, th, match12, nfound);
}

}  // extern "C"

eao_status eao::search::by_sim3(const eao_frame_view* K1, const match::Resident* res1, const float* T1w, const eao_map_points* pts1, const eao_frame_view* K2,
                                const match::Resident* res2, const float* T2w, const eao_map_points* pts2, float fx, float fy, float cx, float cy, float s12,
                                const float* R12, const float* t12, float th, int32_t* match12, int32_t* nfound) {
    EAO_REQUIRE(view_ok(K1) && view_ok(K2) && T1w && T2w && points_ok(pts1, false) && points_ok(pts2, false) && R12 && t12 && match12 && nfound,
                "bad argument");
    EAO_REQUIRE(finite_n(T1w, 16) && finite_n(T2w, 16) && finite_n(R12, 9) && finite_n(t12, 3) && std::isfinite(s12), "a pose / the sim3 holds a NaN / Inf");
    float R1w[9], t1w[3], R2w[9], t2w[3];
    split_pose(T1w, R1w, t1w);
    split_pose(T2w, R2w, t2w);
    float sR12[9], sR21[9], t21[3];
    const float is12 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = s12 * R12[r * 3 + c]; sR21[r * 3 + c] = is12 * R12[c * 3 + r]; }
    affine3(sR21, t12, nullptr, -1.f, t21);
    auto one_way = [&](const eao_map_points* P, const float* Rw, const float* tw, const float* sR, const float* t,
                       const eao_frame_view* K, const match::Resident* resK, std::vector<int>& out) -> eao_status {
        const int n = P->n;
        std::vector<Query> q(n, inactive());
        std::vector<int> level(n, 0);
        for (int i = 0; i < n; i++) {
            if (!P->active[i]) continue;
            float pa[3], pb[3];
            affine3(Rw, P->Xw + 3 * i, tw, 1.f, pa);
            affine3(sR, pa, t, 1.f, pb);
            if (pb[2] < 0.0) continue;
            const float invz = (float)(1.0 / pb[2]);
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!in_image(K, u, v)) continue;
            const float dist3D = norm3(pb);
            if (dist3D < P->min_dist_inv[i] || dist3D > P->max_dist_inv[i]) continue;
            const int lvl = predict_scale(P->max_dist[i], dist3D, K->log_scale_factor);
            if (lvl < 0 || lvl >= K->nlevels) continue;
            level[i] = lvl;
            q[i] = window(u, v, th * K->scale_factors[lvl], -1, -1);
        }
        Lists L;
        eao_status st = eao::match::build_lists(K, q, P->desc, L, resK);
        if (st) return st;
        out.assign(n, -1);
        for (int i = 0; i < n; i++) {
            if (!q[i].active) continue;
            int bestDist = INT_MAX, bestIdx = -1;
            for (int c = 0; c < L.count[i]; c++) {
                const unsigned it = L.items[L.start[i] + c];
                const int k = (int)(it & 0xFFFF), d = (int)(it >> 16);
                if (K->kp_octave[k] < level[i] - 1 || K->kp_octave[k] > level[i]) continue;
                if (d < bestDist) { bestDist = d; bestIdx = k; }
            }
            if (bestDist <= TH_HIGH) out[i] = bestIdx;
        }
        return EAO_OK;
    };
    std::vector<int> m1, m2;
    eao_status st = one_way(pts1, R1w, t1w, sR21, t21, K2, res2, m1);
    if (st) return st;
    if ((st = one_way(pts2, R2w, t2w, sR12, t12, K1, res1, m2))) return st;
    int nf = 0;
    for (int i1 = 0; i1 < pts1->n; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && idx2 < pts2->n && m2[idx2] == i1) { match12[i1] = idx2; nf++; }
    }
    *nfound = nf;
    return EAO_OK;
}

// ---- the two per-frame searches (moved here from match.hip in round 4: this file is the HOST half of every guided search -- no kernel, no HIP call --
//      so that it also builds with g++ -fsanitize=address,undefined against a CPU list provider: tests/test_host_replay_cpu.py, tools/run_sanitizers.sh)
// The search windows of ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono), src/ORBmatcher.cc:1339-1393: the relative
// motion along the optical axis decides the level window (:1343-1347), every valid last-frame point is projected with the current pose (float
// matrices, double accumulation) and gets a window of th x the scale factor of its octave.  Shared by the host-hop entry point below and by the
// device-resident chain (csrc/track.hip).
eao_status eao::match::build_frame_queries(const FrameQueryArgs& A, Query* q) {
    const float* Tcw = A.Tcw; const float* Tlw = A.Tlw;
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double sacc = 0;
        for (int k = 0; k < 3; k++) sacc += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3];
        twc[i] = (float)sacc;
    }
    for (int i = 0; i < 3; i++) {
        double sacc = 0;
        for (int k = 0; k < 3; k++) sacc += (double)Tlw[i * 4 + k] * (double)twc[k];
        tlc[i] = (float)(sacc + (double)Tlw[i * 4 + 3]);
    }
    const bool bForward = tlc[2] > A.mb && !A.mono;
    const bool bBackward = -tlc[2] > A.mb && !A.mono;
    for (int i = 0; i < A.n_last; i++) {
        Query& Q = q[i];
        Q.active = 0; Q.x = Q.y = Q.r = 0; Q.minLevel = 0; Q.maxLevel = -1; Q.urRef = 0; Q.urTol = 0;
        if (!A.valid[i]) continue;
        float xc3[3];
        for (int r = 0; r < 3; r++) {
            double sacc = 0;
            for (int k = 0; k < 3; k++) sacc += (double)Tcw[r * 4 + k] * (double)A.Xw[3 * i + k];
            xc3[r] = (float)(sacc + (double)Tcw[r * 4 + 3]);
        }
        const float invzc = (float)(1.0 / xc3[2]);
        if (!(invzc >= 0)) continue;                   // "if(invzc<0) continue" -- and a NaN projection (a NaN pose or point) matches nothing
        const float u = A.fx * xc3[0] * invzc + A.cx, v = A.fy * xc3[1] * invzc + A.cy;
        if (!(u >= A.min_x && u <= A.max_x)) continue;
        if (!(v >= A.min_y && v <= A.max_y)) continue;
        const int oct = A.last_octave[i];
        EAO_REQUIRE(oct >= 0 && oct < A.nlevels, "last-frame keypoint %d: octave %d out of range", i, oct);
        const float radius = A.th * A.scale_factors[oct];
        Q.active = 1; Q.x = u; Q.y = v; Q.r = radius;
        if (bForward) { Q.minLevel = oct; Q.maxLevel = -1; }
        else if (bBackward) { Q.minLevel = 0; Q.maxLevel = oct; }
        else { Q.minLevel = oct - 1; Q.maxLevel = oct + 1; }
        Q.urRef = u - A.mbf * invzc; Q.urTol = radius;
    }
    return EAO_OK;
}

extern "C" {

eao_status eao_search_by_projection_points(const eao_frame_view* F, int32_t n_mp, const float* proj_x, const float* proj_y,
                                           const float* proj_xr, const float* view_cos, const int32_t* pred_level,
                                           const uint8_t* mp_desc, const uint8_t* skip, float th, float nnratio,
                                           int32_t* match_kp, int32_t* nmatches) {
    EAO_REQUIRE(F && match_kp && nmatches && n_mp >= 0, "null argument");
    EAO_REQUIRE(n_mp == 0 || (proj_x && proj_y && proj_xr && view_cos && pred_level && mp_desc), "null map-point arrays");
    EAO_REQUIRE(F->n == 0 || (F->kp_x && F->kp_y && F->kp_octave && F->u_right && F->descriptors && F->scale_factors), "incomplete frame view");
    const bool bFactor = th != 1.0;
    std::vector<Query> q(n_mp);
    std::vector<float> rs(n_mp, 0.f);
    for (int m = 0; m < n_mp; m++) {
        Query& Q = q[m];
        Q.active = !(skip && skip[m]);
        const int lvl = pred_level[m];
        if (Q.active) EAO_REQUIRE(lvl >= 0 && lvl < F->nlevels, "map point %d: predicted level %d out of range", m, lvl);
        float r = view_cos[m] > refc::VIEWCOS_NARROW ? refc::RADIUS_NARROW : refc::RADIUS_WIDE;          // RadiusByViewingCos, :131-137
        if (bFactor) r *= th;
        rs[m] = Q.active ? r * F->scale_factors[lvl] : 0.f;
        Q.x = proj_x[m]; Q.y = proj_y[m]; Q.r = rs[m];
        Q.minLevel = lvl - 1; Q.maxLevel = lvl;
        Q.urRef = proj_xr[m]; Q.urTol = rs[m];
    }
    Lists L;
    eao_status st = build_lists(F, q, mp_desc, L);
    if (st) return st;
    std::vector<uint8_t> occ(F->n, 0);
    if (F->occupied) std::memcpy(occ.data(), F->occupied, F->n);
    int nm = 0;
    for (int m = 0; m < n_mp; m++) {   // upstream's loop, :51-126
        match_kp[m] = -1;
        if (!q[m].active || L.count[m] == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        const unsigned* it = &L.items[L.start[m]];
        for (int k = 0; k < L.count[m]; k++) {
            const int i = (int)(it[k] & 0xFFFF), d = (int)(it[k] >> 16);
            if (occ[i]) continue;
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = F->kp_octave[i]; bestIdx = i; }
            else if (d < bestDist2) { bestLevel2 = F->kp_octave[i]; bestDist2 = d; }
        }
        if (bestDist <= refc::TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match_kp[m] = bestIdx;
            occ[bestIdx] = 1;
            nm++;
        }
    }
    *nmatches = nm;
    return EAO_OK;
}

eao_status eao_search_by_projection_frames(const eao_frame_view* C, const float* Tcw, const float* Tlw, int32_t n_last,
                                           const uint8_t* valid, const float* Xw, const uint8_t* mp_desc, const int32_t* last_octave,
                                           const float* last_angle, float fx, float fy, float cx, float cy, float mbf, float mb,
                                           float th, int32_t mono, int32_t check_orientation, int32_t* cur_match, int32_t* nmatches) {
    EAO_REQUIRE(C && Tcw && Tlw && cur_match && nmatches && n_last >= 0, "null argument");
    EAO_REQUIRE(finite_n(Tcw, 16) && finite_n(Tlw, 16), "a pose holds a NaN / Inf");
    EAO_REQUIRE(n_last == 0 || (valid && Xw && mp_desc && last_octave && last_angle), "null last-frame arrays");
    EAO_REQUIRE(C->n == 0 || (C->kp_x && C->kp_y && C->kp_octave && C->kp_angle && C->u_right && C->descriptors && C->scale_factors), "incomplete frame view");
    for (int i = 0; i < C->n; i++) cur_match[i] = -1;
    std::vector<Query> q(n_last);
    eao::match::FrameQueryArgs QA{Tcw, Tlw, n_last, valid, Xw, last_octave, fx, fy, cx, cy, mbf, mb, th, mono, C->min_x, C->max_x, C->min_y, C->max_y, C->scale_factors, C->nlevels};
    if (eao_status qs = eao::match::build_frame_queries(QA, q.data())) return qs;
    Lists L;
    eao_status st = build_lists(C, q, mp_desc, L);
    if (st) return st;
    std::vector<uint8_t> occ(C->n, 0);
    if (C->occupied) std::memcpy(occ.data(), C->occupied, C->n);
    constexpr int HISTO = refc::HISTO_LENGTH;
    std::vector<int> rotHist[HISTO];
    const float factor = HISTO / 360.0f;          // this fork's histogram factor for this routine (:1337)
    int nm = 0;
    for (int i = 0; i < n_last; i++) {
        if (!q[i].active || L.count[i] == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        const unsigned* it = &L.items[L.start[i]];
        for (int k = 0; k < L.count[i]; k++) {
            const int i2 = (int)(it[k] & 0xFFFF), d = (int)(it[k] >> 16);
            if (occ[i2]) continue;
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= refc::TH_HIGH) {
            cur_match[bestIdx2] = i;
            occ[bestIdx2] = 1;
            nm++;
            if (check_orientation) {
                float rot = last_angle[i] - C->kp_angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO) bin = 0;
                if (bin >= 0 && bin < HISTO) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {   // keep the three fullest bins (ComputeThreeMaxima, :1603-1644)
        int top[3] = {0, 0, 0}, ind[3] = {-1, -1, -1};
        for (int b = 0; b < HISTO; b++) {
            const int pop = (int)rotHist[b].size();
            for (int rk = 0; rk < 3; rk++)
                if (pop > top[rk]) {
                    for (int z = 2; z > rk; z--) { top[z] = top[z - 1]; ind[z] = ind[z - 1]; }
                    top[rk] = pop; ind[rk] = b;
                    break;
                }
        }
        const float floor10 = 0.1f * (float)top[0];
        if (top[1] < floor10) { ind[1] = -1; ind[2] = -1; }
        else if (top[2] < floor10) ind[2] = -1;
        for (int b = 0; b < HISTO; b++)
            if (b != ind[0] && b != ind[1] && b != ind[2])
                for (int k : rotHist[b]) { cur_match[k] = -1; nm--; }
    }
    *nmatches = nm;
    return EAO_OK;
}

}  // extern "C"
