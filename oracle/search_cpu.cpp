// oracle/search_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into or called by the product).
//
// Sequential restatement, over plain arrays, of the remaining guided searches of the reference matcher:
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)                  src/ORBmatcher.cc:290-403
//   SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)            src/ORBmatcher.cc:1474-1601
//   SearchByBoW(KeyFrame*, Frame&, ...) / SearchByBoW(KeyFrame*, KeyFrame*, ...)  src/ORBmatcher.cc:159-288, 522-655
//   SearchForTriangulation + CheckDistEpipolarLine                               src/ORBmatcher.cc:657-823, 140-157
//   SearchForInitialization                                                      src/ORBmatcher.cc:405-520
//   Fuse(KeyFrame*, vpMapPoints, th) / Fuse(KeyFrame*, Scw, ...) (search half)   src/ORBmatcher.cc:825-975, 977-1100
//   SearchBySim3                                                                 src/ORBmatcher.cc:1102-1326
//   KeyFrame::GetFeaturesInArea / IsInImage, MapPoint::PredictScale              src/KeyFrame.cc:608-652, src/MapPoint.cc:385-394
// Keypoints live in a real grid of index vectors and every loop runs in upstream's order.
// PARITY UNPINNED (no upstream tests/fixtures; the matcher translation unit needs the whole SLAM object model and
// OpenCV/PCL/DBoW2 to compile).  Documented choices for the cv::Mat expressions of float matrices: a product A*x (+ b)
// accumulates in double and rounds once to float (cv::gemm); cv::norm / Mat::dot of floats accumulate in double; a matrix
// scaled by a float scalar is an element-wise float product; PredictScale evaluates logf / division / ceilf in float.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ref_constants.inc"   // GENERATED from the reference text (tools/gen_ref_constants.py): namespace refc

namespace {

struct FrameView {   // same layout as eao_frame_view (include/eao_fusion.h)
    int32_t n;
    const float* kp_x; const float* kp_y; const int32_t* kp_octave; const float* kp_angle; const float* u_right;
    const uint8_t* descriptors; const uint8_t* occupied;
    float min_x, min_y, max_x, max_y, grid_inv_w, grid_inv_h;
    int32_t grid_cols, grid_rows;
    const float* scale_factors; int32_t nlevels;
    float log_scale_factor; const float* level_sigma2; const float* inv_level_sigma2;
};
struct MapPoints {   // same layout as eao_map_points
    int32_t n;
    const uint8_t* active; const float* Xw; const float* normal; const float* min_dist_inv; const float* max_dist_inv;
    const float* max_dist; const uint8_t* desc;
};
struct FeatVec {     // same layout as eao_feature_vector
    int32_t n_nodes; const uint32_t* node_id; const int32_t* node_start; const uint32_t* index;
};

int dist256(const uint8_t* a, const uint8_t* b) {
    uint32_t wa[8], wb[8];
    std::memcpy(wa, a, 32); std::memcpy(wb, b, 32);
    int d = 0;
    for (int i = 0; i < 8; i++) d += __builtin_popcount(wa[i] ^ wb[i]);
    return d;
}

struct Grid {
    const FrameView& F;
    std::vector<std::vector<int>> cell;   // [ix * rows + iy]
    explicit Grid(const FrameView& f) : F(f), cell((size_t)f.grid_cols * f.grid_rows) {
        for (int i = 0; i < F.n; i++) {
            const int px = (int)std::round((F.kp_x[i] - F.min_x) * F.grid_inv_w);
            const int py = (int)std::round((F.kp_y[i] - F.min_y) * F.grid_inv_h);
            if (px < 0 || px >= F.grid_cols || py < 0 || py >= F.grid_rows) continue;
            cell[(size_t)px * F.grid_rows + py].push_back(i);
        }
    }
    // Frame::GetFeaturesInArea (minLevel = -1, maxLevel = -1 gives KeyFrame::GetFeaturesInArea: no level test)
    std::vector<int> area(float x, float y, float r, int minLevel, int maxLevel) const {
        std::vector<int> out;
        const int x0 = std::max(0, (int)std::floor((x - F.min_x - r) * F.grid_inv_w));
        if (x0 >= F.grid_cols) return out;
        const int x1 = std::min(F.grid_cols - 1, (int)std::ceil((x - F.min_x + r) * F.grid_inv_w));
        if (x1 < 0) return out;
        const int y0 = std::max(0, (int)std::floor((y - F.min_y - r) * F.grid_inv_h));
        if (y0 >= F.grid_rows) return out;
        const int y1 = std::min(F.grid_rows - 1, (int)std::ceil((y - F.min_y + r) * F.grid_inv_h));
        if (y1 < 0) return out;
        const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = x0; ix <= x1; ix++)
            for (int iy = y0; iy <= y1; iy++)
                for (int i : cell[(size_t)ix * F.grid_rows + iy]) {
                    if (checkLevels) {
                        if (F.kp_octave[i] < minLevel) continue;
                        if (maxLevel >= 0 && F.kp_octave[i] > maxLevel) continue;
                    }
                    const float dx = F.kp_x[i] - x, dy = F.kp_y[i] - y;
                    if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(i);
                }
        return out;
    }
};

// y = A x + b with A 3x3 row-major (stride ld), double accumulation, one rounding
void affine3(const float* A, int ld, const float* x, const float* b, float alpha, float* y) {
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)A[r * ld + k] * (double)x[k];
        y[r] = (float)((double)alpha * s + (b ? (double)b[r] : 0.0));
    }
}
float norm3(const float* v) { return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
double dot3(const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }   // Mat::dot returns double
int predict_scale(float maxDistance, float currentDist, float logScaleFactor) {
    const float ratio = maxDistance / currentDist;
    return (int)std::ceil(std::log(ratio) / logScaleFactor);
}
bool in_image(const FrameView& K, float x, float y) { return x >= K.min_x && x < K.max_x && y >= K.min_y && y < K.max_y; }

// Scw -> Rcw, tcw, Ow (src/ORBmatcher.cc:298-303)
void decompose_sim3(const float* S, float* Rcw, float* tcw, float* Ow) {
    const float scw = (float)std::sqrt((double)S[0] * S[0] + (double)S[1] * S[1] + (double)S[2] * S[2]);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = S[r * 4 + c] / scw;
        tcw[r] = S[r * 4 + 3] / scw;
    }
    for (int i = 0; i < 3; i++) {   // -Rcw^T tcw
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        Ow[i] = (float)(-s);
    }
}

void three_maxima(const std::vector<int>* hist, int L, int& ind1, int& ind2, int& ind3) {   // src/ORBmatcher.cc:1603-1644
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = (int)hist[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
int rot_bin(float a1, float a2, float factor, int L) {
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == L) bin = 0;
    return bin;
}
const int TH_HIGH = refc::TH_HIGH, TH_LOW = refc::TH_LOW, HISTO = refc::HISTO_LENGTH;

}  // namespace

extern "C" {

int orc_search_by_projection_sim3(const FrameView* K, const float* Scw, float fx, float fy, float cx, float cy,
                                  const MapPoints* P, int th, int32_t* kp_match) {
    Grid grid(*K);
    float Rcw[9], tcw[3], Ow[3];
    decompose_sim3(Scw, Rcw, tcw, Ow);
    std::vector<uint8_t> occ(K->n, 0);
    if (K->occupied) std::memcpy(occ.data(), K->occupied, K->n);
    for (int k = 0; k < K->n; k++) kp_match[k] = -1;
    int nmatches = 0;
    for (int i = 0; i < P->n; i++) {
        if (!P->active[i]) continue;
        const float* Xw = P->Xw + 3 * i;
        float pc[3];
        affine3(Rcw, 3, Xw, tcw, 1.f, pc);
        if (pc[2] < 0.0) continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!in_image(*K, u, v)) continue;
        const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
        const float dist = norm3(PO);
        if (dist < P->min_dist_inv[i] || dist > P->max_dist_inv[i]) continue;
        if (dot3(PO, P->normal + 3 * i) < 0.5 * dist) continue;
        const int lvl = predict_scale(P->max_dist[i], dist, K->log_scale_factor);
        if (lvl < 0 || lvl >= K->nlevels) continue;   // upstream would index out of range here
        const float radius = th * K->scale_factors[lvl];
        const std::vector<int> idx = grid.area(u, v, radius, -1, -1);
        if (idx.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int k : idx) {
            if (occ[k]) continue;
            const int kl = K->kp_octave[k];
            if (kl < lvl - 1 || kl > lvl) continue;
            const int d = dist256(P->desc + 32 * (size_t)i, K->descriptors + 32 * (size_t)k);
            if (d < bestDist) { bestDist = d; bestIdx = k; }
        }
        if (bestDist <= TH_LOW) { kp_match[bestIdx] = i; occ[bestIdx] = 1; nmatches++; }
    }
    return nmatches;
}

int orc_search_by_projection_kf(const FrameView* C, const float* Tcw, float fx, float fy, float cx, float cy, const MapPoints* P,
                                const float* kf_angle, float th, int orb_dist, int check_orientation, int32_t* cur_match) {
    Grid grid(*C);
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        Ow[i] = (float)(-s);
    }
    std::vector<uint8_t> occ(C->n, 0);
    if (C->occupied) std::memcpy(occ.data(), C->occupied, C->n);
    for (int k = 0; k < C->n; k++) cur_match[k] = -1;
    std::vector<int> rotHist[HISTO];
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    for (int i = 0; i < P->n; i++) {
        if (!P->active[i]) continue;
        const float* Xw = P->Xw + 3 * i;
        float xc3[3];
        affine3(Rcw, 3, Xw, tcw, 1.f, xc3);
        const float invzc = (float)(1.0 / xc3[2]);
        const float u = fx * xc3[0] * invzc + cx, v = fy * xc3[1] * invzc + cy;
        if (u < C->min_x || u > C->max_x) continue;
        if (v < C->min_y || v > C->max_y) continue;
        const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < P->min_dist_inv[i] || dist3D > P->max_dist_inv[i]) continue;
        const int lvl = predict_scale(P->max_dist[i], dist3D, C->log_scale_factor);
        if (lvl < 0 || lvl >= C->nlevels) continue;
        const float radius = th * C->scale_factors[lvl];
        const std::vector<int> idx = grid.area(u, v, radius, lvl - 1, lvl + 1);
        if (idx.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : idx) {
            if (occ[i2]) continue;
            const int d = dist256(P->desc + 32 * (size_t)i, C->descriptors + 32 * (size_t)i2);
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur_match[bestIdx2] = i; occ[bestIdx2] = 1; nmatches++;
            if (check_orientation) rotHist[rot_bin(kf_angle[i], C->kp_angle[bestIdx2], factor, HISTO)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(rotHist, HISTO, i1, i2, i3);
        for (int b = 0; b < HISTO; b++)
            if (b != i1 && b != i2 && b != i3)
                for (int k : rotHist[b]) { cur_match[k] = -1; nmatches--; }
    }
    return nmatches;
}

int orc_search_by_bow(int mode, int n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const FeatVec* f1,
                      int n2, const uint8_t* desc2, const float* angle2, const uint8_t* valid2, const FeatVec* f2, float nnratio,
                      int check_orientation, int32_t* match12) {
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> taken2(n2, 0);
    std::vector<int> rotHist[HISTO];
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < f1->n_nodes && b < f2->n_nodes) {
        if (f1->node_id[a] == f2->node_id[b]) {
            for (int p = f1->node_start[a]; p < f1->node_start[a + 1]; p++) {
                const int idx1 = (int)f1->index[p];
                if (!valid1[idx1]) continue;
                int best1 = 256, bestIdx2 = -1, best2 = 256;
                for (int q = f2->node_start[b]; q < f2->node_start[b + 1]; q++) {
                    const int idx2 = (int)f2->index[q];
                    if (taken2[idx2]) continue;
                    if (mode == 1 && !valid2[idx2]) continue;
                    const int d = dist256(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (d < best1) { best2 = best1; best1 = d; bestIdx2 = idx2; }
                    else if (d < best2) { best2 = d; }
                }
                const bool close = mode == 0 ? best1 <= TH_LOW : best1 < TH_LOW;
                if (close && (float)best1 < nnratio * (float)best2) {
                    match12[idx1] = bestIdx2;
                    taken2[bestIdx2] = 1;
                    if (check_orientation) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2], factor, HISTO)].push_back(idx1);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (f1->node_id[a] < f2->node_id[b]) {
            while (a < f1->n_nodes && f1->node_id[a] < f2->node_id[b]) a++;      // lower_bound
        } else {
            while (b < f2->n_nodes && f2->node_id[b] < f1->node_id[a]) b++;
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(rotHist, HISTO, i1, i2, i3);
        for (int h = 0; h < HISTO; h++)
            if (h != i1 && h != i2 && h != i3)
                for (int k : rotHist[h]) { match12[k] = -1; nmatches--; }
    }
    (void)n1;
    return nmatches;
}

int orc_search_for_triangulation(const FrameView* K1, const FeatVec* f1, const FrameView* K2, const FeatVec* f2, const float* F12,
                                 float ex, float ey, int only_stereo, int check_orientation, int32_t* match12) {
    for (int i = 0; i < K1->n; i++) match12[i] = -1;
    std::vector<int> rotHist[HISTO];
    const float factor = 1.0f / HISTO;
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < f1->n_nodes && b < f2->n_nodes) {
        if (f1->node_id[a] == f2->node_id[b]) {
            for (int p = f1->node_start[a]; p < f1->node_start[a + 1]; p++) {
                const int idx1 = (int)f1->index[p];
                if (K1->occupied && K1->occupied[idx1]) continue;
                const bool stereo1 = K1->u_right[idx1] >= 0;
                if (only_stereo && !stereo1) continue;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int q = f2->node_start[b]; q < f2->node_start[b + 1]; q++) {
                    const int idx2 = (int)f2->index[q];
                    if (K2->occupied && K2->occupied[idx2]) continue;     // vbMatched2 is never set upstream
                    const bool stereo2 = K2->u_right[idx2] >= 0;
                    if (only_stereo && !stereo2) continue;
                    const int d = dist256(K1->descriptors + 32 * (size_t)idx1, K2->descriptors + 32 * (size_t)idx2);
                    if (d > TH_LOW || d > bestDist) continue;
                    const float x2 = K2->kp_x[idx2], y2 = K2->kp_y[idx2];
                    if (!stereo1 && !stereo2) {
                        const float dex = ex - x2, dey = ey - y2;
                        if (dex * dex + dey * dey < 100 * K2->scale_factors[K2->kp_octave[idx2]]) continue;
                    }
                    // CheckDistEpipolarLine
                    const float x1 = K1->kp_x[idx1], y1 = K1->kp_y[idx1];
                    const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
                    const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
                    const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
                    const float num = la * x2 + lb * y2 + lc;
                    const float den = la * la + lb * lb;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < refc::EPIPOLAR_CHI2 * K2->level_sigma2[K2->kp_octave[idx2]]) { bestIdx2 = idx2; bestDist = d; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) rotHist[rot_bin(K1->kp_angle[idx1], K2->kp_angle[bestIdx2], factor, HISTO)].push_back(idx1);
                }
            }
            a++; b++;
        } else if (f1->node_id[a] < f2->node_id[b]) {
            while (a < f1->n_nodes && f1->node_id[a] < f2->node_id[b]) a++;
        } else {
            while (b < f2->n_nodes && f2->node_id[b] < f1->node_id[a]) b++;
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(rotHist, HISTO, i1, i2, i3);
        for (int h = 0; h < HISTO; h++)
            if (h != i1 && h != i2 && h != i3)
                for (int k : rotHist[h]) { match12[k] = -1; nmatches--; }
    }
    return nmatches;
}

int orc_search_for_initialization(int n1, const int32_t* octave1, const float* angle1, const uint8_t* desc1, const FrameView* F2,
                                  float* prev_matched, int window, float nnratio, int check_orientation, int32_t* match12) {
    Grid grid(*F2);
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<int> rotHist[HISTO];
    const float factor = 1.0f / HISTO;
    std::vector<int> matchedDistance(F2->n, INT_MAX), matches21(F2->n, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = octave1[i1];
        if (level1 > 0) continue;
        const std::vector<int> idx = grid.area(prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1, level1);
        if (idx.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : idx) {
            const int d = dist256(desc1 + 32 * (size_t)i1, F2->descriptors + 32 * (size_t)i2);
            if (matchedDistance[i2] <= d) continue;
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) { bestDist2 = d; }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) { match12[matches21[bestIdx2]] = -1; nmatches--; }
                match12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) rotHist[rot_bin(angle1[i1], F2->kp_angle[bestIdx2], factor, HISTO)].push_back(i1);
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(rotHist, HISTO, i1, i2, i3);
        for (int h = 0; h < HISTO; h++)
            if (h != i1 && h != i2 && h != i3)
                for (int k : rotHist[h])
                    if (match12[k] >= 0) { match12[k] = -1; nmatches--; }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (match12[i1] >= 0) { prev_matched[2 * i1] = F2->kp_x[match12[i1]]; prev_matched[2 * i1 + 1] = F2->kp_y[match12[i1]]; }
    return nmatches;
}

int orc_fuse_search(const FrameView* K, int use_sim3, const float* pose, float fx, float fy, float cx, float cy, float bf,
                    const MapPoints* P, float th, int32_t* best_kp) {
    Grid grid(*K);
    float Rcw[9], tcw[3], Ow[3];
    if (use_sim3) decompose_sim3(pose, Rcw, tcw, Ow);
    else { std::memcpy(Rcw, pose, 36); std::memcpy(tcw, pose + 9, 12); std::memcpy(Ow, pose + 12, 12); }
    int nFused = 0;
    for (int i = 0; i < P->n; i++) {
        best_kp[i] = -1;
        if (!P->active[i]) continue;
        const float* Xw = P->Xw + 3 * i;
        float pc[3];
        affine3(Rcw, 3, Xw, tcw, 1.f, pc);
        if (pc[2] < 0.0f) continue;
        const float invz = use_sim3 ? (float)(1.0 / pc[2]) : 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!in_image(*K, u, v)) continue;
        const float ur = u - bf * invz;
        const float PO[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < P->min_dist_inv[i] || dist3D > P->max_dist_inv[i]) continue;
        if (dot3(PO, P->normal + 3 * i) < 0.5 * dist3D) continue;
        const int lvl = predict_scale(P->max_dist[i], dist3D, K->log_scale_factor);
        if (lvl < 0 || lvl >= K->nlevels) continue;
        const float radius = th * K->scale_factors[lvl];
        const std::vector<int> idx = grid.area(u, v, radius, -1, -1);
        if (idx.empty()) continue;
        int bestDist = use_sim3 ? INT_MAX : 256, bestIdx = -1;
        for (int k : idx) {
            const int kl = K->kp_octave[k];
            if (kl < lvl - 1 || kl > lvl) continue;
            if (!use_sim3) {
                const float exx = u - K->kp_x[k], eyy = v - K->kp_y[k];
                if (K->u_right[k] >= 0) {
                    const float er = ur - K->u_right[k];
                    const float e2 = exx * exx + eyy * eyy + er * er;
                    if (e2 * K->inv_level_sigma2[kl] > refc::FUSE_CHI2_STEREO) continue;
                } else {
                    const float e2 = exx * exx + eyy * eyy;
                    if (e2 * K->inv_level_sigma2[kl] > refc::FUSE_CHI2_MONO) continue;
                }
            }
            const int d = dist256(P->desc + 32 * (size_t)i, K->descriptors + 32 * (size_t)k);
            if (d < bestDist) { bestDist = d; bestIdx = k; }
        }
        if (bestDist <= TH_LOW) { best_kp[i] = bestIdx; nFused++; }
    }
    return nFused;
}

int orc_search_by_sim3(const FrameView* K1, const float* T1w, const MapPoints* P1, const FrameView* K2, const float* T2w,
                       const MapPoints* P2, float fx, float fy, float cx, float cy, float s12, const float* R12, const float* t12,
                       float th, int32_t* match12) {
    Grid g1(*K1), g2(*K2);
    float R1w[9], t1w[3], R2w[9], t2w[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { R1w[r * 3 + c] = T1w[r * 4 + c]; R2w[r * 3 + c] = T2w[r * 4 + c]; }
        t1w[r] = T1w[r * 4 + 3]; t2w[r] = T2w[r * 4 + 3];
    }
    float sR12[9], sR21[9], t21[3];
    const float is12 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = s12 * R12[r * 3 + c]; sR21[r * 3 + c] = is12 * R12[c * 3 + r]; }
    affine3(sR21, 3, t12, nullptr, -1.f, t21);
    const int N1 = P1->n, N2 = P2->n;
    std::vector<int> m1(N1, -1), m2(N2, -1);
    auto one_way = [&](const MapPoints* P, const float* Rw, const float* tw, const float* sR, const float* t, const FrameView* K,
                       const Grid& g, std::vector<int>& out) {
        for (int i = 0; i < P->n; i++) {
            if (!P->active[i]) continue;
            float pa[3], pb[3];
            affine3(Rw, 3, P->Xw + 3 * i, tw, 1.f, pa);
            affine3(sR, 3, pa, t, 1.f, pb);
            if (pb[2] < 0.0) continue;
            const float invz = (float)(1.0 / pb[2]);
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!in_image(*K, u, v)) continue;
            const float dist3D = norm3(pb);
            if (dist3D < P->min_dist_inv[i] || dist3D > P->max_dist_inv[i]) continue;
            const int lvl = predict_scale(P->max_dist[i], dist3D, K->log_scale_factor);
            if (lvl < 0 || lvl >= K->nlevels) continue;
            const float radius = th * K->scale_factors[lvl];
            const std::vector<int> idx = g.area(u, v, radius, -1, -1);
            if (idx.empty()) continue;
            int bestDist = INT_MAX, bestIdx = -1;
            for (int k : idx) {
                if (K->kp_octave[k] < lvl - 1 || K->kp_octave[k] > lvl) continue;
                const int d = dist256(P->desc + 32 * (size_t)i, K->descriptors + 32 * (size_t)k);
                if (d < bestDist) { bestDist = d; bestIdx = k; }
            }
            if (bestDist <= TH_HIGH) out[i] = bestIdx;
        }
    };
    one_way(P1, R1w, t1w, sR21, t21, K2, g2, m1);
    one_way(P2, R2w, t2w, sR12, t12, K1, g1, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && idx2 < N2 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    return nFound;
}

}  // extern "C"
