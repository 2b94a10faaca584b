// oracle/match_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into or called by the product).
//
// Sequential restatement, over plain arrays, of the two per-frame guided searches of the reference matcher:
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)        src/ORBmatcher.cc:45-129, 131-137
//   ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)    src/ORBmatcher.cc:1328-1472
//   Frame::AssignFeaturesToGrid / GetFeaturesInArea / PosInGrid                 src/Frame.cc:599-614, 696-761
//   ORBmatcher::ComputeThreeMaxima                                              src/ORBmatcher.cc:1603-1644
// The frame's keypoints live in a real grid of index vectors, candidates are visited in the grid's order, and every
// query sees the assignments of the queries before it -- the same control flow as upstream.
// PARITY UNPINNED (no upstream tests/fixtures; the matcher translation unit needs the whole SLAM object model and
// OpenCV/PCL/DBoW2 to compile).  Documented choices: a keypoint assigned during the call counts as occupied
// (upstream: Observations() > 0 of the assigned point); Rcw*Xw+tcw accumulates in double and rounds to float
// (cv::gemm's small-matrix path).
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
    float log_scale_factor; const float* level_sigma2; const float* inv_level_sigma2;   // unused here
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

}  // namespace

extern "C" {

int orc_search_by_projection_points(const FrameView* F, int n_mp, const float* proj_x, const float* proj_y, const float* proj_xr,
                                    const float* view_cos, const int32_t* pred_level, const uint8_t* mp_desc, const uint8_t* skip,
                                    float th, float nnratio, int32_t* match_kp) {
    Grid grid(*F);
    std::vector<uint8_t> occ(F->n, 0);
    if (F->occupied) std::memcpy(occ.data(), F->occupied, F->n);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int m = 0; m < n_mp; m++) {
        match_kp[m] = -1;
        if (skip && skip[m]) continue;
        const int lvl = pred_level[m];
        float r = view_cos[m] > refc::VIEWCOS_NARROW ? refc::RADIUS_NARROW : refc::RADIUS_WIDE;
        if (bFactor) r *= th;
        const float rs = r * F->scale_factors[lvl];
        const std::vector<int> idx = grid.area(proj_x[m], proj_y[m], rs, lvl - 1, lvl);
        if (idx.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int i : idx) {
            if (occ[i]) continue;
            if (F->u_right[i] > 0) {
                const float er = std::fabs(proj_xr[m] - F->u_right[i]);
                if (er > rs) continue;
            }
            const int d = dist256(mp_desc + 32 * (size_t)m, F->descriptors + 32 * (size_t)i);
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = F->kp_octave[i]; bestIdx = i; }
            else if (d < bestDist2) { bestLevel2 = F->kp_octave[i]; bestDist2 = d; }
        }
        if (bestDist <= refc::TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match_kp[m] = bestIdx;
            occ[bestIdx] = 1;
            nmatches++;
        }
    }
    return nmatches;
}

int orc_search_by_projection_frames(const FrameView* C, const float* Tcw, const float* Tlw, int n_last, const uint8_t* valid,
                                    const float* Xw, const uint8_t* mp_desc, const int32_t* last_octave, const float* last_angle,
                                    float fx, float fy, float cx, float cy, float mbf, float mb, float th, int mono,
                                    int check_orientation, int32_t* cur_match) {
    Grid grid(*C);
    std::vector<uint8_t> occ(C->n, 0);
    if (C->occupied) std::memcpy(occ.data(), C->occupied, C->n);
    for (int i = 0; i < C->n; i++) cur_match[i] = -1;
    int nmatches = 0;
    const int HISTO = refc::HISTO_LENGTH;
    std::vector<int> rotHist[HISTO];
    const float factor = HISTO / 360.0f;
    // twc = -Rcw^T tcw ; tlc = Rlw twc + tlw   (float matrices, double accumulation)
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3];
        twc[i] = (float)s;
    }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tlw[i * 4 + k] * (double)twc[k];
        tlc[i] = (float)(s + (double)Tlw[i * 4 + 3]);
    }
    const bool bForward = tlc[2] > mb && !mono;
    const bool bBackward = -tlc[2] > mb && !mono;
    for (int i = 0; i < n_last; i++) {
        if (!valid[i]) continue;
        float xc3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)Tcw[r * 4 + k] * (double)Xw[3 * i + k];
            xc3[r] = (float)(s + (double)Tcw[r * 4 + 3]);   // gemm(A, B, 1, C, 1): one rounding
        }
        const float xc = xc3[0], yc = xc3[1];
        const float invzc = (float)(1.0 / xc3[2]);
        if (invzc < 0) continue;
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < C->min_x || u > C->max_x) continue;
        if (v < C->min_y || v > C->max_y) continue;
        const int oct = last_octave[i];
        const float radius = th * C->scale_factors[oct];
        std::vector<int> idx;
        if (bForward) idx = grid.area(u, v, radius, oct, -1);
        else if (bBackward) idx = grid.area(u, v, radius, 0, oct);
        else idx = grid.area(u, v, radius, oct - 1, oct + 1);
        if (idx.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : idx) {
            if (occ[i2]) continue;
            if (C->u_right[i2] > 0) {
                const float ur = u - mbf * invzc;
                const float er = std::fabs(ur - C->u_right[i2]);
                if (er > radius) continue;
            }
            const int d = dist256(mp_desc + 32 * (size_t)i, C->descriptors + 32 * (size_t)i2);
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= refc::TH_HIGH) {
            cur_match[bestIdx2] = i;
            occ[bestIdx2] = 1;
            nmatches++;
            if (check_orientation) {
                float rot = last_angle[i] - C->kp_angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < HISTO; b++) {
            const int s = (int)rotHist[b].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
            else if (s > max3) { max3 = s; ind3 = b; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int b = 0; b < HISTO; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int k : rotHist[b]) { cur_match[k] = -1; nmatches--; }
    }
    return nmatches;
}

}  // extern "C"

// ---- candidate lists for the HOST half of the product (round 4; tests/test_host_replay_cpu.py, tools/run_sanitizers.sh): what the product's candidate
// kernel hands its host replay -- per query the keypoints of Frame::GetFeaturesInArea (src/Frame.cc:696-749) in the grid's visiting order, with the stereo
// gate the searches apply inside their candidate loops (src/ORBmatcher.cc:95-100, 1405-1411: a keypoint with uRight > 0 whose |urRef - uRight| exceeds the
// tolerance is skipped) and the Hamming distance to the query's descriptor, packed distance << 16 | keypoint.  This lets the product's host replay code
// (eao_fusion_amd/csrc/search.hip compiled as plain C++ with sanitizers, tests/cpp/host_replay_provider.cpp as its list provider) run WITHOUT a GPU.
// Query: the eight 4-byte fields of eao::match::Query (csrc/match_internal.h): x, y, r, minLevel, maxLevel, urRef, urTol, active.
extern "C" int orc_candidate_lists(const FrameView* F, int nq, const void* queries, const uint8_t* qdesc, int32_t* start, int32_t* count, uint32_t* items,
                                   int items_cap) {
    struct Q { float x, y, r; int32_t minLevel, maxLevel; float urRef, urTol; int32_t active; };
    const Q* q = (const Q*)queries;
    const Grid g(*F);
    int total = 0;
    for (int k = 0; k < nq; k++) {
        start[k] = total; count[k] = 0;
        if (!q[k].active) continue;
        for (int i : g.area(q[k].x, q[k].y, q[k].r, q[k].minLevel, q[k].maxLevel)) {
            const float u = F->u_right[i];
            if (u > 0 && std::fabs(q[k].urRef - u) > q[k].urTol) continue;
            if (total >= items_cap) return -1;
            items[total++] = ((uint32_t)dist256(qdesc + 32 * (size_t)k, F->descriptors + 32 * (size_t)i) << 16) | (uint32_t)i;
            count[k]++;
        }
    }
    return total;
}
extern "C" int orc_pair_distance(const uint8_t* a, const uint8_t* b) { return dist256(a, b); }
