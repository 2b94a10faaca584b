// track.hip -- the device-resident tracked frame (second half), MI355X (gfx950).
//
// Tracking::TrackLocalMap's data path -- Frame::ComputeStereoFromRGBD + AssignFeaturesToGrid on the extractor's output,
// Frame::isInFrustum over the local map (Tracking::SearchLocalPoints), ORBmatcher::SearchByProjection(Frame&, local map
// points, th) and Optimizer::PoseOptimization(&frame) -- reference src/Tracking.cc:1717-2231, 2587-2641; src/Frame.cc:
// 599-614, 638-761, 1016-1037; src/ORBmatcher.cc:45-137; src/Optimizer.cc:325-673 -- chained on ONE stream over arrays
// that never leave HBM: the keypoints, descriptors and keypoint count come straight from eao_orb_extract_batch_device, the
// local map is uploaded when it changes, and one copy brings the pose, the matches and the outlier flags back.
//
// Four launches (eight until late in round 2: every launch of this latency chain costs its ~3 us gap):
//   k_track_frame    workgroup 0: cv::KeyPoint records -> coordinate / octave / angle arrays, mvuRight / mvDepth from the
//                    depth image, PosInGrid keys sorted in LDS -> the grid-order walk list k_match_candidates uses;
//                    workgroups 1..: Frame::isInFrustum, one thread per local map point (frustum_point, chain_internal.h -- the
//                    same code as frame.hip's k_is_in_frustum), beside it
//   k_match_candidates (match.hip) one wave per local map point: its search window (RadiusByViewingCos x th x scale factor of the
//                    predicted level -- QueryBuild, chain_internal.h; a launch of its own until late in round 2), then its candidate
//                    list in upstream's order
//   k_track_assign_edges   one workgroup, two steps:
//       assignment   upstream's greedy assignment (a keypoint taken by an earlier map point is skipped by later ones) WITHOUT
//                    walking the map points one by one: rounds in which every undecided point decides from the final claims
//                    so far, and becomes final when no earlier undecided point lists any keypoint its decision depends on
//       edges        mvpMapPoints by keypoint (prior matches + new ones), the PoseOptimization edges in keypoint order
//   k_pose_optimization (lm.hip) with the edge count read on the device
//                    (it also scatters mvbOutlier by keypoint into the result block; the rest of the block is written by the tail of
//                    k_track_assign_edges -- a k_track_finish launch behind PoseOptimization until late in round 2)
#include <cmath>
#include <cstring>
#include <chrono>
#include <atomic>
#include <vector>

#include "chain_internal.h"
#include "common.h"

using eao::match::Query;

struct eao_tracker {
    eao_tracker_cfg cfg;
    std::vector<float> scale, invSigma2;
    int cap = 0, capMp = 0, nMp = 0, nCells = 0;
    std::vector<unsigned char> hActive;      // host copy of the local map's active[]: prior matches are checked against it
    hipStream_t stream = nullptr;
    hipEvent_t evIn = nullptr, evOut = nullptr;
    eao::DevBuf<unsigned char> dev;
    unsigned char* pin = nullptr;      // pinned staging: local map upload, prior matches in, result block out
    size_t pinCap = 0;
    // device slices (offsets into dev)
    float *kx, *ky, *ang, *ur, *dz; int* oct; int* order; unsigned short *cellx, *celly; int* counts;   // counts: n, nOrdered, nEdges, nMatches, err
    int* colStart = nullptr;
    int* prior; float* priorXw; int* kpMp; unsigned char* occ; unsigned char* kpOut;
    float *mXw, *mNormal, *mMin, *mMax, *mNum; unsigned char* mDesc; unsigned char* mActive; unsigned char* mSkip;
    unsigned char* inView; float *projX, *projY, *projXR, *viewCos; int* level;
    Query* q; unsigned* lists; int *segStart, *segCount, *cursor; int* match;
    double *eXw, *eObs, *eInfo, *eErr; unsigned char *eFlags, *eOutl; int* eKp;
    float* dScale; float* dInvSigma2;
    // the LAST frame's map points (TrackWithMotionModel): uploaded per call, apart from the local map, which stays where it is
    float* lXw = nullptr; unsigned char* lDesc = nullptr; float* lAng = nullptr; unsigned char* lSkip = nullptr; Query* lQ = nullptr;      // lQ | lXw | lDesc | lAng: ONE block, one copy
    int capQ = 0;                      // queries / candidate segments / matches: max(max_keypoints, max_map_points)
    unsigned char* res = nullptr;      // result block: device view of resPin, resBytes
    unsigned char* resPin = nullptr;
    unsigned char* resDev = nullptr;   // its DEVICE twin (same layout): every kernel of the chain writes here; the chain's last launch alone copies it to `res`
    size_t resBytes = 0, listCap = 0, assignLds = 0;
    int seq = 0;                       // call counter: the chain's last launch stores it in the result block's done word
    // eao_tracker_set_options (round 5): consumed by the NEXT track call
    int optMinMatches = 0, optPlanes = 0;
    unsigned char* optPlaneOut = nullptr;
    eao::frame::Distortion dist{};     // eao_tracker_set_distortion (persistent): dist.on = the frame set-up undistorts the keypoints first
    double* plPin = nullptr; double* plDev = nullptr;      // the plane edges' records in mapped pinned memory (10 doubles each), as the pose kernel reads them
    long long* dbg = nullptr;          // EAO_DEBUG_STAMPS: phase stamps of k_track_assign_edges (diagnostic runs only)
    ~eao_tracker() {
        if (pin) (void)hipHostFree(pin);
        if (plPin) (void)hipHostFree(plPin);
        if (resPin) (void)hipHostFree(resPin);
        if (dbg) (void)hipFree(dbg);
        if (evIn) (void)hipEventDestroy(evIn);
        if (evOut) (void)hipEventDestroy(evOut);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

struct FrameArrays {
    float *kx, *ky, *ang, *ur, *dz; int* oct; int* order; unsigned short *cellx, *celly; int* counts;
    int* prior; int* kpMp; unsigned char* occ; unsigned char* mSkip;
    int* cursor;      // k_match_candidates' list cursor, zeroed here (instead of a fill launch in front of it)
    int* colStart;    // cols + 1: first walk-list entry of every grid column
};

constexpr int kFrameThreads = 1024;
// ONE workgroup.  Frame::Frame's per-keypoint work for a distortion-free camera (mvKeysUn = mvKeys, as in the reference's
// RGB-D configuration: ros_test/config/TUM3.yaml:13-16): ComputeStereoFromRGBD (src/Frame.cc:1016-1037) and
// AssignFeaturesToGrid / PosInGrid (:599-614, 751-761).  The grid leaves as the walk list of k_match_candidates: keypoints in
// (cell column, cell row, index) order = the order upstream's nested cell loops of GetFeaturesInArea visit them.
__global__ __launch_bounds__(kFrameThreads) void k_track_frame(const eao_keypoint* __restrict__ kps, const int* __restrict__ nPtr, int cap,
                                                               const float* __restrict__ depth, int pitch, int W, int H, float mbf,
                                                               float minX, float minY, float invW, float invH, int cols, int rows, int npow2,
                                                               int nMp, FrameArrays A, eao::frame::FrustumArgs FA, long long* dbg, int countingSort,
                                                               eao::frame::Distortion D) {
    extern __shared__ unsigned tkeys[];
    __shared__ int s_cnt, s_maxc, s_wtot[kFrameThreads / 64];
    const int t = threadIdx.x;
    // Workgroups 1.. : Frame::isInFrustum over the local map points (independent of the keypoints) beside workgroup 0's frame
    // set-up -- as a launch of its own it cost 4.5 us plus a launch gap in front of every tracked frame's searches
    if (blockIdx.x > 0) {
        const int m = ((int)blockIdx.x - 1) * kFrameThreads + t;
        if (m < FA.n) eao::frame::frustum_point(FA, m);
        return;
    }
    if (dbg && t == 0) dbg[24] = clock64();
    const int n = min(max(*nPtr, 0), cap);
    if (t == 0) { s_cnt = 0; *A.cursor = 0; s_maxc = 0; }
    for (int m = t; m < nMp; m += kFrameThreads) A.mSkip[m] = 0;
    if (countingSort) for (int c = t; c <= cols * rows; c += kFrameThreads) tkeys[npow2 + c] = 0;      // (the cell histogram of the counting sort below)
    __syncthreads();
    for (int i = t; i < npow2; i += kFrameThreads) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n) {
            const eao_keypoint kp = kps[i];
            // mvKeysUn (Frame::UndistortKeyPoints, src/Frame.cc:773-806): what the grid, the searches and the pose edges read; the depth image is
            // looked up at the DISTORTED keypoint (:1024-1029)
            float ux = kp.x, uy = kp.y;
            if (D.on) eao::frame::undistort_point(D, kp.x, kp.y, ux, uy);
            A.kx[i] = ux; A.ky[i] = uy; A.ang[i] = kp.angle; A.oct[i] = kp.octave;
            float ur = -1.0f, dz = -1.0f;
            const int xi = (int)kp.x, yi = (int)kp.y;          // Mat::at<float>(float v, float u): truncation
            if (depth && xi >= 0 && xi < W && yi >= 0 && yi < H) {
                const float d = depth[(size_t)yi * pitch + xi];
                if (d > 0) { dz = d; ur = ux - mbf / d; }
            }
            A.ur[i] = ur; A.dz[i] = dz;
            const int px = (int)roundf((ux - minX) * invW);      // PosInGrid, src/Frame.cc:753-757
            const int py = (int)roundf((uy - minY) * invH);
            if (px >= 0 && px < cols && py >= 0 && py < rows) key = ((unsigned)(px * rows + py) << 16) | (unsigned)i;
            // mvpMapPoints as the caller hands it over: a keypoint that already has a map point is occupied, and that map
            // point is not searched again (Tracking::SearchLocalPoints: mnLastFrameSeen == frame id).  The host entry point
            // has validated the array: -1 (free), -2 (a map point outside the local map; its position travels in priorXw)
            // or the index of an ACTIVE local map point
            const int pm = A.prior ? A.prior[i] : -1;
            A.kpMp[i] = pm;
            A.occ[i] = pm != -1 ? 1 : 0;
            if (pm >= 0 && pm < nMp) A.mSkip[pm] = 1;
        }
        tkeys[i] = key;
    }
    __syncthreads();
    if (dbg && t == 0) dbg[25] = clock64();
    // ---- The keys (cell << 16 | keypoint) in ascending order = the keypoints cell by cell, by index inside a cell.  Cells hold a keypoint or two
    //      (1100 keypoints over 3072 cells), so a COUNTING sort does it in five barrier phases: histogram of the cells (LDS atomics), exclusive scan
    //      (three cells per thread, DPP wave scan, sixteen wave totals), scatter through per-cell cursors (arrival order), and every cell's few
    //      entries put in index order by the thread that owns the cell.  The column starts of the walk lists ARE the scan.  (The bitonic network
    //      below took 31 k of this workgroup's 46 k cycles for 2048 keys: 51 shuffle steps and 14 LDS steps with two barriers each.)  A cell with
    //      more than 32 keypoints (or a grid beyond the LDS budget) sends the frame through the network instead.
    const int cells = cols * rows;
    bool counted = false;
    const unsigned* sorted = tkeys;      // where the sorted keys end up
    int inGrid = -1;                     // counting sort: the number of keypoints inside the grid (entries of `sorted` beyond it are not keys)
    if (countingSort) {
        unsigned* hist = tkeys + npow2;              // cells + 1 counters, then the scan
        unsigned* outk = hist + cells + 1;           // npow2 sorted keys
        for (int i = t; i < npow2; i += kFrameThreads) { const unsigned key = tkeys[i]; if (key != 0xFFFFFFFFu) atomicAdd(&hist[key >> 16], 1u); }
        __syncthreads();
        // exclusive scan over the cells: thread t owns cells [t * per, t * per + per)
        const int per = (cells + kFrameThreads - 1) / kFrameThreads;
        unsigned loc[8], sum = 0, mx = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) { const int c = t * per + q; loc[q] = (q < per && c < cells) ? hist[c] : 0u; sum += loc[q]; mx = max(mx, loc[q]); }
        unsigned inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = (unsigned)__shfl_up((int)inc, d); if ((t & 63) >= d) inc += o; }
        if ((t & 63) == 63) s_wtot[t >> 6] = (int)inc;
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d));
        if ((t & 63) == 0 && mx) atomicMax(&s_maxc, (int)mx);      // (one LDS atomic per wave, not per thread)
        __syncthreads();
        unsigned base = inc - sum;
        for (int w = 0; w < (t >> 6); w++) base += (unsigned)s_wtot[w];
        if (s_maxc <= 32 && per <= 8) {
            counted = true;
#pragma unroll
            for (int q = 0; q < 8; q++) { const int c = t * per + q; if (q < per && c < cells) { hist[c] = base; base += loc[q]; } }
            if (t == kFrameThreads - 1) hist[cells] = base;      // (the last thread's running sum = the number of keypoints inside the grid)
            __syncthreads();
            // scatter: outk[start(cell) + arrival], the cell's cursor = its start counted up (restored from the next cell's start afterwards)
            for (int i = t; i < npow2; i += kFrameThreads) { const unsigned key = tkeys[i]; if (key != 0xFFFFFFFFu) outk[atomicAdd(&hist[key >> 16], 1u)] = key; }
            __syncthreads();
            // every cell's entries by index (insertion sort of <= 32 keys by the cell's owner); hist[c] now holds the cell's END
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int c = t * per + q;
                if (q < per && c < cells && loc[q] > 1) {
                    const unsigned e = hist[c], b = e - loc[q];
                    for (unsigned a2 = b + 1; a2 < e; a2++) {
                        const unsigned v = outk[a2];
                        unsigned z = a2;
                        while (z > b && outk[z - 1] > v) { outk[z] = outk[z - 1]; z--; }
                        outk[z] = v;
                    }
                }
            }
            __syncthreads();
            inGrid = (int)hist[cells];
            sorted = outk;
        }
    }
    if (counted) {
    } else if (npow2 >= 128 && npow2 <= 2 * kFrameThreads) {
        // Two keys per thread in REGISTERS (elements t and t + npow2 / 2, both exchange with thread t ^ j): the 51 of 66 steps
        // whose partner sits in the same wave are shuffles, the step j = npow2 / 2 is a swap of the thread's own pair, and
        // only the 14 steps with 64 <= j < npow2 / 2 go through LDS and a workgroup barrier (all 66 did: 30 us for 2048 keys).
        const int half = npow2 >> 1;
        const bool own = t < half;
        unsigned a = own ? tkeys[t] : 0xFFFFFFFFu, b = own ? tkeys[t + half] : 0xFFFFFFFFu;
        __syncthreads();
        for (int k = 2; k <= npow2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j == half) {           // k == npow2: ascending
                    const unsigned lo = min(a, b), hi = max(a, b);
                    a = lo; b = hi;
                    continue;
                }
                unsigned pa, pb;
                if (j < 64) {
                    pa = (unsigned)__shfl_xor((int)a, j);
                    pb = (unsigned)__shfl_xor((int)b, j);
                } else {
                    if (own) { tkeys[t] = a; tkeys[t + half] = b; }
                    __syncthreads();
                    pa = own ? tkeys[t ^ j] : 0xFFFFFFFFu;
                    pb = own ? tkeys[(t ^ j) + half] : 0xFFFFFFFFu;
                    __syncthreads();
                }
                const bool lower = (t & j) == 0;
                const bool upA = (t & k) == 0, upB = ((t + half) & k) == 0;
                a = (upA == lower) ? min(a, pa) : max(a, pa);
                b = (upB == lower) ? min(b, pb) : max(b, pb);
            }
        }
        if (own) { tkeys[t] = a; tkeys[t + half] = b; }
        __syncthreads();
    } else
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < npow2; i += kFrameThreads) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned a = tkeys[i], b = tkeys[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { tkeys[i] = b; tkeys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (dbg && t == 0) dbg[26] = clock64();
    int mine = 0;
    for (int i = t; i < npow2; i += kFrameThreads) {
        const unsigned key = (inGrid < 0 || i < inGrid) ? sorted[i] : 0xFFFFFFFFu;
        if (key != 0xFFFFFFFFu) {
            const int cell = (int)(key >> 16);
            A.order[i] = (int)(key & 0xFFFFu);
            A.cellx[i] = (unsigned short)(cell / rows);
            A.celly[i] = (unsigned short)(cell % rows);
            mine++;
        }
    }
    // the number of keypoints inside the grid: the counting sort's scan has it; otherwise one LDS atomic per WAVE (a thousand lanes adding
    // to one word are served one after the other: 4 k cycles)
    if (counted) { if (t == 0) s_cnt = inGrid; }
    else {
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if ((t & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    }
    // first walk-list entry of every grid column (binary search over the sorted keys; keys of keypoints outside the grid are
    // 0xFFFFFFFF and sort behind every cell): a search window then walks its one or two columns, not the whole frame
    for (int cI = t; cI <= cols; cI += kFrameThreads) {
        if (counted) { A.colStart[cI] = cI ? (int)tkeys[npow2 + cI * rows - 1] : 0; continue; }      // the END of the previous column's last cell (the scan, counted up by the scatter)
        const unsigned want = (unsigned)(cI * rows);
        int lo = 0, hi = npow2;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((tkeys[mid] >> 16) < want) lo = mid + 1; else hi = mid; }
        A.colStart[cI] = lo;
    }
    __syncthreads();
    if (t == 0) { A.counts[0] = n; A.counts[1] = s_cnt; A.counts[2] = 0; A.counts[3] = 0; A.counts[4] = 0; }
    if (dbg && t == 0) dbg[27] = clock64();
}

// ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th), src/ORBmatcher.cc:51-81: the window of every point in view

// Upstream walks the map points in index order; a keypoint assigned to an earlier point is skipped by every later one
// (src/ORBmatcher.cc:87-89, 123).  A point's decision -- the two smallest (distance, list position) among its unoccupied
// candidates, the TH_HIGH / ratio tests -- depends only on the occupancy of the candidates up to its second best.  Rounds:
//   1. minq[k] = the smallest UNDECIDED point that lists keypoint k within TH_HIGH (a point can only ever take a keypoint at a distance
//      <= TH_HIGH, so a more distant lister is no competitor: with every lister counted the rounds took 8 passes instead of 4 on the benchmark frame)
//   2. every undecided point m decides from the claims that are final so far; it becomes final iff minq[k] >= m for every
//      keypoint its decision depended on (no earlier undecided point can still take one of them) -- the smallest undecided
//      point always does, and a claim never lands in the list of an earlier undecided point, so final decisions are exactly
//      upstream's
// ONE workgroup; occupancy and minq in LDS.  Converges in a handful of rounds (overlaps are local).
constexpr int kAssignThreads = 1024;
constexpr int kWalk = 4;              // candidate-list entries loaded together by the walks of the assignment step (8: no further gain)
constexpr int kLdsLists = 28 * 1024;  // candidate-list entries the assignment step can keep in LDS (112 KB of the CU's 160)
// ListT: `const unsigned*` into HBM, or into the LDS copy of the whole (compact) candidate array -- every round walks every undecided
// point's list three times, and each walk step was a dependent L2 round trip (8 rounds: 44.6 us for 1000 points; from LDS: see DESIGN.md)
template <int PER, typename ListT>
__device__ __forceinline__ void track_assign_body(int nMp, int cap, const Query* __restrict__ q, ListT lists,
                                                  const int* __restrict__ segStart, const int* __restrict__ segCount,
                                                  const int* __restrict__ oct, unsigned char* occG, float nnratio,
                                                  int* match, int* counts, int allListers, long long* dbg = nullptr) {
    extern __shared__ int asm_[];
    int* minq = asm_;                                           // cap
    // own[k]: who holds keypoint k -- kFree, kPrior (occupied before the search: every point sees it) or the map point that claimed it, which
    // only LATER points see: upstream walks the points in index order, so a claim by point p is invisible to every point before p.  (With a
    // plain occupancy flag a point could only become final when no earlier undecided point listed its keypoint AT ALL -- an earlier point's
    // second-best distance, hence its ratio test, would otherwise see the keypoint vanish.)
    constexpr unsigned short kFree = 0xFFFF, kPrior = 0xFFFE;
    unsigned short* own = reinterpret_cast<unsigned short*>(minq + cap);   // cap
    unsigned char* octL = reinterpret_cast<unsigned char*>(own + cap);     // cap: the keypoints' octaves (read per candidate in every round)
    __shared__ int s_left, s_nm;
    const int t = threadIdx.x;
    // PER = map points per thread: 4, 8 or 16 (local maps of up to 4096 / 8192 / 16384 points)
    int st_[PER], cn_[PER], mine[PER];      // mine: the point's match (written to memory once, after the rounds)
    bool open[PER];
    for (int i = t; i < cap; i += kAssignThreads) { own[i] = occG[i] ? kPrior : kFree; octL[i] = (unsigned char)oct[i]; }
    if (t == 0) s_nm = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int m = t + u * kAssignThreads;
        open[u] = false; st_[u] = 0; cn_[u] = 0; mine[u] = -1;
        if (m < nMp) {
            if (q[m].active && segCount[m] > 0) { open[u] = true; st_[u] = segStart[m]; cn_[u] = segCount[m]; }
        }
    }
    __syncthreads();
    long long ph[4] = {0, 0, 0, 0}, pt0 = dbg ? clock64() : 0;
    auto lap = [&](int k) { if (dbg && t == 0) { const long long now = clock64(); ph[k] += now - pt0; pt0 = now; } };
    const int listerBound = allListers ? 256 : refc::TH_HIGH;      // (debug A/B: EAO_TRACK_ALL_LISTERS=1 counts every lister)
    int tag = 0x1FFFE;          // (the initial fill 0x7FFFFFFF reads as round 0x1FFFF: never the current one)
    constexpr int kTagShift = 14, kPointMask = (1 << kTagShift) - 1;
    static_assert(kPointMask >= 16384 - 1, "map-point indices must fit the key");
    // upstream's scan over the candidates of point m that are free (:83-115) -- and, in the SAME walk, the smallest distance among the
    // candidates an earlier undecided point also lists: the decision is final iff that distance is beyond everything it looked
    // at with an effect (the second best; every candidate when fewer than two are free).  One walk instead of two, and the three
    // LDS reads a candidate needs (occupancy, octave, earliest undecided lister) are independent of each other.
    // Returns whether the decision is final; best = the keypoint the point takes (-1: none).
    auto decide = [&](int m, int st, int cn, int& best) -> bool {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, earlierD = 0x7FFFFFFF;
        for (int k = 0; k < cn; k += kWalk) {
            unsigned itw[kWalk];
#pragma unroll
            for (int j = 0; j < kWalk; j++) itw[j] = k + j < cn ? lists[st + k + j] : 0u;
            unsigned char ol[kWalk];
            unsigned short ow[kWalk];
            int mq[kWalk];
#pragma unroll
            for (int j = 0; j < kWalk; j++) { const int i = (int)(itw[j] & 0xFFFF); ow[j] = own[i]; ol[j] = octL[i]; mq[j] = minq[i]; }
#pragma unroll
            for (int j = 0; j < kWalk; j++) {
                if (k + j >= cn) continue;
                const int i = (int)(itw[j] & 0xFFFF), d = (int)(itw[j] >> 16);
                if ((mq[j] >> kTagShift) == tag && (mq[j] & kPointMask) < m) earlierD = min(earlierD, d);
                if (ow[j] != kFree && (ow[j] == kPrior || (int)ow[j] < m)) continue;      // occupied when point m is reached
                if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = ol[j]; bestIdx = i; }
                else if (d < bestDist2) { bestLevel2 = ol[j]; bestDist2 = d; }
            }
        }
        best = -1;
        if (earlierD <= bestDist2) return false;
        if (bestDist <= refc::TH_HIGH && !(bestLevel == bestLevel2 && bestDist > nnratio * bestDist2)) best = bestIdx;
        return true;
    };
    // minq entries carry the round in their upper bits, counted DOWN: this round's claims are smaller than anything an earlier round left
    // behind, so the array is never reset (a reset was a pass over `cap` entries and a barrier per round); every entry a decision reads
    // has been claimed in the same round (at least by the point that reads it), i.e. its low 14 bits are this round's smallest lister.
    for (int i = t; i < cap; i += kAssignThreads) minq[i] = 0x7FFFFFFF;
    __syncthreads();
    for (int round = 0; round <= nMp; round++, tag--) {      // (every round finalises at least the smallest undecided point: at most nMp rounds)
        if (t == 0) s_left = 0;
        lap(0);
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (open[u]) {
                const int m = t + u * kAssignThreads;
                // (the candidate lists are walked four entries at a time: four independent loads in flight instead of one
                //  load latency per candidate -- these walks were 60 % of the kernel)
                for (int k = 0; k < cn_[u]; k += kWalk) {
                    unsigned itw[kWalk];
#pragma unroll
                    for (int j = 0; j < kWalk; j++) itw[j] = k + j < cn_[u] ? lists[st_[u] + k + j] : 0u;
#pragma unroll
                    for (int j = 0; j < kWalk; j++)
                        if (k + j < cn_[u] && (int)(itw[j] >> 16) <= listerBound) atomicMin(&minq[itw[j] & 0xFFFF], (tag << kTagShift) | m);
                }
            }
        __syncthreads();
        lap(1);
        int claim[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            claim[u] = -1;
            if (!open[u]) continue;
            int best = -1;
            if (!decide(t + u * kAssignThreads, st_[u], cn_[u], best)) continue;        // not final yet
            open[u] = false;
            claim[u] = best;
            mine[u] = best;
        }
        __syncthreads();      // every decision of the round was taken from the same occupancy
        lap(2);
        // (counts by ballot: a thousand lanes adding to one LDS word are served one after the other)
        int wl = 0, wn = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (claim[u] >= 0) own[claim[u]] = (unsigned short)(t + u * kAssignThreads);
            wn += (int)__popcll(__ballot(claim[u] >= 0));
            wl += (int)__popcll(__ballot(open[u]));
        }
        if ((t & 63) == 0) {
            if (wl) atomicAdd(&s_left, wl);
            if (wn) atomicAdd(&s_nm, wn);
        }
        __syncthreads();
        if (dbg && t == 0 && round < 16) dbg[16 + round] = s_left;
        if (s_left == 0) { if (t == 0) counts[5] = round + 1; break; }
        if (s_left <= 64) {
            // ---- The undecided points halve from round to round (373, 182, 96, 52, 27, 12, 5 of 1000 on the benchmark frame) while a round of
            //      the whole workgroup costs its five barriers and the reset of minq whatever is left (8 k cycles): the last 64 points are
            //      handed to ONE wave, a lane each, which runs the SAME rounds with wavefront fences instead of workgroup barriers and resets
            //      only the minq entries its lists name.
            __shared__ int tM[64], tSt[64], tCn[64], tRes[64];
            __shared__ int s_tail;
            __syncthreads();
            if (t == 0) s_tail = 0;
            __syncthreads();
            int slot[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                slot[u] = -1;
                if (open[u]) {
                    const int sl = atomicAdd(&s_tail, 1);
                    slot[u] = sl; tM[sl] = t + u * kAssignThreads; tSt[sl] = st_[u]; tCn[sl] = cn_[u]; tRes[sl] = -1;
                }
            }
            __syncthreads();
            if (t < 64) {
                const int nT = s_tail;
                bool op = t < nT;
                const int m = op ? tM[t] : 0, st = op ? tSt[t] : 0, cn = op ? tCn[t] : 0;
                int res = -1, more = 0;
                auto wave_sync = [] { eao::wave_sync(); };
                while (__any(op)) {
                    more++;
                    tag--;
                    if (op)
                        for (int k = 0; k < cn; k += kWalk) {
                            unsigned itw[kWalk];
#pragma unroll
                            for (int j = 0; j < kWalk; j++) itw[j] = k + j < cn ? lists[st + k + j] : 0u;
#pragma unroll
                            for (int j = 0; j < kWalk; j++)
                                if (k + j < cn && (int)(itw[j] >> 16) <= listerBound) atomicMin(&minq[itw[j] & 0xFFFF], (tag << kTagShift) | m);
                        }
                    wave_sync();
                    int best = -1;
                    const bool fin = op && decide(m, st, cn, best);
                    wave_sync();          // every decision of the round was taken from the same occupancy
                    if (fin) { op = false; res = best; if (best >= 0) own[best] = (unsigned short)m; }
                    wave_sync();
                }
                if (t < nT) tRes[t] = res;
                const int nmw = (int)__popcll(__ballot(res >= 0));
                if (t == 0) { s_nm += nmw; counts[5] = round + 1 + more; }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; u++)
                if (slot[u] >= 0) mine[u] = tRes[slot[u]];
            break;
        }
        __syncthreads();
        lap(3);
    }
    if (dbg && t == 0) { dbg[8] = ph[0]; dbg[9] = ph[1]; dbg[10] = ph[2]; dbg[11] = ph[3]; }
    for (int i = t; i < cap; i += kAssignThreads) occG[i] = own[i] != kFree ? 1 : 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int m = t + u * kAssignThreads;
        if (m < nMp) match[m] = mine[u];
    }
    if (t == 0) counts[3] = s_nm;
}

// mvpMapPoints by keypoint (the prior matches + this search's), and the edges of Optimizer::PoseOptimization in keypoint
// order (src/Optimizer.cc:361-447): Xw, (u, v, uR), invSigma2 of the octave, stereo / robust flags.  ONE workgroup.
struct EdgeArrays { double *Xw, *obs, *info; unsigned char* flags; int* eKp; };
struct ResultBlock { int* counts; int* kpMp; unsigned char* outl; float* ur; float* dz; unsigned char* inView; };      // slices of the host-visible result block
constexpr int kEdgeThreads = 1024;
__device__ __forceinline__ void track_edges_body(int nMp, int cap, const int* match, int* kpMp,
                                                 const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ ur,
                                                 const int* __restrict__ oct, const float* __restrict__ mXw, const float* __restrict__ priorXw,
                                                 const float* __restrict__ invSigma2, const EdgeArrays& E, int edgeCap, int* counts,
                                                 unsigned char* eOutl) {
    __shared__ int s_wsum[kEdgeThreads / 64], s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = counts[0];
    for (int k = t; k < cap; k += kEdgeThreads) eOutl[k] = 0;      // (instead of a fill launch in front of PoseOptimization)
    for (int m = t; m < nMp; m += kEdgeThreads) {
        const int k = match[m];
        if (k >= 0) kpMp[k] = m;          // (a keypoint is claimed by at most one point)
    }
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += kEdgeThreads) {
        const int k = k0 + t;
        const int m = k < n ? kpMp[k] : -1;
        const bool has = m >= 0 || m == -2;       // -2: a prior match outside the local map (an edge all the same, :361-447)
        const unsigned long long bal = __ballot(has);
        if (lane == 0) s_wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; w++) off += s_wsum[w];
        const int e = off + __popcll(bal & ((1ull << lane) - 1));
        if (has && e < edgeCap) {
            const float* X = m >= 0 ? mXw + 3 * m : priorXw + 3 * k;
            E.Xw[3 * e] = X[0]; E.Xw[3 * e + 1] = X[1]; E.Xw[3 * e + 2] = X[2];
            const float u_r = ur[k];
            E.obs[3 * e] = kx[k]; E.obs[3 * e + 1] = ky[k]; E.obs[3 * e + 2] = u_r;
            E.info[e] = invSigma2[oct[k]];
            E.flags[e] = (unsigned char)((!(u_r < 0) ? 1 : 0) | 4);
            E.eKp[e] = k;
        }
        __syncthreads();
        if (t == 0) { int tot = 0; for (int w = 0; w < kEdgeThreads / 64; w++) tot += s_wsum[w]; s_base += tot; }
        __syncthreads();
    }
    if (t == 0) {
        if (s_base > edgeCap) atomicOr(&counts[4], 2);
        counts[2] = min(s_base, edgeCap);
    }
}

// SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches), src/ORBmatcher.cc:159-288, for the chain of TrackReferenceKeyFrame: the features of the keyframe and of the
// frame that fell into the same vocabulary node are compared node by node.  Upstream's order only matters INSIDE a node (a frame keypoint lies in one node, so
// "if(vpMapPointMatches[realIdxF]) continue" :199-200 can only ever see keypoints taken by earlier keyframe features of the same node): a wave owns a node pair
// and walks its keyframe features in order, the lanes share the frame side -- the two smallest (distance, list position) keys of the keypoints still free,
// i.e. the scan's "first of equal distances wins" (:206-215), then "bestDist1<=TH_LOW" and the ratio test (:218-220).
struct BowDev {
    const int4* pairs;               // per common node: {first keyframe entry, count, first frame entry, count} (NULL: not this stage)
    int nPairs;
    const int* kfIdx; const int* fIdx;       // the nodes' index lists, common nodes only
    const unsigned char* valid;      // by keyframe keypoint: its map point exists and is not bad (:183-189)
    const uint4* kfDesc; const uint4* fDesc;
    float nnratio;
};
__device__ __forceinline__ void track_bow_body(int nKf, const BowDev& B, int* match, int* counts) {
    __shared__ int s_nmB;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = counts[0];
    for (int m = t; m < nKf; m += kAssignThreads) match[m] = -1;
    if (t == 0) s_nmB = 0;
    __syncthreads();
    int nm = 0, bad = 0;
    for (int pi = wv; pi < B.nPairs; pi += kAssignThreads / 64) {
        const int4 pr = B.pairs[pi];
        unsigned long long taken = 0;      // bit j: list position lane + 64 j of this node's frame side was matched (<= 4096 keypoints: 64 per lane)
        for (int a = 0; a < pr.y; a++) {
            const int iKF = B.kfIdx[pr.x + a];      // (wave-uniform)
            if (!B.valid[iKF]) continue;
            const uint4 d0 = B.kfDesc[2 * (size_t)iKF], d1 = B.kfDesc[2 * (size_t)iKF + 1];
            unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;      // distance << 16 | list position
            for (int p = lane, j = 0; p < pr.w; p += 64, j++) {
                if ((taken >> j) & 1) continue;
                const int iF = B.fIdx[pr.z + p];
                if (iF >= n) { bad = 1; continue; }
                const uint4 f0 = B.fDesc[2 * (size_t)iF], f1 = B.fDesc[2 * (size_t)iF + 1];
                const unsigned d = __popc(d0.x ^ f0.x) + __popc(d0.y ^ f0.y) + __popc(d0.z ^ f0.z) + __popc(d0.w ^ f0.w) + __popc(d1.x ^ f1.x) + __popc(d1.y ^ f1.y) +
                                   __popc(d1.z ^ f1.z) + __popc(d1.w ^ f1.w);
                const unsigned key = (d << 16) | (unsigned)p;
                if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {      // the wave's two smallest keys (all distinct: the position is part of the key)
                const unsigned o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
                const unsigned lo = min(k1, o1), hi = max(k1, o1);
                k2 = min(hi, min(k2, o2)); k1 = lo;
            }
            const int best1 = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 16), best2 = k2 == 0xFFFFFFFFu ? 256 : (int)(k2 >> 16);
            if (best1 <= refc::TH_LOW && (float)best1 < B.nnratio * (float)best2) {
                const int p = (int)(k1 & 0xFFFFu);
                if ((p & 63) == lane) taken |= 1ull << (p >> 6);
                if (lane == 0) { match[iKF] = B.fIdx[pr.z + p]; nm++; }
            }
        }
    }
    if (lane == 0 && nm) atomicAdd(&s_nmB, nm);
    if (bad) atomicOr(&counts[4], 8);      // a frame index beyond the keypoints the extractor left
    __syncthreads();
    if (t == 0) counts[3] = s_nmB;
}

// ONE launch for the greedy assignment and the edge list behind it (both are single-workgroup steps of 1024 threads; as two
// launches the second one waited a launch gap for the first).  nMp = 0: no search ran, only the prior matches become edges.
static_assert(kAssignThreads == kEdgeThreads, "one workgroup runs both steps");
template <int PER>
__global__ __launch_bounds__(kAssignThreads) void k_track_assign_edges(int nMp, int cap, const Query* __restrict__ q, const unsigned* __restrict__ lists,
                                                                       const int* __restrict__ segStart, const int* __restrict__ segCount, const int* __restrict__ cursor,
                                                                       const int* __restrict__ oct, unsigned char* occG, float nnratio, int* match, int* counts,
                                                                       int* kpMp, const float* __restrict__ kx, const float* __restrict__ ky,
                                                                       const float* __restrict__ ur, const float* __restrict__ mXw, const float* __restrict__ priorXw,
                                                                       const float* __restrict__ invSigma2, EdgeArrays E, int edgeCap, unsigned char* eOutl,
                                                                       const float* __restrict__ dz, const unsigned char* __restrict__ inView, ResultBlock R,
                                                                       long long* dbg, int allListers, const float* __restrict__ mAngle, const float* __restrict__ kAngle,
                                                                       float rotFactor, BowDev bow, int minMatches) {
    extern __shared__ unsigned char asm_raw[];
    if (dbg && threadIdx.x == 0) dbg[0] = clock64();
    if (nMp > 0) {
        if (bow.pairs) track_bow_body(nMp, bow, match, counts);
        else {
            // the candidate lists are ONE compact array (k_match_candidates fills it through an atomic cursor): when it fits, the
            // workgroup copies it into LDS once (coalesced) and the rounds walk it there
            const int total = *cursor;
            unsigned* ldsLists = reinterpret_cast<unsigned*>(asm_raw + ((7 * (size_t)cap + 15) & ~(size_t)15));
            if (total <= kLdsLists) {
                for (int i = threadIdx.x; i < total; i += kAssignThreads) ldsLists[i] = lists[i];
                __syncthreads();
                if (dbg && threadIdx.x == 0) { dbg[1] = clock64(); dbg[7] = total; }
                track_assign_body<PER, const unsigned*>(nMp, cap, q, ldsLists, segStart, segCount, oct, occG, nnratio, match, counts, allListers, dbg);
            } else {
                track_assign_body<PER, const unsigned* __restrict__>(nMp, cap, q, lists, segStart, segCount, oct, occG, nnratio, match, counts, allListers);
            }
        }
        __syncthreads();      // match[] is complete (and visible to the whole workgroup)
        if (rotFactor > 0.f) {
            // rotation consistency of SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1426-1468): the matches vote for the bin of their angle difference,
            // the three fullest bins stay (ComputeThreeMaxima, :1603-1644: strictly greater in bin order, the second / third only from a tenth of the first)
            constexpr int HISTO = refc::HISTO_LENGTH;
            __shared__ int s_hist[HISTO], s_keep[3], s_gone;
            if (threadIdx.x < HISTO) s_hist[threadIdx.x] = 0;
            if (threadIdx.x == 0) s_gone = 0;
            __syncthreads();
            int bins[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int m = threadIdx.x + u * kAssignThreads;
                bins[u] = -1;
                const int k = m < nMp ? match[m] : -1;
                if (k >= 0) {
                    float rot = mAngle[m] - kAngle[k];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * rotFactor);
                    if (bin == HISTO) bin = 0;
                    if (bin >= 0 && bin < HISTO) { bins[u] = bin; atomicAdd(&s_hist[bin], 1); }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int top[3] = {0, 0, 0}, ind[3] = {-1, -1, -1};
                for (int b = 0; b < HISTO; b++) {
                    const int pop = s_hist[b];
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
                s_keep[0] = ind[0]; s_keep[1] = ind[1]; s_keep[2] = ind[2];
            }
            __syncthreads();
            int gone = 0;
#pragma unroll
            for (int u = 0; u < PER; u++)
                if (bins[u] >= 0 && bins[u] != s_keep[0] && bins[u] != s_keep[1] && bins[u] != s_keep[2]) { match[threadIdx.x + u * kAssignThreads] = -1; gone++; }
            if (gone) atomicAdd(&s_gone, gone);
            __syncthreads();
            if (threadIdx.x == 0) counts[3] -= s_gone;
            __syncthreads();
        }
    }
    if (dbg && threadIdx.x == 0) dbg[2] = clock64();
    track_edges_body(nMp, cap, match, kpMp, kx, ky, ur, oct, mXw, priorXw, invSigma2, E, edgeCap, counts, eOutl);
    __syncthreads();          // counts[2], kpMp[] are final
    // A search that returned fewer than min_matches matches (upstream: "if(nmatches<20)" retry / return false, src/Tracking.cc:1756-1763; "if(nmatches<10) return
    // false", :1580-1581) is NOT followed by the pose optimisation: the edge count the pose kernel reads becomes zero, its launch leaves at once
    if (threadIdx.x == 0 && counts[3] < minMatches) counts[2] = 0;
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[3] = clock64();
    // everything the host needs except the pose and mvbOutlier, which PoseOptimization itself writes (by keypoint, through
    // the edge -> keypoint table): the block lies in mapped host memory
    const int t = threadIdx.x, n = counts[0];
    if (t < 8) R.counts[t] = t < 5 ? counts[t] : 0;
    for (int i = t; i < cap; i += kAssignThreads) {
        R.kpMp[i] = i < n ? kpMp[i] : -1;
        R.ur[i] = i < n ? ur[i] : -1.f;
        R.dz[i] = i < n ? dz[i] : -1.f;
        R.outl[i] = 0;
    }
    // Frame::isInFrustum(pMP, 0.5) of every local map point (workgroups 1.. of k_track_frame): the caller's visibility counters
    // (the motion-model / reference-keyframe stages pass inView = nullptr: their nMp counts last-frame / keyframe keypoints, which may exceed the
    //  max_map_points the inView slice is sized for, and the host does not read the table there)
    if (inView) for (int m = t; m < nMp; m += kAssignThreads) R.inView[m] = inView[m];
    if (dbg && threadIdx.x == 0) dbg[4] = clock64();
}

// everything the host needs, in one block: [SE3 | result ints | counts | kpMp | kpOutlier | uRight | depth]

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

eao_status eao_tracker_create(const eao_tracker_cfg* cfg, eao_tracker** out) {
    EAO_REQUIRE(cfg && out && cfg->scale_factors && cfg->inv_level_sigma2, "null argument");
    EAO_REQUIRE(cfg->nlevels >= 1 && cfg->nlevels <= 64 && cfg->grid_cols > 0 && cfg->grid_rows > 0 && (long long)cfg->grid_cols * cfg->grid_rows < 65535, "bad geometry");
    EAO_REQUIRE(cfg->max_keypoints >= 1 && cfg->max_keypoints <= 4096, "max_keypoints must be in 1..4096 (the grid sort runs in the LDS of one workgroup)");
    EAO_REQUIRE(cfg->max_map_points >= 1 && cfg->max_map_points <= 16384, "max_map_points must be in 1..16384 (one assignment workgroup, up to sixteen points per thread)");
    EAO_REQUIRE(cfg->max_x > cfg->min_x && cfg->max_y > cfg->min_y, "empty image bounds");
    eao_status st = eao::require_device();
    if (st) return st;
    eao_tracker* h = new eao_tracker();
    h->cfg = *cfg;
    h->scale.assign(cfg->scale_factors, cfg->scale_factors + cfg->nlevels);
    h->invSigma2.assign(cfg->inv_level_sigma2, cfg->inv_level_sigma2 + cfg->nlevels);
    h->cfg.scale_factors = nullptr; h->cfg.inv_level_sigma2 = nullptr;
    h->cap = cfg->max_keypoints; h->capMp = cfg->max_map_points; h->nCells = cfg->grid_cols * cfg->grid_rows;
    const size_t C = h->cap, M = h->capMp, Q = std::max(C, M);      // Q: either stage's queries (local map points / last-frame keypoints)
    h->capQ = (int)Q;
    h->listCap = std::min<size_t>(C * Q, (size_t)1 << 24);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = al256(off + bytes); return o; };
    const size_t oKx = take(4 * C), oKy = take(4 * C), oAng = take(4 * C), oUr = take(4 * C), oDz = take(4 * C), oOct = take(4 * C), oOrd = take(4 * C),
                 oCx = take(2 * C), oCy = take(2 * C), oCnt = take(64), oPrior = take(4 * C), oPriorX = take(12 * C), oKpMp = take(4 * C), oOcc = take(C), oKpOut = take(C),
                 oMX = take(12 * M), oMN = take(12 * M), oMMin = take(4 * M), oMMax = take(4 * M), oMNum = take(4 * M), oMD = take(32 * M), oMA = take(M),
                 oMS = take(M), oIn = take(M), oPx = take(4 * M), oPy = take(4 * M), oPxr = take(4 * M), oVc = take(4 * M), oLv = take(4 * M),
                 oQ = take(sizeof(Query) * Q), oLists = take(4 * h->listCap), oSS = take(4 * Q), oSC = take(4 * Q), oCur = take(64), oMatch = take(4 * Q),
                 oLQ = take(sizeof(Query) * C), oLX = take(12 * C), oLD = take(32 * C), oLA = take(4 * C), oLS = take(Q),
                 oEX = take(24 * C), oEO = take(24 * C), oEI = take(8 * C), oEE = take(24 * C), oEF = take(C), oEOu = take(C), oEK = take(4 * C),
                 oSc = take(4 * 64), oIs = take(4 * 64), oCol = take(4 * ((size_t)cfg->grid_cols + 1));
    const size_t se3 = al256(eao::lm::pose_se3_bytes());
    h->resBytes = se3 + al256(16) + al256(192 * 8) + al256(32) + al256(eao::lm::kPoseChainMaxPlanes) + al256(4 * C) + al256(C) + al256(4 * C) + al256(4 * C) + al256(Q) + 256;      // (+ the done word)
    const size_t oRes = take(h->resBytes);
    if ((st = h->dev.reserve(off))) { delete h; return st; }
    unsigned char* b = h->dev.p;
    h->resDev = b + oRes;
    EAO_HIP(hipMemset(h->resDev, 0, h->resBytes));
    h->kx = (float*)(b + oKx); h->ky = (float*)(b + oKy); h->ang = (float*)(b + oAng); h->ur = (float*)(b + oUr); h->dz = (float*)(b + oDz);
    h->oct = (int*)(b + oOct); h->order = (int*)(b + oOrd); h->cellx = (unsigned short*)(b + oCx); h->celly = (unsigned short*)(b + oCy);
    h->counts = (int*)(b + oCnt); h->prior = (int*)(b + oPrior); h->priorXw = (float*)(b + oPriorX); h->kpMp = (int*)(b + oKpMp); h->occ = b + oOcc; h->kpOut = b + oKpOut;
    h->mXw = (float*)(b + oMX); h->mNormal = (float*)(b + oMN); h->mMin = (float*)(b + oMMin); h->mMax = (float*)(b + oMMax); h->mNum = (float*)(b + oMNum);
    h->mDesc = b + oMD; h->mActive = b + oMA; h->mSkip = b + oMS;
    h->inView = b + oIn; h->projX = (float*)(b + oPx); h->projY = (float*)(b + oPy); h->projXR = (float*)(b + oPxr); h->viewCos = (float*)(b + oVc);
    h->level = (int*)(b + oLv); h->q = (Query*)(b + oQ); h->lists = (unsigned*)(b + oLists); h->segStart = (int*)(b + oSS); h->segCount = (int*)(b + oSC);
    h->cursor = (int*)(b + oCur); h->match = (int*)(b + oMatch);
    h->eXw = (double*)(b + oEX); h->eObs = (double*)(b + oEO); h->eInfo = (double*)(b + oEI); h->eErr = (double*)(b + oEE); h->eFlags = b + oEF;
    h->eOutl = b + oEOu; h->eKp = (int*)(b + oEK); h->dScale = (float*)(b + oSc); h->dInvSigma2 = (float*)(b + oIs);
    h->colStart = (int*)(b + oCol);
    h->lQ = (Query*)(b + oLQ); h->lXw = (float*)(b + oLX); h->lDesc = b + oLD; h->lAng = (float*)(b + oLA); h->lSkip = b + oLS;
    // the result block is MAPPED PINNED HOST memory: the LAST kernel of the chain copies the device twin into it over PCIe (~30 KB) and the host
    // reads it after the one synchronisation -- no device-to-host copy behind the chain
    if (hipHostMalloc((void**)&h->resPin, h->resBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { delete h; eao::set_error("pinned allocation failed"); return EAO_ERR_NO_DEVICE; }
    std::memset(h->resPin, 0, h->resBytes);
    if (hipHostMalloc((void**)&h->plPin, 10 * sizeof(double) * eao::lm::kPoseChainMaxPlanes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->plDev, h->plPin, 0) != hipSuccess) { delete h; eao::set_error("pinned allocation failed"); return EAO_ERR_NO_DEVICE; }
    if (hipHostGetDevicePointer((void**)&h->res, h->resPin, 0) != hipSuccess) { delete h; eao::set_error("hipHostGetDevicePointer failed"); return EAO_ERR_NO_DEVICE; }
    if (eao::create_stream(&h->stream, eao::StreamClass::Latency) != hipSuccess || hipEventCreateWithFlags(&h->evIn, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evOut, hipEventDisableTiming) != hipSuccess) { delete h; eao::set_error("stream / event creation failed"); return EAO_ERR_NO_DEVICE; }
    h->pinCap = std::max(h->resBytes, al256(12 * M) * 2 + al256(4 * M) * 3 + al256(32 * M) + al256(M) + al256(4 * C) + al256(12 * C) +
                                      al256(sizeof(Query) * C) + al256(32 * C) + al256(4 * C)) + 4096;
    if (hipHostMalloc((void**)&h->pin, h->pinCap, hipHostMallocDefault) != hipSuccess) { delete h; eao::set_error("pinned allocation failed"); return EAO_ERR_NO_DEVICE; }
    // the assignment workgroup's LDS: claims / occupancy / octaves by keypoint, then the staged candidate lists (beyond the default 64 KB)
    if (getenv("EAO_DEBUG_STAMPS")) { EAO_HIP(hipMalloc((void**)&h->dbg, 256)); EAO_HIP(hipMemset(h->dbg, 0, 256)); }
    h->assignLds = ((7 * C + 15) & ~(size_t)15) + 4 * (size_t)kLdsLists;
    EAO_HIP(hipFuncSetAttribute((const void*)k_track_assign_edges<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->assignLds));
    EAO_HIP(hipFuncSetAttribute((const void*)k_track_assign_edges<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->assignLds));
    EAO_HIP(hipFuncSetAttribute((const void*)k_track_assign_edges<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->assignLds));
    EAO_HIP(hipMemcpyAsync(h->dScale, h->scale.data(), 4 * (size_t)cfg->nlevels, hipMemcpyHostToDevice, h->stream));
    EAO_HIP(hipMemcpyAsync(h->dInvSigma2, h->invSigma2.data(), 4 * (size_t)cfg->nlevels, hipMemcpyHostToDevice, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    *out = h;
    return EAO_OK;
}

void eao_tracker_destroy(eao_tracker* h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
}

eao_status eao_tracker_set_local_map(eao_tracker* h, const eao_map_points* pts) {
    EAO_REQUIRE(h && pts && pts->n >= 0 && pts->n <= h->capMp, "bad argument (at most %d map points)", h ? h->capMp : 0);
    const int n = pts->n;
    h->nMp = n;
    h->hActive.clear();
    if (n == 0) return EAO_OK;
    EAO_REQUIRE(pts->active && pts->Xw && pts->normal && pts->min_dist_inv && pts->max_dist_inv && pts->max_dist && pts->desc, "incomplete map-point arrays");
    const size_t M = n;
    h->hActive.assign(pts->active, pts->active + n);
    size_t off = 0;
    auto put = [&](void* dst, const void* src, size_t bytes) -> eao_status {
        std::memcpy(h->pin + off, src, bytes);
        EAO_HIP(hipMemcpyAsync(dst, h->pin + off, bytes, hipMemcpyHostToDevice, h->stream));
        off = al256(off + bytes);
        return EAO_OK;
    };
    eao_status st;
    EAO_HIP(eao::wait_latency(h->stream));      // the staging block may still feed the previous upload
    if ((st = put(h->mXw, pts->Xw, 12 * M)) || (st = put(h->mNormal, pts->normal, 12 * M)) || (st = put(h->mMin, pts->min_dist_inv, 4 * M)) ||
        (st = put(h->mMax, pts->max_dist_inv, 4 * M)) || (st = put(h->mNum, pts->max_dist, 4 * M)) || (st = put(h->mDesc, pts->desc, 32 * M)) ||
        (st = put(h->mActive, pts->active, M))) return st;
    EAO_HIP(eao::wait_latency(h->stream));
    return EAO_OK;
}

}  // extern "C"

namespace {
// TrackWithMotionModel's inputs (NULL: TrackLocalMap over the uploaded local map): the last frame's map points as plain host arrays
struct MotionArgs {
    const float* Tcw_last; int n_last; const uint8_t* valid; const float* Xw; const uint8_t* mp_desc; const int32_t* octave; const float* angle;
    int mono, check_orientation;
};

// TrackReferenceKeyFrame's inputs: the reference keyframe's keypoints (map point present and good, position, descriptor, mvKeysUn angle) and the two
// DBoW2 feature vectors (the frame's one comes from Frame::ComputeBoW on the host: the vocabulary tree is not part of this library)
struct BowArgs {
    int n_kf; const uint8_t* valid; const float* Xw; const uint8_t* kf_desc; const float* kf_angle;
    const eao_feature_vector* fv_kf; const eao_feature_vector* fv_cur; int check_orientation;
};

eao_status check_feature_vector(const eao_feature_vector* fv, int n, const char* who) {
    EAO_REQUIRE(fv && fv->n_nodes >= 0, "%s: null feature vector", who);
    EAO_REQUIRE(fv->n_nodes == 0 || (fv->node_id && fv->node_start && fv->index), "%s: null feature-vector arrays", who);
    if (fv->n_nodes) EAO_REQUIRE(fv->node_start[0] == 0, "%s: node_start[0] must be 0", who);
    for (int i = 0; i < fv->n_nodes; i++) {
        EAO_REQUIRE(i == 0 || fv->node_id[i] > fv->node_id[i - 1], "%s: node ids must ascend strictly (node %d)", who, i);
        EAO_REQUIRE(fv->node_start[i + 1] >= fv->node_start[i], "%s: node_start must not decrease (node %d)", who, i);
    }
    EAO_REQUIRE(fv->n_nodes == 0 || fv->node_start[fv->n_nodes] <= n, "%s: %d indices for %d keypoints (a keypoint lies in ONE node)", who, fv->n_nodes ? fv->node_start[fv->n_nodes] : 0, n);
    return EAO_OK;
}

eao_status track_chain(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                       const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_prior,
                       const int32_t* prior_kp_map_point, const float* prior_kp_Xw, float th, float nnratio, eao_track_result* out,
                       void* stream, const MotionArgs* mm, const BowArgs* bw = nullptr) {
    EAO_REQUIRE(h && d_kps && d_desc && d_n && Tcw_prior && out && out->kp_map_point && out->kp_outlier, "null argument");
    for (int i = 0; i < 16; i++) EAO_REQUIRE(std::isfinite(Tcw_prior[i]), "the pose prior holds a NaN / Inf (entry %d)", i);
    EAO_REQUIRE(!d_depth || (depth_pitch >= width && width > 0 && height > 0), "bad depth image geometry");
    EAO_REQUIRE(((uintptr_t)d_desc & 15) == 0, "descriptors must be 16-byte aligned");
    eao::Range rg("track: frame -> frustum -> search -> pose");
    const eao_tracker_cfg& c = h->cfg;
    const int C = h->cap, nMp = bw ? bw->n_kf : mm ? mm->n_last : h->nMp;      // the stage's queries: keyframe keypoints / last-frame keypoints / local map points
    const bool plain = !mm && !bw;
    // the one-shot options of eao_tracker_set_options
    const int minMatches = plain ? 0 : h->optMinMatches, nPlanes = h->optPlanes;
    unsigned char* planeOut = h->optPlaneOut;
    h->optMinMatches = 0; h->optPlanes = 0; h->optPlaneOut = nullptr;
    // The chain runs on the CALLER's stream itself: it is ordered behind whatever produced the inputs there (the extraction)
    // without an event hand-over to a private stream and back (~10 us each on this runtime).  The handle's own stream only
    // carries the local-map uploads, which eao_tracker_set_local_map waits for.
    hipStream_t s = (hipStream_t)stream;
    int nPrior = 0;
    if (prior_kp_map_point) {
        // Validated HERE: the kernels index the local map with these values.  An index beyond the uploaded map (e.g. a stale table
        // from before a smaller eao_tracker_set_local_map) is an error; a prior on an INACTIVE point is dropped -- the keypoint is free
        // again, as upstream's "if(pMP->isBad()) *vit = NULL" (src/Tracking.cc:2596-2599); -2 = the keypoint's map point is not in the
        // local map: it stays occupied and becomes a pose edge from prior_kp_Xw (PoseOptimization takes every mvpMapPoints entry)
        int32_t* pr = reinterpret_cast<int32_t*>(h->pin);
        bool outside = false;
        for (int k = 0; k < C; k++) {
            int32_t pm = prior_kp_map_point[k];
            EAO_REQUIRE(pm >= -2 && pm < nMp, "prior_kp_map_point[%d] = %d names no point of the local map (%d points)", k, (int)pm, nMp);
            if (pm >= 0 && !h->hActive[pm]) pm = -1;
            outside = outside || pm == -2;
            nPrior += pm != -1;
            pr[k] = pm;
        }
        EAO_REQUIRE(!outside || prior_kp_Xw, "prior_kp_map_point holds -2 but prior_kp_Xw is NULL");
        EAO_HIP(hipMemcpyAsync(h->prior, h->pin, 4 * (size_t)C, hipMemcpyHostToDevice, s));
        if (outside) {
            unsigned char* px = h->pin + al256(4 * (size_t)C);
            std::memcpy(px, prior_kp_Xw, 12 * (size_t)C);
            EAO_HIP(hipMemcpyAsync(h->priorXw, px, 12 * (size_t)C, hipMemcpyHostToDevice, s));
        }
    }
    if (mm) {
        // ---- the last frame's points: windows on the host (a thousand projections), then ONE staging block -> four asynchronous copies on the caller's
        //      stream (the staging block is free: every call ends with its results on the host)
        EAO_REQUIRE(mm->n_last >= 0 && mm->n_last <= C, "at most max_keypoints (%d) last-frame keypoints", C);
        EAO_REQUIRE(mm->n_last == 0 || (mm->valid && mm->Xw && mm->mp_desc && mm->octave && mm->angle && mm->Tcw_last), "null last-frame arrays");
        if (nMp > 0) {
            for (int i = 0; i < 16; i++) EAO_REQUIRE(std::isfinite(mm->Tcw_last[i]), "the last frame's pose holds a NaN / Inf (entry %d)", i);
            unsigned char* pq = h->pin; Query* hq = reinterpret_cast<Query*>(pq);
            eao::match::FrameQueryArgs QA{Tcw_prior, mm->Tcw_last, nMp, mm->valid, mm->Xw, mm->octave, c.fx, c.fy, c.cx, c.cy, c.mbf, c.mbf / c.fx, th, mm->mono,
                                          c.min_x, c.max_x, c.min_y, c.max_y, h->scale.data(), c.nlevels};
            eao_status qs = eao::match::build_frame_queries(QA, hq);
            if (qs) return qs;
            // the staging block mirrors the device block lQ | lXw | lDesc | lAng (offsets by CAPACITY, as eao_tracker_create laid them out): one copy
            const size_t C2 = (size_t)C, oX = al256(sizeof(Query) * C2), oD = oX + al256(12 * C2), oA = oD + al256(32 * C2);
            std::memcpy(h->pin + oX, mm->Xw, 12 * (size_t)nMp); std::memcpy(h->pin + oD, mm->mp_desc, 32 * (size_t)nMp); std::memcpy(h->pin + oA, mm->angle, 4 * (size_t)nMp);
            EAO_HIP(hipMemcpyAsync(h->lQ, h->pin, oA + 4 * (size_t)nMp, hipMemcpyHostToDevice, (hipStream_t)stream));
        }
    }
    BowDev BD;
    std::memset(&BD, 0, sizeof(BD));
    if (bw) {
        // ---- the node pairs both feature vectors hold (the merge of :176-263), their index lists, the keyframe's arrays: ONE staging block, one copy.  The tables
        //      take the place of the motion-model stage's windows (25 bytes per keypoint at most against the 32 of a window).
        EAO_REQUIRE(bw->n_kf >= 0 && bw->n_kf <= C, "at most max_keypoints (%d) keyframe keypoints", C);
        EAO_REQUIRE(bw->n_kf == 0 || (bw->valid && bw->Xw && bw->kf_desc && bw->kf_angle), "null keyframe arrays");
        eao_status fs;
        if ((fs = check_feature_vector(bw->fv_kf, bw->n_kf, "keyframe")) || (fs = check_feature_vector(bw->fv_cur, C, "frame"))) return fs;
        if (nMp > 0) {
            const eao_feature_vector &K = *bw->fv_kf, &F = *bw->fv_cur;
            int nPairs = 0, nKI = 0, nFI = 0;
            for (int a = 0, b = 0; a < K.n_nodes && b < F.n_nodes;) {
                if (K.node_id[a] == F.node_id[b]) { nPairs++; nKI += K.node_start[a + 1] - K.node_start[a]; nFI += F.node_start[b + 1] - F.node_start[b]; a++; b++; }
                else if (K.node_id[a] < F.node_id[b]) a++; else b++;
            }
            const size_t oKI = 16 * (size_t)nPairs, oFI = oKI + 4 * (size_t)nKI, oV = oFI + 4 * (size_t)nFI;
            EAO_REQUIRE(oV + (size_t)C <= sizeof(Query) * (size_t)C, "feature vectors beyond the tracker's capacity");
            int* hp = reinterpret_cast<int*>(h->pin);
            int* hk = reinterpret_cast<int*>(h->pin + oKI);
            int* hf = reinterpret_cast<int*>(h->pin + oFI);
            unsigned char* seen = h->pin + oV;      // (first as the "index already listed" table of either side, then the valid flags)
            std::memset(seen, 0, (size_t)C);
            int kk = 0, ff = 0, pp = 0;
            for (int a = 0, b = 0; a < K.n_nodes && b < F.n_nodes;) {
                if (K.node_id[a] == F.node_id[b]) {
                    hp[4 * pp] = kk; hp[4 * pp + 1] = K.node_start[a + 1] - K.node_start[a]; hp[4 * pp + 2] = ff; hp[4 * pp + 3] = F.node_start[b + 1] - F.node_start[b]; pp++;
                    for (int i = K.node_start[a]; i < K.node_start[a + 1]; i++) {
                        const uint32_t v = K.index[i];
                        EAO_REQUIRE(v < (uint32_t)nMp, "keyframe feature-vector index %u beyond its %d keypoints", v, nMp);
                        EAO_REQUIRE(!(seen[v] & 1), "keyframe keypoint %u is listed twice", v);
                        seen[v] |= 1; hk[kk++] = (int)v;
                    }
                    for (int i = F.node_start[b]; i < F.node_start[b + 1]; i++) {
                        const uint32_t v = F.index[i];
                        EAO_REQUIRE(v < (uint32_t)C, "frame feature-vector index %u beyond max_keypoints", v);
                        EAO_REQUIRE(!(seen[v] & 2), "frame keypoint %u is listed twice", v);
                        seen[v] |= 2; hf[ff++] = (int)v;
                    }
                    a++; b++;
                } else if (K.node_id[a] < F.node_id[b]) a++; else b++;
            }
            std::memcpy(h->pin + oV, bw->valid, (size_t)nMp);
            const size_t C2 = (size_t)C, oX = al256(sizeof(Query) * C2), oD = oX + al256(12 * C2), oA = oD + al256(32 * C2);
            std::memcpy(h->pin + oX, bw->Xw, 12 * (size_t)nMp); std::memcpy(h->pin + oD, bw->kf_desc, 32 * (size_t)nMp); std::memcpy(h->pin + oA, bw->kf_angle, 4 * (size_t)nMp);
            EAO_HIP(hipMemcpyAsync(h->lQ, h->pin, oA + 4 * (size_t)nMp, hipMemcpyHostToDevice, s));
            unsigned char* dq = reinterpret_cast<unsigned char*>(h->lQ);
            BD.pairs = reinterpret_cast<const int4*>(dq); BD.nPairs = nPairs; BD.kfIdx = reinterpret_cast<const int*>(dq + oKI); BD.fIdx = reinterpret_cast<const int*>(dq + oFI);
            BD.valid = dq + oV; BD.kfDesc = reinterpret_cast<const uint4*>(h->lDesc); BD.fDesc = reinterpret_cast<const uint4*>(d_desc); BD.nnratio = nnratio;
        }
    }
    FrameArrays A;
    A.kx = h->kx; A.ky = h->ky; A.ang = h->ang; A.ur = h->ur; A.dz = h->dz; A.oct = h->oct; A.order = h->order; A.cellx = h->cellx; A.celly = h->celly;
    A.counts = h->counts; A.prior = prior_kp_map_point ? h->prior : nullptr; A.kpMp = h->kpMp; A.occ = h->occ; A.mSkip = plain ? h->mSkip : h->lSkip; A.cursor = h->cursor; A.colStart = h->colStart;
    int npow2 = 64;
    while (npow2 < C) npow2 <<= 1;
    const float invW = (float)c.grid_cols / (c.max_x - c.min_x), invH = (float)c.grid_rows / (c.max_y - c.min_y);   // src/Frame.cc:258-259
    eao::frame::FrustumArgs FA;
    std::memset(&FA, 0, sizeof(FA));
    if (nMp > 0 && plain) {
        eao::frame::FrustumDevArgs F;
        F.n = nMp; F.Xw = h->mXw; F.normal = h->mNormal; F.minDist = h->mMin; F.maxDist = h->mMax; F.maxDistNum = h->mNum;
        std::memcpy(F.Tcw, Tcw_prior, 64);
        for (int i = 0; i < 3; i++) {   // mOw = -Rcw^T tcw (Frame::UpdatePoseMatrices, src/Frame.cc:630-636): float matrices, double accumulation
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)Tcw_prior[4 * k + i] * (double)Tcw_prior[4 * k + 3];
            F.Ow[i] = (float)(-acc);
        }
        F.fx = c.fx; F.fy = c.fy; F.cx = c.cx; F.cy = c.cy; F.mbf = c.mbf; F.minX = c.min_x; F.maxX = c.max_x; F.minY = c.min_y; F.maxY = c.max_y;
        F.logScale = c.log_scale_factor; F.cosLimit = 0.5f;          // Tracking::SearchLocalPoints: isInFrustum(pMP, 0.5)
        F.inView = h->inView; F.projX = h->projX; F.projY = h->projY; F.projXR = h->projXR; F.viewCos = h->viewCos; F.level = h->level;
        eao::frame::fill_frustum_args(F, FA);
    }
    static const int envCount = getenv("EAO_TRACK_COUNTING_SORT") ? atoi(getenv("EAO_TRACK_COUNTING_SORT")) : 1;      // (A/B switch)
    const size_t cells = (size_t)c.grid_cols * c.grid_rows, countLds = (2 * (size_t)npow2 + cells + 1) * 4;
    const int countingSort = envCount && cells <= 8 * kFrameThreads && countLds <= 60 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(k_track_frame, dim3(1 + (nMp > 0 && plain ? eao::cdiv(nMp, kFrameThreads) : 0)), dim3(kFrameThreads), countingSort ? countLds : (size_t)npow2 * 4, s, d_kps, d_n, C, d_depth,
                       depth_pitch, width, height, c.mbf, c.min_x, c.min_y, invW, invH, c.grid_cols, c.grid_rows, npow2, nMp, A, FA, h->dbg, countingSort, h->dist);
    eao_status st;
    if (nMp > 0 && !bw) {
        // the search windows are built by the candidate kernel itself (one wave per map point), which leaves them in h->q for the assignment
        eao::match::QueryBuild QB;
        QB.active = h->mActive; QB.skip = h->mSkip; QB.inView = h->inView; QB.projX = h->projX; QB.projY = h->projY; QB.projXR = h->projXR;
        QB.viewCos = h->viewCos; QB.level = h->level; QB.scale = h->dScale; QB.nlevels = c.nlevels; QB.th = th; QB.errFlags = h->counts + 4; QB.qOut = h->q;
        eao::match::FrameDevArgs FD;
        FD.cap = C; FD.nOrdered = h->counts + 1; FD.kx = h->kx; FD.ky = h->ky; FD.oct = h->oct; FD.ur = h->ur; FD.desc = d_desc;
        FD.order = h->order; FD.cellx = h->cellx; FD.celly = h->celly; FD.colStart = h->colStart;
        FD.minX = c.min_x; FD.minY = c.min_y; FD.invW = invW; FD.invH = invH; FD.cols = c.grid_cols; FD.rows = c.grid_rows;
        if ((st = eao::match::enqueue_candidates_device(FD, mm ? h->lQ : h->q, mm ? h->lDesc : h->mDesc, nMp, h->lists, (int)std::min(h->listCap, (size_t)0x7FFFFFFF), h->segStart,
                                                        h->segCount, h->cursor, s, true, mm ? nullptr : &QB))) return st;
    }
    EdgeArrays E;
    E.Xw = h->eXw; E.obs = h->eObs; E.info = h->eInfo; E.flags = h->eFlags; E.eKp = h->eKp;
    const int edgeCap = std::min(C, 2048);
    // result block layout
    // (kernels write the DEVICE twin; the chain's last launch publishes it: see PoseDev::pubSrc, csrc/lm.hip)
    const size_t se3 = al256(eao::lm::pose_se3_bytes());
    unsigned char* r = h->resDev;
    size_t ro = 0;
    void* rSE3 = r + ro; ro += se3;
    int* rRes = (int*)(r + ro); ro += al256(16);
    double* rTrace = (double*)(r + ro); ro += al256(192 * 8);
    int* rCounts = (int*)(r + ro); ro += al256(32);
    unsigned char* rPlOut = r + ro; ro += al256(eao::lm::kPoseChainMaxPlanes);      // mvbPlaneOutlier of the frame's plane edges (eao_tracker_set_options)
    int* rKpMp = (int*)(r + ro); ro += al256(4 * (size_t)C);
    unsigned char* rOutl = r + ro; ro += al256(C);
    float* rUr = (float*)(r + ro); ro += al256(4 * (size_t)C);
    float* rDz = (float*)(r + ro); ro += al256(4 * (size_t)C);
    unsigned char* rInView = r + ro;
    const size_t pubBytes = (ro + (size_t)std::max(nMp, 0) + 15) & ~(size_t)15;      // everything up to the last in-view flag
    ro += al256((size_t)h->capQ);
    const size_t oDone = ro; ro += 256;
    int* rDone = (int*)(h->res + oDone);                                            // the done word itself only exists in host memory
    ResultBlock RB{rCounts, rKpMp, rOutl, rUr, rDz, rInView};
    static const int envAll = getenv("EAO_TRACK_ALL_LISTERS") ? atoi(getenv("EAO_TRACK_ALL_LISTERS")) : 0;
    auto launch_assign = [&](auto kern) {
        // the rotation histogram's factor: this fork's HISTO_LENGTH / 360 in SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1337), 1 / HISTO_LENGTH in SearchByBoW (:171)
        const float rotFactor = mm && mm->check_orientation ? (float)refc::HISTO_LENGTH / 360.0f : bw && bw->check_orientation ? 1.0f / refc::HISTO_LENGTH : 0.f;
        hipLaunchKernelGGL(kern, dim3(1), dim3(kAssignThreads), h->assignLds, s, nMp, C, plain ? h->q : h->lQ, h->lists, h->segStart, h->segCount, h->cursor, h->oct, h->occ,
                           mm ? INFINITY : nnratio, h->match, h->counts, h->kpMp, h->kx, h->ky, h->ur, plain ? h->mXw : h->lXw, h->priorXw, h->dInvSigma2, E, edgeCap, h->eOutl, h->dz,
                           plain ? h->inView : nullptr, RB, h->dbg, envAll, h->lAng, h->ang, rotFactor, BD, minMatches);
    };
    if (nMp <= 4 * kAssignThreads) launch_assign(k_track_assign_edges<4>);
    else if (nMp <= 8 * kAssignThreads) launch_assign(k_track_assign_edges<8>);
    else launch_assign(k_track_assign_edges<16>);
    eao::lm::PoseChainArgs PA;
    PA.nEdges = h->counts + 2; PA.cap = edgeCap;
    PA.maxEdges = nMp + nPrior;        // every edge is a prior match or a new match, and new matches go to distinct local map points
    PA.Xw = h->eXw; PA.obs = h->eObs; PA.info = h->eInfo; PA.flags = h->eFlags; PA.err = h->eErr; PA.outlier = h->eOutl;
    std::memcpy(PA.Tcw0, Tcw_prior, 64);
    PA.fx = c.fx; PA.fy = c.fy; PA.cx = c.cx; PA.cy = c.cy; PA.bf = c.mbf;
    PA.outSE3 = rSE3; PA.outResult = rRes; PA.outTrace = rTrace;
    PA.scatterIdx = h->eKp; PA.scatterOut = rOutl;      // mvbOutlier by keypoint, straight into the result block
    PA.nPlanes = nPlanes; PA.planes = nPlanes ? h->plDev : nullptr; PA.planeOutlier = nPlanes ? rPlOut : nullptr;
    // The results are in mapped host memory when the chain's last launch has stored this call's number in the done word: that launch is the
    // ONLY writer of the host block -- it copies the device twin out, every thread fences at system scope, and behind a barrier one thread
    // stores the word (round 4; until then two kernels wrote the host block and the host trusted the earlier kernel's posted writes to have
    // landed when the later kernel's word arrived: one wrong frame in ~10^4).  The host polls the word instead of
    // waiting for the runtime to notice the end of the stream; EAO_TRACK_POLL=0 goes back to hipStreamSynchronize, which is also what a
    // call falls back to after 50 ms without the word.
    static const int envPoll = getenv("EAO_TRACK_POLL") ? atoi(getenv("EAO_TRACK_POLL")) : 1;
    const int seq = ++h->seq;
    *reinterpret_cast<volatile int*>(h->resPin + oDone) = 0;
    PA.done = rDone; PA.doneSeq = seq; PA.pubSrc = h->resDev; PA.pubDst = h->res; PA.pubN16 = (int)(pubBytes / 16);
    if ((st = eao::lm::enqueue_pose_device(PA, s))) return st;
    // ---- the results are in host memory when the done word says so (or when the stream has drained)
    bool seen = false;
    eao::note_latency_call();
    if (envPoll) {
        const volatile int* done = reinterpret_cast<const volatile int*>(h->resPin + oDone);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !(seen = *done == seq); spins++)
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    if (h->dbg) {
        long long st[32];
        EAO_HIP(hipMemcpy(st, h->dbg, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[eao track stamps] frame set-up (workgroup 0): records + depth + keys %lld, sort %lld, order / cells / column starts %lld ticks\n", st[25] - st[24], st[26] - st[25], st[27] - st[26]);
        fprintf(stderr, "[eao track stamps]   points still undecided after each round:");
        for (int k = 0; k < 16 && (k == 0 || st[16 + k - 1] > 0); k++) fprintf(stderr, " %lld", st[16 + k]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[eao track stamps]   rounds: init + reset %lld, claim walk %lld, decide %lld, commit %lld ticks\n", st[8], st[9], st[10], st[11]);
        int rounds = 0;
        EAO_HIP(hipMemcpy(&rounds, h->counts + 5, sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "[eao track stamps] assign + edges: stage lists %lld (%lld entries), %d rounds %lld, edge list %lld, result block %lld ticks\n", st[1] - st[0], st[7], rounds,
                st[2] - st[1], st[3] - st[2], st[4] - st[3]);
    }
    const unsigned char* p = h->resPin;
    const int* oc = (const int*)(p + ((unsigned char*)rCounts - r));
    const int n = oc[0], nEdges = oc[2];
    EAO_REQUIRE(!(oc[4] & 1), "a map point in view has a predicted level outside the pyramid (upstream would index mvScaleFactors out of range)");
    EAO_REQUIRE(!(oc[4] & 2), "more than %d correspondences: beyond the chained PoseOptimization's capacity", edgeCap);
    EAO_REQUIRE(!(oc[4] & 8), "the frame's feature vector names a keypoint beyond the %d the extractor left", oc[0]);
    out->n_keypoints = n; out->n_matches = oc[3]; out->n_edges = nEdges;
    std::memcpy(out->kp_map_point, p + ((unsigned char*)rKpMp - r), 4 * (size_t)C);
    if (out->kp_u_right) std::memcpy(out->kp_u_right, p + ((unsigned char*)rUr - r), 4 * (size_t)C);
    if (out->kp_depth) std::memcpy(out->kp_depth, p + ((unsigned char*)rDz - r), 4 * (size_t)C);
    if (out->map_in_view && nMp > 0 && plain) {
        const unsigned char* iv = p + ((unsigned char*)rInView - r);
        for (int m = 0; m < nMp; m++) out->map_in_view[m] = (iv[m] && h->hActive[m]) ? 1 : 0;      // (an inactive point's arrays are stale)
    }
    if (nEdges < 3) {   // "if(nInitialCorrespondences<3) return 0" (src/Optimizer.cc:453-454): pose untouched -- also the exit of a search below min_matches
        std::memcpy(out->Tcw, Tcw_prior, 64);
        out->n_inliers = 0;
        std::memset(out->kp_outlier, 0, C);
        if (planeOut) std::memset(planeOut, 0, nPlanes);      // (upstream resets mvbPlaneOutlier only behind that test: untouched there, cleared here)
    } else {
        if (planeOut) std::memcpy(planeOut, p + (rPlOut - r), nPlanes);
        eao::lm::pose_se3_to_Tcw(p, out->Tcw);
        const int* rr = (const int*)(p + ((unsigned char*)rRes - r));
        out->n_inliers = nEdges + nPlanes - rr[0];      // nInitialCorrespondences counts the plane edges too (src/Optimizer.cc:456-535, 672)
        std::memcpy(out->kp_outlier, p + ((unsigned char*)rOutl - r), C);
    }
    return EAO_OK;
}
}  // namespace

extern "C" {

eao_status eao_tracker_set_options(eao_tracker* h, const eao_track_options* opt) {
    EAO_REQUIRE(h, "null handle");
    h->optMinMatches = 0; h->optPlanes = 0; h->optPlaneOut = nullptr;
    if (!opt) return EAO_OK;
    EAO_REQUIRE(opt->min_matches >= 0 && opt->n_planes >= 0 && opt->n_planes <= eao::lm::kPoseChainMaxPlanes, "min_matches >= 0, at most %d plane edges", eao::lm::kPoseChainMaxPlanes);
    EAO_REQUIRE(opt->n_planes == 0 || (opt->plane_world && opt->plane_obs && opt->plane_seen && opt->plane_outlier), "plane arrays missing");
    for (int i = 0; i < 4 * opt->n_planes; i++) EAO_REQUIRE(std::isfinite(opt->plane_world[i]) && std::isfinite(opt->plane_obs[i]), "a plane coefficient is NaN / Inf");
    h->optMinMatches = opt->min_matches;
    h->optPlanes = opt->n_planes; h->optPlaneOut = opt->plane_outlier;
    if (opt->n_planes) eao::lm::pose_plane_records(opt->n_planes, opt->plane_world, opt->plane_obs, opt->plane_seen, h->plPin);
    return EAO_OK;
}

eao_status eao_tracker_set_distortion(eao_tracker* h, const float* dist_coef, int32_t n_coef) {
    EAO_REQUIRE(h && n_coef >= 0 && n_coef <= 5 && (n_coef == 0 || dist_coef), "bad argument");
    for (int i = 0; i < n_coef; i++) EAO_REQUIRE(std::isfinite(dist_coef[i]), "a distortion coefficient is NaN / Inf");
    eao::frame::fill_distortion(h->dist, h->cfg.fx, h->cfg.fy, h->cfg.cx, h->cfg.cy, dist_coef, n_coef);
    return EAO_OK;
}

eao_status eao_tracker_track_local_map(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                       const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_prior,
                                       const int32_t* prior_kp_map_point, const float* prior_kp_Xw, float th, float nnratio, eao_track_result* out,
                                       void* stream) {
    return track_chain(h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_prior, prior_kp_map_point, prior_kp_Xw, th, nnratio, out, stream, nullptr);
}

eao_status eao_tracker_track_with_motion_model(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                               const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_cur,
                                               const float* Tcw_last, int32_t n_last, const uint8_t* valid, const float* Xw, const uint8_t* mp_desc,
                                               const int32_t* last_octave, const float* last_angle, float th, int32_t mono, int32_t check_orientation,
                                               int32_t discard_outliers, eao_track_result* out, void* stream) {
    EAO_REQUIRE(h && Tcw_last && n_last >= 0, "null argument");
    MotionArgs mm{Tcw_last, n_last, valid, Xw, mp_desc, last_octave, last_angle, mono ? 1 : 0, check_orientation ? 1 : 0};
    eao_status st = track_chain(h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_cur, nullptr, nullptr, th, 0.f, out, stream, &mm);
    if (st) return st;
    if (discard_outliers) {      // "Discard outliers", src/Tracking.cc:2188-2207: the match is dropped, the flag cleared (n_inliers = what is left)
        for (int k = 0; k < h->cap; k++)
            if (out->kp_outlier[k]) { out->kp_map_point[k] = -1; out->kp_outlier[k] = 0; }
    }
    return EAO_OK;
}

eao_status eao_tracker_track_reference_keyframe(eao_tracker* h, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                                const float* d_depth, int32_t depth_pitch, int32_t width, int32_t height, const float* Tcw_last,
                                                int32_t n_kf, const uint8_t* valid, const float* Xw, const uint8_t* kf_desc, const float* kf_angle,
                                                const eao_feature_vector* fv_kf, const eao_feature_vector* fv_cur, float nnratio, int32_t check_orientation,
                                                int32_t discard_outliers, eao_track_result* out, void* stream) {
    EAO_REQUIRE(h && n_kf >= 0 && std::isfinite(nnratio), "null / bad argument");
    BowArgs bw{n_kf, valid, Xw, kf_desc, kf_angle, fv_kf, fv_cur, check_orientation ? 1 : 0};
    eao_status st = track_chain(h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_last, nullptr, nullptr, 0.f, nnratio, out, stream, nullptr, &bw);
    if (st) return st;
    if (discard_outliers) {      // "Discard outliers", src/Tracking.cc:1593-1612
        for (int k = 0; k < h->cap; k++)
            if (out->kp_outlier[k]) { out->kp_map_point[k] = -1; out->kp_outlier[k] = 0; }
    }
    return EAO_OK;
}

int32_t eao_abi_version(void) { return EAO_ABI_VERSION; }

}  // extern "C"
