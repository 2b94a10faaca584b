// lba.hip -- the local-BA kernels of the Levenberg-Marquardt engine on MI355X (gfx950) and every launch of the engine (BALaunch).  Shared pieces: lm_internal.h;
// PoseOptimization: pose.hip; the map-scale path: gba.hip; the host side: lm_host.hip.  (Round 6: split out of csrc/lm.hip.)
//
// Stands behind Optimizer::PoseOptimization (reference src/Optimizer.cc:325-673) and
// Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:675-1138), i.e. the arithmetic the reference
// delegates to its vendored g2o (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189,
// core/block_solver.hpp:354-604, types/types_six_dof_expmap.cpp, types/se3quat.h).  g2o's object graph is
// not reproduced; the maths is re-decomposed for the GPU:
//
//  PoseOptimization  : ONE persistent 512-thread workgroup runs all 4 rounds x <=10 LM iterations x <=10
//                      trials on the device (no host round trip): per-edge 2x6/3x6 Jacobians in registers,
//                      fixed-order shuffle/LDS reductions of the 28 accumulators, 6x6 LDLT by lane 0.
//  LocalBundleAdjustment : multi-workgroup kernels per LM trial, all reductions in a fixed order
//      k_ba_linearize   role A (eight lanes per map point, one edge per lane): residual Jacobians, Hll/bl, per-edge 6x3
//                       Hpl blocks; role B (one workgroup per free camera): Hpp/bp by a fixed-order tree over its edges
//      k_ba_schur_pairs_mfma (round 4, the default of the tile-solver windows; k_ba_schur_pairs / _b, the VALU forms over unscaled blocks, serve windows with
//                       plane landmarks; the slab assembly k_ba_schur + k_ba_reduce_* of round 1 was removed in round 5): the
//                       linearisation leaves Cholesky-scaled blocks W = Hpl C^-T (C C^T = Hll + lambda I), ONE workgroup per camera pair adds
//                       W(l, i1) W(l, i2)^T over the landmarks both observe on v_mfma_f64_4x4x4 (16-byte fetches staged through wave-private LDS)
//      k_ba_solve_tiles the 6*nFree square system as register-resident 16x16 tiles: LDL^T in 4-column panels, trailing
//                       update by v_mfma_f64_16x16x4_f64 (the ONE GEMM-shaped piece of the path), forward substitution
//                       folded in as an extra matrix row, single-wave back substitution, pose update exp(dx)*T.
//                       (windows of more than 30 free keyframes take the map-scale path k_bal_*; the single-workgroup LDS solver k_ba_solve is gone)
//      k_ba_backsub     per point: x_l = Dinv (bl - Hpl^T x_p), new point, residuals + robust chi2 at the trial state
//      k_ba_decide      fixed-order sums, rho, lambda / nu update; status lands in pinned host memory
//  The Jacobian blocks are 6x6 / 6x3 / 3x3: fp64 VALU + LDS, latency/bandwidth bound (the pair products are the exception: four 4x4x4 blocks).  LM state
//  and control flow live on the device: the host enqueues all iterations of an optimize() call in bulk, syncs once, and
//  replays an iteration trial by trial only when its first trial was rejected.
#include "lm_internal.h"

namespace {

// ============================================================================================ LocalBundleAdjustment
__device__ inline void ba_project(const Cam& c, bool stereo, const double p[3], double r[3]) {
    if (!stereo) {
        if (plain_den(p[2])) { const DivBy<true> z(p[2]); r[0] = z(p[0]) * c.fx + c.cx; r[1] = z(p[1]) * c.fy + c.cy; }
        else { r[0] = p[0] / p[2] * c.fx + c.cx; r[1] = p[1] / p[2] * c.fy + c.cy; }
        r[2] = 0;
    } else {
        const float invz = (float)(1.0 / p[2]);   // types_six_dof_expmap.cpp:150-156
        r[0] = p[0] * invz * c.fx + c.cx;
        r[1] = p[1] * invz * c.fy + c.cy;
        r[2] = r[0] - (double)(c.bf_f * invz);     // "const float& bf": float x float product
    }
}
__device__ inline double ba_chi2(const double* e, double info, bool stereo) {
    double s = e[0] * (info * e[0]) + e[1] * (info * e[1]);
    if (stereo) s += e[2] * (info * e[2]);
    return s;
}
template <bool SHARED>
__device__ __forceinline__ void ba_jacobians_t(const Cam& c, bool stereo, const double R[9], double X, double Y, double Zd, double A[3][3], double B[3][6]) {
    const DivBy<SHARED> Z(Zd), z2(Zd * Zd);      // "x / Z" is spelled Z(x) below; every expression keeps upstream's order of operations
    if (!stereo) {
        const double tmp[2][3] = {{c.fx, 0, Z(-X) * c.fx}, {0, c.fy, Z(-Y) * c.fy}};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++)
                A[i][j] = (Z(-1.) * tmp[i][0]) * R[j] + (Z(-1.) * tmp[i][1]) * R[3 + j] + (Z(-1.) * tmp[i][2]) * R[6 + j];
        for (int j = 0; j < 3; j++) A[2][j] = 0;
    } else {
        for (int j = 0; j < 3; j++) {
            A[0][j] = Z(-c.fx * R[j]) + z2(c.fx * X * R[6 + j]);
            A[1][j] = Z(-c.fy * R[3 + j]) + z2(c.fy * Y * R[6 + j]);
            A[2][j] = A[0][j] - z2(c.bf * R[6 + j]);
        }
    }
    B[0][0] = z2(X * Y) * c.fx; B[0][1] = -(1 + (z2(X * X))) * c.fx; B[0][2] = Z(Y) * c.fx;
    B[0][3] = Z(-1.) * c.fx; B[0][4] = 0; B[0][5] = z2(X) * c.fx;
    B[1][0] = (1 + z2(Y * Y)) * c.fy; B[1][1] = z2(-X * Y) * c.fy; B[1][2] = Z(-X) * c.fy;
    B[1][3] = 0; B[1][4] = Z(-1.) * c.fy; B[1][5] = z2(Y) * c.fy;
    if (stereo) {
        B[2][0] = B[0][0] - z2(c.bf * Y); B[2][1] = B[0][1] + z2(c.bf * X); B[2][2] = B[0][2];
        B[2][3] = B[0][3]; B[2][4] = 0; B[2][5] = B[0][5] - z2(c.bf);
    } else {
        for (int j = 0; j < 6; j++) B[2][j] = 0;
    }
}
// Jacobians wrt point (A, Dx3) and pose (B, Dx6): types_six_dof_expmap.cpp:103-139,188-234
__device__ inline void ba_jacobians(const Cam& c, bool stereo, const SE3& T, const double* pw, double A[3][3], double B[3][6]) {
    double p[3], R[9];
    se3_map(T, pw, p);
    quat_to_matrix(T.r, R);
    if (plain_den(p[2]) && plain_den(p[2] * p[2])) ba_jacobians_t<true>(c, stereo, R, p[0], p[1], p[2], A, B);
    else ba_jacobians_t<false>(c, stereo, R, p[0], p[1], p[2], A, B);
}

__device__ inline double plane_chi2(const BADev& P, const double* e) {
    return e[0] * (P.infoAngle * e[0]) + e[1] * (P.infoAngle * e[1]) + e[2] * (P.infoDist * e[2]);
}
// g2o's numeric Jacobians of an EdgePlane (central differences, delta = 1e-9, core/base_binary_edge.hpp:131-205):
// with respect to the plane vertex (VertexPlane::oplusImpl) ...
__device__ inline void plane_jac_plane(const SE3& T, const double* pl, const double* meas, double A[3][3]) {
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double add[3] = {0, 0, 0}, q[4], ep[3], em[3];
        add[d] = delta;  plane_oplus(pl, add, q); plane_error(T, q, meas, ep);
        add[d] = -delta; plane_oplus(pl, add, q); plane_error(T, q, meas, em);
        for (int k = 0; k < 3; k++) A[k][d] = scalar * (ep[k] - em[k]);
    }
}
// ... and with respect to the keyframe (exp(delta) * T)
__device__ inline void plane_jac_pose(const SE3& T, const double* pl, const double* meas, double B[3][6]) {
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    for (int d = 0; d < 6; d++) {
        double add[6] = {0, 0, 0, 0, 0, 0}, ep[3], em[3];
        add[d] = delta;  plane_error(se3_mul(se3_exp(add), T), pl, meas, ep);
        add[d] = -delta; plane_error(se3_mul(se3_exp(add), T), pl, meas, em);
        for (int k = 0; k < 3; k++) B[k][d] = scalar * (ep[k] - em[k]);
    }
}

// residuals + robust chi2 of every active edge at the CURRENT state (start of an optimize() call)
// Eight lanes per landmark, one edge per lane (points with more than 8 active edges loop): the per-point sums are
// 3-step xor trees inside the lane group.
template <bool PL>
__global__ __launch_bounds__(256) void k_ba_errors(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    const int l = (bx * 256 + threadIdx.x) >> 3, slot = threadIdx.x & 7;
    if (P.ctl[kCtlHalt]) return;
    const bool live = l < P.nL;
    const int pt = live ? P.actPt[l] : 0;
    const SE3* cams = cur_cams(P);
    const double* pts = cur_pts(P);
    const int beg = live ? P.ptStart[l] : 0, end = live ? P.ptStart[l + 1] : 0;
    double chi = 0;
    for (int k = beg + slot; k < end; k += 8) {
        const int e = P.ptEdges[k];
        const unsigned char fl = P.eflag[e];
        if (fl & 2) continue;
        const bool stereo = fl & 1;
        double* er = &P.err[3 * e];
        if (PL && (fl & 8)) {
            plane_error(cams[P.ecam[e]], &P.plBuf[P.ctl[kCtlCur]][4 * (pt - P.nPtsOnly)], &P.pmeas[4 * (e - P.nEdgesPt)], er);
            const double c2 = plane_chi2(P, er);
            double r0 = c2, w;
            if (fl & 4) huber(c2, P.deltaPlane, r0, w);
            chi += r0;
            continue;
        }
        double p[3], r[3];
        se3_map(cams[P.ecam[e]], &pts[3 * pt], p);
        ba_project(P.cam, stereo, p, r);
        er[0] = P.obs[3 * e] - r[0]; er[1] = P.obs[3 * e + 1] - r[1]; er[2] = stereo ? P.obs[3 * e + 2] - r[2] : 0;
        const double c2 = ba_chi2(er, P.info[e], stereo);
        double r0 = c2, w;
        if (fl & 4) huber(c2, stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
        chi += r0;
    }
    chi = group8_sum(chi);
    if (live && slot == 0) P.partChi[l] = chi;
}

// fixed-order sum of part[0..n) by one 256-thread block
__device__ inline double ordered_sum(const double* part, int n, double* red, double* out1) {
    double v[1] = {0};
    const int per = (n + 255) / 256;
    const int b = min((int)threadIdx.x * per, n), e = min(b + per, n);
    for (int i = b; i < e; i++) v[0] += part[i];
    block_sum<1, 256>(v, red, out1);
    return *out1;
}

__global__ __launch_bounds__(256) void k_ba_chi_init(const BADev* __restrict__ W, int wpar, int seq) {
    BA_WIN(P);
    BAStatus* const st = P.status;
    __shared__ double red[4], out1;
    if (P.ctl[kCtlHalt]) return;
    if (P.ctl[kCtlPhase] == 1 && !P.ctl[kCtlAnyActive]) {   // initializeOptimization(0) found nothing: optimize() returns -1
        __syncthreads();
        if (threadIdx.x == 0) {
            P.ctl[kCtlHalt] = 1; P.ctl[kCtlStatus] = kStEmpty;
            st->ph[1].status = kStEmpty; st->ph[1].iters = 0; st->ph[1].cur = P.ctl[kCtlCur]; st->ph[1].chi = 0; st->ph[1].touched = 1;
            st->status = kStEmpty; st->seq = seq;
        }
        return;
    }
    const double chi = ordered_sum(P.partChi, P.nL, red, &out1);
    if (threadIdx.x == 0) {
        P.lm[2] = chi;
        P.lm[3] = 0;          // max |diagonal| accumulator of the first linearisation (lambda_0)
        st->chi = chi; st->tempChi = chi; st->accepted = 1; st->solveOk = 1; st->rho = 0; st->lambda = P.lm[0];
        st->cur = P.ctl[kCtlCur];
        BAPhase& ph = st->ph[P.ctl[kCtlPhase] & 1];
        ph.chi = chi; ph.cur = P.ctl[kCtlCur]; ph.iters = P.ctl[kCtlIters]; ph.status = kStRunning; ph.nBad = P.ctl[kCtlNBad]; ph.accepted = 1;
        ph.rho = 0; ph.touched = 1;
        __threadfence_system();
        st->seq = seq;
    }
}

// The decision is a pure function of the previous LM state and three sums (BADecision); committing it writes the next LM
// state.  It runs either in k_ba_decide (state updated in place) or, for every trial but the last of a bulk segment, at the
// head of the NEXT trial's k_ba_linearize: every workgroup of that kernel derives the same decision from the OLD control
// block (which nobody writes during the kernel), workgroup 0 commits it into the OTHER control block, and the rest of
// the trial reads that one -- one launch less per trial, no inter-workgroup synchronisation.
struct BADecision {
    double lambda, nu, chi, rho, tempChi, currentChi;
    int accepted, cur, iters, nBad, halt, status, ok2;
};
__device__ inline BADecision ba_decision(const double* lmOld, const int* ctlOld, double tempSum, double scale, int ok2, int bulk) {
    BADecision d;
    const double lambda = lmOld[0];
    d.ok2 = ok2;
    d.tempChi = ok2 ? tempSum : DBL_MAX;
    d.currentChi = lmOld[2];
    double rho = d.currentChi - d.tempChi;
    scale += 1e-3;
    rho /= scale;
    d.rho = rho;
    d.cur = ctlOld[kCtlCur]; d.iters = ctlOld[kCtlIters]; d.nBad = ctlOld[kCtlNBad]; d.halt = ctlOld[kCtlHalt];
    d.chi = d.currentChi;
    d.accepted = 0;
    if (rho > 0 && isfinite(d.tempChi)) {
        const double y = 2 * rho - 1;
        double alpha = 1. - y * y * y;
        alpha = fmin(alpha, 2. / 3.);
        d.lambda = lambda * fmax(1. / 3., alpha);
        d.nu = 2;
        d.chi = d.tempChi;
        d.accepted = 1;
        d.cur ^= 1;                         // discardTop(): the trial buffer becomes the estimate
    } else {
        d.lambda = lambda * lmOld[1];
        d.nu = lmOld[1] * 2;
    }
    d.status = kStRunning;
    if (bulk) {
        if (d.accepted) {
            d.iters += 1;
            if ((d.currentChi - d.tempChi) * 1e3 < d.currentChi) d.nBad++; else d.nBad = 0;
            if (d.nBad >= 3) { d.status = kStTerminate; d.halt = 1; }
        } else {
            d.status = kStTakeover;
            d.halt = 1;
        }
    }
    return d;
}
// one thread; lm / ctl may be the old blocks themselves (in place) or the other pair
__device__ inline void ba_commit(const BADecision& d, const double* lmOld, const int* ctlOld, double* lm, int* ctl, BAStatus* st, int seq, int bulk) {
    const int phase = ctlOld[kCtlPhase] & 1;
    const int itersOld = ctlOld[kCtlIters], statusOld = ctlOld[kCtlStatus], anyActive = ctlOld[kCtlAnyActive], phaseRaw = ctlOld[kCtlPhase];
    const double maxDiag = lmOld[3];
    lm[0] = d.lambda; lm[1] = d.nu; lm[2] = d.chi; lm[3] = maxDiag;
    ctl[kCtlHalt] = d.halt; ctl[kCtlCur] = d.cur; ctl[kCtlIters] = d.iters; ctl[kCtlNBad] = d.nBad;
    ctl[kCtlStatus] = bulk ? d.status : statusOld; ctl[kCtlPhase] = phaseRaw; ctl[kCtlAnyActive] = anyActive;
    if (bulk && d.accepted && itersOld < 32) {
        const int slot = 32 * phase + itersOld;
        st->trLambda[slot] = d.lambda; st->trChi[slot] = d.tempChi; st->trTrials[slot] = 1;
    }
    st->lambda = d.lambda; st->rho = d.rho; st->chi = d.chi; st->tempChi = d.tempChi; st->accepted = d.accepted; st->solveOk = d.ok2;
    st->cur = d.cur; st->iters = d.iters; st->status = d.status; st->nBad = d.nBad;
    BAPhase& ph = st->ph[phase];
    ph.lambda = d.lambda; ph.rho = d.rho; ph.chi = d.chi; ph.accepted = d.accepted; ph.cur = d.cur; ph.iters = d.iters;
    ph.status = d.status; ph.nBad = d.nBad; ph.touched = 1;
    st->seq = seq;
}

// wmode 1: instead of (Hll + lambda I)^-1 the landmark keeps the inverse Cholesky factor.  M = Hll + lambda I = C C^T (C lower), T = C^-T
// (upper), so M^-1 = T T^T and   Hpl M^-1 Hpl'^T = (Hpl T)(Hpl' T)^T,   Hpl M^-1 bl = (Hpl T)(C^-1 bl),   x_l = T (C^-1 bl - sum (Hpl T)^T x_p):
// every product the Schur complement needs is a plain inner product of pre-scaled blocks W = Hpl T -- what the matrix-core assembly
// accumulates.  (block_solver.hpp:381-432 forms the same matrices through Eigen's 3x3 inverse(); the two differ in rounding only.)
// H: upper triangle 00 01 02 11 12 22 of Hll.  1 / sqrt by v_rsq_f64 + two Newton steps (~1 ulp).
__device__ __forceinline__ double ba_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(y * 0.5, e, y);
}
__device__ __forceinline__ void ba_chol3(const double H[6], double lambda, double T[6]) {
    // Pivots: Hll is a sum of J^T W J terms and lambda > 0, so every pivot is positive in exact arithmetic; for a landmark whose observations are (nearly)
    // collinear the subtraction can cancel to <= 0, where v_rsq_f64 would return Inf / NaN and poison the whole window's Schur system.  A pivot is therefore
    // kept at no less than one rounding unit of its diagonal entry -- the magnitude the cancellation noise has, and what upstream's explicit 3 x 3 inverse
    // (finite, huge) amounts to; fmax leaves every healthy pivot bit for bit as it was.
    const double m00 = fmax(H[0] + lambda, 2.2250738585072014e-308), m11 = H[3] + lambda, m22 = H[5] + lambda;
    const double li00 = ba_rsqrt(m00);
    const double c10 = H[1] * li00, c20 = H[2] * li00;
    const double li11 = ba_rsqrt(fmax(m11 - c10 * c10, fmax(m11 * 0x1p-53, 2.2250738585072014e-308)));
    const double c21 = (H[4] - c20 * c10) * li11;
    const double li22 = ba_rsqrt(fmax(m22 - c20 * c20 - c21 * c21, fmax(m22 * 0x1p-53, 2.2250738585072014e-308)));
    const double li10 = -(c10 * li00) * li11;
    const double li21 = -(c21 * li11) * li22;
    const double li20 = -(c20 * li00 + c21 * li10) * li22;
    T[0] = li00; T[1] = li10; T[2] = li20; T[3] = li11; T[4] = li21; T[5] = li22;      // T[a][b] = C^-1[b][a]
}
// row (h0 h1 h2) of Hpl -> row of W = Hpl T
__device__ __forceinline__ void ba_scale_row(const double T[6], double h0, double h1, double h2, double& w0, double& w1, double& w2) {
    w0 = h0 * T[0];
    w1 = h0 * T[1] + h1 * T[3];
    w2 = h0 * T[2] + h1 * T[4] + h2 * T[5];
}

// role A: blocks [0, ptBlocks): eight lanes per landmark.  role B: blocks [ptBlocks, ptBlocks + nFree): one block per free
// camera, one edge per thread; the 27 sums (21 of Hpp's upper triangle + 6 of bp) go through LDS in a fixed order: lane
// quads first (DPP), then 27 x 8 column threads over 32 quad leaders each, then the last 8.  (Sixty-four-lane shuffle
// trees for the 27 values cost 6.4 us here; this costs about one.)
constexpr int kLinThreads = 1024;      // workgroup size of a BATCH's linearisation (and of the classic, non-pre-scaled one).  Rounds 2-3 measured 512- and 256-thread
                                       //  workgroups in batches without a difference (3.27 / 3.29 / 3.29 ms for 25 windows; the launch alone 43.8 -> 37.6 us with 512, the call
                                       //  2.94-2.98 ms either way); round 4, pre-scaled blocks: a batch 48.9 us with 1024 against 51.8 with 512, ONE window 1.079 ms with 1024,
                                       //  1.028 with 512, 1.062 with 256 -- BALaunch::lin_w picks.  (The role-B sum's tree depends on the size: last-bit differences only.)
// first = 1 on the first linearisation of an optimize() call: the largest |diagonal entry| of Hpp / Hll (lambda_0 = 1e-5 x
// that, optimization_algorithm_levenberg.cpp:166-180) is accumulated with one atomic max per workgroup -- a max does not
// depend on the order, and non-negative doubles order like their bit patterns.
// ctlOld != nullptr: the previous trial's decision has not been taken yet (bulk segment): see BADecision.  P.ctl / P.lm
// are then the blocks this trial runs on (written by workgroup 0 here), ctlOld / lmOld the ones the previous trial ran on.
// WM (wmode 1, point landmarks only): the blocks are stored pre-scaled, W = Hpl C^-T (ba_chol3), which needs the landmark's COMPLETE Hll before
// the first block can be written: role A runs over its edges twice -- sums first, then the Jacobians again (a hundred instructions from the edge's
// inputs, which are in L1 by then) for the blocks.  Holding the first pass's 6 x 3 block in registers instead spilled (128 VGPRs at 1024 threads).
// diagOnly (WM, first linearisation of an optimize() call): only the largest diagonal entry is wanted -- lambda_0 must be known before
// the blocks can be scaled, so that call runs the launch twice: once for the maximum, once for everything else.
template <bool PL, int NT, bool WM>
__global__ __launch_bounds__(NT) void k_ba_linearize(const BADev* __restrict__ W, int wpar, int ptBlocks, int first, int diagOnly) {
    static_assert(!(PL && WM), "pre-scaled blocks: point landmarks only");
    BA_WIN(P);
    if ((int)bx >= ptBlocks + P.nFree) return;      // (a batch is launched with the largest window's grid)
    __shared__ double red[(NT / 4) * 27], part[8 * 27];
    __shared__ double s_wmax[NT / 64];
    if (P.ctl[kCtlHalt]) return;            // (the previous trial's decision is already in this control block: k_ba_backsub's last workgroup)
    const int cur = P.ctl[kCtlCur];
    const SE3* cams = P.camsBuf[cur];
    const double* pts = P.ptsBuf[cur];
    const bool stampL = P.dbg && threadIdx.x == 0 && ((int)bx == 0 || (int)bx == ptBlocks);
    const long long tl0 = stampL ? clock64() : 0;
    if ((int)bx < ptBlocks) {
        if constexpr (WM) {
            const int l = (bx * NT + threadIdx.x) >> 3, slot = threadIdx.x & 7;
            const bool live = l < P.nL;
            const int4 rec = live ? P.slot[(size_t)l * 8 + slot] : make_int4(-1, 0, -1, 0);    // the lane's first edge (see BADev::slot)
            const int pt = rec.w & 0x7FFFFFFF;
            const bool more = rec.w < 0;          // a landmark with more than eight edges walks its list as before
            const int kFirst = more ? P.ptStart[l] + slot : 0, end = more ? P.ptStart[l + 1] : 0;
            auto each_edge = [&](auto&& body) {
                int e = rec.x, cam = rec.y, ci = rec.z, k = kFirst;
                while (e >= 0) {
                    body(e, cam, ci);
                    e = -1;
                    if (more) {
                        k += 8;
                        if (k < end) { e = P.ptEdges[k]; cam = P.ecam[e]; ci = P.camIdx[cam]; }
                    }
                }
            };
            // robust weight and Jacobians of one edge, exactly as the classic role A forms them
            auto edge_terms = [&](int e, int cam, double (&A)[3][3], double (&B)[3][6], double (&wi)[3], double (&omr)[3]) {
                const unsigned char fl = P.eflag[e];
                const bool stereo = fl & 1;
                const double* er = &P.err[3 * e];
                const double info = P.info[e];
                double w = 1.0, r0;
                ba_jacobians(P.cam, stereo, cams[cam], &pts[3 * pt], A, B);
                if (fl & 4) huber(ba_chi2(er, info, stereo), stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
                for (int r = 0; r < 3; r++) { wi[r] = w * info; omr[r] = w * (-(info * er[r])); }
            };
            double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};   // upper triangle 00 01 02 11 12 22
            // (round 5, VERDICT r4 next #6b, measured and not kept: forming the unscaled block B^T (w Omega) A of the lane's first edge in THIS pass and keeping it
            //  in registers until the landmark's Cholesky factor exists, instead of evaluating the edge's Jacobians a second time below -- 180 VGPRs, 1.030 ms against
            //  1.030 ms per window; 70 spilled registers in the 1024-thread form of a batch, 3.34 ms against 2.62: profiles/r05_ba_pair_ablation.txt)
            each_edge([&](int e, int cam, int) {
                if (P.eflag[e] & 2) return;
                double A[3][3], B[3][6], wi[3], omr[3];
                edge_terms(e, cam, A, B, wi, omr);
                int q = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double sb = 0;
#pragma unroll
                    for (int r = 0; r < 3; r++) sb += A[r][i] * omr[r];
                    b[i] += sb;
#pragma unroll
                    for (int j = i; j < 3; j++) { double h = 0; for (int r = 0; r < 3; r++) h += A[r][i] * wi[r] * A[r][j]; H[q++] += h; }
                }
            });
#pragma unroll
            for (int i = 0; i < 6; i++) H[i] = group8_sum(H[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) b[i] = group8_sum(b[i]);
            if (first) {
                double m = (live && slot == 0) ? fmax(fmax(fabs(H[0]), fabs(H[3])), fabs(H[5])) : 0.0;
                for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_down(m, d));
                if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = m;
                __syncthreads();
                if (threadIdx.x == 0) {
                    for (int w = 1; w < NT / 64; w++) m = fmax(m, s_wmax[w]);
                    atomicMax((unsigned long long*)&P.lm[3], (unsigned long long)__double_as_longlong(m));
                }
            }
            if (!diagOnly) {
                // every lane of the group holds the landmark's sums: each factors M = Hll + lambda I itself and scales its own block(s)
                const double lambda = first ? refc::LM_TAU * P.lm[3] : P.lm[0];
                double T[6];
                ba_chol3(H, lambda, T);
                each_edge([&](int e, int cam, int ci) {
                    if (ci < 0 || (P.eflag[e] & 2)) return;
                    double A[3][3], B[3][6], wi[3], omr[3];
                    edge_terms(e, cam, A, B, wi, omr);
                    double* Wx = &P.Hpl[(size_t)e * 18];
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        double h[3];
#pragma unroll
                        for (int j = 0; j < 3; j++) { double v = 0; for (int r = 0; r < 3; r++) v += B[r][i] * wi[r] * A[r][j]; h[j] = v; }
                        double w0, w1, w2;
                        ba_scale_row(T, h[0], h[1], h[2], w0, w1, w2);
                        Wx[i * 3] = w0; Wx[i * 3 + 1] = w1; Wx[i * 3 + 2] = w2;
                    }
                });
                if (live && slot == 0) {
                    double* Ho = &P.Hll[(size_t)l * 9];
                    Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
                    for (int i = 0; i < 3; i++) P.bl[(size_t)l * 3 + i] = b[i];
                    double* To = &P.Tl[(size_t)l * 6];
#pragma unroll
                    for (int i = 0; i < 6; i++) To[i] = T[i];
                    P.ul[(size_t)l * 3] = T[0] * b[0];
                    P.ul[(size_t)l * 3 + 1] = T[1] * b[0] + T[3] * b[1];
                    P.ul[(size_t)l * 3 + 2] = T[2] * b[0] + T[4] * b[1] + T[5] * b[2];
                }
            }
        } else {
        // role A: eight lanes per landmark, one edge per lane; Hll / bl are xor-tree sums over the lane group
        const int l = (bx * NT + threadIdx.x) >> 3, slot = threadIdx.x & 7;
        const bool live = l < P.nL;
        const int4 rec = live ? P.slot[(size_t)l * 8 + slot] : make_int4(-1, 0, -1, 0);    // the lane's first edge (see BADev::slot)
        const int pt = rec.w & 0x7FFFFFFF;
        const bool more = rec.w < 0;          // a landmark with more than eight edges walks its list as before
        int e = rec.x, cam = rec.y, ci = rec.z;
        int k = 0, end = 0;
        if (more) { k = P.ptStart[l] + slot; end = P.ptStart[l + 1]; }
        double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};   // upper triangle 00 01 02 11 12 22
        while (e >= 0) {
            const unsigned char fl = P.eflag[e];
            if (!(fl & 2)) {
                const bool stereo = fl & 1;
                constexpr int D = 3;   // monocular edges carry a zero third row / residual: static loops, no scratch
                double A[3][3], B[3][6];
                const double* er = &P.err[3 * e];
                double info = P.info[e], info2 = info;     // rows 0, 1 / row 2 of the (diagonal) information matrix
                double w = 1.0, r0;
                const bool camFree = ci >= 0;
                if (PL && (fl & 8)) {
                    const double* pl = &P.plBuf[cur][4 * (pt - P.nPtsOnly)];
                    const double* meas = &P.pmeas[4 * (e - P.nEdgesPt)];
                    plane_jac_plane(cams[cam], pl, meas, A);
                    if (camFree) plane_jac_pose(cams[cam], pl, meas, B);
                    info = P.infoAngle; info2 = P.infoDist;
                    if (fl & 4) huber(plane_chi2(P, er), P.deltaPlane, r0, w);
                } else {
                    ba_jacobians(P.cam, stereo, cams[cam], &pts[3 * pt], A, B);
                    if (fl & 4) huber(ba_chi2(er, info, stereo), stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
                }
                const double wi[3] = {w * info, w * info, PL ? w * info2 : w * info};
                double omr[3];
                for (int r = 0; r < 3; r++) omr[r] = w * (-((PL && r == 2 ? info2 : info) * er[r]));
                int q = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double sb = 0;
#pragma unroll
                    for (int r = 0; r < D; r++) sb += A[r][i] * omr[r];
                    b[i] += sb;
#pragma unroll
                    for (int j = i; j < 3; j++) { double h = 0; for (int r = 0; r < D; r++) h += A[r][i] * wi[r] * A[r][j]; H[q++] += h; }
                }
                if (camFree) {
                    double* Hx = &P.Hpl[(size_t)e * 18];
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) { double h = 0; for (int r = 0; r < D; r++) h += B[r][i] * wi[r] * A[r][j]; Hx[i * 3 + j] = h; }
                }
            }
            e = -1;
            if (more) {
                k += 8;
                if (k < end) { e = P.ptEdges[k]; cam = P.ecam[e]; ci = P.camIdx[cam]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) H[i] = group8_sum(H[i]);
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = group8_sum(b[i]);
        if (live && slot == 0) {
            double* Ho = &P.Hll[(size_t)l * 9];
            Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
            for (int i = 0; i < 3; i++) P.bl[(size_t)l * 3 + i] = b[i];
        }
        if (first) {
            double m = (live && slot == 0) ? fmax(fmax(fabs(H[0]), fabs(H[3])), fabs(H[5])) : 0.0;
            for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_down(m, d));
            if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < NT / 64; w++) m = fmax(m, s_wmax[w]);
                atomicMax((unsigned long long*)&P.lm[3], (unsigned long long)__double_as_longlong(m));
            }
        }
        }
        if (stampL) P.dbg[14] = clock64() - tl0;
    } else {
        const int ci = bx - ptBlocks;
        const int cam = P.actCam[ci];
        const SE3 T = cams[cam];
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; k++) acc[k] = 0;
        const int beg = P.camStart[ci], end = P.camStart[ci + 1];
        for (int k = beg + (int)threadIdx.x; k < end; k += NT) {
            const int e = P.camEdges[k];
            const unsigned char fl = P.eflag[e];
            if (fl & 2) continue;
            const bool stereo = fl & 1;
            constexpr int D = 3;
            double A[3][3], B[3][6];
            const double* er = &P.err[3 * e];
            double info = P.info[e], info2 = info;
            double w = 1.0, r0;
            if (PL && (fl & 8)) {
                plane_jac_pose(T, &P.plBuf[cur][4 * (P.ept[e] - P.nPtsOnly)], &P.pmeas[4 * (e - P.nEdgesPt)], B);
                info = P.infoAngle; info2 = P.infoDist;
                if (fl & 4) huber(plane_chi2(P, er), P.deltaPlane, r0, w);
            } else {
                ba_jacobians(P.cam, stereo, T, &pts[3 * P.ept[e]], A, B);
                if (fl & 4) huber(ba_chi2(er, info, stereo), stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
            }
            const double wi[3] = {w * info, w * info, PL ? w * info2 : w * info};
            double omr[3];
            for (int r = 0; r < 3; r++) omr[r] = w * (-((PL && r == 2 ? info2 : info) * er[r]));
            int q = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double s = 0;
                for (int r = 0; r < D; r++) s += B[r][i] * omr[r];
                acc[21 + i] += s;
#pragma unroll
                for (int j = i; j < 6; j++) { double h = 0; for (int r = 0; r < D; r++) h += B[r][i] * wi[r] * B[r][j]; acc[q++] += h; }
            }
        }
        block_sum_lds<27, NT>(acc, red, part);
        if (threadIdx.x == 0) {
            const double* sums = part;
            int q = 0;
            if (!(WM && diagOnly)) {
                for (int i = 0; i < 6; i++)
                    for (int j = i; j < 6; j++) { P.Hpp[(size_t)ci * 36 + i * 6 + j] = sums[q]; P.Hpp[(size_t)ci * 36 + j * 6 + i] = sums[q]; q++; }
                for (int i = 0; i < 6; i++) P.bp[(size_t)ci * 6 + i] = sums[21 + i];
            }
            if (first) {
                double m = 0;
                q = 0;
                for (int i = 0; i < 6; i++) { m = fmax(m, fabs(sums[q])); q += 6 - i; }
                atomicMax((unsigned long long*)&P.lm[3], (unsigned long long)__double_as_longlong(m));
            }
        }
        if (stampL) P.dbg[15] = clock64() - tl0;
    }
}

// lambda_0 = 1e-5 * max |diag| over every active vertex (optimization_algorithm_levenberg.cpp:166-180)
__global__ __launch_bounds__(256) void k_ba_lambda_init(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    __shared__ double red[4];
    if (P.ctl[kCtlHalt]) return;
    double m = 0;
    for (int i = threadIdx.x; i < P.nFree * 6; i += 256) m = fmax(m, fabs(P.Hpp[(size_t)(i / 6) * 36 + (i % 6) * 7]));
    for (int i = threadIdx.x; i < P.nL * 3; i += 256) m = fmax(m, fabs(P.Hll[(size_t)(i / 3) * 9 + (i % 3) * 4]));
    for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_down(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        P.lm[0] = refc::LM_TAU * m; P.lm[1] = refc::LM_NI; P.lm[3] = m;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Tile-resident solver (up to kTileMaxFree free keyframes): the dense Schur system sits in REGISTERS as 16x16 accumulator
// tiles of v_mfma_f64_16x16x4_f64 (lower-triangle tiles, dealt round-robin to the 16 waves; lane l of a tile holds rows
// (l>>4)+4*reg, column l&15), with the right-hand side as one extra matrix row so that the forward substitution is part
// of the factorisation.  Right-looking LDL^T without pivoting in panels of FOUR columns (= the k extent of the MFMA):
//   export   the tile column under the panel writes its four columns to LDS                           -> barrier
//   panel    one thread per remaining row: factor the 4x4 diagonal block (redundantly, 10 broadcast reads), solve its
//            row  w = a L_kk^-T,  l = w D^-1;  publish w and l (zero for finished rows), archive l    -> barrier
//   update   every live tile: C -= W_tile L_tile^T, ONE MFMA, operands = one LDS double per lane each
// Measured on MI355X a lone wave issues one fp64 VALU instruction per 8 cycles whatever the dependences, so the cost
// of a step is its instruction count on the longest wave: ~90 for the panel thread, ~10 per tile for the update.
// L is archived row-packed in LDS; the back substitution L^T x = z walks it bottom-up in blocks of sixteen rows.

// ---------------------------------------------------------------------------------------------------------------------
// Pair-owned Schur assembly for the tile solver.  Instead of partial rows per (camera, edge chunk) that a second kernel has to
// sum, ONE workgroup owns the 6x6 block S(i1, i2) of a camera pair i1 <= i2 (and, for i1 == i2, the camera's six right-hand
// side entries): its threads split the ascending list of landmarks both cameras observe, each thread accumulates
//   -Y(i1,l) Hpl(i2,l)^T  with  Y = Hpl(i1,l) (Hll_l + lambda I)^-1     (+ Hpl(i1,l) (Hll_l + lambda I)^-1 bl_l for the rhs)
// over its landmarks in order, a fixed-order LDS reduction adds the threads, and the block goes straight into the solver's
// register-tile layout (Hpp + lambda I added on the diagonal).  No slabs, no reduction launch, no atomics on data.
constexpr int kPairThreads = 512;    // 1024 threads (one landmark per thread on the diagonal pairs) spill the 42 accumulators: 36.6 us vs 13.5

// landmark lists of the camera pairs (once per window; the outlier pass only clears table entries, which the assembly re-checks)
__global__ __launch_bounds__(256) void k_ba_pairs(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    if ((int)bx >= P.nFree * (P.nFree + 1) / 2) return;
    __shared__ int s_base;
    const int nF = P.nFree, t = threadIdx.x, lane = t & 63;
    int i1 = 0, rem = bx;
    while (rem >= nF - i1) { rem -= nF - i1; i1++; }
    const int i2 = i1 + rem;
    int* out = P.pairPts + (size_t)bx * P.nL;
    int4* items = P.wmode ? P.pairItems + (size_t)bx * P.nL : nullptr;      // wmode 1: {edge (l, i1), edge (l, i2), l} instead of the bare landmark list
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int l0 = 0; l0 < P.nL; l0 += 256) {
        const int l = l0 + t;
        const int e1 = l < P.nL ? P.table[(size_t)l * nF + i1] : -1, e2 = l < P.nL ? P.table[(size_t)l * nF + i2] : -1;
        const bool hit = e1 >= 0 && e2 >= 0;
        // ordered compaction: waves in order, lanes in order
        __shared__ int s_w[4];
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_w[t >> 6] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < (t >> 6); w++) off += s_w[w];
        if (hit) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1));
            if (items) items[pos] = make_int4(e1 * 144, e2 * 144, l * 24, l); else out[pos] = l;      // byte offsets into Hpl (W) / ul, see k_ba_schur_pairs_mfma
        }
        __syncthreads();
        if (t == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (t == 0) P.pairCnt[bx] = s_base;
}

// zero / identity background of the tile system (entries no pair block owns never change during a window)
__global__ __launch_bounds__(256) void k_ba_tiles_init(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    const TileGeom g = tile_geom(P.nFree);
    const int o = bx * 256 + threadIdx.x;
    if (o >= g.nTiles * 256) return;
    const int idx = o >> 8, reg = (o >> 6) & 3, lane = o & 63;
    int ti, tj;
    tile_of(g, idx, ti, tj);
    const int row = ti * 16 + (lane >> 4) + 4 * reg, col = tj * 16 + (lane & 15);
    P.sys[o] = (row >= g.n && row < g.n4 && col == row) ? 1.0 : 0.0;
}

__device__ __forceinline__ void tile_store(const BADev& P, const TileGeom& g, int row, int col, double v) {   // col <= row
    const int ti = row >> 4, tj = col >> 4;
    const int tileIdx = tj * g.Tr - tj * (tj - 1) / 2 + (ti - tj);
    const int rr = row & 15, cc = col & 15;
    P.sys[(size_t)tileIdx * 256 + (rr >> 2) * 64 + (rr & 3) * 16 + cc] = v;
}

// (a) ONE window: one thread per landmark, 512 threads -- the 210 workgroups of a 20-keyframe window have the chip to
//     themselves, so the 194 VGPRs (two waves per SIMD) of 42 accumulators per thread cost nothing and the launch is as
//     short as its slowest workgroup (13.8 us).
__global__ __launch_bounds__(kPairThreads) void k_ba_schur_pairs(const BADev* __restrict__ W, int wpar, int first) {
    BA_WIN(P);
    __shared__ double red[(kPairThreads / 4) * 42], part[8 * 42];
    if (P.ctl[kCtlHalt]) return;
    const int nF = P.nFree, t = threadIdx.x;
    int i1 = 0, rem = bx;
    while (rem >= nF - i1) { rem -= nF - i1; i1++; }
    const int i2 = i1 + rem;
    const bool diag = i1 == i2;
    const double lambda = first ? refc::LM_TAU * P.lm[3] : P.lm[0];
    if (first && bx == 0 && t == 0) { P.lm[0] = lambda; P.lm[1] = 2; }
    const int cnt = P.pairCnt[bx];
    const int* pts = P.pairPts + (size_t)bx * P.nL;
    const bool stampS = P.dbg && bx == 0 && t == 0;
    long long sp0 = 0, sp1 = 0, sp2 = 0;
    if (stampS) sp0 = clock64();
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; q++) acc[q] = 0;
    for (int k = t; k < cnt; k += kPairThreads) {
        const int l = pts[k];
        const int e1 = P.table[(size_t)l * nF + i1], e2 = P.table[(size_t)l * nF + i2];
        if (e1 < 0 || e2 < 0) continue;                      // deactivated by the outlier pass
        double Di[9];
        dinv3(&P.Hll[(size_t)l * 9], lambda, Di);
        const double* B1 = &P.Hpl[(size_t)e1 * 18];
        const double* B2 = &P.Hpl[(size_t)e2 * 18];
        double b2[18];
#pragma unroll
        for (int q = 0; q < 18; q++) b2[q] = B2[q];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double h0 = B1[r * 3], h1 = B1[r * 3 + 1], h2 = B1[r * 3 + 2];
            const double y0 = h0 * Di[0] + h1 * Di[3] + h2 * Di[6];
            const double y1 = h0 * Di[1] + h1 * Di[4] + h2 * Di[7];
            const double y2 = h0 * Di[2] + h1 * Di[5] + h2 * Di[8];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[r * 6 + c] -= y0 * b2[c * 3] + y1 * b2[c * 3 + 1] + y2 * b2[c * 3 + 2];
            if (diag) {
                const double* bl = &P.bl[(size_t)l * 3];
                acc[36 + r] += y0 * bl[0] + y1 * bl[1] + y2 * bl[2];     // Hpl (Dinv bl), row r
            }
        }
    }
    if (stampS) sp1 = clock64();
    block_sum_lds<42, kPairThreads>(acc, red, part);
    if (stampS) { sp2 = clock64(); P.dbg[13] = ((sp1 - sp0) << 20) | (sp2 - sp1); }
    const TileGeom g = tile_geom(nF);
    if (t < 36) {
        const int r = t / 6, c = t - r * 6;
        if (!diag) {
            tile_store(P, g, i2 * 6 + c, i1 * 6 + r, part[t]);        // lower element (row of i2, column of i1) = S(i1,i2)[r][c]
        } else if (c >= r) {
            const double v = part[t] + P.Hpp[(size_t)i1 * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
            tile_store(P, g, i1 * 6 + c, i1 * 6 + r, v);              // the upper-triangle value, mirrored into the lower tile
        }
    } else if (t < 42 && diag) {
        const int r = t - 36;
        tile_store(P, g, g.n4, i1 * 6 + r, P.bp[i1 * 6 + r] - part[t]);   // right-hand side row
    }
}

// (b) A BATCH of windows (5 250 workgroups for 25 windows) is bound by workgroup latency x rounds: two threads per landmark
//     (ADJACENT lanes 2k / 2k + 1 accumulate rows 0..2 / 3..5 of the pair's 6x6 block and of its right-hand side: 21
//     accumulators, 128 VGPRs) in 256-thread workgroups, four of which fit a CU instead of two.  The launch is bound by the
//     texture-address unit walking the distinct cache lines of each scattered 16-byte load (about one line per cycle): with
//     the two lanes of a landmark adjacent their Hll / Hpl(e2) loads share lines -- 32 lines per wave load instead of 64,
//     -20 us per launch against the halves in separate waves.  Four lanes per landmark (3x3 sub-blocks, 16 lines per load but
//     three passes per pair instead of two): +14 us.  One window: no difference to (a) at 256 / 512 / 1024 threads.  Measured, 25
//     windows per launch: (a) 152 us, (a) with 256 threads 106 us, two threads per landmark in 512-thread workgroups 112 us,
//     this 58 us; staging the operands through LDS with cooperative loads 156 us, pinning windows to XCDs no change; eight
//     lanes per landmark (lane r owns row r: a wave load touches 8 cache lines instead of 64, but three times the load
//     instructions) +12 us.  With the loop body removed the launch costs 19 us (5 250 workgroups: prologue, reduction, stores),
//     with the index loads only the same -- the time is the 30 scattered 16-byte loads per landmark-thread.
constexpr int kPairThreadsB = 256;
template <int NT>
__global__ __launch_bounds__(NT, 1024 / NT) void k_ba_schur_pairs_b(const BADev* __restrict__ W, int wpar, int first) {
    BA_WIN(P);
    if ((int)bx >= P.nFree * (P.nFree + 1) / 2) return;
    constexpr int kHalf = NT / 2, kRows = NT / 16;      // DPP rows of 16 lanes: lanes 0 / 1 of a row lead its two halves
    __shared__ double red[2 * kRows * 21], part[4 * 42];
    // The loads a workgroup needs first -- halt flag, lambda, the pair's landmark count and its first list entry -- are independent
    // of each other: all four are issued before the halt flag is looked at, so a workgroup starts its first Hll / Hpl loads after
    // three memory round trips instead of five (flag -> count -> list entry -> table -> blocks).
    const int nF = P.nFree, t = threadIdx.x;
    const int halt = P.ctl[kCtlHalt];
    const double lm0 = P.lm[0], lm3 = P.lm[3];
    const int cnt = P.pairCnt[bx];
    const int* pts = P.pairPts + (size_t)bx * P.nL;
    const int lFirst = (t >> 1) < P.nL ? pts[t >> 1] : 0;      // (inside the pair's nL slots whatever cnt turns out to be)
    if (halt) return;
    int i1 = 0, rem = bx;
    while (rem >= nF - i1) { rem -= nF - i1; i1++; }
    const int i2 = i1 + rem;
    const bool diag = i1 == i2;
    const double lambda = first ? refc::LM_TAU * lm3 : lm0;
    if (first && bx == 0 && t == 0) { P.lm[0] = lambda; P.lm[1] = 2; }
    const int half = t & 1;        // adjacent lanes share a landmark: their Hll / Hpl(e2) loads hit the same cache lines
    double acc[21];
#pragma unroll
    for (int q = 0; q < 21; q++) acc[q] = 0;
    for (int k = t >> 1; k < cnt; k += kHalf) {
        const int l = k == (t >> 1) ? lFirst : pts[k];
        const int e1 = P.table[(size_t)l * nF + i1], e2 = P.table[(size_t)l * nF + i2];
        if (e1 < 0 || e2 < 0) continue;                      // deactivated by the outlier pass
        double Di[9];
        dinv3(&P.Hll[(size_t)l * 9], lambda, Di);
        const double* B1 = &P.Hpl[(size_t)e1 * 18 + 9 * half];
        const double* B2 = &P.Hpl[(size_t)e2 * 18];
        double b2[18];
#pragma unroll
        for (int q = 0; q < 18; q++) b2[q] = B2[q];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const double h0 = B1[r * 3], h1 = B1[r * 3 + 1], h2 = B1[r * 3 + 2];
            const double y0 = h0 * Di[0] + h1 * Di[3] + h2 * Di[6];
            const double y1 = h0 * Di[1] + h1 * Di[4] + h2 * Di[7];
            const double y2 = h0 * Di[2] + h1 * Di[5] + h2 * Di[8];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[r * 6 + c] -= y0 * b2[c * 3] + y1 * b2[c * 3 + 1] + y2 * b2[c * 3 + 2];
            if (diag) {
                const double* bl = &P.bl[(size_t)l * 3];
                acc[18 + r] += y0 * bl[0] + y1 * bl[1] + y2 * bl[2];     // Hpl (Dinv bl), row 3 half + r
            }
        }
    }
    // fixed-order sum: the eight landmarks of a DPP row per half (lanes i, i+2 of a quad, then the row's four quads), then
    // 42 x 4 column threads over a quarter of their half's row leaders each, then the last four.  Value q < 18:
    // S[3 half + q / 6][q % 6]; q = 18..20: right-hand side row 3 half + q - 18.
#pragma unroll
    for (int q = 0; q < 21; q++) acc[q] = row_half_sum(acc[q]);
    if ((t & 14) == 0) {
        double* dst = red + (half * kRows + (t >> 4)) * 21;
#pragma unroll
        for (int q = 0; q < 21; q++) dst[q] = acc[q];
    }
    __syncthreads();
    if (t < 42 * 4) {
        const int v = t % 42, seg = t / 42, h = v >= 21 ? 1 : 0, q = v - 21 * h;
        constexpr int kSeg = kRows / 4;
        const double* src = red + (size_t)(h * kRows + seg * kSeg) * 21 + q;
        double sacc = 0;
        for (int j = 0; j < kSeg; j++) sacc += src[j * 21];
        part[seg * 42 + v] = sacc;
    }
    __syncthreads();
    double tot = 0;
    if (t < 42) tot = ((part[t] + part[42 + t]) + part[84 + t]) + part[126 + t];
    const TileGeom g = tile_geom(nF);
    if (t < 42) {
        const int h = t >= 21 ? 1 : 0, q = t - 21 * h;
        if (q < 18) {
            const int r = 3 * h + q / 6, c = q % 6;
            if (!diag) {
                tile_store(P, g, i2 * 6 + c, i1 * 6 + r, tot);            // lower element (row of i2, column of i1) = S(i1,i2)[r][c]
            } else if (c >= r) {
                const double v = tot + P.Hpp[(size_t)i1 * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
                tile_store(P, g, i1 * 6 + c, i1 * 6 + r, v);              // the upper-triangle value, mirrored into the lower tile
            }
        } else if (diag) {
            const int r = 3 * h + q - 18;
            tile_store(P, g, g.n4, i1 * 6 + r, P.bp[i1 * 6 + r] - tot);   // right-hand side row
        }
    }
}

// (c) wmode 1 (round 4): the pair's block on the MATRIX CORES.  With the blocks stored pre-scaled (W = Hpl C^-T, see ba_chol3)
//     S(i1, i2) = Hpp - sum_l W(i1, l) W(i2, l)^T   and   rhs(i1) = bp - sum_l W(i1, l) u_l
//     are plain inner products over k = (landmark, component): a [6 x 3 n] by [3 n x 7] product per pair -- too thin for a GEMM tile, but exactly
//     what v_mfma_f64_4x4x4_4b_f64 takes: FOUR independent 4 x 4 x 4 products per instruction, used as the four quadrants of an 8 x 8 output
//     (rows 0..5 = the rows of W(i1), columns 0..5 = the rows of W(i2), column 6 = u_l on diagonal pairs; the rest is padding).  One
//     instruction consumes one shared landmark (k = 0..2 its three components, k = 3 padding).  Lane maps of the instruction
//     (tools/ubench/mfma_probe.hip -> profiles/r04_mfma_probe.txt):
//       A: block (l >> 2) & 3, row l & 3, k = l >> 4      B: block (l >> 2) & 3, column l & 3, k = l >> 4      D: block (l >> 2) & 3, row l >> 4, column l & 3
//     with block = 2 (row quadrant) + (column quadrant).
//     What bounds every variant of this assembly is the path from L2 into the CU, not the arithmetic (a launch with the products removed takes as long
//     as with them): the texture-address unit walks a wave's load quad by quad and line by line, so a load is only cheap when ALL its lanes carry
//     16 useful bytes that fall on few lines.  Hence the fetch is shaped for the memory path, not for the instruction: a wave fetches the two blocks of
//     SEVEN landmarks with two 16-byte-per-lane loads (lane 9 j + c: chunk c of landmark j's 144-byte block; ~15 cache lines per load), parks them in
//     its own 2.4 KB of LDS, and reads each landmark's operands back in the instruction's layout (two 8-byte LDS reads with constant per-lane offsets;
//     the padding lanes read a zero slot).  The first version -- operands loaded element-wise from global memory, 8 bytes per lane, 36 of 64 lanes useful,
//     two loads per landmark -- took 105 us per 25 windows against the VALU kernel's 73: profiles/r04_ba_pair_ablation.txt.
//     The loads of the next seven landmarks are in flight while the current seven are multiplied (eight VGPRs per stage); the records
//     {144 * edge (l, i1), 144 * edge (l, i2), 24 * l} come as one small coalesced load two steps ahead and are handed to the lanes by ds_bpermute.
//     A workgroup owns a pair: wave w takes the groups w, w + NW, ... of the pair's landmark list (four accumulators per wave, in turn), the
//     partial sums are added in a fixed order -- no atomics, reproducible.  The list's last group is filled up with records of the ZERO block (block nEdges
//     of the array, entry nL of u: k_ba_prepare clears them).  An edge the outlier pass switched off keeps its list entries: its W block is zero
//     (k_ba_classify), its products vanish.
constexpr int kPairG = 7;                       // landmarks per fetch
constexpr int kPairSlot = 160;                  // LDS bytes per staged block: 144 data + a zero slot for the padding lanes
constexpr int kPairWaveLds = 3 * kPairG * kPairSlot;      // A blocks, B blocks, u entries (24 bytes each, at the blocks' pitch: one immediate offset per landmark)
template <int NW>      // waves per pair: 4 in a batch (5 250 workgroups: throughput), 16 for a single window (210 workgroups: latency of the longest pair)
#ifndef EAO_PAIR_OCC
#define EAO_PAIR_OCC 6
#endif
__global__ __launch_bounds__(64 * NW, NW == 4 ? EAO_PAIR_OCC : 1) void k_ba_schur_pairs_mfma(const BADev* __restrict__ W, int wpar, int first) {
    BA_WIN(P);
    if ((int)bx >= P.nFree * (P.nFree + 1) / 2) return;
    __shared__ double part[NW][64];
    __shared__ __attribute__((aligned(16))) unsigned char stage[NW][kPairWaveLds];
    const int nF = P.nFree, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // A launch of this kernel lasts (rounds of workgroups) x (a workgroup's chain of dependent memory round trips), not its arithmetic: everything a
    // workgroup can ask for before it knows anything else is asked for at once -- halt flag, lambda, the pair's landmark count, the wave's FIRST
    // group of records (wave w takes groups w, w + NW, ...: where they lie does not depend on the count) and, for the last step, its share of Hpp / bp.
    int i1 = 0, rem = bx;
    while (rem >= nF - i1) { rem -= nF - i1; i1++; }
    const int i2 = i1 + rem;
    const bool diag = i1 == i2;
    const int blk = (lane >> 2) & 3, k = lane >> 4;
    const int dR = 4 * (blk >> 1) + (lane >> 4), dC = 4 * (blk & 1) + (lane & 3);       // D: row of W(i1), column = row of W(i2) (6: the right-hand side)
    const int* __restrict__ irec = reinterpret_cast<const int*>((const int4*)P.pairItems + (size_t)bx * P.nL);
    const int recCap = 4 * P.nL;
    auto load_rec = [&](int g) { const int o = g * (4 * kPairG) + lane; return lane < 4 * kPairG && o < recCap ? irec[o] : 0; };      // (inside the pair's nL slots whatever the count)
    const int halt = P.ctl[kCtlHalt];
    const double lm0 = P.lm[0], lm3 = P.lm[3];
    const int cnt = P.pairCnt[bx];
    int recRaw = load_rec(wave);
    double hb = 0;
    if (diag && wave == 0 && dR < 6) { if (dC < 6) hb = P.Hpp[(size_t)i1 * 36 + dR * 6 + dC]; else if (dC == 6) hb = P.bp[i1 * 6 + dR]; }
    if (halt) return;
    const double lambda = (first & 1) ? refc::LM_TAU * lm3 : lm0;
    if ((first & 1) && bx == 0 && t == 0) { P.lm[0] = lambda; P.lm[1] = 2; }
    // ---- the instruction's view: which element of the 8 x 4 / 4 x 8 operands this lane supplies, as a byte offset into a staged block
    const int aRow = 4 * (blk >> 1) + (lane & 3), bCol = 4 * (blk & 1) + (lane & 3);
    const bool aOn = aRow < 6 && k < 3, bOn = bCol < 6 && k < 3, uOn = diag && bCol == 6 && k < 3;
    unsigned char* const sA = stage[wave];
    unsigned char* const sB = sA + kPairG * kPairSlot;
    unsigned char* const sU = sB + kPairG * kPairSlot;
    const unsigned char* const rdA = sA + (aOn ? (aRow * 3 + k) * 8 : 144);
    const unsigned char* const rdB = bOn ? sB + (bCol * 3 + k) * 8 : (uOn ? sU + k * 8 : sB + 144);
    const unsigned rdA32 = (unsigned)reinterpret_cast<size_t>(rdA), rdB32 = (unsigned)reinterpret_cast<size_t>(rdB);      // LDS byte addresses
    // ---- the memory path's view: lane 9 j + c fetches 16-byte chunk c of landmark j's blocks (lane 63 idles); lanes 3 j + d fetch u_l[d]
    const int fj = lane / 9, fc = lane - 9 * fj;
    const bool fOn = lane < 9 * kPairG, fuOn = diag && lane < 3 * kPairG;
    unsigned char* const wrA = sA + fj * kPairSlot + fc * 16;
    unsigned char* const wrB = sB + fj * kPairSlot + fc * 16;
    unsigned char* const wrU = sU + (lane / 3) * kPairSlot + (lane % 3) * 8;
    if (lane < kPairG) {      // the zero slots (never written again)
        *reinterpret_cast<uint4*>(sA + lane * kPairSlot + 144) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(sB + lane * kPairSlot + 144) = make_uint4(0, 0, 0, 0);
    }
    const char* const Wb = reinterpret_cast<const char*>((const double*)P.Hpl);
    const char* const ub = reinterpret_cast<const char*>((const double*)P.ul);
    const int nGall = (cnt + kPairG - 1) / kPairG;                    // groups of the pair; this wave: wave, wave + NW, ...
    const int padRec = (lane & 3) == 2 ? P.nL * 24 : P.nEdges * 144;
    auto fix_rec = [&](int raw, int g) { return lane < 4 * min(kPairG, cnt - g * kPairG) ? raw : padRec; };      // a short last group: the rest are zero-block records
    struct Fetch { uint4 a, b; double u; };
    auto fetch_data = [&](int rec) {
        Fetch f;
        f.a = make_uint4(0, 0, 0, 0); f.b = f.a; f.u = 0;
        const unsigned oa = (unsigned)__shfl(rec, 4 * fj), ob = (unsigned)__shfl(rec, 4 * fj + 1);
        if (fOn) {
            f.a = *reinterpret_cast<const uint4*>(Wb + oa + fc * 16);
            f.b = *reinterpret_cast<const uint4*>(Wb + ob + fc * 16);
        }
        if (diag) {
            const unsigned ou = (unsigned)__shfl(rec, 4 * (lane / 3) + 2);
            if (fuOn) f.u = *reinterpret_cast<const double*>(ub + ou + (lane % 3) * 8);
        }
        return f;
    };
    auto wave_fence = [] { eao::wave_sync(); };      // (csrc/common.h)
    double acc[4] = {0, 0, 0, 0};
    if (wave < nGall) {
        Fetch cur = fetch_data(fix_rec(recRaw, wave));
        int recN = wave + NW < nGall ? load_rec(wave + NW) : 0;
        for (int g = wave; g < nGall; g += NW) {
            // park group g (the reads of the group before it were issued ahead of these writes: LDS operations of a wave execute in order)
            if (fOn) { *reinterpret_cast<uint4*>(wrA) = cur.a; *reinterpret_cast<uint4*>(wrB) = cur.b; }
            if (fuOn) *reinterpret_cast<double*>(wrU) = cur.u;
            // the next group's blocks on their way, the records of the one after it behind them
            if (g + NW < nGall) {
                cur = fetch_data(fix_rec(recN, g + NW));
                if (g + 2 * NW < nGall) recN = load_rec(g + 2 * NW);
            }
            wave_fence();
            // the operands: fourteen 8-byte reads, one instruction each.  (Left to the compiler they pair up into ds_read2_b64, which the LDS pipe serves at
            // 8 - 10 cycles against 2 x 2.2 for the two plain reads -- tools/ubench/lds_ops.hip -- and this kernel runs at the pace of the CU's LDS pipe.)
            double a0, a1, a2, a3, a4, a5, a6, b0, b1, b2, b3, b4, b5, b6;
            static_assert(kPairG == 7 && kPairSlot == 160, "the immediate offsets below");
            asm volatile("ds_read_b64 %0, %14\n ds_read_b64 %7, %15\n ds_read_b64 %1, %14 offset:160\n ds_read_b64 %8, %15 offset:160\n"
                         "ds_read_b64 %2, %14 offset:320\n ds_read_b64 %9, %15 offset:320\n ds_read_b64 %3, %14 offset:480\n ds_read_b64 %10, %15 offset:480\n"
                         "ds_read_b64 %4, %14 offset:640\n ds_read_b64 %11, %15 offset:640\n ds_read_b64 %5, %14 offset:800\n ds_read_b64 %12, %15 offset:800\n"
                         "ds_read_b64 %6, %14 offset:960\n ds_read_b64 %13, %15 offset:960\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6)
                         : "v"(rdA32), "v"(rdB32) : "memory");
            acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b3, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4, b4, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a5, b5, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a6, b6, acc[2], 0, 0, 0);
            wave_fence();
        }
    }
    part[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (t < 64) {
        double tot = part[0][t];
#pragma unroll
        for (int w = 1; w < NW; w++) tot += part[w][t];
        const TileGeom g = tile_geom(nF);
        if (dR < 6 && dC < 6) {
            if (!diag) {
                tile_store(P, g, i2 * 6 + dC, i1 * 6 + dR, -tot);           // lower element (row of i2, column of i1) = S(i1,i2)[r][c]
            } else if (dC >= dR) {
                tile_store(P, g, i1 * 6 + dC, i1 * 6 + dR, (hb + (dR == dC ? lambda : 0.0)) - tot);      // the upper-triangle value, mirrored into the lower tile
            }
        } else if (diag && dR < 6 && dC == 6) {
            tile_store(P, g, g.n4, i1 * 6 + dR, hb - tot);                  // right-hand side row
        }
    }
}

template <int TPW>
__global__ __launch_bounds__(kTileThreads) void k_ba_solve_tiles(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    int* const solveOk = P.solveOk;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_fail;
    if (P.ctl[kCtlHalt]) return;
    const TileGeom g = tile_geom(P.nFree);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: tile tests become scalar branches
    constexpr int NW = kTileThreads / 64;
    const int rowsPad = g.Tr * 16;
    typedef double v2d __attribute__((ext_vector_type(2)));
    double* Lall = sm;                                          // tile_lrow(): rows 0 .. n4 (row n4 = z)
    double* sP = Lall + ((tile_lrow(g.n4 + 1) + 1) & ~1);       // [rowsPad][4] exported panel (16-byte aligned rows)
    double* sW = sP + (size_t)rowsPad * 4;                      // [rowsPad][4]
    double* sL = sW + (size_t)rowsPad * 4;                      // [rowsPad][4]
    double* sx = sL + (size_t)rowsPad * 4;                      // n4
    const bool stamp = P.dbg && t == 0;
    long long acc0 = 0, acc1 = 0, acc2 = 0, ts = 0;
    if (stamp) { P.dbg[0] = clock64(); P.dbg[1] = wall_clock64(); }
    if (t == 0) s_fail = 0;
    v4d c[TPW];
    int tJ[TPW], aOff[TPW], bOff[TPW], eOff[TPW];
    const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        const int idx = wave + NW * q;
        int ti = 0, tj = 0;
        tJ[q] = -1;
        c[q] = v4d{0, 0, 0, 0};
        if (idx < g.nTiles) {
            tile_of(g, idx, ti, tj);
            tJ[q] = tj;
            const double* src = P.sys + (size_t)idx * 256 + lane;
            c[q] = v4d{src[0], src[64], src[128], src[192]};
        }
        aOff[q] = (ti * 16 + lc) * 4 + lr;       // MFMA operands: W(row of the tile), L(column of the tile)
        bOff[q] = (tj * 16 + lc) * 4 + lr;
        eOff[q] = (ti * 16 + lr) * 4 + lc;       // export: element (row lr + 4 reg, column lc)
    }
    for (int q = t; q < rowsPad * 4; q += kTileThreads) { sW[q] = 0; sL[q] = 0; sP[q] = 0; }
    // panel role: thread t owns matrix row t (row n4 = right-hand side) for the whole factorisation
    const bool hasRow = t <= g.n4;
    const v2d* myP = (const v2d*)(sP + t * 4);
    v2d* myW = (v2d*)(sW + t * 4);
    v2d* myL = (v2d*)(sL + t * 4);
    double* myArch = Lall + tile_lrow(hasRow ? t : 0);
    __syncthreads();
    if (stamp) { P.dbg[2] = clock64(); P.dbg[3] = wall_clock64(); }
    for (int kc = 0; kc < g.n4; kc += 4) {
        const int tk = kc >> 4, co = kc & 15;
        if (stamp) ts = clock64();
        // ---- export the four panel columns of the tile column under the panel (all 16 rows of each tile: rows above
        //      the panel land in slots nobody reads)
        if (lc >= co && lc < co + 4) {
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                if (tJ[q] == tk) {
                    double* dst = sP + eOff[q] - co;
                    dst[0] = c[q][0]; dst[16] = c[q][1]; dst[32] = c[q][2]; dst[48] = c[q][3];
                }
            }
        }
        __syncthreads();
        if (stamp) { const long long now = clock64(); acc0 += now - ts; ts = now; }
        // ---- panel: one thread per remaining row.  The four diagonal rows run the same row solve: it reproduces their
        //      L entries left of the diagonal; what they publish in sW / sL only ever reaches matrix entries of finished
        //      rows and columns, which are never read again.  A zero pivot leaves Inf / NaN that reach x (checked there).
        if (hasRow && t >= kc) {
            const v2d* dg = (const v2d*)(sP + kc * 4);
            const double d00 = dg[0].x;
            const v2d r1 = dg[2], r2a = dg[4], r2b = dg[5], r3a = dg[6], r3b = dg[7];
            const v2d pa = myP[0], pb = myP[1];
            const double d10 = r1.x, d11 = r1.y, d20 = r2a.x, d21 = r2a.y, d22 = r2b.x, d30 = r3a.x, d31 = r3a.y, d32 = r3b.x, d33 = r3b.y;
            const double i0 = frcp1(d00);
            const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
            const double e11 = fma(-l10, d10, d11), e21 = fma(-l20, d10, d21), e31 = fma(-l30, d10, d31);
            const double e22 = fma(-l20, d20, d22), e32 = fma(-l30, d20, d32), e33 = fma(-l30, d30, d33);
            const double i1 = frcp1(e11);
            const double l21 = e21 * i1, l31 = e31 * i1;
            const double f22 = fma(-l21, e21, e22), f32 = fma(-l31, e21, e32), f33 = fma(-l31, e31, e33);
            const double i2 = frcp1(f22);
            const double l32 = f32 * i2;
            const double h33 = fma(-l32, f32, f33);
            const double i3 = frcp1(h33);
            const double w0 = pa.x;
            const double w1 = fma(-w0, l10, pa.y);
            const double w2 = fma(-w1, l21, fma(-w0, l20, pb.x));
            const double w3 = fma(-w2, l32, fma(-w1, l31, fma(-w0, l30, pb.y)));
            const double m0 = w0 * i0, m1 = w1 * i1, m2 = w2 * i2, m3 = w3 * i3;
            myW[0] = v2d{w0, w1}; myW[1] = v2d{w2, w3};
            myL[0] = v2d{m0, m1}; myL[1] = v2d{m2, m3};
            double* dst = myArch + kc;
            dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3;
        }
        __syncthreads();
        if (stamp) { const long long now = clock64(); acc1 += now - ts; ts = now; }
        // ---- trailing update: tiles whose columns reach past the panel; operands first, then the MFMAs back to back
        const int tkNext = (kc + 4) >> 4;
        double av[TPW], bv[TPW];
#pragma unroll
        for (int q = 0; q < TPW; q++) { av[q] = -sW[aOff[q]]; bv[q] = sL[bOff[q]]; }
#pragma unroll
        for (int q = 0; q < TPW; q++)
            if (tJ[q] >= tkNext) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], c[q], 0, 0, 0);
        if (stamp) acc2 += clock64() - ts;
    }
    if (stamp) { P.dbg[4] = clock64(); P.dbg[5] = wall_clock64(); P.dbg[10] = acc0; P.dbg[11] = acc1; P.dbg[12] = acc2; }
    __syncthreads();
    // ---- L^T x = z bottom-up in blocks of 16 rows.  Thread i keeps z(i) for the whole substitution.  The wave that holds a block's
    //      rows solves its triangle: x(r) is broadcast with a readlane (and parked in lane r of `xv`: the products
    //      run over the whole register, so entries at and right of the diagonal go on disturbing z values that were consumed already),
    //      the block's 16 rows of L are fetched from LDS before the chain starts.  Rows held by LOWER waves take the block's 16
    //      updates from the published x after a barrier, while the owner is already in its next block; blocks of wave 0 need no
    //      barrier at all.  Every z(i) sees the same fused multiply-adds in the same order (r descending) as in a row-by-row sweep.
    //      (Round 3: the sweep by one wave -- three registers of z, one row per step, 151 cycles per row -- took 18.1 k of the solver's
    //      76 k cycles at 120 rows.)
    {
        auto bcast = [](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
        double z = 0, xv = 0;
        if (wave <= ((g.n4 - 1) >> 6)) z = Lall[tile_lrow(g.n4) + t];      // (slack + the following arrays keep the padding lanes in bounds)
        // one block of CNT rows [lo, lo + CNT): straight-line code per block size (n4 is a multiple of four: only the top block is short)
        auto block = [&](auto cntTag, int lo) {
            constexpr int CNT = decltype(cntTag)::value;
            const int wB = lo >> 6;      // workgroup-uniform
            double cL[CNT];
            if (wave <= wB) {
                const double* row = Lall + tile_lrow(lo + CNT - 1) + t;      // row lo + CNT - 1 - k, column t
#pragma unroll
                for (int k = 0; k < CNT; k++) { cL[k] = *row; row -= lo + CNT - 1 - k + 3; }      // tile_lrow(r) - tile_lrow(r - 1) = r + 3
            }
            if (wave == wB) {
#pragma unroll
                for (int k = 0; k < CNT; k++) {
                    const int rl = (lo + CNT - 1 - k) & 63;
                    const double xr = __hiloint2double(bcast(__double2hiint(z), rl), bcast(__double2loint(z), rl));
                    xv = lane == rl ? z : xv;
                    z = fma(-cL[k], xr, z);
                }
                const int lb = lo & 63;
                if (lane >= lb && lane < lb + CNT) sx[lo + lane - lb] = xv;
            }
            if (wB > 0) {
                __syncthreads();
                if (wave < wB) {
#pragma unroll
                    for (int k = 0; k < CNT; k++) z = fma(-cL[k], sx[lo + CNT - 1 - k], z);
                }
            }
        };
        int B = (g.n4 - 1) >> 4;
        switch (g.n4 - (B << 4)) {      // rows of the top block
            case 4: block(std::integral_constant<int, 4>(), B << 4); break;
            case 8: block(std::integral_constant<int, 8>(), B << 4); break;
            case 12: block(std::integral_constant<int, 12>(), B << 4); break;
            default: block(std::integral_constant<int, 16>(), B << 4); break;
        }
        for (B--; B >= 0; B--) block(std::integral_constant<int, 16>(), B << 4);
    }
    __syncthreads();
    if (stamp) { P.dbg[6] = clock64(); P.dbg[7] = wall_clock64(); }
    bool bad = false;
    for (int q = t; q < g.n; q += kTileThreads) { const double v = sx[q]; P.xp[q] = v; bad |= !isfinite(v); }
    if (bad) s_fail = 1;
    __syncthreads();
    const bool failed = s_fail != 0;
    if (failed) {   // like a failed SimplicialLDLT: the step is void
        for (int q = t; q < g.n; q += kTileThreads) { sx[q] = 0; P.xp[q] = 0; }
        __syncthreads();
    }
    const SE3* cams = cur_cams(P);
    SE3* camsT = trial_cams(P);
    for (int cI = t; cI < P.nCams; cI += kTileThreads) {
        const int ci = P.camIdx[cI];
        if (ci >= 0) {
            double u[6];
            for (int q = 0; q < 6; q++) u[q] = sx[ci * 6 + q];
            camsT[cI] = se3_mul(se3_exp(u), cams[cI]);
        } else {
            camsT[cI] = cams[cI];
        }
    }
    if (t == 0) *solveOk = failed ? 0 : 1;
    if (stamp) { P.dbg[8] = clock64(); P.dbg[9] = wall_clock64(); }
}

// rho, accept / reject, lambda & nu update (optimization_algorithm_levenberg.cpp:118-147) by one 256-thread block.
// bulk = 1: the host has enqueued every remaining iteration of this optimize() call back to back (one trial each).  A
// clean iteration -- first trial accepted -- is finished right here (trace entry, iteration count, the "3 bad
// iterations" stop); anything else (rejected trial, rho == 0 or NaN) raises the halt flag so that the kernels still in
// the stream do nothing, and the host takes that iteration over trial by trial.  The status block is pinned host memory;
// the host reads it after a stream synchronisation, so no fence is needed.  (The camera part of the gain denominator,
// sum x (lambda x + b), is left in lm[4] by k_ba_backsub: the next linearisation overwrites b.)
__device__ inline void ba_decide_block(const BADev& P, const int* solveOk, BAStatus* st, int seq, int bulk) {
    __shared__ double red[2 * 4], out2[2];
    double v[2] = {0, 0};
#pragma unroll 4
    for (int i = threadIdx.x; i < P.nL; i += 256) { v[0] += P.partChi[i]; v[1] += P.partScale[i]; }
    block_sum<2, 256>(v, red, out2);
    if (threadIdx.x == 0) {
        const BADecision d = ba_decision(P.lm, P.ctl, out2[0], P.lm[4] + out2[1], *solveOk, bulk);
        ba_commit(d, P.lm, P.ctl, P.lm, P.ctl, st, seq, bulk);
    }
}

// a double made visible to every XCD before this thread goes on: an agent-scope atomic exchange whose result is waited for (see k_ba_backsub's memory model)
__device__ __forceinline__ void publish_f64(double* p, double v) {
    const unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old) : "memory");      // the returning form of the instruction: its completion is what orders it
}

// per landmark (eight lanes, one edge per lane): x_l = Dinv (bl - Hpl^T x_p); trial point; residuals + robust chi2 of
// its edges at the trial state; scale partial
template <bool PL>
__global__ __launch_bounds__(256) void k_ba_backsub(const BADev* __restrict__ W, int wpar, int decide, int seq) {
    BA_WIN(P);
    const int l = (bx * 256 + threadIdx.x) >> 3, slot = threadIdx.x & 7;
    int* const ctlNext = P.ctl0 + 8 * ((wpar & 1) ^ 1);
    double* const lmNext = P.lm0 + 8 * ((wpar & 1) ^ 1);
    if (P.ctl[kCtlHalt]) {      // frozen stream: hand the control block on unchanged
        if (decide && bx == 0 && threadIdx.x < 8) { ctlNext[threadIdx.x] = P.ctl[threadIdx.x]; lmNext[threadIdx.x] = P.lm[threadIdx.x]; }
        return;
    }
    const bool live = l < P.nL;
    const int4 rec = live ? P.slot[(size_t)l * 8 + slot] : make_int4(-1, 0, -1, 0);      // the lane's first edge (BADev::slot)
    const int pt = rec.w & 0x7FFFFFFF;
    const bool more = rec.w < 0;                // more than eight edges: the lane walks on through the landmark's list
    const double* pts = cur_pts(P);
    double* ptsT = trial_pts(P);
    const SE3* camsT = trial_cams(P);
    const double lambda = P.lm[0];
    const int kFirst = more ? P.ptStart[l] + slot : 0, end = more ? P.ptStart[l + 1] : 0;
    auto each_edge = [&](auto&& body) {         // body(edge, camera, free-camera index) for every edge of this lane
        int e = rec.x, cam = rec.y, ci = rec.z, k = kFirst;
        while (e >= 0) {
            body(e, cam, ci);
            e = -1;
            if (more) {
                k += 8;
                if (k < end) { e = P.ptEdges[k]; cam = P.ecam[e]; ci = P.camIdx[cam]; }
            }
        }
    };
    double cl[3] = {0, 0, 0};
    each_edge([&](int e, int, int ci) {
        if (ci < 0 || (P.eflag[e] & 2)) return;
        const double* Bi = &P.Hpl[(size_t)e * 18];
        const double* x = &P.xp[ci * 6];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double sacc = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) sacc += Bi[r * 3 + c] * (-x[r]);
            cl[c] += sacc;
        }
    });
    double bl[3] = {0, 0, 0}, Di[9], xl[3], np[3];
    const bool wm = !PL && P.wmode != 0;
    if (live) { bl[0] = P.bl[(size_t)l * 3]; bl[1] = P.bl[(size_t)l * 3 + 1]; bl[2] = P.bl[(size_t)l * 3 + 2]; }
    if (wm) {
        // the edges' blocks are stored pre-scaled (W = Hpl C^-T):  x_l = T (u - sum W^T x_p),  T = C^-T, u = C^-1 bl  (see ba_chol3)
        double T[6] = {0, 0, 0, 0, 0, 0}, u[3] = {0, 0, 0};
        if (live) {
#pragma unroll
            for (int i = 0; i < 6; i++) T[i] = P.Tl[(size_t)l * 6 + i];
            u[0] = P.ul[(size_t)l * 3]; u[1] = P.ul[(size_t)l * 3 + 1]; u[2] = P.ul[(size_t)l * 3 + 2];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) cl[c] = u[c] + group8_sum(cl[c]);
        xl[0] = T[0] * cl[0] + T[1] * cl[1] + T[2] * cl[2];
        xl[1] = T[3] * cl[1] + T[4] * cl[2];
        xl[2] = T[5] * cl[2];
    } else {
        if (live) dinv3(&P.Hll[(size_t)l * 9], lambda, Di);
        else for (int i = 0; i < 9; i++) Di[i] = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) cl[c] = bl[c] + group8_sum(cl[c]);
#pragma unroll
        for (int i = 0; i < 3; i++) xl[i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) np[i] = (live ? pts[3 * pt + i] : 0.0) + xl[i];
    const bool isPl = PL && live && pt >= P.nPtsOnly;
    double npl[4] = {1, 0, 0, 0};
    if (isPl) plane_oplus(&P.plBuf[P.ctl[kCtlCur]][4 * (pt - P.nPtsOnly)], xl, npl);      // VertexPlane::oplusImpl
    double chi = 0;
    each_edge([&](int e, int cam, int) {
        const unsigned char fl = P.eflag[e];
        if (fl & 2) return;                // level-1 edges keep the residual they last computed
        const bool stereo = fl & 1;
        double* er = &P.err[3 * e];
        if (isPl) {
            plane_error(camsT[cam], npl, &P.pmeas[4 * (e - P.nEdgesPt)], er);
            const double c2 = plane_chi2(P, er);
            double r0 = c2, w;
            if (fl & 4) huber(c2, P.deltaPlane, r0, w);
            chi += r0;
            return;
        }
        double p[3], r[3];
        se3_map(camsT[cam], np, p);
        ba_project(P.cam, stereo, p, r);
        er[0] = P.obs[3 * e] - r[0]; er[1] = P.obs[3 * e + 1] - r[1]; er[2] = stereo ? P.obs[3 * e + 2] - r[2] : 0;
        const double c2 = ba_chi2(er, P.info[e], stereo);
        double r0 = c2, w;
        if (fl & 4) huber(c2, stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
        chi += r0;
    });
    chi = group8_sum(chi);
    if (isPl && slot == 0) {
        double* plT = &P.plBuf[P.ctl[kCtlCur] ^ 1][4 * (pt - P.nPtsOnly)];
        for (int i = 0; i < 4; i++) plT[i] = npl[i];
    }
    double mine[2] = {0, 0};      // this landmark's terms of the trial chi2 and of the gain denominator (group leaders only)
    if (live && slot == 0) {
        for (int i = 0; i < 3; i++) { P.xl[(size_t)l * 3 + i] = xl[i]; ptsT[3 * pt + i] = np[i]; }
        mine[0] = chi;
        mine[1] = xl[0] * (lambda * xl[0] + bl[0]) + xl[1] * (lambda * xl[1] + bl[1]) + xl[2] * (lambda * xl[2] + bl[2]);
        P.partChi[l] = mine[0];
        P.partScale[l] = mine[1];
    }
    __shared__ double red[8], out2[2];
    if (bx == 0) {   // camera part of the gain denominator, sum x (lambda x + b), for the decision (fixed order)
        double v[1] = {0};
        for (int i = threadIdx.x; i < P.nFree * 6; i += 256) { const double x = P.xp[i]; v[0] += x * (lambda * x + P.bp[i]); }
        block_sum<1, 256>(v, red, &out2[0]);
        if (threadIdx.x == 0) publish_f64(&P.lm[4], out2[0]);
    }
    if (!decide) return;
    // The accept / reject decision of this trial, by the LAST workgroup of the window to get here (every other one has
    // published its partial sums by then): the next linearisation finds it in the other control block.  Leaving it to the
    // head of that launch instead -- every workgroup re-deriving it from the 2 nL partial sums -- cost each of its 1 100
    // workgroups (25 windows) 2.6 - 4 us of an 8 - 9 us life.
    // Memory model: a workgroup publishes ITS partial sums (wgPart) with agent-scope atomic EXCHANGES whose results it waits for, and only then takes its ticket; the
    // last workgroup reads the partial sums with agent-scope atomic loads.  The same goes for the camera part in lm[4].
    // (Round 6.  Rounds 2-5 published with agent-scope atomic STORES behind a workgroup-scope release -- no __threadfence(): its buffer_wbl2 per workgroup made the
    //  launch 3x slower.  A written-through store is acknowledged before it has landed at the point where the eight XCDs' L2s meet, and the ticket -- an atomic on
    //  another address -- can overtake it: beside another thread's bundle adjustment (a busy fabric) the last workgroup read ONE stale partial sum -- the previous
    //  trial's -- in about one call of 10 000, an LM decision flipped, and the call returned one of a handful of wrong-but-repeatable results; never seen alone, nor
    //  beside the extractor: tools/dbg_lba_beside_gba.py, tests/cpp/mixed_load.cpp's results_identical.  A read-modify-write atomic is performed AT that point and
    //  returns only afterwards, so waiting for the exchange's result orders it before the ticket; two extra atomics per workgroup, no cache write-back.)
    __shared__ int s_last;
    __syncthreads();
    block_sum<2, 256>(mine, red, out2);         // this workgroup's 32 landmarks, fixed order
    if (threadIdx.x == 0) {
        publish_f64(&P.wgPart[2 * bx], out2[0]);
        publish_f64(&P.wgPart[2 * bx + 1], out2[1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        s_last = __hip_atomic_fetch_add((int*)P.doneCnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    double v[2] = {0, 0};
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
        v[0] += __hip_atomic_load(&P.wgPart[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[1] += __hip_atomic_load(&P.wgPart[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    block_sum<2, 256>(v, red, out2);
    if (threadIdx.x == 0) {
        {   // (the counter back to zero for the next launch: an exchange as well -- performed where the next launch's tickets are counted)
            const int was = __hip_atomic_exchange((int*)P.doneCnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(was) : "memory");
        }
        const double camPart = __hip_atomic_load(&P.lm[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const BADecision d = ba_decision(P.lm, P.ctl, out2[0], camPart + out2[1], *P.solveOk, 1);
        ba_commit(d, P.lm, P.ctl, lmNext, ctlNext, P.status, seq, 1);
    }
}

// One block: the decision of the LAST trial of a bulk segment and of host-stepped trials (the others are decided by
// k_ba_backsub's last workgroup).
// ctl0 / lm0: the canonical control blocks.  The last trial of a bulk segment may have run on the other pair (see
// BADecision); whatever happened, the state is back in the canonical pair when this kernel ends.
__global__ __launch_bounds__(256) void k_ba_decide(const BADev* __restrict__ W, int wpar, int seq, int bulk) {
    BA_WIN(P);
    const int* const solveOk = P.solveOk;
    BAStatus* const st = P.status;
    int* const ctl0 = P.ctl0;
    double* const lm0 = P.lm0;
    if (!P.ctl[kCtlHalt]) ba_decide_block(P, solveOk, st, seq, bulk);
    if (P.ctl != ctl0) {
        __syncthreads();
        if (threadIdx.x < 8) { ctl0[threadIdx.x] = P.ctl[threadIdx.x]; lm0[threadIdx.x] = P.lm[threadIdx.x]; }
    }
}

// per edge: chi2 (of the stored residual) > threshold or non-positive depth at the current state.  update = 1 is the
// outlier pass between the two optimize() calls (src/Optimizer.cc:978-1008): flagged edges go to level 1 (bit1) and every
// edge loses its robust kernel (bit2) -- on the device, so the window's structure is never rebuilt.
__global__ void k_ba_classify(const BADev* __restrict__ W, int wpar, int update) {
    BA_WIN(P);
    unsigned char* const out = P.cls;
    const int nEdges = P.nEdges;
    const int e = bx * blockDim.x + threadIdx.x;
    if (e >= nEdges) return;
    // enqueued speculatively behind the first optimize(): a pending host takeover freezes the rest of the stream
    if (update && P.ctl[kCtlStatus] == kStTakeover) return;
    const unsigned char fl = P.eflag[e];
    if (fl & 8) { out[e] = 0; return; }      // (EdgePlane edges only exist in BundleAdjustment, which has no outlier pass)
    const bool stereo = fl & 1;
    const double c2 = ba_chi2(&P.err[3 * e], P.info[e], stereo);
    double p[3];
    se3_map(cur_cams(P)[P.ecam[e]], &cur_pts(P)[3 * P.ept[e]], p);
    const unsigned char bad = (c2 > (stereo ? refc::LBA_CHI2_STEREO : refc::LBA_CHI2_MONO) || !(p[2] > 0.0)) ? 1 : 0;
    out[e] = bad;
    if (update) {
        if (e == 0) {   // fresh control block for the second optimize() (the current-buffer index carries over); a first
                        // call that stopped early ("3 bad iterations") left the halt flag up -- it ends here
            P.ctl[kCtlHalt] = 0; P.ctl[kCtlIters] = 0; P.ctl[kCtlNBad] = 0; P.ctl[kCtlPhase] = 1;
        }
        P.eflag[e] = (unsigned char)((fl | (bad ? 2 : 0)) & ~4);
        if (!bad && !(fl & 2)) P.ctl[kCtlAnyActive] = 1;     // somebody survives: the second optimize() has something to do
        const int l = P.ptIdx[P.ept[e]], ci = P.camIdx[P.ecam[e]];
        if (bad && l >= 0 && ci >= 0) {
            if (P.table) P.table[(size_t)l * P.nFree + ci] = -1;
            if (P.wmode && !(fl & 2)) {      // the pair lists keep their entries: a switched-off edge contributes a zero block from now on
                double* Wx = &P.Hpl[(size_t)e * 18];
#pragma unroll
                for (int q = 0; q < 18; q++) Wx[q] = 0;
            }
        }
    }
}

// Device-side part of the set-up: the dense (point x free camera) edge table from the point adjacency, and the second
// copy of the state (a vertex no active edge reaches keeps its value in BOTH buffers).
__global__ __launch_bounds__(256) void k_ba_prepare(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    const int idx = bx * 256 + threadIdx.x;
    if (P.table && idx < P.nL * P.nFree) {      // (the map-scale path has no table: its pair lists carry the edges)
        const int l = idx / P.nFree, ci = idx - l * P.nFree;
        int found = -1;
        for (int k = P.ptStart[l]; k < P.ptStart[l + 1]; k++) {
            const int e = P.ptEdges[k];
            if (P.camIdx[P.ecam[e]] == ci) found = e;
        }
        P.table[idx] = found;
    }
    if (idx < P.nL * 8) {
        const int l = idx >> 3, beg = P.ptStart[l], end = P.ptStart[l + 1], k = beg + (idx & 7);
        int4 r = make_int4(-1, 0, -1, P.actPt[l] | (end - beg > 8 ? (int)0x80000000 : 0));
        if (k < end) { r.x = P.ptEdges[k]; r.y = P.ecam[r.x]; r.z = P.camIdx[r.y]; }
        P.slot[idx] = r;
    }
    if (idx < P.nPts * 3) P.ptsBuf[1][idx] = P.ptsBuf[0][idx];
    if (idx < (P.nPts - P.nPtsOnly) * 4) P.plBuf[1][idx] = P.plBuf[0][idx];
    if (idx < P.nCams) P.camsBuf[1][idx] = P.camsBuf[0][idx];
    if (idx < P.camStart[P.nFree]) P.camEdgeL[idx] = P.ptIdx[P.ept[P.camEdges[idx]]];
    if (idx == 0) *P.doneCnt = 0;
    if (P.wmode && idx < 18) P.Hpl[(size_t)P.nEdges * 18 + idx] = 0;      // the zero block / zero u entry that pad the pair lists' last blocks
    if (P.wmode && idx < 3) P.ul[(size_t)P.nL * 3 + idx] = 0;
}

// Results straight into pinned host memory: final state + the per-edge outlier flags.
__global__ __launch_bounds__(256) void k_ba_finish(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    SE3* const outCams = P.outCams; double* const outPts = P.outPts; unsigned char* const outCls = P.outCls; double* const outPlanes = P.outPlanes;
    const int idx = bx * 256 + threadIdx.x;
    if (idx < P.nCams) outCams[idx] = cur_cams(P)[idx];
    if (idx < P.nPtsOnly * 3) outPts[idx] = cur_pts(P)[idx];
    if (idx < (P.nPts - P.nPtsOnly) * 4) outPlanes[idx] = P.plBuf[P.ctl[kCtlCur]][idx];
    if (idx < P.nEdgesPt) {
        const int e = idx;
        const bool stereo = P.eflag[e] & 1;
        const double c2 = ba_chi2(&P.err[3 * e], P.info[e], stereo);
        double p[3];
        se3_map(cur_cams(P)[P.ecam[e]], &cur_pts(P)[3 * P.ept[e]], p);
        outCls[e] = (c2 > (stereo ? refc::LBA_CHI2_STEREO : refc::LBA_CHI2_MONO) || !(p[2] > 0.0)) ? 1 : 0;
    }
}

}  // namespace

namespace eao {
namespace lm {

eao_status BALaunch::attributes() const {
    // (per-function, process-wide state: only ever raised -- two threads or two windows of different sizes must not lower
    //  each other's limit between the set and the launch)
    auto raise = [](const void* fn, std::atomic<int>& cur, size_t want) -> eao_status {
        int have = cur.load();
        while ((int)want > have) {
            EAO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
            if (cur.compare_exchange_weak(have, (int)want)) break;
        }
        return EAO_OK;
    };
    static std::atomic<int> aT3{0}, aT5{0};
    eao_status st;
    if (d.solveTiles) {
        if (d.tiles3) { if ((st = raise((const void*)k_ba_solve_tiles<3>, aT3, d.tileLds))) return st; }
        else if ((st = raise((const void*)k_ba_solve_tiles<5>, aT5, d.tileLds))) return st;
    }
    if (d.bigPath && (st = gba_attributes())) return st;
    return EAO_OK;
}
int BALaunch::linBlocks() const { return eao::cdiv(std::max(d.nL, 1) * 8, kLinThreads); }
// the pre-scaled-block linearisation: 1024-thread workgroups in a batch (48.9 us per 25 windows against 51.8 with 512), 512 / 256 for one window -- its 24 landmark
// workgroups of 1024 leave nine tenths of the chip idle (EAO_BA_LIN_THREADS for A/B runs; profiles/r04_ba_pair_ablation.txt)
template <int NT> void BALaunch::lin_launch(int par, int first, int diagOnly) const {
    const int lb = eao::cdiv(std::max(d.nL, 1) * 8, NT);
    hipLaunchKernelGGL((k_ba_linearize<false, NT, true>), dim3(lb + d.nF, 1, gz()), dim3(NT), 0, s, W, wp(par), lb, first, diagOnly);
}
void BALaunch::lin_w(int par, int first, int diagOnly) const {
    static const int envLinT = getenv("EAO_BA_LIN_THREADS") ? atoi(getenv("EAO_BA_LIN_THREADS")) : 0;
    const int nt = envLinT ? envLinT : (nz > 1 ? 1024 : 512);
    if (nt == 256) lin_launch<256>(par, first, diagOnly);
    else if (nt == 512) lin_launch<512>(par, first, diagOnly);
    else lin_launch<kLinThreads>(par, first, diagOnly);
}
void BALaunch::setup() const {      // device-side part of the set-up (once per window)
    hipLaunchKernelGGL(k_ba_prepare, dim3(eao::cdiv(std::max(std::max(std::max(std::max(std::max(d.bigPath ? 0 : d.nL * d.nF, d.nL * 8), d.nP * 3), d.nC), d.E), d.nPl * 4), 256), 1, gz()), dim3(256), 0, s, W, wp(0));
    if (d.usePairs) {
        hipLaunchKernelGGL(k_ba_pairs, dim3(d.nF * (d.nF + 1) / 2, 1, gz()), dim3(256), 0, s, W, wp(0));
        hipLaunchKernelGGL(k_ba_tiles_init, dim3(tile_geom(d.nF).nTiles, 1, gz()), dim3(256), 0, s, W, wp(0));
    }
    if (d.bigPath) gba_enqueue_pair_fill(*this);
}
// one LM trial behind a linearisation: Schur assembly, solve, back substitution + residuals, (decision)
void BALaunch::trial(int par, int bulk, bool firstTrial, bool withDecide) {
    const int nF = d.nF, nL = d.nL;
    if (nF && d.bigPath) {
        gba_enqueue_trial(*this, par, firstTrial);
    } else if (nF && d.usePairs && d.solveTiles && d.wmode) {
        if (nz > 1) hipLaunchKernelGGL(k_ba_schur_pairs_mfma<4>, dim3(nF * (nF + 1) / 2, 1, gz()), dim3(256), 0, s, W, wp(par), firstTrial ? 1 : 0);
        else hipLaunchKernelGGL(k_ba_schur_pairs_mfma<16>, dim3(nF * (nF + 1) / 2, 1, 1), dim3(1024), 0, s, W, wp(par), firstTrial ? 1 : 0);
    } else if (nF && d.usePairs && d.solveTiles) {
        if (nz > 1) hipLaunchKernelGGL(k_ba_schur_pairs_b<kPairThreadsB>, dim3(nF * (nF + 1) / 2, 1, gz()), dim3(kPairThreadsB), 0, s, W, wp(par), firstTrial ? 1 : 0);
        else hipLaunchKernelGGL(k_ba_schur_pairs, dim3(nF * (nF + 1) / 2, 1, 1), dim3(kPairThreads), 0, s, W, wp(par), firstTrial ? 1 : 0);
    }
    // (round 5: the slab assembly k_ba_schur + k_ba_reduce_* and the single-workgroup LDS / global-scratch solver k_ba_solve of rounds 1-2 are gone -- dominated
    //  at every size by the pair assembly + register-tile solver up to 30 free keyframes and by the map-scale path beyond: 5.8 / 29 ms against 2.8 / 5.2 ms at
    //  31 / 64 free keyframes; commit 2249a31 holds the code)
    if (d.bigPath) {}
    else if (d.solveTiles && d.tiles3) hipLaunchKernelGGL(k_ba_solve_tiles<3>, dim3(1, 1, gz()), dim3(kTileThreads), d.tileLds, s, W, wp(par));
    else if (d.solveTiles) hipLaunchKernelGGL(k_ba_solve_tiles<5>, dim3(1, 1, gz()), dim3(kTileThreads), d.tileLds, s, W, wp(par));
    const int decideHere = bulk && !withDecide ? 1 : 0, sqHere = decideHere ? ++seq : 0;      // (see k_ba_backsub)
    if (nL && d.hasPl) hipLaunchKernelGGL(k_ba_backsub<true>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(par), decideHere, sqHere);
    else if (nL) hipLaunchKernelGGL(k_ba_backsub<false>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(par), decideHere, sqHere);
    if (withDecide) hipLaunchKernelGGL(k_ba_decide, dim3(1, 1, gz()), dim3(256), 0, s, W, wp(par), ++seq, bulk);
}
// wmode 1, a host-stepped RETRIAL (lambda changed, state unchanged): the stored blocks carry the previous lambda's Cholesky factor, and the residuals
// array holds the rejected trial's -- both are rebuilt at the current state (the same values the rejected trial started from; upstream keeps
// its linearisation and only re-adds lambda, block_solver.hpp:564-589)
void BALaunch::relinearize() {
    if (!d.wmode || !d.nL) return;
    hipLaunchKernelGGL(k_ba_errors<false>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(0));
    lin_w(0, 0, 0);
}
// iterations [from, to) of an optimize() call, one trial each, no host round trip
void BALaunch::bulk(int from, int to, bool withErrors) {
    const int nF = d.nF, nL = d.nL, lb = linBlocks();
    if (withErrors) {   // computeActiveErrors + activeRobustChi2 (otherwise the accepted trial already left them)
        if (nL && d.hasPl) hipLaunchKernelGGL(k_ba_errors<true>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(0));
        else if (nL) hipLaunchKernelGGL(k_ba_errors<false>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(0));
        hipLaunchKernelGGL(k_ba_chi_init, dim3(1, 1, gz()), dim3(256), 0, s, W, wp(0), ++seq);
    }
    // every trial but the last is decided (accept / reject) by the last workgroup of its k_ba_backsub, into the other
    // control block (BADecision), where the next linearisation finds it; the last one is decided by k_ba_decide, which
    // also brings the state back to the canonical block
    int par = 0;
    for (int it = from; it < to; it++) {
        if (it != from && nL) par ^= 1;
        if (d.hasPl) hipLaunchKernelGGL((k_ba_linearize<true, kLinThreads, false>), dim3(lb + nF, 1, gz()), dim3(kLinThreads), 0, s, W, wp(par), lb, it == 0 ? 1 : 0, 0);
        else if (d.wmode) {
            // first linearisation of the call: lambda_0 (the largest diagonal entry) must be known before the blocks can be stored pre-scaled --
            // one pass for the maximum, one for everything else
            if (it == 0) lin_w(par, 1, 1);
            lin_w(par, it == 0 ? 1 : 0, 0);
        } else hipLaunchKernelGGL((k_ba_linearize<false, kLinThreads, false>), dim3(lb + nF, 1, gz()), dim3(kLinThreads), 0, s, W, wp(par), lb, it == 0 ? 1 : 0, 0);
        if (it == 0 && !nF) hipLaunchKernelGGL(k_ba_lambda_init, dim3(1), dim3(256), 0, s, W, wp(par));   // no Schur kernel to do it
        trial(par, 1, it == 0 && nF, it == to - 1 || !nL);      // (no landmark, no k_ba_backsub: k_ba_decide after every trial)
    }
}
void BALaunch::classify() const { hipLaunchKernelGGL(k_ba_classify, dim3(eao::cdiv(std::max(d.E, 1), 256), 1, gz()), dim3(256), 0, s, W, wp(0), 1); }
void BALaunch::finish() const {
    hipLaunchKernelGGL(k_ba_finish, dim3(eao::cdiv(std::max(std::max(std::max(std::max(d.E, d.nP * 3), d.nC), d.nPl * 4), 1), 256), 1, gz()), dim3(256), 0, s, W, wp(0));
}
// Both optimize() calls of the reference, the outlier pass between them and the result copy in ONE enqueue.  A rejected LM
// trial freezes the rest of ITS window's work (halt flag + the takeover check of the outlier pass); the host then
// continues that window from where the device stopped.
void BALaunch::chain(int mode, int itsFirst, int itsSecond) {
    eao::Range rg("lm: optimize x2 + outlier pass, enqueue");
    bulk(0, itsFirst, true);
    if (mode == 0) { classify(); bulk(0, itsSecond, true); }
    finish();
}

}  // namespace lm
}  // namespace eao
