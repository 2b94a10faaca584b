// lm.hip -- Levenberg-Marquardt on MI355X (gfx950) for the two optimiser entry points of the hot path.
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
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "common.h"

namespace {

// 1/x: v_rcp_f64 + two Newton steps on the device (~1 ulp, a third of the instructions of an IEEE division), a plain
// division in host code
__host__ __device__ inline double recip(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// Several IEEE divisions by ONE denominator.  The compiler's sequence per fp64 division is v_div_scale x 2, v_rcp_f64 (quarter
// rate), four FMAs that refine the reciprocal, v_mul, v_fma, v_div_fmas, v_div_fixup: eleven instructions, seven of which
// depend on the denominator alone.  DivBy<true> runs those once (rcp + 4 FMA) and spends mul + 2 FMA per quotient -- the same
// operations on the same values, so the quotient is the SAME correctly rounded double, as long as v_div_scale would not have
// rescaled and v_div_fixup would not have patched anything: plain_den() admits denominators of magnitude 2^-400 .. 2^400 (not
// zero, NaN or infinite; depths in metres and their squares), the numerators are coordinates and products of coordinates
// (a quotient in the denormal range or a -0 numerator's sign could differ -- neither reaches a result).  The Jacobians of
// one BA edge divide 22 times by z or z^2 (types_six_dof_expmap.cpp:103-139,188-234 written out as upstream writes them):
// 242 -> 76 instructions per edge and role.  DivBy<false> is the plain division, taken lane by lane for any other denominator.
__device__ __forceinline__ bool plain_den(double d) { const double a = fabs(d); return a > 0x1p-400 && a < 0x1p400; }
template <bool SHARED> struct DivBy {
    double d, r;
    __device__ __forceinline__ explicit DivBy(double den) : d(den), r(0) {
        if (SHARED) {
            r = __builtin_amdgcn_rcp(den);
            r = fma(r, fma(-den, r, 1.0), r);
            r = fma(r, fma(-den, r, 1.0), r);
        }
    }
    __device__ __forceinline__ double operator()(double a) const {
        if (!SHARED) return a / d;
        const double q = a * r;
        return fma(fma(-d, q, a), r, q);
    }
};

// ============================================================================================ SE3 helpers
struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };

__host__ __device__ inline Quat quat_from_matrix(const double m[9]) {
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 * recip(t);
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {   // i = 0 (Eigen picks the largest diagonal; ties go to the lower index)
        t = sqrt(m[0] - m[4] - m[8] + 1.0);
        q.x = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[7] - m[5]) * t; q.y = (m[3] + m[1]) * t; q.z = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {    // i = 1
        t = sqrt(m[4] - m[8] - m[0] + 1.0);
        q.y = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[2] - m[6]) * t; q.z = (m[7] + m[5]) * t; q.x = (m[1] + m[3]) * t;
    } else {                                        // i = 2
        t = sqrt(m[8] - m[0] - m[4] + 1.0);
        q.z = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[3] - m[1]) * t; q.x = (m[2] + m[6]) * t; q.y = (m[5] + m[7]) * t;
    }
    return q;
}
__host__ __device__ inline void quat_normalize_pos(Quat& q) {
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double in = recip(sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w));
    q.x *= in; q.y *= in; q.z *= in; q.w *= in;
}
__host__ __device__ inline Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
__host__ __device__ inline void quat_rotate(const Quat& q, const double v[3], double out[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
__host__ __device__ inline void quat_to_matrix(const Quat& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void se3_map(const SE3& s, const double p[3], double out[3]) {
    quat_rotate(s.r, p, out);
    out[0] += s.t[0]; out[1] += s.t[1]; out[2] += s.t[2];
}
__host__ __device__ inline SE3 se3_exp(const double u[6]) {  // (omega, upsilon), types/se3quat.h:223-259
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double Om2[9], R[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { const double id = (i % 4 == 0) ? 1.0 : 0.0; R[i] = id + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        double st, ct;
#ifdef __HIP_DEVICE_COMPILE__
        // LM steps are small rotations: below half a radian the Taylor polynomials to x^15 / x^16 are exact to the last bit or two
        // (remainder < 2^-70) and cost 17 fused multiply-adds; the library's sincos (argument reduction, two kernels, ~150 fp64
        // instructions on ONE wave while the workgroup waits) only runs for larger angles.
        if (theta < 0.5) {
            const double z = theta * theta;
            double ps = -1.0 / 1307674368000.0;                     // -1/15!
            ps = fma(ps, z, 1.0 / 6227020800.0);                    // +1/13!
            ps = fma(ps, z, -1.0 / 39916800.0);                     // -1/11!
            ps = fma(ps, z, 1.0 / 362880.0);                        // +1/9!
            ps = fma(ps, z, -1.0 / 5040.0);                         // -1/7!
            ps = fma(ps, z, 1.0 / 120.0);                           // +1/5!
            ps = fma(ps, z, -1.0 / 6.0);                            // -1/3!
            st = fma(theta * z, ps, theta);
            double pc = 1.0 / 20922789888000.0;                     // +1/16!
            pc = fma(pc, z, -1.0 / 87178291200.0);                  // -1/14!
            pc = fma(pc, z, 1.0 / 479001600.0);                     // +1/12!
            pc = fma(pc, z, -1.0 / 3628800.0);                      // -1/10!
            pc = fma(pc, z, 1.0 / 40320.0);                         // +1/8!
            pc = fma(pc, z, -1.0 / 720.0);                          // -1/6!
            pc = fma(pc, z, 1.0 / 24.0);                            // +1/4!
            pc = fma(pc, z, -0.5);                                  // -1/2!
            ct = fma(pc, z, 1.0);
        } else
#endif
        sincos(theta, &st, &ct);                // one range reduction for both
        const double it = recip(theta), it2 = it * it;
        const double a = st * it, b = (1 - ct) * it2;
        const double c = (theta - st) * (it2 * it);
        for (int i = 0; i < 9; i++) {
            const double id = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = id + a * Om[i] + b * Om2[i];
            V[i] = id + b * Om[i] + c * Om2[i];
        }
    }
    SE3 s;
    s.r = quat_from_matrix(R);
    for (int i = 0; i < 3; i++) s.t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    quat_normalize_pos(s.r);
    return s;
}
__host__ __device__ inline SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 r;
    double rt[3];
    quat_rotate(a.r, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
    r.r = quat_mul(a.r, b.r);
    quat_normalize_pos(r.r);
    return r;
}
inline SE3 se3_from_Tcw_f32(const float* T) {  // Converter::toSE3Quat, reference src/Converter.cc:28-38
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    SE3 s;
    s.r = quat_from_matrix(R);
    s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
    quat_normalize_pos(s.r);
    return s;
}
inline void se3_to_Tcw_f32(const SE3& s, float* T) {  // Converter::toCvMat(SE3Quat), reference src/Converter.cc:40-59
    double R[9];
    quat_to_matrix(s.r, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)s.t[i];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// 1/x by v_rcp_f64 + two Newton steps (~1 ulp): the solver's pivots
// one Newton step on v_rcp_f64: ~2^-46 relative error (the hardware seed carries single-precision accuracy)
__device__ inline double frcp1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ inline double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ inline void huber(double e, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho0 = e; rho1 = 1.; }
    else { const double s = sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

struct Cam { double fx, fy, cx, cy, bf; float bf_f; double deltaMono, deltaStereo; };

// v + (v of the lane a DPP control selects): the building block of the cross-lane sums below -- VALU only, no LDS round trip
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
// sum over the 8 lanes of an aligned lane group; every lane gets the result.  The same tree as xor-shuffles by 1, 2, 4
// (pairs, quads, then the mirrored quad of the other half), so the sums are bit-identical to those -- but each step is a
// DPP add instead of a ds_bpermute round trip (66 of them per k_ba_linearize before).
__device__ __forceinline__ double group8_sum(double v) {
    v = dpp_add_f64<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp_add_f64<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]: lane ^ 2
    v = dpp_add_f64<0x141, 0xF>(v);    // row_half_mirror: lane i <-> 7 - i of its 8-lane half, i.e. the other quad's sum
    return v;
}

// ---- block-wide fixed-order sum of NV doubles per thread; result valid in thread 0 (and in `out` LDS after a barrier)
__device__ __forceinline__ double quad_sum(double v) {   // sum over the four lanes of a quad, same value in all four
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));          // quad_perm [2,3,0,1]
    return v + o;
}
// Sum over the eight lanes of a 16-lane DPP row that share this lane's parity (valid in lanes 0 and 1 of the row).
__device__ __forceinline__ double row_half_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x124, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, true));        // row_ror:4
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true));        // row_ror:8
    return v + o;
}
// Fixed-order sum of NV accumulators over a block of NT threads through LDS: lane quads first (DPP), then NV x 8 column
// threads over NT/32 quad leaders each, then the last 8.  red: (NT/4)*NV doubles, part: 8*NV doubles; the totals land in
// part[0 .. NV).  (A 64-lane shuffle tree per value costs ~230 cycles per value; this is ~10x cheaper for NV ~ 28.)
template <int NV, int NT, int SEGS = 8>
__device__ inline void block_sum_lds(double (&acc)[NV], double* red, double* part) {
    static_assert(NV * SEGS <= NT && SEGS <= 8 && (NT / 4) % SEGS == 0, "column threads");
#pragma unroll
    for (int q = 0; q < NV; q++) acc[q] = quad_sum(acc[q]);
    if ((threadIdx.x & 3) == 0) {
        double* dst = red + (threadIdx.x >> 2) * NV;
#pragma unroll
        for (int q = 0; q < NV; q++) dst[q] = acc[q];
    }
    __syncthreads();
    constexpr int kSeg = NT / 4 / SEGS;
    if (threadIdx.x < NV * SEGS) {
        const int q = threadIdx.x % NV, seg = threadIdx.x / NV;
        double sacc = 0;
        for (int j = 0; j < kSeg; j++) sacc += red[(seg * kSeg + j) * NV + q];
        part[seg * NV + q] = sacc;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double sacc = 0;
        for (int seg = 0; seg < SEGS; seg++) sacc += part[seg * NV + threadIdx.x];
        part[threadIdx.x] = sacc;                  // only this thread reads or writes these eight slots
    }
    __syncthreads();
}

// Sum over the 64 lanes of a wave on the VALU: DPP quad permutes, half-row and row mirrors, then the row broadcasts 15 / 31;
// the total ends up in lane 63.  (A shuffle tree is six dependent ds_bpermute round trips per value: the single-value sums
// on the LM kernels' critical paths cost ~4.5 k cycles each that way.)
__device__ __forceinline__ double wave_sum_f64_lane63(double v) {
    v = dpp_add_f64<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141, 0xF>(v);    // row_half_mirror
    v = dpp_add_f64<0x140, 0xF>(v);    // row_mirror: every lane of a row holds the row sum
    v = dpp_add_f64<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3
    return v;
}
template <int NV, int NT>
__device__ inline void block_sum(double (&v)[NV], double* lds /* (NT/64)*NV */, double* out /* NV */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const double x = wave_sum_f64_lane63(v[k]);
        if (lane == 63) lds[wv * NV + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int w = 0; w < NT / 64; w++) s += lds[w * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// 6x6 LDLT solve of the damped pose system.  g2o's dense solver (solvers/linear_solver_dense.h:104-112) uses Eigen's
// diagonally pivoted LDLT and reports failure when the matrix is not positive; an unpivoted factorisation has the same
// inertia (Sylvester), hence the same success/failure decision, and the same solution up to rounding -- and it keeps
// every index static (registers, no scratch).
__device__ inline bool ldlt6_solve(const double* A, const double* b, double* x) {
    double a[6][6], inv[6], y[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) a[r][c] = A[r * 6 + c];
    bool positive = true;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const double d = a[r][r];
        if (!(d > 0)) positive = false;
        inv[r] = recip(d);
#pragma unroll
        for (int i = r + 1; i < 6; i++) {
            const double l = a[r][i] * inv[r];
#pragma unroll
            for (int c = i; c < 6; c++) a[i][c] = fma(-l, a[r][c], a[i][c]);
            a[i][r] = l;          // keep the multiplier below the diagonal
        }
    }
    if (!positive) return false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v = fma(-a[i][k], y[k], v);
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i] * inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v = fma(-a[k][i], x[k], v);
        x[i] = v;
    }
    return true;
}

// ------------------------------------------------------------------ planes (src/g2oAddition/Plane3D.h, EdgePlane.h)
// Plane3D keeps (n, -d) normalised with a non-negative fourth coefficient; the edge's error is (azimuth, elevation,
// distance) of the measured plane in the frame that rotates the predicted plane's normal onto +x.
__host__ __device__ inline void plane_normalize(double c[4]) {                 // Plane3D::normalize
    const double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double s = 1. / n;
    for (int k = 0; k < 4; k++) c[k] = c[k] * s;
    if (c[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
}
inline void plane_from_f32(const float* v, double c[4]) {                      // Converter::toPlane3D, src/Converter.cc:215-225
    for (int k = 0; k < 4; k++) c[k] = v[k];
    if (v[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
    plane_normalize(c);
}
// Plane3D::rotation: AngleAxis(azimuth, Z) * AngleAxis(-elevation, Y) as a quaternion product
__device__ inline void plane_rotation(const double* v, double Rn[9]) {
    const double az = atan2(v[1], v[0]);
    const double el = atan2(v[2], sqrt(v[0] * v[0] + v[1] * v[1]));
    const double ha = 0.5 * az, hb = 0.5 * (-el);
    const Quat qa{0, 0, sin(ha) * 1.0, cos(ha)}, qb{0, sin(hb) * 1.0, 0, cos(hb)};
    quat_to_matrix(quat_mul(qa, qb), Rn);
}
// Plane3D::oplus (VertexPlane::oplusImpl, src/g2oAddition/Plane3D.h:73-89, VertexPlane.h:35-38)
__device__ inline void plane_oplus(const double* c, const double* v, double out[4]) {
    const double sn = sin(v[1]), cs = cos(v[1]);
    const double n[3] = {cs * cos(v[0]), cs * sin(v[0]), sn};
    double R[9];
    plane_rotation(c, R);
    const double d = (-c[3]) + v[2];
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = R[r * 3] * n[0] + R[r * 3 + 1] * n[1] + R[r * 3 + 2] * n[2];
    out[3] = -d;
    plane_normalize(out);
}
__device__ inline void plane_error(const SE3& T, const double* world, const double* meas, double err[3]) {   // EdgePlane::computeError
    double R[9];
    quat_to_matrix(T.r, R);
    double v2[4];
#pragma unroll
    for (int r = 0; r < 3; r++) v2[r] = R[r * 3] * world[0] + R[r * 3 + 1] * world[1] + R[r * 3 + 2] * world[2];
    v2[3] = world[3] - (T.t[0] * v2[0] + T.t[1] * v2[1] + T.t[2] * v2[2]);
    if (v2[3] < 0.0) { v2[0] = -v2[0]; v2[1] = -v2[1]; v2[2] = -v2[2]; v2[3] = -v2[3]; }
    plane_normalize(v2);
    double Rn[9];
    plane_rotation(v2, Rn);
    double n[3];
#pragma unroll
    for (int r = 0; r < 3; r++) n[r] = Rn[r] * meas[0] + Rn[3 + r] * meas[1] + Rn[6 + r] * meas[2];   // rotation^T * normal
    err[0] = atan2(n[1], n[0]);
    err[1] = atan2(n[2], sqrt(n[0] * n[0] + n[1] * n[1]));
    err[2] = (-v2[3]) - (-meas[3]);
}
constexpr int kPoseMaxPlanes = 32;

// ============================================================================================ PoseOptimization
// A pointer member of a record that kernels read FROM MEMORY (the window records of the batched LM kernels).  A plain `T*`
// loaded from memory is a generic pointer to the compiler, and every access through it becomes a FLAT instruction: it
// takes the LDS path's counters as well, so the waits behind it are `vmcnt(0) lgkmcnt(0)` instead of counted ones, and it
// cannot use the scalar-base addressing mode.  On the device the member is declared in the global address space (same
// size and layout); converting it to `T*` is then a global -> generic cast the optimiser sees through, and the kernels'
// accesses become GLOBAL instructions with no change at their use sites.  Only device / pinned buffers are ever stored here.
template <typename T> struct GP {
#ifdef __HIP_DEVICE_COMPILE__
    __attribute__((address_space(1))) T* p;
#else
    T* p;
#endif
    __host__ __device__ __forceinline__ operator T*() const { return (T*)p; }
    __host__ __device__ __forceinline__ GP& operator=(T* q) { p = (decltype(p))q; return *this; }
};

struct PoseDev {
    int n;
    GP<const int> nDev;     // when set: the number of edges lives on the device (chained tracking: eao_tracker); n is then the capacity
    int devLo, devHi;       // ... and this launch serves device counts in (devLo, devHi] only (the chain enqueues one launch per register variant)
    GP<const double> Xw;    // n*3
    GP<const double> obs;   // n*3
    GP<const double> info;  // n
    GP<double> err;         // n*3 (last computed residual of each edge, g2o's _error)
    GP<unsigned char> flags;  // bit0 stereo, bit1 level (1 = excluded), bit2 robust kernel present
    GP<unsigned char> outlier;
    SE3 T0;
    Cam cam;
    // outputs
    GP<SE3> Tout;
    GP<int> result;         // [0] nBad of the last round, [1] LM iterations, [2] trace count
    GP<double> trace;       // 3 * 64: lambda, chi2, trials
    GP<long long> dbg;      // optional phase stamps (diagnostic runs of the harness only)
    // plane edges: nPlanes x { world[4], meas[4], infoAngle, infoDistance } doubles; flags as for the points
    int nPlanes;
    GP<const double> planes;
    GP<unsigned char> planeOutlier;
    double deltaPlane;
    GP<const int> scatterIdx;        // chained tracking: edge -> keypoint; the final outlier flags also land in scatterOut by keypoint
    GP<unsigned char> scatterOut;    // (null otherwise)
    GP<int> done; int doneSeq;       // chained tracking: the LAST launch of the chain stores doneSeq here (mapped host memory) when everything
                                     // the chain writes is visible to the host, which polls it instead of paying a stream synchronisation
    // chained tracking: every kernel of the chain (this one included) writes its results into a DEVICE block; the chain's last launch -- and
    // nobody else -- copies that block into its mapped host twin (pubN16 x 16 bytes), fences at system scope and only then stores the done
    // word.  One writer, one fence: the host never depends on how stores of two different kernels to host memory are ordered.
    GP<const uint4> pubSrc; GP<uint4> pubDst; int pubN16;
};
// the tail of the chain's last launch (every thread of the workgroup calls it; all of the workgroup's own stores to the device block are done)
__device__ __forceinline__ void pose_publish(const PoseDev& P) {
    if (!P.done) return;
    __syncthreads();                                  // this workgroup's stores into the device block (other waves') are visible
    const uint4* src = P.pubSrc; uint4* dst = P.pubDst;
    for (int i = threadIdx.x; i < P.pubN16; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();                           // every thread: its stores have left the device ...
    __syncthreads();
    if (threadIdx.x == 0) {
        // ... which is NOT yet "arrived in host memory, in order": the stores are posted PCIe writes, and with the single-writer copy and the
        // fence alone the host still found one stale 16-byte piece of the block behind a fresh done word once in 10^5 calls
        // (profiles/r04_track_poll_stress.txt) -- posted writes may pass each other (relaxed ordering), the word overtook a piece of
        // the data.  A READ of the host block cannot pass posted writes and its completion cannot be returned before they are done
        // (PCIe ordering: non-posted requests push posted ones), so the word is stored only after a load from the block has come back.
        const int seen = __hip_atomic_load((int*)P.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (the host zeroed it before the call)
        if (seen != P.doneSeq) __hip_atomic_store((int*)P.done, P.doneSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

constexpr int kPoseThreads = 512;

__device__ inline void pose_edge_error(const PoseDev& P, const SE3& est, int i, bool stereo) {
    double p[3];
    se3_map(est, &P.Xw[3 * i], p);
    const Cam& c = P.cam;
    if (!stereo) {
        P.err[3 * i] = P.obs[3 * i] - (p[0] / p[2] * c.fx + c.cx);
        P.err[3 * i + 1] = P.obs[3 * i + 1] - (p[1] / p[2] * c.fy + c.cy);
        P.err[3 * i + 2] = 0;
    } else {
        const float invz = (float)(1.0 / p[2]);  // types_six_dof_expmap.cpp:335-342 ("const float invz")
        const double r0 = p[0] * invz * c.fx + c.cx;
        const double r1 = p[1] * invz * c.fy + c.cy;
        const double r2 = r0 - c.bf * invz;
        P.err[3 * i] = P.obs[3 * i] - r0; P.err[3 * i + 1] = P.obs[3 * i + 1] - r1; P.err[3 * i + 2] = P.obs[3 * i + 2] - r2;
    }
}
__device__ inline double pose_edge_chi2(const PoseDev& P, int i, bool stereo) {
    const double w = P.info[i];
    double s = P.err[3 * i] * (w * P.err[3 * i]) + P.err[3 * i + 1] * (w * P.err[3 * i + 1]);
    if (stereo) s += P.err[3 * i + 2] * (w * P.err[3 * i + 2]);
    return s;
}

// ---- sum of 32 per-lane doubles over the 64 lanes of a wave WITHOUT reducing every value through its own tree: in step s the lane
// pairs (L, L ^ 2^s) split the values they still hold -- the lane whose bit s is clear keeps the lower half, its partner the upper
// half, each adds what the other one held of its half -- so the live values halve every step (16 + 8 + 4 + 2 + 1 exchanges instead
// of 6 x 32), and after five steps lane L holds ONE value: the sum, over the lanes that agree with L in bit 5, of value
// bitrev5(L & 31); the sixth step adds the two halves of the wave.  Steps 0 / 1 are DPP quad permutes, the others ds_bpermute.
// A fixed tree: the result does not depend on anything but the 64 x 32 inputs.
__device__ __forceinline__ double dpp_quad_f64(double v, bool xor2) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return xor2 ? __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true))
                : __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));
}
// H += w Omega J_r^T J_r (upper triangle, row-major in acc[0 .. 21)) and b -= J_r^T (w Omega e_r) (acc[21 .. 27)) for ONE Jacobian row whose
// column ZERO is a structural zero; every index is a compile-time constant.
template <int ZERO>
__device__ __forceinline__ void pose_accumulate_row(double (&acc)[32], const double (&J)[6], double wi, double we) {
#pragma unroll
    for (int a = 0; a < 6; a++) {
        if (a == ZERO) continue;
        const double wJ = wi * J[a];
        acc[21 + a] = fma(-J[a], we, acc[21 + a]);
#pragma unroll
        for (int b = a; b < 6; b++) {
            if (b == ZERO) continue;
            const int q = a * 6 - a * (a - 1) / 2 + (b - a);
            acc[q] = fma(wJ, J[b], acc[q]);
        }
    }
}
// Halving steps at lane distance 4 / 8 (inside a 16-lane DPP row) and 16 / 32 (across rows) WITHOUT selects or ds_bpermute (round 5; rounds 3-4 built `keep` / `send`
// with four v_cndmask per pair and fetched `send` through two ds_bpermute round trips).  With A = v[j], B = v[j + M]: the lane whose distance bit is clear
// wants A + A(partner), its partner B + B(partner).
//   * distance 4 / 8: two bank-masked DPP moves per 32-bit half.  X = B overwritten with A(partner) in the banks whose bit is clear, Y = A overwritten with
//     B(partner) in the banks whose bit is set; X + Y is `recv + keep` in the former and `keep + recv` in the latter -- the same sums as before (addition
//     commutes), bit for bit.
//   * distance 16 / 32: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of the first operand with the even rows (lower half)
//     of the second -- after the swap of (A, B) every lane holds `keep` in one register and `recv` in the other.
template <int CTRL_LO, int CTRL_HI, int BANK_LO>      // CTRL_LO: the control that reads lane + distance, CTRL_HI: lane - distance; BANK_LO: banks whose distance bit is clear
__device__ __forceinline__ double dpp_halving_pair(double A, double B) {
    const int al = __double2loint(A), ah = __double2hiint(A), bl = __double2loint(B), bh = __double2hiint(B);
    const double X = __hiloint2double(__builtin_amdgcn_update_dpp(bh, ah, CTRL_LO, 0xF, BANK_LO, false), __builtin_amdgcn_update_dpp(bl, al, CTRL_LO, 0xF, BANK_LO, false));
    const double Y = __hiloint2double(__builtin_amdgcn_update_dpp(ah, bh, CTRL_HI, 0xF, 0xF ^ BANK_LO, false), __builtin_amdgcn_update_dpp(al, bl, CTRL_HI, 0xF, 0xF ^ BANK_LO, false));
    return X + Y;
}
template <bool ROW32>      // (A, B) -> keep + recv across DPP rows (distance 16) or wave halves (distance 32)
__device__ __forceinline__ double swap_halving_pair(double A, double B) {
    const int al = __double2loint(A), ah = __double2hiint(A), bl = __double2loint(B), bh = __double2hiint(B);
    if constexpr (ROW32) {
        const auto lo = __builtin_amdgcn_permlane32_swap(al, bl, false, false), hi = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
        return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap(al, bl, false, false), hi = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
        return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
}
__device__ __forceinline__ double wave_transpose_sum32(double (&v)[32]) {
    const int lane = threadIdx.x & 63;
    {
        const bool up = lane & 1;
#pragma unroll
        for (int j = 0; j < 16; j++) { const double keep = up ? v[j + 16] : v[j], send = up ? v[j] : v[j + 16]; v[j] = keep + dpp_quad_f64(send, false); }
    }
    {
        const bool up = lane & 2;
#pragma unroll
        for (int j = 0; j < 8; j++) { const double keep = up ? v[j + 8] : v[j], send = up ? v[j] : v[j + 8]; v[j] = keep + dpp_quad_f64(send, true); }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = dpp_halving_pair<0x104, 0x114, 0x5>(v[j], v[j + 4]);      // row_shl:4 / row_shr:4, banks 0 and 2
#pragma unroll
    for (int j = 0; j < 2; j++) v[j] = dpp_halving_pair<0x128, 0x128, 0x3>(v[j], v[j + 2]);      // row_ror:8 either way, banks 0 and 1
    v[0] = swap_halving_pair<false>(v[0], v[1]);
    return swap_halving_pair<true>(v[0], v[0]);       // both halves of the wave hold the total
}
// which of the 32 values a lane holds after wave_transpose_sum32
__device__ __forceinline__ int transpose_sum_index(int lane) {
    return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
}
__device__ __forceinline__ double lane_bcast_f64(double v, int src) {      // v of lane `src` in every lane (src wave-uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// exp(dx) * T for the LM's pose update (SE3Quat::exp, types/se3quat.h:223-259, then operator*), specialised for ONE wave computing it
// while a workgroup waits: the rotation as the quaternion (sin(theta/2) / theta * omega, cos(theta/2)) -- what the rotation matrix
// upstream builds and converts back (Eigen's Quaterniond(R)) represents, without the detour (R, trace branch, two square roots); the
// V matrix as upstream writes it (I + b Omega + c Omega^2).  Differs from the matrix route by rounding (~1e-16); the composed quaternion
// is normalised as upstream's operator* does.
__device__ __forceinline__ SE3 se3_exp_mul(const double u[6], const SE3& T) {
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double theta = sqrt(th2);
    double b, cc, sh_over, ch;      // (1 - cos) / theta^2, (theta - sin) / theta^3, sin(theta / 2) / theta, cos(theta / 2)
    if (theta < 0.00001) {          // upstream's small-angle branch: R = I + Omega + Omega^2, V = R
        b = 1.0; cc = 1.0;
        // quaternion of I + Omega + Omega^2 to first order: (omega / 2, 1), normalised below like every result
        sh_over = 0.5; ch = 1.0;
    } else {
        const double h = 0.5 * theta;
        double sh, st, ct;
        if (theta < 0.5) {          // Taylor polynomials on the half angle (remainder < 2^-80): 15 fused multiply-adds
            const double z = h * h;
            double ps = -1.0 / 1307674368000.0;
            ps = fma(ps, z, 1.0 / 6227020800.0); ps = fma(ps, z, -1.0 / 39916800.0); ps = fma(ps, z, 1.0 / 362880.0);
            ps = fma(ps, z, -1.0 / 5040.0); ps = fma(ps, z, 1.0 / 120.0); ps = fma(ps, z, -1.0 / 6.0);
            sh = fma(h * z, ps, h);
            double pc = 1.0 / 20922789888000.0;
            pc = fma(pc, z, -1.0 / 87178291200.0); pc = fma(pc, z, 1.0 / 479001600.0); pc = fma(pc, z, -1.0 / 3628800.0);
            pc = fma(pc, z, 1.0 / 40320.0); pc = fma(pc, z, -1.0 / 720.0); pc = fma(pc, z, 1.0 / 24.0); pc = fma(pc, z, -0.5);
            ch = fma(pc, z, 1.0);
        } else sincos(h, &sh, &ch);
        st = 2.0 * sh * ch;                 // sin(theta), 1 - cos(theta) = 2 sin^2(theta / 2)
        ct = 2.0 * sh * sh;
        const double it = recip(theta), it2 = it * it;
        b = ct * it2;
        cc = (theta - st) * (it2 * it);
        sh_over = sh * it;
    }
    SE3 e;
    e.r.x = sh_over * w0; e.r.y = sh_over * w1; e.r.z = sh_over * w2; e.r.w = ch;
    // V u = u + b (omega x u) + c omega x (omega x u)
    const double ux = u[3], uy = u[4], uz = u[5];
    const double c1x = w1 * uz - w2 * uy, c1y = w2 * ux - w0 * uz, c1z = w0 * uy - w1 * ux;
    const double c2x = w1 * c1z - w2 * c1y, c2y = w2 * c1x - w0 * c1z, c2z = w0 * c1y - w1 * c1x;
    e.t[0] = ux + b * c1x + cc * c2x; e.t[1] = uy + b * c1y + cc * c2y; e.t[2] = uz + b * c1z + cc * c2z;
    return se3_mul(e, T);       // (normalises the product quaternion, sign w >= 0)
}

// Optimizer::PoseOptimization in ONE workgroup (round 3: rewritten around what the cycle stamps showed -- in round 2 two thirds of an LM
// trial were serial sections and barrier hand-overs, and the kernel sat at 256 VGPRs with 500 bytes of scratch).
//   * thread t owns edges t, t + NT, ... (EPT of them) for the whole call -- point, observation, information, flags and the last computed
//     residual (g2o's _error) never leave its registers;
//   * an iteration's FIRST trial is evaluated by a HEAVY pass: residuals, robust chi2 AND the Jacobians / H / b at the trial pose, so an
//     accepted first trial (the common case) IS the next iteration's linear system -- g2o recomputes exactly these values at the top of
//     the next iteration from the same state (optimization_algorithm_levenberg.cpp:64-72);
//   * after a rejected trial the next lambdas are known in advance (lambda *= ni, ni *= 2 while rho < 0, :143-145): up to FOUR retrials
//     are solved side by side, one per wave, and evaluated together by one LIGHT pass (residuals + chi2 only, four poses per edge);
//     the decisions are then taken in upstream's order and stop at the first accepted one (the run of rejections that closes every
//     converged round costs three passes instead of nine).  An accepted retrial is followed by a heavy pass at its pose;
//   * the 28 sums of a heavy pass go through wave_transpose_sum32 and ONE barrier; the control flow that follows (totals, accept / reject,
//     LM bookkeeping) is replicated in every wave from the same LDS totals, so there is no broadcast of decisions -- two barriers per pass;
//   * the plane edges (rare: RGB-D frames with associated PEAC planes) live in a separate instantiation.
// Same LM control flow as the round-2 kernel; the order of the 28 sums differs (a tree over lanes and waves instead of quads / columns /
// segments), products are accumulated with fused multiply-adds, and exp(dx) goes through the quaternion directly.  The single call, the
// batch and the tracker chain share this body and the same edge -> (thread, slot) map, so they agree bit for bit.
constexpr int kPoseCand = 4;      // retrials evaluated per light pass (round 5, measured and not kept: a second batch of up to five -- two solves in wave 0 -- so that a run
                                  // of nine rejections takes two light passes instead of three: the runs that close a round end on rho == 0 long before that, 36 -> 35 passes
                                  // at 1000 correspondences, and the fifth candidate's registers slow every light pass: 0.182 -> 0.193 ms)
template <int EPT, bool PLANES>
__device__ __forceinline__ void pose_lm_fused(const PoseDev& P) {
    constexpr int kMaxWaves = kPoseThreads / 64;
    __shared__ double red[2][kMaxWaves * 32];           // per-wave totals of a pass, double-buffered by pass parity (ONE barrier per heavy pass: a wave that is
                                                        // a pass ahead writes the other buffer while slower waves still add this one up)
    __shared__ double s_sys[kMaxWaves][28];             // per wave: the linear system at the CURRENT estimate -- H (upper triangle, 21), b (6), chi2
    // Round 5: the first trial of an iteration is solved by EVERY wave for itself (same system, same instructions, same result -- a wave per SIMD, so the
    // copies cost no time), which takes barrier (A) and the LDS round trips of the hand-over (publish, copy, read back) out of every heavy pass.  A wave keeps
    // two pose records of its own: the current (last accepted) estimate in s_pose[wv][ei], the trial in the other one -- accepting the trial flips `ei`, nothing is
    // copied -- and carries the trial's rotation matrix in registers into the next pass (the evaluation needs it there anyway).  (Holding the poses themselves
    // in wave-uniform registers was measured first: 28 more SGPRs spill into VGPR lanes, +316 v_readlane in the kernel, 0.183 -> 0.191 ms.)  Only the poses of
    // a LIGHT pass (retrials solved side by side, one per wave) still go through shared records:
    // the candidate records are DOUBLE-BUFFERED by light-pass parity: a wave that has taken its decisions publishes the next light pass's candidates
    // while slower waves still read this one's (nothing but barrier (A) separates the two)
    __shared__ SE3 s_cpose[2][kPoseCand];               // the poses a light pass evaluates
    __shared__ SE3 s_pose[kMaxWaves][2];                // per wave: estimate and trial (see above)
    __shared__ double s_cscale[2][kPoseCand];           // computeScale() of each: sum x (lambda x + b)
    __shared__ int s_cok[2][kPoseCand];                 // did its LDL^T succeed
    __shared__ double s_nb[kMaxWaves];
    // plane edges: thread 13 p + v evaluates plane p at the pose perturbed along variant v (0: none, 1 + 2d / 2 + 2d: +-1e-9
    // along dimension d) -- g2o's central-difference Jacobian (core/base_binary_edge.hpp:131-205); thread p owns the edge
    __shared__ double s_pvar[PLANES ? kPoseMaxPlanes * 13 * 3 : 1];
    __shared__ double s_perr[PLANES ? kPoseMaxPlanes * 3 : 1];          // the edge's _error (last computed)
    __shared__ unsigned char s_pflag[PLANES ? kPoseMaxPlanes : 1], s_pout[PLANES ? kPoseMaxPlanes : 1];
    const int M = PLANES ? P.nPlanes : 0;
    const int NT = blockDim.x, nw = NT >> 6;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = P.nDev ? min(*P.nDev, P.n) : P.n;
    if (P.nDev && (n <= P.devLo || n > P.devHi)) {            // another variant's frame (workgroup-uniform)
        pose_publish(P);      // (the frame's own variant ran in front of this launch and left its results in the device block)
        return;
    }
    const float chi2Mono = refc::POSE_CHI2_MONO, chi2Stereo = refc::POSE_CHI2_STEREO;
    const Cam c = P.cam;
    double eX[EPT][3], eO[EPT][3], eI[EPT], eE[EPT][3];
    unsigned char eF[EPT], eOut[EPT];
    bool eLive[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int i = t + k * NT;
        eLive[k] = i < n;
        const int j = eLive[k] ? i : 0;
#pragma unroll
        for (int d = 0; d < 3; d++) { eX[k][d] = P.Xw[3 * j + d]; eO[k][d] = P.obs[3 * j + d]; eE[k][d] = 0; }
        eI[k] = P.info[j];
        eF[k] = eLive[k] ? P.flags[j] : (unsigned char)2;     // a slot without an edge behaves like a level-1 edge
        eOut[k] = 0;
    }
    // A pose as the kernel maps points with it: the rotation MATRIX of the unit quaternion and the translation, built once per pose and
    // thread (20 instructions) instead of rotating every point through the quaternion product (30 per point; the two agree to rounding).
    struct PoseRt { double R[9], t[3]; };
    auto pose_rt = [&](const SE3& T) { PoseRt m; quat_to_matrix(T.r, m.R); m.t[0] = T.t[0]; m.t[1] = T.t[1]; m.t[2] = T.t[2]; return m; };
    // residual of edge slot k at a pose (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose::computeError); pc = T * Xw, invz = 1 / z.
    // The quotients x / z, y / z (monocular edge) and 1 / z (stereo edge, Jacobians) share ONE refined reciprocal (DivBy: the same
    // correctly rounded quotients as three IEEE divisions for every depth of magnitude 2^-400 .. 2^400, the plain divisions otherwise).
    auto edge_error = [&](const PoseRt& m, int k, double (&pc)[3], double& invz) {
        const bool stereo = eF[k] & 1;
        const double x = eX[k][0], y = eX[k][1], z = eX[k][2];
        pc[0] = fma(m.R[2], z, fma(m.R[1], y, fma(m.R[0], x, m.t[0])));
        pc[1] = fma(m.R[5], z, fma(m.R[4], y, fma(m.R[3], x, m.t[1])));
        pc[2] = fma(m.R[8], z, fma(m.R[7], y, fma(m.R[6], x, m.t[2])));
        double qx, qy;
        if (plain_den(pc[2])) { const DivBy<true> Z(pc[2]); invz = Z(1.0); qx = Z(pc[0]); qy = Z(pc[1]); }
        else { invz = 1.0 / pc[2]; qx = pc[0] / pc[2]; qy = pc[1] / pc[2]; }
        if (!stereo) {
            eE[k][0] = eO[k][0] - (qx * c.fx + c.cx);
            eE[k][1] = eO[k][1] - (qy * c.fy + c.cy);
            eE[k][2] = 0;
        } else {
            const float invzf = (float)invz;  // types_six_dof_expmap.cpp:335-342 ("const float invz")
            const double r0 = pc[0] * invzf * c.fx + c.cx;
            const double r1 = pc[1] * invzf * c.fy + c.cy;
            const double r2 = r0 - c.bf * invzf;
            eE[k][0] = eO[k][0] - r0; eE[k][1] = eO[k][1] - r1; eE[k][2] = eO[k][2] - r2;
        }
    };
    auto edge_chi2 = [&](int k) {
        const double w = eI[k];
        double s2 = eE[k][0] * (w * eE[k][0]) + eE[k][1] * (w * eE[k][1]);
        if (eF[k] & 1) s2 += eE[k][2] * (w * eE[k][2]);
        return s2;
    };
    auto plane_chi2 = [&](int p) {
        const double* pl = P.planes + 10 * p;
        const double* e = s_perr + 3 * p;
        return e[0] * (pl[8] * e[0]) + e[1] * (pl[8] * e[1]) + e[2] * (pl[9] * e[2]);
    };
    if (PLANES && t < M) { s_pflag[t] = 4; s_pout[t] = 0; s_perr[3 * t] = s_perr[3 * t + 1] = s_perr[3 * t + 2] = 0; }
    const bool stamp = P.dbg && t == 0;
    long long pa[6] = {0, 0, 0, 0, 0, 0}, pts = 0;
    auto lap = [&](int slot) { if (stamp) { const long long now = clock64(); pa[slot] += now - pts; pts = now; } };
    const bool wstamp = P.dbg != nullptr;                // per-wave stamps: evaluation (heavy / light) and reduction tree of every wave
    long long wH = 0, wL = 0, wT = 0, wt0 = 0;
    int ntrace = 0, iters = 0, npass = 0, nlight = 0;
    int pb = 0;                                          // parity of the light pass in flight (which candidate buffer it reads)
    int rb = 0;                                          // parity of the pass in flight (which buffer of red[] it fills)
    double* const mySys = s_sys[wv];
    int ei = 0;                                          // which of s_pose[wv][] holds the current estimate (wave-uniform)
    int ci = 0;                                          // ... and which one the heavy pass in flight evaluates (ei: re-linearisation, ei ^ 1: a trial)
    double cScale = 0; int cOk = 0;                      // computeScale() / the LDL^T's success of the trial in s_pose[wv][ei ^ 1]
    // solve (H + lambda I) x = b of THIS wave's copy of the system, x -> trial pose exp(x) * est
    auto solve_candidate = [&](double lambda, SE3& trialOut, double& scOut, int& okOut) {
        eao::wave_sync();               // mySys / s_pose[wv] were written by lanes of this wave
        double A[36], b[6];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b2 = a; b2 < 6; b2++) { const double h = mySys[q]; A[a * 6 + b2] = h; A[b2 * 6 + a] = h; q++; }
#pragma unroll
        for (int a = 0; a < 6; a++) { b[a] = mySys[21 + a]; A[a * 7] += lambda; }
        double x[6] = {0, 0, 0, 0, 0, 0};
        const long long q0 = stamp ? clock64() : 0;
        const int ok = ldlt6_solve(A, b, x) ? 1 : 0;
        double sc = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) sc += x[a] * (lambda * x[a] + b[a]);
        const long long q1 = stamp ? clock64() : 0;
        trialOut = se3_exp_mul(x, s_pose[wv][ei]);
        scOut = sc; okOut = ok;
        if (stamp) { const long long q2 = clock64(); pa[3] += q1 - q0; pa[4] += q2 - q1; pa[5] += 1; }
    };
    auto publish = [&](int buf, int slot, const SE3& T, double sc, int ok) {      // a light pass's candidate, by lane 0 of the calling wave
        if (lane == 0) { s_cpose[buf][slot] = T; s_cscale[buf][slot] = sc; s_cok[buf][slot] = ok; }
    };
    __syncthreads();
    for (int round = 0; round < refc::POSE_ROUNDS; round++) {
        int any = 0;                                  // any level-0 edge?  (g2o: optimize() returns -1 when the index mapping is empty)
#pragma unroll
        for (int k = 0; k < EPT; k++) any |= !(eF[k] & 2);
        if (PLANES && t < M) any |= !(s_pflag[t] & 2);
        if (lane == 0) s_pose[wv][ei] = P.T0;                                           // every round restarts from the frame's pose (src/Optimizer.cc:547)
        ci = ei;
        PoseRt Tm = pose_rt(P.T0);                                                      // rotation matrix + translation of the pose the next heavy pass evaluates
        const int active = __syncthreads_or(any);
        if (active) {
            // ---- the LM of this optimize() call.  Every wave carries the SAME state (wave-uniform registers), derived from the same LDS
            //      totals by the same instructions; only the side effects (trace, result) are wave 0's.
            enum { kInit = 0, kTrial1 = 1, kRetry = 2, kRefresh = 3 };
            int phase = kInit, it = 0, qmax = 0, nbad = 0, nb = 1;      // nb: poses of the pass in flight
            bool heavy = true, needA = false;
            double lambda = 0, ni = 2, curChi = 0, iniChi = 0;
            for (;;) {
                if (needA) { __syncthreads(); pb ^= 1; }                                     // (A) the candidate poses of a light pass are published
                if (stamp) pts = clock64();
                npass++;
                double tot = 0;       // after the pass: lane q of every wave holds total q (heavy: 28 sums; light: chi2 of candidate q)
                if (heavy) {
                    if (wstamp) wt0 = clock64();
                    // ---- computeActiveErrors + activeRobustChi2 + buildSystem at T
                    double acc[32];
#pragma unroll
                    for (int q = 0; q < 32; q++) acc[q] = 0;
                    bool work = false;
#pragma unroll
                    for (int k = 0; k < EPT; k++) {
                        const unsigned char fl = eF[k];
                        if (fl & 2) continue;
                        work = true;
                        const bool stereo = fl & 1;
                        double p[3], invz;
                        edge_error(Tm, k, p, invz);
                        const double c2 = edge_chi2(k);
                        double w = 1.0, r0 = c2;
                        if (fl & 4) huber(c2, stereo ? c.deltaStereo : c.deltaMono, r0, w);
                        acc[27] += r0;
                        // Jacobian rows of the edge (types_six_dof_expmap.cpp:103-139, 188-234), written out as upstream writes them.  Columns
                        // 4 of rows 0 / 2 and 3 of row 1 are structural zeros: their products are left out of the sums below (adding +0.0
                        // changes no bit of a sum), which takes a quarter of the pass's arithmetic away; each remaining product is
                        // accumulated with ONE fused multiply-add (the 28 accumulators were two thirds of the pass).
                        const double X = p[0], Y = p[1], invz2 = invz * invz;
                        double J0[6], J1[6], J2[6];
                        J0[0] = X * Y * invz2 * c.fx; J0[1] = -(1 + (X * X * invz2)) * c.fx; J0[2] = Y * invz * c.fx;
                        J0[3] = -invz * c.fx; J0[4] = 0; J0[5] = X * invz2 * c.fx;
                        J1[0] = (1 + Y * Y * invz2) * c.fy; J1[1] = -X * Y * invz2 * c.fy; J1[2] = -X * invz * c.fy;
                        J1[3] = 0; J1[4] = -invz * c.fy; J1[5] = Y * invz2 * c.fy;
                        if (stereo) {
                            J2[0] = J0[0] - c.bf * Y * invz2; J2[1] = J0[1] + c.bf * X * invz2; J2[2] = J0[2];
                            J2[3] = J0[3]; J2[4] = 0; J2[5] = J0[5] - c.bf * invz2;
                        } else {   // monocular edge: a zero third row keeps every loop static (registers, no scratch)
#pragma unroll
                            for (int a = 0; a < 6; a++) J2[a] = 0;
                        }
                        const double info = eI[k], wi = w * info;
                        pose_accumulate_row<4>(acc, J0, wi, w * (info * eE[k][0]));
                        pose_accumulate_row<3>(acc, J1, wi, w * (info * eE[k][1]));
                        pose_accumulate_row<4>(acc, J2, wi, w * (info * eE[k][2]));
                    }
                    if (PLANES && M) {
                        eao::wave_sync();
                        const SE3 T = s_pose[wv][ci];
                        for (int pq = t; pq < 13 * M; pq += NT) {      // (a launch may have fewer threads than 13 M: 64 for a frame of 40 points)
                            const int p = pq / 13, v = pq - 13 * p;
                            if (!(s_pflag[p] & 2)) {
                                SE3 Tp = T;
                                if (v) {
                                    double add[6] = {0, 0, 0, 0, 0, 0};
                                    const int d = (v - 1) >> 1;
                                    const double step = ((v - 1) & 1) ? -1e-9 : 1e-9;
#pragma unroll
                                    for (int q = 0; q < 6; q++) if (q == d) add[q] = step;
                                    Tp = se3_mul(se3_exp(add), T);
                                }
                                plane_error(Tp, P.planes + 10 * p, P.planes + 10 * p + 4, &s_pvar[(p * 13 + v) * 3]);
                            }
                        }
                        __syncthreads();
                        if (t < M && !(s_pflag[t] & 2)) {
                            work = true;
                            const double* pl = P.planes + 10 * t;
                            const double* pv = &s_pvar[t * 13 * 3];
                            s_perr[3 * t] = pv[0]; s_perr[3 * t + 1] = pv[1]; s_perr[3 * t + 2] = pv[2];
                            const double scalar = 1.0 / (2 * 1e-9);
                            double J[3][6];
#pragma unroll
                            for (int d = 0; d < 6; d++)
#pragma unroll
                                for (int r = 0; r < 3; r++) J[r][d] = scalar * (pv[(1 + 2 * d) * 3 + r] - pv[(2 + 2 * d) * 3 + r]);
                            const double c2 = plane_chi2(t);
                            double w = 1.0, r0 = c2;
                            if (s_pflag[t] & 4) huber(c2, P.deltaPlane, r0, w);
                            acc[27] += r0;
                            const double info[3] = {pl[8], pl[8], pl[9]};
                            int q = 0;
#pragma unroll
                            for (int a = 0; a < 6; a++) {
                                double sb = J[0][a] * (info[0] * pv[0]) + J[1][a] * (info[1] * pv[1]);
                                sb += J[2][a] * (info[2] * pv[2]);
                                acc[21 + a] -= w * sb;
#pragma unroll
                                for (int b = a; b < 6; b++) {
                                    double hh = J[0][a] * (w * info[0]) * J[0][b] + J[1][a] * (w * info[1]) * J[1][b];
                                    hh += J[2][a] * (w * info[2]) * J[2][b];
                                    acc[q++] += hh;
                                }
                            }
                        }
                    }
                    lap(0);
                    if (wstamp) { const long long now = clock64(); wH += now - wt0; wt0 = now; }
                    // ---- the wave's 28 totals (a wave without a level-0 edge contributes zeros without walking the tree)
                    if (__any(work)) {
                        const double ws = wave_transpose_sum32(acc);
                        if (lane < 32) red[rb][wv * 32 + transpose_sum_index(lane)] = ws;
                    } else if (lane < 32) red[rb][wv * 32 + lane] = 0;
                    if (wstamp) wT += clock64() - wt0;
                } else {
                    // ---- LIGHT pass: the robust chi2 of up to four candidate poses (computeActiveErrors + activeRobustChi2 of each, in
                    //      candidate order: the residuals every edge keeps are the LAST candidate's, as after upstream's last trial)
                    nlight++;
                    if (wstamp) wt0 = clock64();
                    double chi[kPoseCand];
#pragma unroll
                    for (int cnd = 0; cnd < kPoseCand; cnd++) {
                        chi[cnd] = 0;
                        if (cnd >= nb) continue;
                        const SE3 T = s_cpose[pb][cnd];
                        const PoseRt Tm = pose_rt(T);
#pragma unroll
                        for (int k = 0; k < EPT; k++) {
                            const unsigned char fl = eF[k];
                            if (fl & 2) continue;
                            double p[3], invz;
                            edge_error(Tm, k, p, invz);
                            const double c2 = edge_chi2(k);
                            double w, r0 = c2;
                            if (fl & 4) huber(c2, (fl & 1) ? c.deltaStereo : c.deltaMono, r0, w);
                            chi[cnd] += r0;
                        }
                        if (PLANES && t < M && !(s_pflag[t] & 2)) {
                            plane_error(T, P.planes + 10 * t, P.planes + 10 * t + 4, &s_perr[3 * t]);
                            const double c2 = plane_chi2(t);
                            double w, r0 = c2;
                            if (s_pflag[t] & 4) huber(c2, P.deltaPlane, r0, w);
                            chi[cnd] += r0;
                        }
                    }
                    lap(0);
                    if (wstamp) wL += clock64() - wt0;
#pragma unroll
                    for (int cnd = 0; cnd < kPoseCand; cnd++) {
                        if (cnd >= nb) continue;
                        const double ws = wave_sum_f64_lane63(chi[cnd]);
                        if (lane == 63) red[rb][wv * 32 + cnd] = ws;
                    }
                }
                __syncthreads();                                                             // (B) red[] is complete
                lap(1);
                // ---- totals: lane q adds value q of the waves in wave order (every wave does, for itself)
                if (lane < 28)
                    for (int w = 0; w < nw; w++) tot += red[rb][w * 32 + lane];
                rb ^= 1;
                bool solve1 = false, done = false, toInit = false, toRefresh = false;
                int refreshFrom = 0;
                if (phase == kInit) {
                    // the pass linearised the current estimate (first pass of the round, after an accepted retrial, or g2o's recomputation
                    // after an iteration that ended on a rejected trial)
                    if (lane < 28) mySys[lane] = tot;
                    curChi = lane_bcast_f64(tot, 27);
                    if (it == 0) {      // lambda_0 = tau * max |diagonal| (:166-180), at the first iteration of every optimize()
                        double md = 0;
                        md = fmax(md, fabs(lane_bcast_f64(tot, 0))); md = fmax(md, fabs(lane_bcast_f64(tot, 6))); md = fmax(md, fabs(lane_bcast_f64(tot, 11)));
                        md = fmax(md, fabs(lane_bcast_f64(tot, 15))); md = fmax(md, fabs(lane_bcast_f64(tot, 18))); md = fmax(md, fabs(lane_bcast_f64(tot, 20)));
                        lambda = refc::LM_TAU * md;
                        ni = 2;
                        nbad = 0;
                    }
                    iniChi = curChi;
                    qmax = 0;
                    solve1 = true;
                } else if (phase == kRefresh) {
                    done = true;        // (the pass only restored the residuals of the trial the LM stopped on)
                } else {
                    // ---- decisions on the candidates just evaluated, in upstream's order (optimization_algorithm_levenberg.cpp:118-149)
                    bool closed = false, accepted = false;
                    double r = 0;
                    int cnd = 0;
                    for (; cnd < nb; cnd++) {
                        const double chiT = lane_bcast_f64(tot, phase == kTrial1 ? 27 : cnd);
                        const double tempChi = (phase == kTrial1 ? cOk : s_cok[pb][cnd]) ? chiT : DBL_MAX;
                        r = curChi - tempChi;
                        r /= (phase == kTrial1 ? cScale : s_cscale[pb][cnd]) + 1e-3;
                        qmax++;
                        if (r > 0 && isfinite(tempChi)) {
                            const double y = 2 * r - 1;
                            double alpha = 1. - y * y * y;
                            alpha = fmin(alpha, 2. / 3.);
                            lambda *= fmax(1. / 3., alpha);
                            ni = 2;
                            curChi = tempChi;
                            if (phase == kTrial1) {
                                ei ^= 1;                                                        // est = this trial
                                if (lane < 28) mySys[lane] = tot;                               // its system is the next iteration's
                            } else {                                                            // est = this retrial (seven lanes copy the record)
                                eao::wave_sync();
                                if (lane < 7) reinterpret_cast<double*>(&s_pose[wv][ei])[lane] = reinterpret_cast<const double*>(&s_cpose[pb][cnd])[lane];
                            }
                            accepted = true;
                            closed = true;
                            break;
                        }
                        lambda *= ni;
                        ni *= 2;
                        if (!(r < 0 && qmax < refc::LM_MAX_TRIALS)) { closed = true; break; }
                    }
                    if (!closed) {
                        // every candidate was rejected with rho < 0 and trials remain: the next batch of retrials (their lambdas are known)
                        phase = kRetry; heavy = false;
                        nb = min(min(kPoseCand, nw), refc::LM_MAX_TRIALS - qmax);
                        needA = true;
                        if (wv < nb) {
                            double lc = lambda, nc = ni;
                            for (int k = 0; k < wv; k++) { lc *= nc; nc *= 2; }
                            SE3 tr; double sc; int ok;
                            solve_candidate(lc, tr, sc, ok);
                            publish(pb ^ 1, wv, tr, sc, ok);
                        }
                    } else {
                        // ---- the iteration closes here
                        if (wv == 0 && lane == 0 && ntrace < 64) { P.trace[ntrace] = lambda; P.trace[64 + ntrace] = curChi; P.trace[128 + ntrace] = qmax; }
                        if (ntrace < 64) ntrace++;
                        iters++;
                        bool term = qmax == refc::LM_MAX_TRIALS || r == 0;
                        if (!term) {
                            if ((iniChi - curChi) * 1e3 < iniChi) nbad++; else nbad = 0;
                            if (nbad >= 3) term = true;
                        }
                        it++;
                        const bool stop = term || it >= refc::POSE_ITS;
                        // the residuals the edges keep must be those of the LAST trial upstream evaluated: candidate `cnd` -- a light pass ends
                        // on its last candidate, so a run that stops earlier WITHOUT continuing re-evaluates that one pose
                        const bool stale = phase == kRetry && cnd < nb - 1;
                        if (stop) {
                            if (stale) { toRefresh = true; refreshFrom = cnd; }
                            else done = true;
                        } else if (!accepted) toInit = true;        // (only a NaN rho ends an iteration on a rejected trial without terminating:
                                                                    //  g2o then recomputes errors and system at the restored estimate)
                        else if (phase == kTrial1) { iniChi = curChi; qmax = 0; solve1 = true; }      // the heavy pass already linearised the new estimate
                        else toInit = true;                         // an accepted retrial: linearise its pose
                    }
                }
                if (solve1) {
                    phase = kTrial1; heavy = true; nb = 1; needA = false;
                    SE3 tr;
                    solve_candidate(lambda, tr, cScale, cOk);                               // every wave, for itself
                    ci = ei ^ 1;
                    if (lane == 0) s_pose[wv][ci] = tr;
                    Tm = pose_rt(tr);
                } else if (toInit) {
                    phase = kInit; heavy = true; nb = 1; needA = false;
                    eao::wave_sync();
                    ci = ei;
                    Tm = pose_rt(s_pose[wv][ei]);
                } else if (toRefresh) {
                    phase = kRefresh; heavy = false; nb = 1; needA = true;
                    if (wv == 0) { const SE3 Tr = s_cpose[pb][refreshFrom]; publish(pb ^ 1, 0, Tr, 0, 1); }
                }
                lap(2);
                if (done) break;
            }
        }
        // ---- inlier / outlier classification of every edge (src/Optimizer.cc:554-621) at the round's estimate
        eao::wave_sync();
        const SE3 fin = s_pose[wv][ei];
        const PoseRt finm = pose_rt(fin);
        double nbo = 0;
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            if (!eLive[k]) continue;
            unsigned char fl = eF[k];
            const bool stereo = fl & 1;
            if (eOut[k]) { double p[3], invz; edge_error(finm, k, p, invz); }
            const float c2 = (float)edge_chi2(k);
            if (c2 > (stereo ? chi2Stereo : chi2Mono)) { eOut[k] = 1; fl |= 2; nbo += 1; }
            else { eOut[k] = 0; fl &= ~2; }
            if (!stereo) fl &= ~4;            // mono: kernel removed after every round
            else if (round == refc::POSE_UNROBUST_ROUND) fl &= ~4;    // stereo: at it == 2
            eF[k] = fl;
        }
        if (PLANES && t < M) {   // src/Optimizer.cc:626-658
            unsigned char fl = s_pflag[t];
            if (s_pout[t]) plane_error(fin, P.planes + 10 * t, P.planes + 10 * t + 4, &s_perr[3 * t]);
            const float c2 = (float)plane_chi2(t);
            if (c2 > refc::PLANE_CHI2) { s_pout[t] = 1; fl |= 2; nbo += 1; }
            else { s_pout[t] = 0; fl &= ~2; }
            if (round == refc::POSE_UNROBUST_ROUND) fl &= ~4;
            s_pflag[t] = fl;
        }
        {
            const double xs = wave_sum_f64_lane63(nbo);
            if (lane == 63) s_nb[wv] = xs;
        }
        __syncthreads();
        if (t == 0) {
            double totb = 0;
            for (int w = 0; w < nw; w++) totb += s_nb[w];
            P.result[0] = (int)totb;
            *P.Tout = fin;
        }
        if (n + M < 10) break;
    }
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (eLive[k]) {
            P.outlier[t + k * NT] = eOut[k];
            if (P.scatterOut) P.scatterOut[P.scatterIdx[t + k * NT]] = eOut[k];
        }
    if (PLANES && t < M) P.planeOutlier[t] = s_pout[t];
    if (t == 0) { P.result[1] = iters; P.result[2] = ntrace; P.result[3] = npass | (nlight << 16); }
    pose_publish(P);         // (behind a barrier: the copy reads what other threads of this workgroup stored -- outlier flags by keypoint, the pose, the counts)
    if (stamp) for (int q = 0; q < 6; q++) P.dbg[q] = pa[q];
    if (wstamp && lane == 0) { P.dbg[8 + wv] = wH; P.dbg[24 + wv] = wL; P.dbg[40 + wv] = wT; if (wv == 0) P.dbg[7] = nw; }
}
// MAXT = the launch's thread count bound: the four-wave classes are compiled for ONE wave per SIMD, i.e. with the SIMD's whole
// register file (256 VGPRs + 256 AGPRs) per wave -- what does not fit the VGPRs is parked in AGPRs, not in scratch memory.
template <int EPT, bool PLANES, int MAXT>
__global__ __launch_bounds__(MAXT) void k_pose_optimization(PoseDev P) { pose_lm_fused<EPT, PLANES>(P); }
// One workgroup per frame: eao_pose_optimization_batch (the record is read through scalar loads, never copied to registers).
template <int EPT, bool PLANES, int MAXT>
__global__ __launch_bounds__(MAXT) void k_pose_optimization_batch(const PoseDev* __restrict__ W) {
    pose_lm_fused<EPT, PLANES>(W[blockIdx.x]);
}
// Launch geometry of the register kernels.  A lone wave issues an fp64 instruction every 5.45 cycles, the SIMD's fp64 pipe takes one every
// 4.3 (tools/ubench/f64_simd.hip, profiles/r03_ubench_f64.txt): one wave per SIMD (four per workgroup) is within 20 % of what the CU's
// fp64 pipes can do, and the per-wave stamps of the evaluation say the same (tools/dbg_pose_waves.py, 1000 correspondences, 23 heavy
// passes: four waves x 4 edges 114 k ticks; eight waves x 2 edges: waves 0-3 59 k, waves 4-7 -- the younger wave of every SIMD, served
// when the older one has nothing to issue -- 108 k).  Every further wave adds its own reduction tree.  Hence FOUR waves with up to four
// edges per thread for frames of up to 1024 correspondences (edge i lives in thread i mod 256, slot i / 256), eight waves x four edges beyond.
constexpr int kPoseWaves4 = 256;
inline int pose_threads(int n) { return n <= 4 * kPoseWaves4 ? std::min(kPoseWaves4, std::max(64, (n + 63) / 64 * 64)) : kPoseThreads; }
inline int pose_ept(int n) { return n <= kPoseWaves4 ? 1 : (n <= 2 * kPoseWaves4 ? 2 : 4); }
inline int pose_class(int n) { return n <= kPoseWaves4 ? 0 : n <= 2 * kPoseWaves4 ? 1 : n <= 4 * kPoseWaves4 ? 2 : 3; }
inline void launch_pose_registers(const PoseDev& P, int n, bool planes, hipStream_t s) {
    static const bool env8 = getenv("EAO_POSE_WAVES") && atoi(getenv("EAO_POSE_WAVES")) == 8;      // A/B switch: eight waves, fewer edges per thread
    if (env8 && !planes && n <= 4 * kPoseThreads) {
        const dim3 b8(kPoseThreads);
        if (n <= kPoseThreads) hipLaunchKernelGGL((k_pose_optimization<1, false, kPoseThreads>), dim3(1), b8, 0, s, P);
        else if (n <= 2 * kPoseThreads) hipLaunchKernelGGL((k_pose_optimization<2, false, kPoseThreads>), dim3(1), b8, 0, s, P);
        else hipLaunchKernelGGL((k_pose_optimization<4, false, kPoseThreads>), dim3(1), b8, 0, s, P);
        return;
    }
    const dim3 b(planes && n <= 4 * kPoseWaves4 ? std::max(pose_threads(n), std::min(kPoseWaves4, (13 * P.nPlanes + 63) / 64 * 64)) : pose_threads(n));
    switch (pose_class(n) + (planes ? 4 : 0)) {
        case 0: hipLaunchKernelGGL((k_pose_optimization<1, false, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        case 1: hipLaunchKernelGGL((k_pose_optimization<2, false, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        case 2: hipLaunchKernelGGL((k_pose_optimization<4, false, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        case 3: hipLaunchKernelGGL((k_pose_optimization<4, false, kPoseThreads>), dim3(1), b, 0, s, P); break;
        case 4: hipLaunchKernelGGL((k_pose_optimization<1, true, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        case 5: hipLaunchKernelGGL((k_pose_optimization<2, true, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        case 6: hipLaunchKernelGGL((k_pose_optimization<4, true, kPoseWaves4>), dim3(1), b, 0, s, P); break;
        default: hipLaunchKernelGGL((k_pose_optimization<4, true, kPoseThreads>), dim3(1), b, 0, s, P); break;
    }
}

// Generic variant: edges stay in global memory (frames with more than 4 * kPoseThreads correspondences).
__global__ __launch_bounds__(kPoseThreads) void k_pose_optimization_mem(PoseDev P) {
    __shared__ double red[(kPoseThreads / 4) * 28];
    __shared__ double sums[8 * 28];
    __shared__ SE3 s_est, s_backup;
    __shared__ double s_x[6];
    __shared__ double s_lambda, s_ni, s_rho, s_cur, s_tmp;
    __shared__ int s_ok, s_flag, s_nbad, s_ntrace, s_iters, s_active;
    // plane edges exactly as in the register variant: thread 13 p + v evaluates plane p at the pose perturbed along variant v
    // (g2o's central-difference Jacobian, core/base_binary_edge.hpp:131-205); thread p owns the edge
    __shared__ double s_pvar[kPoseMaxPlanes * 13 * 3];
    __shared__ double s_perr[kPoseMaxPlanes * 3];
    __shared__ unsigned char s_pflag[kPoseMaxPlanes], s_pout[kPoseMaxPlanes];
    const int M = P.nPlanes;
    const int t = threadIdx.x, n = P.nDev ? min(*P.nDev, P.n) : P.n;
    const float chi2Mono = refc::POSE_CHI2_MONO, chi2Stereo = refc::POSE_CHI2_STEREO;
    if (t == 0) { s_ntrace = 0; s_iters = 0; s_est = P.T0; }
    if (t < M) { s_pflag[t] = 4; s_pout[t] = 0; s_perr[3 * t] = s_perr[3 * t + 1] = s_perr[3 * t + 2] = 0; }
    auto plane_chi2 = [&](int p) {
        const double* pl = P.planes + 10 * p;
        const double* e = s_perr + 3 * p;
        return e[0] * (pl[8] * e[0]) + e[1] * (pl[8] * e[1]) + e[2] * (pl[9] * e[2]);
    };
    __syncthreads();
    for (int round = 0; round < refc::POSE_ROUNDS; round++) {
        if (t == 0) { s_est = P.T0; s_active = 0; }
        __syncthreads();
        {   // any level-0 edge?  (g2o: optimize() returns -1 when the index mapping is empty)
            int any = 0;
            for (int i = t; i < n; i += kPoseThreads) any |= !(P.flags[i] & 2);
            if (t < M) any |= !(s_pflag[t] & 2);
            if (any) s_active = 1;
        }
        __syncthreads();
        const int active = s_active;
        int lmNBad = 0;  // uniform copies of the LM state that only thread 0 updates live in LDS
        if (active) {
            bool ok = true;
            for (int it = 0; it < refc::POSE_ITS && ok; it++) {
                // ---- computeActiveErrors + activeRobustChi2 + buildSystem at the current estimate
                const SE3 est = s_est;
                double acc[28];
#pragma unroll
                for (int k = 0; k < 28; k++) acc[k] = 0;
                for (int i = t; i < n; i += kPoseThreads) {
                    const unsigned char fl = P.flags[i];
                    if (fl & 2) continue;
                    const bool stereo = fl & 1;
                    pose_edge_error(P, est, i, stereo);
                    const double c2 = pose_edge_chi2(P, i, stereo);
                    double w = 1.0, r0 = c2;
                    if (fl & 4) huber(c2, stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
                    acc[27] += r0;
                    double p[3];
                    se3_map(est, &P.Xw[3 * i], p);
                    const Cam& c = P.cam;
                    const double X = p[0], Y = p[1], invz = 1.0 / p[2], invz2 = invz * invz;
                    double J[3][6];
                    J[0][0] = X * Y * invz2 * c.fx; J[0][1] = -(1 + (X * X * invz2)) * c.fx; J[0][2] = Y * invz * c.fx;
                    J[0][3] = -invz * c.fx; J[0][4] = 0; J[0][5] = X * invz2 * c.fx;
                    J[1][0] = (1 + Y * Y * invz2) * c.fy; J[1][1] = -X * Y * invz2 * c.fy; J[1][2] = -X * invz * c.fy;
                    J[1][3] = 0; J[1][4] = -invz * c.fy; J[1][5] = Y * invz2 * c.fy;
                    if (stereo) {
                        J[2][0] = J[0][0] - c.bf * Y * invz2; J[2][1] = J[0][1] + c.bf * X * invz2; J[2][2] = J[0][2];
                        J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - c.bf * invz2;
                    } else {   // monocular edge: a zero third row keeps every loop static (registers, no scratch)
#pragma unroll
                        for (int a = 0; a < 6; a++) J[2][a] = 0;
                    }
                    const double info = P.info[i], wi = w * info;
                    const double e0 = info * P.err[3 * i], e1 = info * P.err[3 * i + 1], e2 = info * P.err[3 * i + 2];
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        double sb = J[0][a] * e0 + J[1][a] * e1;
                        sb += J[2][a] * e2;
                        acc[21 + a] -= w * sb;
#pragma unroll
                        for (int b = a; b < 6; b++) {
                            double hh = J[0][a] * wi * J[0][b] + J[1][a] * wi * J[1][b];
                            hh += J[2][a] * wi * J[2][b];
                            acc[q++] += hh;
                        }
                    }
                }
                if (M) {
                    if (t < 13 * M) {
                        const int p = t / 13, v = t - 13 * p;
                        if (!(s_pflag[p] & 2)) {
                            SE3 Tp = est;
                            if (v) {
                                double add[6] = {0, 0, 0, 0, 0, 0};
                                const int d = (v - 1) >> 1;
                                const double step = ((v - 1) & 1) ? -1e-9 : 1e-9;
#pragma unroll
                                for (int q = 0; q < 6; q++) if (q == d) add[q] = step;
                                Tp = se3_mul(se3_exp(add), est);
                            }
                            plane_error(Tp, P.planes + 10 * p, P.planes + 10 * p + 4, &s_pvar[(p * 13 + v) * 3]);
                        }
                    }
                    __syncthreads();
                    if (t < M && !(s_pflag[t] & 2)) {
                        const double* pl = P.planes + 10 * t;
                        const double* pv = &s_pvar[t * 13 * 3];
                        s_perr[3 * t] = pv[0]; s_perr[3 * t + 1] = pv[1]; s_perr[3 * t + 2] = pv[2];
                        const double scalar = 1.0 / (2 * 1e-9);
                        double J[3][6];
#pragma unroll
                        for (int d = 0; d < 6; d++)
#pragma unroll
                            for (int r = 0; r < 3; r++) J[r][d] = scalar * (pv[(1 + 2 * d) * 3 + r] - pv[(2 + 2 * d) * 3 + r]);
                        const double c2 = plane_chi2(t);
                        double w = 1.0, r0 = c2;
                        if (s_pflag[t] & 4) huber(c2, P.deltaPlane, r0, w);
                        acc[27] += r0;
                        const double info[3] = {pl[8], pl[8], pl[9]};
                        int q = 0;
#pragma unroll
                        for (int a = 0; a < 6; a++) {
                            double sb = J[0][a] * (info[0] * pv[0]) + J[1][a] * (info[1] * pv[1]);
                            sb += J[2][a] * (info[2] * pv[2]);
                            acc[21 + a] -= w * sb;
#pragma unroll
                            for (int b = a; b < 6; b++) {
                                double hh = J[0][a] * (w * info[0]) * J[0][b] + J[1][a] * (w * info[1]) * J[1][b];
                                hh += J[2][a] * (w * info[2]) * J[2][b];
                                acc[q++] += hh;
                            }
                        }
                    }
                }
                block_sum_lds<28, kPoseThreads>(acc, red, sums);
                if (t == 0) {
                    s_cur = sums[27];
                    if (it == 0) {
                        double md = 0;
                        int q = 0;
                        for (int a = 0; a < 6; a++) { md = fmax(md, fabs(sums[q])); q += 6 - a; }
                        s_lambda = refc::LM_TAU * md;
                        s_ni = 2;
                        s_nbad = 0;
                    }
                }
                __syncthreads();
                const double iniChi = s_cur;
                int qmax = 0;
                double rho = 0;
                do {
                        if (t == 0) {
                        s_backup = s_est;
                        double A[36], b[6];
                        int q = 0;
                        for (int a = 0; a < 6; a++)
                            for (int b2 = a; b2 < 6; b2++) { A[a * 6 + b2] = sums[q]; A[b2 * 6 + a] = sums[q]; q++; }
                        for (int a = 0; a < 6; a++) { A[a * 7] += s_lambda; b[a] = sums[21 + a]; }
                        double x[6] = {0, 0, 0, 0, 0, 0};
                        s_ok = ldlt6_solve(A, b, x) ? 1 : 0;
                        for (int a = 0; a < 6; a++) s_x[a] = x[a];
                        s_est = se3_mul(se3_exp(x), s_est);
                    }
                    __syncthreads();
                    const SE3 tr = s_est;
                    double chi[1] = {0};
                    for (int i = t; i < n; i += kPoseThreads) {
                        const unsigned char fl = P.flags[i];
                        if (fl & 2) continue;
                        const bool stereo = fl & 1;
                        pose_edge_error(P, tr, i, stereo);
                        const double c2 = pose_edge_chi2(P, i, stereo);
                        double w, r0 = c2;
                        if (fl & 4) huber(c2, stereo ? P.cam.deltaStereo : P.cam.deltaMono, r0, w);
                        chi[0] += r0;
                    }
                    if (t < M && !(s_pflag[t] & 2)) {
                        plane_error(tr, P.planes + 10 * t, P.planes + 10 * t + 4, &s_perr[3 * t]);
                        const double c2 = plane_chi2(t);
                        double w, r0 = c2;
                        if (s_pflag[t] & 4) huber(c2, P.deltaPlane, r0, w);
                        chi[0] += r0;
                    }
                    block_sum<1, kPoseThreads>(chi, red, &s_tmp);
                    if (t == 0) {
                        double tempChi = s_tmp;
                        if (!s_ok) tempChi = DBL_MAX;
                        double r = s_cur - tempChi;
                        double scale = 0;
                        for (int a = 0; a < 6; a++) scale += s_x[a] * (s_lambda * s_x[a] + sums[21 + a]);
                        scale += 1e-3;
                        r /= scale;
                        if (r > 0 && isfinite(tempChi)) {
                            const double y = 2 * r - 1;
                            double alpha = 1. - y * y * y;
                            alpha = fmin(alpha, 2. / 3.);
                            s_lambda *= fmax(1. / 3., alpha);
                            s_ni = 2;
                            s_cur = tempChi;
                        } else {
                            s_lambda *= s_ni;
                            s_ni *= 2;
                            s_est = s_backup;
                        }
                        s_rho = r;
                    }
                    __syncthreads();
                    rho = s_rho;
                    qmax++;
                } while (rho < 0 && qmax < refc::LM_MAX_TRIALS);
                if (t == 0) {
                    if (s_ntrace < 64) { P.trace[s_ntrace] = s_lambda; P.trace[64 + s_ntrace] = s_cur; P.trace[128 + s_ntrace] = qmax; s_ntrace++; }
                    s_iters++;
                    int term = (qmax == refc::LM_MAX_TRIALS || rho == 0) ? 1 : 0;
                    if (!term) {
                        if ((iniChi - s_cur) * 1e3 < iniChi) s_nbad++; else s_nbad = 0;
                        if (s_nbad >= 3) term = 1;
                    }
                    s_flag = term;
                }
                __syncthreads();
                ok = !s_flag;
                (void)lmNBad;
            }
        }
        // ---- inlier / outlier classification of every edge (src/Optimizer.cc:554-621)
        const SE3 est = s_est;
        double nb[1] = {0};
        for (int i = t; i < n; i += kPoseThreads) {
            unsigned char fl = P.flags[i];
            const bool stereo = fl & 1;
            if (P.outlier[i]) pose_edge_error(P, est, i, stereo);
            const float c2 = (float)pose_edge_chi2(P, i, stereo);
            if (c2 > (stereo ? chi2Stereo : chi2Mono)) { P.outlier[i] = 1; fl |= 2; nb[0] += 1; }
            else { P.outlier[i] = 0; fl &= ~2; }
            if (!stereo) fl &= ~4;            // mono: kernel removed after every round
            else if (round == refc::POSE_UNROBUST_ROUND) fl &= ~4;    // stereo: at it == 2
            P.flags[i] = fl;
        }
        if (t < M) {   // src/Optimizer.cc:626-658
            unsigned char fl = s_pflag[t];
            if (s_pout[t]) plane_error(est, P.planes + 10 * t, P.planes + 10 * t + 4, &s_perr[3 * t]);
            const float c2 = (float)plane_chi2(t);
            if (c2 > refc::PLANE_CHI2) { s_pout[t] = 1; fl |= 2; nb[0] += 1; }
            else { s_pout[t] = 0; fl &= ~2; }
            if (round == refc::POSE_UNROBUST_ROUND) fl &= ~4;
            s_pflag[t] = fl;
        }
        block_sum<1, kPoseThreads>(nb, red, &s_tmp);
        if (t == 0) P.result[0] = (int)s_tmp;
        __syncthreads();
        if (n + M < 10) break;
    }
    if (t < M) P.planeOutlier[t] = s_pout[t];
    if (t == 0) { *P.Tout = s_est; P.result[1] = s_iters; P.result[2] = s_ntrace; }
}

// ============================================================================================ LocalBundleAdjustment
struct BADev {
    int nCams, nPts, nEdges, nFree, nL;   // nFree / nL: active free cameras / active points of the current pass
    Cam cam;
    // problem (device)
    GP<const float> obs;       // E*3 as handed over (promoted to double where used, exactly like Converter / Eigen)
    GP<const float> info;      // E
    GP<const int> ecam;        // E
    GP<const int> ept;         // E
    GP<unsigned char> eflag;   // bit0 stereo, bit1 inactive (level 1: set on the device by the outlier pass), bit2 robust
    GP<const int> camIdx;      // nCams -> free block index or -1
    GP<const int> ptIdx;       // nPts  -> landmark block index or -1
    GP<const int> actCam;      // nFree -> camera
    GP<const int> actPt;       // nL    -> point
    // adjacency of the edges that were active when the window was set up; kernels skip edges whose bit1 was set since
    GP<const int> ptStart;     // nL+1   CSR by landmark block: all active edges of the point, insertion order
    GP<const int> ptEdges;
    GP<const int> camStart;    // nFree+1 CSR by free camera block
    GP<const int> camEdges;
    GP<int> camEdgeL;          // landmark block of each camEdges entry (resolved by k_ba_prepare)
    GP<int> pairCnt;           // per camera pair (i1 <= i2): number of landmarks both observe ...
    GP<int> pairPts;           // ... and their landmark blocks, ascending, nL slots per pair (k_ba_pairs)
    GP<int> table;             // nL * nFree: edge id of (point, free camera) or -1 (built and maintained on the device)
    GP<int4> slot;             // nL * 8: {edge, camera, free-camera index, point | more-than-8-edges << 31} of the landmark's k-th edge, edge = -1
                            // beyond its last one (k_ba_prepare): what the eight lanes of a landmark would otherwise chase through
                            // ptStart -> ptEdges -> ecam -> camIdx, four dependent round trips at the head of every launch
    // state: two buffers; ctl[1] says which one holds the current estimate, the other receives the trial
    GP<SE3> camsBuf[2];
    GP<double> ptsBuf[2];
    GP<int> ctl;               // [0] halt  [1] current buffer  [2] iterations done  [3] status  [4] nBad
    GP<double> err;            // E*3, last computed
    // system
    GP<double> Hpp;            // nFree*36
    GP<double> bp;             // nFree*6
    GP<double> Hll;            // nL*9
    GP<double> bl;             // nL*3
    GP<double> Hpl;            // E*18 (pose row block 6x3) for active edges with a free camera.  wmode = 1: the block is stored PRE-SCALED,
                               // W = Hpl C^-T with C C^T = Hll + lambda I of its landmark (see ba_chol3), so that Hpl (Hll + lambda I)^-1 Hpl'^T = W W'^T
    int wmode;                 // 1: windows on the matrix-core pair assembly (k_ba_schur_pairs_mfma): W / Tl / ul instead of Hpl / dinv3
    GP<double> Tl;             // nL*6: T = C^-T (upper triangle 00 01 02 11 12 22) per landmark block   (wmode 1)
    GP<double> ul;             // nL*3: u = C^-1 bl                                                     (wmode 1)
    GP<int4> pairItems;        // per camera pair (i1 <= i2), nL slots: {144 * edge (l, i1), 144 * edge (l, i2), 24 * l, l} of the landmarks both observe, ascending (k_ba_pairs)
    GP<double> sys;            // n*(n+1): assembled Schur system (upper triangle + rhs column)
    GP<int> doneCnt;           // workgroups of the running k_ba_backsub that have published their partial sums (zero between launches)
    GP<double> wgPart;         // their partial sums: 2 per workgroup
    GP<long long> dbg;         // optional phase stamps of k_ba_solve (diagnostic builds of the harness only)
    GP<double> xp;             // nFree*6
    GP<double> xl;             // nL*3
    GP<double> partChi;        // nL (robust chi2 of the point's edges at the last evaluated state)
    GP<double> partScale;      // nL
    GP<double> lm;             // [0] lambda [1] ni [2] currentChi [3] maxdiag
    // MapPlane vertices / EdgePlane edges of Optimizer::BundleAdjustment (src/Optimizer.cc:203-252): landmarks nPtsOnly.. are
    // planes (4 coefficients each, two state buffers like the points), edges nEdgesPt.. are plane edges (eflag bit3)
    int nPtsOnly, nEdgesPt;
    GP<double> plBuf[2];
    GP<const double> pmeas;    // 4 per plane edge: the measured plane, normalised
    double deltaPlane, infoAngle, infoDist;
    // map-scale path (k_bal_*): dense lower-triangular system in HBM, panel workspace, factored diagonal blocks, pair CSR
    // Round 5: BLOCK-SPARSE.  The reference factors this system with a sparse LDL^T (SimplicialLDLT + AMD, solvers/linear_solver_eigen.h:95-112) because a
    // map's covisibility is sparse; here the lower triangle lives as 64 x 64 TILES and only the tiles that the covisibility structure -- and the fill-in of its
    // elimination, worked out on the host at tile level -- can ever make non-zero exist: bigTile[ti * bigT + tj] = the tile's slot in both pools, or -1.
    // Memory and work follow the non-zero structure; a map in which every keyframe sees every other one keeps every tile and runs as before.
    GP<double> big;            // bigTiles x 64 x 64: the working matrix (row-major inside a tile)
    GP<double> bigL;           // bigTiles x 64 x 64: the factor L (rows below each panel's diagonal block); row N = z
    GP<double> bigDiag;        // (N / 32) * 32 * 32: unit-lower diagonal blocks, column-major (1 / d on the diagonal)
    GP<double> bigLinv;        // (N / 32) * 32 * 32: their inverses as unit-lower blocks (k_bal_linv, for the back substitution)
    GP<int> bigFail;
    GP<const int> bigTile;     // bigT * bigT
    int bigT, bigTiles, bigDense;
    GP<const int4> bigWork;    // per 32-column panel, ONE record pair per tile its launch updates -- {ti, tj, slot of tile (ti, tj), slot of the panel's tile in row ti},
                               // {slot of the panel's tile in row tj, 0, 0, 0} (-1: dead, zeros) -- so a workgroup finds its tiles with one load; the first record of a
                               // panel is always the tile of the NEXT diagonal block (its workgroup factors it on the spot)
    GP<const int> lpStart;     // nPairsNZ + 1
    GP<const int> lpPair;      // 2 * nPairsNZ: (i1, i2), i1 <= i2
    GP<const int> lpOrder;     // launch slots: the pairs with more than kBigPairLong entries first (four waves each), then the others (one wave each); -1 = idle slot
    GP<const int> lpPts;       // landmark blocks of each pair, ascending
    GP<const int> lpE1;        // ... and the landmark's edges in camera i1 / i2 (the dense point x camera table of the window path would be
    GP<const int> lpE2;        //     nP x nC ints: 400 MB for a 1000-keyframe map)
    int nPairsNZ;
    // per-window addresses every kernel finds HERE (the kernels take an array of windows and blockIdx.z, see BA_WIN)
    GP<int> ctl0;              // the two control blocks (8 ints each); a launch runs on ctl0 + 8 * par
    GP<double> lm0;            // the two LM blocks (8 doubles each)
    GP<int> solveOk;
    GP<struct BAStatus> status;   // pinned host memory
    GP<unsigned char> cls;     // E: outlier table of the pass between the two optimize() calls
    GP<SE3> outCams; GP<double> outPts; GP<unsigned char> outCls; GP<double> outPlanes;   // pinned results (k_ba_finish)
};

// Every BA kernel takes the device array of window records: workgroup (x, y, z) works on window z, on the control / LM block
// pair `wpar` of that window (BADecision).  The array holds TWO records per window that differ only in ctl / lm (pair 0 and
// pair 1), so a kernel reads its record in place -- uniform, read-only loads on the scalar unit, exactly like kernel
// arguments (a local copy with the two pointers patched went through scratch memory: 1.17 -> 2.2 ms per window).  A single
// window is a batch of one.
// Batches pin every window to one XCD (speed only, any placement gives the same results): workgroups are dealt round-robin
// over the 8 XCDs in dispatch order, so with the plain (x = block, z = window) numbering the ~5 MB a window keeps re-reading
// (Hpl blocks, residuals, the edge table) would be pulled into all eight L2s -- 25 windows are then fabric-bound (the pair
// assembly alone moved 350 MB per launch).  Window w is served by XCD w % 8 only; the launch pads grid.z to a multiple of 8.
// The windows of an INCOMPLETE last row of eight share all eight XCDs (slot x serves window x mod rem with the other slots of that residue, the
// window's workgroups dealt round-robin among them): with 25 windows on 3 + 3 + ... + 4 the XCD that held four set the pace of every launch -- the
// kernels took as long for 25 windows as for 32 (k_ba_linearize 53.7 / 55.3 us, 43.0 for 24; profiles/r04_ba_xcd_balance.txt).
// wpar = block pair | rot << 4 | number of windows << 8 (rot: XCD of the group's first window -- the groups of a batch run
// concurrently and together should load the XCDs evenly);  bx = this workgroup's block index inside its window.
#define BA_WIN(P)                                                                                                   \
    unsigned bx = blockIdx.x, wz_ = blockIdx.z; (void)bx;                                                                  \
    {                                                                                                               \
        const unsigned nz_ = (unsigned)wpar >> 8;                                                                   \
        if (nz_ > 1) {                                                                                              \
            const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.z, s_ = b_ >> 3;                                  \
            const unsigned xs_ = (b_ - ((unsigned)wpar >> 4)) & 7, row_ = s_ / gridDim.x;      /* XCD x: slot (x - rot) mod 8 */  \
            const unsigned rem_ = nz_ & 7;                                                                          \
            bx = s_ % gridDim.x;                                                                                    \
            wz_ = xs_ + 8 * row_;                                                                                   \
            if (rem_ && row_ == (nz_ >> 3)) {      /* the incomplete last row: its rem_ windows over all eight slots */   \
                const unsigned wq_ = xs_ % rem_, nsh_ = (8 - wq_ + rem_ - 1) / rem_;                                \
                if (bx % nsh_ != xs_ / rem_) return;                                                                \
                wz_ = 8 * row_ + wq_;                                                                               \
            }                                                                                                       \
            if (wz_ >= nz_) return;                                                                                 \
        }                                                                                                           \
    }                                                                                                               \
    const BADev& P = W[2 * wz_ + (wpar & 1)]

enum { kCtlHalt = 0, kCtlCur = 1, kCtlIters = 2, kCtlStatus = 3, kCtlNBad = 4, kCtlPhase = 5, kCtlAnyActive = 6 };   // phase: 0 / 1 = first / second optimize()
enum { kStRunning = 0, kStTakeover = 1, kStTerminate = 2, kStEmpty = 3 };   // empty: no level-0 edge left (g2o's optimize() returns -1)

struct BAPhase { double lambda, rho, chi; int accepted, cur, iters, status, nBad, touched; };
struct BAStatus {            // pinned host memory, written by k_ba_decide / k_ba_chi_init
    double lambda, rho, chi, tempChi;
    int accepted, solveOk, seq, cur;
    int iters, status, nBad, ntrace;
    double trLambda[64], trChi[64];   // [32 * phase + iteration]
    int trTrials[64];
    BAPhase ph[2];           // where each optimize() stood after its last decision (both calls may run in ONE enqueue)
};

__device__ __forceinline__ const SE3* cur_cams(const BADev& P) { return P.camsBuf[P.ctl[kCtlCur]]; }
__device__ __forceinline__ const double* cur_pts(const BADev& P) { return P.ptsBuf[P.ctl[kCtlCur]]; }
__device__ __forceinline__ SE3* trial_cams(const BADev& P) { return P.camsBuf[P.ctl[kCtlCur] ^ 1]; }
__device__ __forceinline__ double* trial_pts(const BADev& P) { return P.ptsBuf[P.ctl[kCtlCur] ^ 1]; }

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

// (Hll + lambda I)^-1 by cofactors / determinant, as Eigen's fixed-size 3x3 inverse() (block_solver.hpp:392)
__device__ inline void dinv3(const double* Hll, double lambda, double Di[9]) {
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; i++) A[i] = Hll[i];
    A[0] += lambda; A[4] += lambda; A[8] += lambda;
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    Di[0] = c00 * id; Di[1] = (A[2] * A[7] - A[1] * A[8]) * id; Di[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Di[3] = c01 * id; Di[4] = (A[0] * A[8] - A[2] * A[6]) * id; Di[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Di[6] = c02 * id; Di[7] = (A[1] * A[6] - A[0] * A[7]) * id; Di[8] = (A[0] * A[4] - A[1] * A[3]) * id;
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
constexpr int kTileMaxFree = 30;
constexpr int kTileThreads = 1024;
typedef double v4d __attribute__((ext_vector_type(4)));

struct TileGeom { int n, n4, R, Tr, Tc, nTiles; };
__host__ __device__ inline TileGeom tile_geom(int nF) {
    TileGeom g;
    g.n = nF * 6; g.n4 = (g.n + 3) & ~3; g.R = g.n4 + 1;
    g.Tr = (g.R + 15) >> 4; g.Tc = (g.n4 + 15) >> 4;
    g.nTiles = g.Tc * g.Tr - g.Tc * (g.Tc - 1) / 2;      // column tj holds tile rows tj .. Tr-1
    return g;
}
__host__ __device__ inline void tile_of(const TileGeom& g, int idx, int& ti, int& tj) {
    int j = 0, off = 0;
    while (idx >= off + (g.Tr - j)) { off += g.Tr - j; j++; }
    tj = j; ti = j + (idx - off);
}
// L archive: row r (r = n4 is the right-hand side, i.e. z) starts at r (r - 1) / 2 + 4 r: r entries + 4 slack so that
// the panel threads store their four values unconditionally
__host__ __device__ inline int tile_lrow(int r) { return r * (r - 1) / 2 + 4 * r; }
__host__ inline size_t tile_solver_lds(int nF) {
    const TileGeom g = tile_geom(nF);
    return ((size_t)((tile_lrow(g.n4 + 1) + 1) & ~1) + 3 * (size_t)g.Tr * 16 * 4 + (size_t)g.n4 + 8) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair-owned Schur assembly for the tile solver.  Instead of partial rows per (camera, edge chunk) that a second kernel has to
// sum, ONE workgroup owns the 6x6 block S(i1, i2) of a camera pair i1 <= i2 (and, for i1 == i2, the camera's six right-hand
// side entries): its threads split the ascending list of landmarks both cameras observe, each thread accumulates
//   -Y(i1,l) Hpl(i2,l)^T  with  Y = Hpl(i1,l) (Hll_l + lambda I)^-1     (+ Hpl(i1,l) (Hll_l + lambda I)^-1 bl_l for the rhs)
// over its landmarks in order, a fixed-order LDS reduction adds the threads, and the block goes straight into the solver's
// register-tile layout (Hpp + lambda I added on the diagonal).  No slabs, no reduction launch, no atomics on data.
__host__ __device__ inline int pair_index(int i1, int i2, int nF) { return i1 * nF - i1 * (i1 - 1) / 2 + (i2 - i1); }   // i1 <= i2
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

// ---------------------------------------------------------------------------------------------------------------------
// Map-scale path (more than kTileMaxFree free keyframes: Optimizer::BundleAdjustment / GlobalBundleAdjustemnt over a whole
// map, src/Optimizer.cc:47-323, and oversized local windows).  The reduced camera system no longer fits one workgroup, so
// it lives in HBM as a dense lower triangle and is factorised by the whole chip:
//   layout   S: RP x RP doubles, row-major, only col <= row is meaningful.  Rows / columns 0..n-1 = the 6 nFree pose
//            unknowns, n..N-1 = identity padding up to a multiple of the panel width, row N = the right-hand side (so the
//            forward substitution is part of the factorisation, as in the tile solver), RP = N + 1 rounded up to the
//            64-wide update tiles.  Anything above the diagonal or beyond row N is scratch nobody reads.
//   assembly k_bal_schur_pairs: ONE WAVEFRONT per camera pair (i1 <= i2) that shares at least one landmark -- the pair
//            lists come from the host as a CSR (counting sort over the landmarks' observer lists, ascending landmark
//            order); the 6x6 block -Sum Y(i1,l) Hpl(i2,l)^T (+ Hpp + lambda I and the right-hand side on diagonal pairs)
//            is reduced through LDS in lane order and stored straight into S.  Pairs without a common landmark stay zero
//            from the memset that precedes every trial (the factorisation fills in place).
//   factor   right-looking LDL^T without pivoting in panels of 32 columns, ONE launch per panel (k_bal_step; it started as two):
//            k_bal_panel   one thread per row below the diagonal block solves  w = a L_kk^-T  (496 FMAs on registers, the
//                          factored L_kk as LDS broadcast reads), keeps w for the update and stores l = w D^-1 in place;
//            k_bal_update  64 x 64 tiles of the trailing lower triangle, C -= W L^T over the 32 panel columns, 4 x 4
//                          outputs per thread from transposed LDS tiles; the tile holding the NEXT diagonal block factors
//                          it on the spot (32 steps, one barrier each), off the critical path of the other tiles.
//   solve    k_bal_backsolve: L^T x = z bottom-up in super-blocks of 256 columns, one launch each (see there), then exp(dx) * T.
// A zero or non-finite pivot fails the trial like SimplicialLDLT (`bigFail`).
constexpr int kBigMaxFree = 2048;
constexpr int kBigNB = 32;
struct BigGeom { int n, N, RP; };
__host__ __device__ inline BigGeom big_geom(int nF) {
    BigGeom g;
    g.n = nF * 6; g.N = (g.n + kBigNB - 1) / kBigNB * kBigNB; g.RP = (g.N + 1 + 63) & ~63;
    return g;
}

// (every tile live -- a map in which every keyframe sees every other one: the slots are the row-major numbering of the lower triangle, no table look-up)
__device__ __forceinline__ int big_slot(const BADev& P, int ti, int tj) { return P.bigDense ? ti * (ti + 1) / 2 + tj : P.bigTile[ti * P.bigT + tj]; }
__device__ __forceinline__ double* big_tile(double* pool, int slot) { return pool + ((size_t)slot << 12); }
// element (r, c), c <= r, of a pool; the caller knows the tile exists
__device__ __forceinline__ double* big_elem(const BADev& P, double* pool, int r, int c) {
    return big_tile(pool, big_slot(P, r >> 6, c >> 6)) + ((r & 63) << 6) + (c & 63);
}

// Round 5: ONE WAVEFRONT per pair, FOUR for the long ones (rounds 3-4 ran 512 threads per pair: 192 VGPRs each, so ONE workgroup per CU, and a block-wide reduction
// of 42 values per pair -- 90 us for the 1 593 pairs of the 200-keyframe benchmark map, six rounds of workgroups).  A pair's entries are a latency chain -- landmark
// index -> edge flags -> three blocks -- of ~4 us per 64 of them: a covisible pair shares tens to a few hundred landmarks, a DIAGONAL pair carries every landmark its
// keyframe sees; the host lists the long pairs first (lpOrder) and they get their own launch.  42 sums per pair: DPP tree per wave, the waves' totals through LDS in wave order.
constexpr int kBigPairLong = 2048;
template <int NT>
__global__ __launch_bounds__(NT) void k_bal_schur_pairs(const BADev* __restrict__ W, int wpar, int first, int pairOff) {
    BA_WIN(P);
    __shared__ double red[NT / 64][42];
    if (P.ctl[kCtlHalt]) return;
    const int nF = P.nFree, t = threadIdx.x;
    const BigGeom g = big_geom(nF);
    const bool lead = pairOff + (int)bx == 0;      // the first workgroup of the first launch also resets the trial's flags and padding
    {
        const int pi = P.lpOrder[pairOff + bx];
        if (pi < 0) return;                        // (an idle slot of the XCD deal: never the first)
        bx = pi;
    }
    const int i1 = P.lpPair[2 * bx], i2 = P.lpPair[2 * bx + 1];
    const bool diag = i1 == i2;
    const double lambda = first ? refc::LM_TAU * P.lm[3] : P.lm[0];
    if (lead) {
        if (first && t == 0) { P.lm[0] = lambda; P.lm[1] = 2; }
        if (t == 0) *P.bigFail = 0;
        for (int r = g.n + t; r < g.N; r += NT) *big_elem(P, P.big, r, r) = 1.0;     // identity padding
    }
    const int beg = P.lpStart[bx], cnt = P.lpStart[bx + 1] - beg;
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; q++) acc[q] = 0;
    for (int k = t; k < cnt; k += NT) {
        const int l = P.lpPts[beg + k];
        const int e1 = P.lpE1[beg + k], e2 = P.lpE2[beg + k];
        if ((P.eflag[e1] | P.eflag[e2]) & 2) continue;       // deactivated by the outlier pass
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
                acc[36 + r] += y0 * bl[0] + y1 * bl[1] + y2 * bl[2];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 42; q++) {
        const double ws = wave_sum_f64_lane63(acc[q]);
        if ((t & 63) == 63) red[t >> 6][q] = ws;
    }
    __syncthreads();
    if (t >= 42) return;
    double s = 0;
#pragma unroll
    for (int wv = 0; wv < NT / 64; wv++) s += red[wv][t];
    if (t < 36) {
        const int r = t / 6, c = t - r * 6;
        if (!diag) *big_elem(P, P.big, i2 * 6 + c, i1 * 6 + r) = s;          // lower element (row of i2, column of i1) = S(i1,i2)[r][c]
        else if (c >= r) *big_elem(P, P.big, i1 * 6 + c, i1 * 6 + r) = s + P.Hpp[(size_t)i1 * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
    } else if (diag) {
        const int r = t - 36;
        *big_elem(P, P.big, g.N, i1 * 6 + r) = P.bp[i1 * 6 + r] - s;         // right-hand side row
    }
}

// The two serial pieces of a panel -- the row solves w = a L_kk^-T and the LDL^T of the next diagonal block -- are chains of broadcast FMAs: every lane needs
// the same coefficient (row solve) or the pivot column's entry of row j (factorisation).  Rounds 3-4 fed them through LDS broadcast reads (a 32-column row solve:
// 496 reads + FMAs on one thread per row, 8.5 k cycles) and v_readlane (the factorisation in one wavefront's registers, lane i = row i: 496 FMAs + 1 056 v_readlane,
// 11.1 k cycles) -- 3.5 + 4.6 of a panel's 14 us (profiles/r05_bal_step_stamps.txt; commit 786e893 holds that code).
// Round 5: gfx90a+ gives 64-bit VALU operations ONE DPP control,
// row_newbcast:K (lane K of every 16-lane row to the whole row), and v_fmac_f64 has a VOP2 encoding that takes it: one instruction is
//      acc = fma(-x[lane K of my row], y, acc)
// -- the broadcast FMA both pieces consist of (tools/ubench/dpp_f64.hip: result and rate on gfx950).  The compiler has no builtin for 64-bit DPP, so
// these are inline assembly.  (The compiler's hazard recogniser does see them -- it puts `s_nop 0` between two dependent ones -- but the wait states a DPP source
// needs behind a VALU write of the same VGPR (2) or a VALU write of EXEC (5) are spelled out where they can occur, and tools/isa_census.py --dpp-hazards checks
// the compiled kernels for them.)  Every lane of the wave must be active.
template <int K> __device__ __forceinline__ void bal_fmac_nb(double& acc, double src, double y) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(y), "n"(K));
}
template <int K> __device__ __forceinline__ double bal_mov_nb(double src) {
    double d;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(K));
    return d;
}
template <int B, int... I, typename F> __device__ __forceinline__ void bal_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, B + I>{}), ...); }
template <int B, int E, typename F> __device__ __forceinline__ void bal_static_for(F&& f) { if constexpr (E > B) bal_static_for_impl<B>(f, std::make_integer_sequence<int, E - B>{}); }

// LDL^T of a 32 x 32 block, DPP form: the matrix lives in EVERY 16-lane row of the wave -- lane l holds rows l % 16 (X0: columns 0..15) and 16 + l % 16 (X1: columns
// 0..31), four identical replicas -- so a_jp, the un-normalised column p at row j, is lane j % 16 of the register that holds column p, and the update
// row[j] -= l_ip a_jp is ONE v_fmac_f64_dpp per row set (616 of them per block against 496 FMAs + 1 056 v_readlane).  The same products in the same order as
// the v_readlane version of rounds 3-4: the factors agree bit for bit.  On return X0 / X1 [j] = L(row, j) below the diagonal and 1 / d on it.
__device__ inline bool bal_factor_diag_dpp(double (&X0)[16], double (&X1)[kBigNB]) {
    const int l16 = threadIdx.x & 15;
    bool bad = false;
    asm volatile("s_nop 4");
    bal_static_for<0, kBigNB>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        asm volatile("s_nop 1");      // column p's last update may be the instruction before this one (pivot 30 -> 31)
        double d;
        if constexpr (p < 16) d = bal_mov_nb<p % 16>(X0[p]); else d = bal_mov_nb<p % 16>(X1[p]);
        bad |= (d == 0.0 || !isfinite(d));
        const double id = frcp(d);
        double lip0 = 0;
        if constexpr (p < 16) lip0 = X0[p] * id;
        const double lip1 = X1[p] * id;
        bal_static_for<p + 1, kBigNB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < 16) { bal_fmac_nb<j % 16>(X0[j], X0[p], lip0); bal_fmac_nb<j % 16>(X1[j], X0[p], lip1); }
            else if constexpr (p < 16) bal_fmac_nb<j % 16>(X1[j], X1[p], lip1);
            else bal_fmac_nb<j % 16>(X1[j], X1[p], lip1);
        });
        if constexpr (p < 16) { X0[p] = l16 == p ? id : lip0; X1[p] = lip1; }
        else X1[p] = l16 == p - 16 ? id : lip1;
    });
    return bad;
}
// rows l % 16 and 16 + l % 16 of the block at `src` (row stride ld doubles)
__device__ __forceinline__ void bal_load_rows_dpp(const double* src, int ld, double (&X0)[16], double (&X1)[kBigNB]) {
    const int l16 = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < 16; j++) X0[j] = src[l16 * ld + j];
#pragma unroll
    for (int j = 0; j < kBigNB; j++) X1[j] = src[(16 + l16) * ld + j];
}
// The factored block as the next panel's launch reads it, COLUMN-major: Ld[c][r] = L(r, c) below the diagonal, 1 / d on it, 0 above.  Every 16-lane row of the
// wave holds the whole block, so lanes 0..31 write one column (rows 0..15 | 16..31) and lanes 32..63 the next one: 16 stores of 512 contiguous bytes.  (Row-major,
// every lane wrote its own 256-byte row -- 32 partial lines per instruction; a store per pivot inside the factorisation loop costs more than it hides: 7.6 k + 0.9 k
// cycles against 4.8 k + 2.3 k.)
__device__ inline void bal_store_diag_dpp(double* bigDiag, int* bigFail, int kb, const double (&X0)[16], const double (&X1)[kBigNB], bool bad) {
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    const bool up = lane & 16, odd = lane & 32;
    double* Ld = bigDiag + (size_t)kb * kBigNB * kBigNB + (lane & 31) + (odd ? kBigNB : 0);
    auto entry = [&](auto jc) {                       // L(row, j) as stored: this lane's row of the upper / lower half
        constexpr int j = decltype(jc)::value;
        double lo = 0.0;
        if constexpr (j < 16) lo = j <= l16 ? X0[j] : 0.0;
        const double hi = j <= 16 + l16 ? X1[j] : 0.0;
        return up ? hi : lo;
    };
    bal_static_for<0, kBigNB / 2>([&](auto hc) {
        constexpr int j = 2 * decltype(hc)::value;
        const double e0 = entry(std::integral_constant<int, j>{}), e1 = entry(std::integral_constant<int, j + 1>{});
        Ld[j * kBigNB] = odd ? e1 : e0;
    });
    if (lane == 0 && bad) *bigFail = 1;
}

// The factorisation launches take what they need BY VALUE: the pool pointers, the halt flag's address and -- for the look-ahead workgroup, whose record is
// the first of the panel's list -- the work record itself.  Read through the window record like the other LM kernels, a panel started with three
// dependent round trips to memory (record -> work list -> tiles) before its first useful load; 38 launches per trial start with that chain.
struct BigStepArgs {
    double* big; double* bigL; double* bigDiag; int* bigFail; const int4* bigWork; const int* ctl; long long* dbg;
    int N; int4 wa0; int wb0x;
};

// first diagonal block of a trial (the following ones are factored by the update kernel of the panel before them)
__global__ __launch_bounds__(64) void k_bal_diag0(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    if (P.ctl[kCtlHalt]) return;
    const double* T0 = big_tile(P.big, big_slot(P, 0, 0));
    double X0[16], X1[kBigNB];
    bal_load_rows_dpp(T0, 64, X0, X1);
    const bool bad = bal_factor_diag_dpp(X0, X1);
    bal_store_diag_dpp(P.bigDiag, P.bigFail, 0, X0, X1, bad);
}

// One launch per 32-column panel: every 64 x 64 tile of the trailing lower triangle first solves the panel rows it needs ITSELF
// -- 64 rows of its tile row (w = a L_kk^-T, kept as W) and 64 of its tile column (l = w D^-1) on 128 threads, 496 FMAs each on
// registers against the factored diagonal block in LDS; a row is solved by every tile that uses it, which costs no wall time --
// then C -= W L^T (4 x 4 outputs per thread from transposed LDS tiles).  The first tile column archives l in `bigL` (the
// factor proper: the working matrix keeps its stale panel columns, nobody reads them again, so there is no race between a
// tile that still reads a and one that would overwrite it with l).  The tile that holds the NEXT diagonal block factors it on
// the spot (one wavefront, see above), so the next launch starts from a finished L_kk.  The first version ran the row solves
// as a launch of their own: two dependent launches per panel instead of one (12 + 13 us at 40 free keyframes).
__global__ __launch_bounds__(256) void k_bal_step(BigStepArgs A, int kb, int last, int workOff) {
    const unsigned bx = blockIdx.x;
    __shared__ __attribute__((aligned(16))) double Wt[kBigNB][68];
    __shared__ __attribute__((aligned(16))) double Lt[kBigNB][68];
    __shared__ double dv[kBigNB];                      // 1 / d of the panel's diagonal block
    __shared__ double da[kBigNB][kBigNB + 1];
    constexpr size_t ld = 64;                          // (inside a tile)
    const int k0 = kb * kBigNB, t = threadIdx.x;
    // the workgroup's tile: record bx of the panel's work list (built by the host from the tile structure; a dense system lists every trailing tile).
    // The halt flag travels with the record: one round trip to memory for both, not one behind the other (38 launches per trial start with this chain)
    const int halted = A.ctl[kCtlHalt];
    int4 wa = A.wa0;
    int wbx = A.wb0x;
    if (bx != 0) { wa = A.bigWork[2 * (size_t)(workOff + bx)]; wbx = A.bigWork[2 * (size_t)(workOff + bx) + 1].x; }
    if (halted) return;
    const bool stp = A.dbg && bx == 0 && t == 0 && kb == 2;      // phase stamps of the look-ahead workgroup of panel 2 (EAO_DEBUG_STAMPS)
    if (stp) A.dbg[16] = clock64();
    const int ti = wa.x, tj = wa.y;
    const int r0 = ti * 64, c0 = tj * 64;
    // panel rows of this tile: threads 0..63 the tile's rows (W), 64..127 its columns (L); a tile row that is dead in the panel's tile column (only ever the
    // first list entry: the next diagonal block's row is always listed) contributes zeros
    const bool roleW = t < 64;
    const int prow = roleW ? r0 + t : c0 + (t - 64);
    const int pslot = roleW ? wa.w : wbx;
    const bool act = t < 128 && pslot >= 0 && prow >= k0 + kBigNB && prow <= A.N;      // rows of the panel itself / beyond the system: zero
    double w[kBigNB];
    if (act) {
        const double* src = big_tile(A.big, pslot) + ((prow & 63) << 6) + (k0 & 63);
#pragma unroll
        for (int c = 0; c < kBigNB; c++) w[c] = src[c];
    } else {
#pragma unroll
        for (int c = 0; c < kBigNB; c++) w[c] = 0.0;
    }
    const int tx = t & 15, ty = t >> 4;
    const bool crit = !last && bx == 0;            // the look-ahead workgroup (see below): it walks its tile quadrant by quadrant
    double c[4][4];
    double* const Ct = big_tile(A.big, wa.z);      // (the host's symbolic elimination made sure the tile exists)
    double* C = Ct + (size_t)(ty * 4) * ld + tx * 4;
    const int oq = k0 + kBigNB - c0;               // (look-ahead) 0 or 32: offset of the next diagonal block inside this tile
    double accq[4];                                // (look-ahead) this thread's four entries of that block
    if (!crit) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) c[i][j] = C[(size_t)i * ld + j];
    } else {
        const double* Cq = Ct + (size_t)(oq + (t >> 3)) * ld + oq + (t & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) accq[j] = Cq[j];
    }
    // the row solves' coefficients in registers, every 16-lane row of the two solving waves a replica: lane l holds rows l % 16 and 16 + l % 16 of L_kk, so
    // L(cc, q) is lane cc % 16 of register q -- a row_newbcast operand (see bal_fmac_nb) instead of an LDS broadcast read per FMA (8.5 k -> 5.3 k cycles per step)
    double A0[15], A1[31];
    {
        const double* Ld = A.bigDiag + (size_t)kb * kBigNB * kBigNB;      // column-major: Ld[c][r]
        if (t < kBigNB) dv[t] = Ld[t * kBigNB + t];
        if (t < 128) {
#pragma unroll
            for (int q = 0; q < 15; q++) A0[q] = Ld[q * kBigNB + (t & 15)];
#pragma unroll
            for (int q = 0; q < 31; q++) A1[q] = Ld[q * kBigNB + 16 + (t & 15)];
        }
    }
    __syncthreads();
    if (stp) A.dbg[17] = clock64();
    if (t < 128) {
        // (every lane takes part: a row that is not `act` solves zeros)
        asm volatile("s_nop 4");
        // columns in PAIRS (c, c + 1): a dependent v_fmac_f64_dpp issues every 12 cycles, an independent one every 8.5 (tools/ubench/dpp_f64.hip), and
        // the two chains only meet in the pair's last step; per column still the FMAs w[c] -= w[q] L(c, q) in ascending q
        auto step = [&](auto ccc, auto qc) {
            constexpr int cc = decltype(ccc)::value, q = decltype(qc)::value;
            if constexpr (cc < 16) bal_fmac_nb<cc>(w[cc], A0[q], w[q]);
            else bal_fmac_nb<cc - 16>(w[cc], A1[q], w[q]);
        };
        bal_static_for<0, kBigNB / 2>([&](auto hc) {
            constexpr int c0 = 2 * decltype(hc)::value - 1;      // pairs (1, 2), (3, 4), ... (29, 30); column 31 alone
            if constexpr (c0 < 0) {}
            else if constexpr (c0 + 1 < kBigNB) {
                bal_static_for<0, c0>([&](auto qc) { step(std::integral_constant<int, c0>{}, qc); step(std::integral_constant<int, c0 + 1>{}, qc); });
                step(std::integral_constant<int, c0 + 1>{}, std::integral_constant<int, c0>{});
            }
        });
        bal_static_for<0, kBigNB - 1>([&](auto qc) { step(std::integral_constant<int, kBigNB - 1>{}, qc); });
        if (roleW) {
#pragma unroll
            for (int k = 0; k < kBigNB; k++) Wt[k][t] = w[k];
            if (ti == tj && act) {      // a tile row's DIAGONAL workgroup archives its panel rows (every listed row has one)
                double* ldst = big_tile(A.bigL, pslot) + ((prow & 63) << 6) + (k0 & 63);
#pragma unroll
                for (int k = 0; k < kBigNB; k++) ldst[k] = w[k] * dv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kBigNB; k++) Lt[k][t - 64] = w[k] * dv[k];
        }
    }
    __syncthreads();
    if (stp) A.dbg[18] = clock64();
    if (!crit) {
#pragma unroll 8
        for (int k = 0; k < kBigNB; k++) {
            double wv[4], lv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { wv[i] = Wt[k][ty * 4 + i]; lv[i] = Lt[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) c[i][j] = fma(-wv[i], lv[j], c[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) C[(size_t)i * ld + j] = c[i][j];
        return;
    }
    // look-ahead: the tile that holds the NEXT panel's diagonal block (the first one dispatched) factors it right away.  Only that 32 x 32 block is on the
    // critical path of the whole factorisation, so it is updated FIRST, by all four waves (four outputs per thread); then wave 0 factors it while waves 1
    // and 2 update the tile's two other quadrants below the diagonal in the shadow of the factorisation (the quadrant above it is read by nobody; when the block
    // is the tile's lower-right quadrant the others are the panel's own rows and columns: no update at all).  The same FMAs per output as the full-tile loop.
    {
        const int o = oq;
        {
            const int qr = t >> 3, qc = (t & 7) * 4;
            double (&acc)[4] = accq;
#pragma unroll 8
            for (int k = 0; k < kBigNB; k++) {
                const double wv = Wt[k][o + qr];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = fma(-wv, Lt[k][o + qc + j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) da[qr][qc + j] = acc[j];
        }
        __syncthreads();
        if (stp) A.dbg[19] = clock64();
        if (t >= 64 && t < 192 && o == 0) {
            const int v = (t >> 6) - 1, lane = t & 63;          // wave 1: rows 32..63 x columns 0..31, wave 2: rows 32..63 x columns 32..63
            const int rb = 32 + (lane >> 3) * 4, cb = v * 32 + (lane & 7) * 4;
            double* Cq = Ct + (size_t)rb * ld + cb;
            double q4[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) q4[i][j] = Cq[(size_t)i * ld + j];
#pragma unroll 8
            for (int k = 0; k < kBigNB; k++) {
                double wv[4], lv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { wv[i] = Wt[k][rb + i]; lv[i] = Lt[k][cb + i]; }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) q4[i][j] = fma(-wv[i], lv[j], q4[i][j]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) Cq[(size_t)i * ld + j] = q4[i][j];
        }
        if (t < 64) {
            double X0[16], X1[kBigNB];
            bal_load_rows_dpp(&da[0][0], kBigNB + 1, X0, X1);
            if (stp) A.dbg[20] = clock64();
            const bool bad = bal_factor_diag_dpp(X0, X1);
            if (stp) A.dbg[21] = clock64();
            bal_store_diag_dpp(A.bigDiag, A.bigFail, kb + 1, X0, X1, bad);
            if (stp) A.dbg[22] = clock64();
        }
    }
}

// The inverses of the 32 x 32 unit-lower diagonal blocks, for the back substitution: one wavefront per pair of blocks (one launch behind the
// factorisation), lane j of each half forward-substitutes column j of its block's inverse in registers (the coefficients as LDS broadcasts).
// Out: bigLinv[block][r][c] = (L_bb^-1)(r, c).
__global__ __launch_bounds__(64) void k_bal_linv(const BADev* __restrict__ W, int wpar, int nbk) {
    BA_WIN(P);
    __shared__ __attribute__((aligned(16))) double Ls[2 * kBigNB * kBigNB];
    if (P.ctl[kCtlHalt]) return;
    const int t = threadIdx.x, hb = t >> 5, j = t & 31, blk = min(2 * (int)bx + hb, nbk - 1);      // (an odd block count: the last wave's upper half repeats its lower one)
    {
        const double* Ld = P.bigDiag + (size_t)(2 * bx) * kBigNB * kBigNB;
        const int cnt = min(2, nbk - 2 * (int)bx) * kBigNB * kBigNB / 2;
        for (int idx = t; idx < cnt; idx += 64) reinterpret_cast<double2*>(Ls)[idx] = reinterpret_cast<const double2*>(Ld)[idx];
    }
    __syncthreads();
    const double* L = Ls + (blk - 2 * (int)bx) * kBigNB * kBigNB;      // column-major: L[c][r]
    double x[kBigNB];
#pragma unroll
    for (int i = 0; i < kBigNB; i++) x[i] = i == j ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < kBigNB - 1; m++) {
#pragma unroll
        for (int i = m + 1; i < kBigNB; i++) x[i] = fma(-L[m * kBigNB + i], x[m], x[i]);
    }
    double* dst = P.bigLinv + ((size_t)blk << 10);
#pragma unroll
    for (int i = 0; i < kBigNB; i++) dst[i * kBigNB + j] = x[i];
}

// Back substitution L^T x = z in super-blocks of 256 columns, one launch each, bottom-up.  Every workgroup solves the
// super-block's own triangle redundantly (8 blocks of 32: column sums over the rows already solved split across the
// waves; the 32 x 32 triangle is ONE product with the block's inverse from k_bal_linv -- 32 independent dot products instead of
// the 31 dependent shuffle steps of rounds 3-4), then removes the super-block's contribution from ITS 64 columns of z
// to the left: z_j -= sum_i L(i, j) x_i.  L is read once, by as many workgroups as there are column chunks; the last launch
// (super-block 0) also applies exp(dx) * T.
constexpr int kBigSB = 256;
__global__ __launch_bounds__(256) void k_bal_backsolve(const BADev* __restrict__ W, int wpar, int J) {
    BA_WIN(P);
    int* const solveOk = P.solveOk;
    __shared__ double xl[kBigSB];      // z of the super-block on entry, x on exit
    extern __shared__ double sdiag[];  // (kBigSB / 32) x 32 x 32: the inverses of the super-block's unit-lower diagonal blocks
    __shared__ double part2[4][64];
    if (P.ctl[kCtlHalt]) return;
    const BigGeom g = big_geom(P.nFree);
    const int n = g.n, N = g.N, t = threadIdx.x;
    double* S = P.bigL;            // the archived factor (rows below each panel's diagonal block, z = row N), as tiles
    const int J0 = J * kBigSB, w = min(kBigSB, N - J0);
    auto zat = [&](int col) { return big_elem(P, S, N, col); };      // z = row N: its tile row is dense by construction
    if (J > 0 && bx > 0) {      // a column chunk in which none of the super-block's tile rows holds a tile has nothing to remove (workgroup-uniform; workgroup 0 also delivers x)
        bool any = false;
        for (int r = J0 >> 6; r <= (J0 + w - 1) >> 6; r++) any = any || big_slot(P, r, bx) >= 0;
        if (!any) return;
    }
    const bool stb = P.dbg && bx == 0 && t == 0 && J == 1;      // phase stamps (EAO_DEBUG_STAMPS): super-block 1, workgroup 0
    if (stb) P.dbg[24] = clock64();
    if (t < w) xl[t] = *zat(J0 + t);
    {   // the super-block's (up to eight) inverted diagonal blocks: 64 KB of LDS, every load of a thread in flight at once
        // (a fixed trip count: as a loop up to `cnt` this copy ran ONE load at a time -- 22 k cycles of a 60 k-cycle launch, EAO_DEBUG_STAMPS)
        const double2* Ld0 = reinterpret_cast<const double2*>(P.bigLinv + (size_t)(J0 / kBigNB) * kBigNB * kBigNB);
        const int cnt2 = (w / kBigNB) * kBigNB * kBigNB / 2;
        double2 v[kBigSB / kBigNB * 2];
#pragma unroll
        for (int u = 0; u < kBigSB / kBigNB * 2; u++) v[u] = t + 256 * u < cnt2 ? Ld0[t + 256 * u] : make_double2(0, 0);
#pragma unroll
        for (int u = 0; u < kBigSB / kBigNB * 2; u++) reinterpret_cast<double2*>(sdiag)[t + 256 * u] = v[u];
    }
    // this thread's column J0 + t: its tile in each of the super-block's (up to four) tile rows, looked up once
    const int zsl = big_slot(P, N >> 6, (J0 + min(t, w - 1)) >> 6);      // (the z row's tile of that column: always there)
    int mysl[kBigSB / 64];
#pragma unroll
    for (int r = 0; r < kBigSB / 64; r++) mysl[r] = (r << 6) < w ? big_slot(P, (J0 >> 6) + r, (J0 + min(t, w - 1)) >> 6) : -1;
    // the super-block's triangle, right-looking: the top block is solved by wave 0 (column c of L_kk in registers, the
    // solved entries handed down by v_readlane), then every thread owning a column to the left removes the block's 32 rows
    // from its z entry -- the 32 loads of a thread are independent of x, so they are in flight before the chain ends
    // the rows a thread needs for block b - 2 are requested while block b is being solved (three register buffers)
    auto fetch = [&](int b, double (&dst)[kBigNB]) {
        if (b >= 0 && t < b * kBigNB) {
            const int cb = J0 + b * kBigNB;
            const int tr = b >> 1;                                   // the block's 32 rows lie in one tile row
            const int sl = tr == 0 ? mysl[0] : tr == 1 ? mysl[1] : tr == 2 ? mysl[2] : mysl[3];
            const double* src = big_tile(S, sl < 0 ? zsl : sl) + ((cb & 63) << 6) + ((J0 + t) & 63);      // (no tile: the z tile's bytes, masked below)
#pragma unroll
            for (int i = 0; i < kBigNB; i++) dst[i] = src[i << 6];
            if (sl < 0) {
#pragma unroll
                for (int i = 0; i < kBigNB; i++) dst[i] = 0.0;
            }
        }
    };
    auto do_block = [&](int b, double (&cur)[kBigNB], double (&nxt)[kBigNB]) {
        fetch(b - 2, nxt);
        __syncthreads();                 // xl[] of this block is final (initial load / previous block's updates)
        if (t < 64) {                    // x_b = L_bb^-T y_b: lane c sums column c of the inverse against y (zeros above the diagonal), four partial sums
            const int c = t & 31;
            const double* Li = sdiag + (size_t)b * kBigNB * kBigNB + c;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int q = 0; q < kBigNB; q += 4) {
                a0 = fma(Li[q * kBigNB], xl[b * kBigNB + q], a0);
                a1 = fma(Li[(q + 1) * kBigNB], xl[b * kBigNB + q + 1], a1);
                a2 = fma(Li[(q + 2) * kBigNB], xl[b * kBigNB + q + 2], a2);
                a3 = fma(Li[(q + 3) * kBigNB], xl[b * kBigNB + q + 3], a3);
            }
            const double v = (a0 + a1) + (a2 + a3);
            eao::wave_sync();            // every lane has read y before any lane overwrites it
            if (t < 32) xl[b * kBigNB + t] = v;
        }
        __syncthreads();
        if (t < b * kBigNB) {
            double acc = 0;
#pragma unroll
            for (int i = 0; i < kBigNB; i++) acc = fma(cur[i], xl[b * kBigNB + i], acc);
            xl[t] -= acc;
        }
    };
    {
        // (three register buffers: a block's rows are requested TWO blocks ahead -- one block of ~2 k cycles does not cover a load from the factor pool)
        double rowA[kBigNB], rowB[kBigNB], rowC[kBigNB];
        const int top = w / kBigNB - 1;
        fetch(top, rowA);
        fetch(top - 1, rowB);
        if (stb) P.dbg[25] = clock64();
        for (int b = top; b >= 0; b -= 3) {
            do_block(b, rowA, rowC);
            if (b - 1 >= 0) do_block(b - 1, rowB, rowA);
            if (b - 2 >= 0) do_block(b - 2, rowC, rowB);
        }
    }
    __syncthreads();
    if (stb) P.dbg[26] = clock64();
    if (bx == 0)
        for (int i = t; i < w; i += 256) if (J0 + i < n) P.xp[J0 + i] = xl[i];
    if (J > 0) {
        // this workgroup's 64 columns to the left of the super-block
        // (rows rg, rg + 4, ... of the super-block, ALL 64 loads of a thread in flight at once behind four tile look-ups: in batches of eight, each behind its own
        //  look-ups, this loop was 21 k cycles of a 60 k-cycle launch)
        const int j = bx * 64 + (t & 63), rg = t >> 6;
        // (a tile that does not exist reads the z tile instead and is masked afterwards: unconditional loads, no branch per load)
        const double* const safe = big_tile(S, big_slot(P, N >> 6, bx)) + (t & 63);
        const double* tl[kBigSB / 64];
        bool ok[kBigSB / 64];
#pragma unroll
        for (int r = 0; r < kBigSB / 64; r++) {
            const int sl = (r << 6) < w ? big_slot(P, (J0 >> 6) + r, bx) : -1;
            ok[r] = sl >= 0;
            tl[r] = ok[r] ? big_tile(S, sl) + (t & 63) : safe;
        }
        double v[kBigSB / 4];
#pragma unroll
        for (int m = 0; m < kBigSB / 4; m++) v[m] = tl[m >> 4][((rg + 4 * m) & 63) << 6];
#pragma unroll
        for (int m = 0; m < kBigSB / 4; m++) v[m] = (rg + 4 * m < w && ok[m >> 4]) ? v[m] : 0.0;
        double acc = 0;
#pragma unroll
        for (int m = 0; m < kBigSB / 4; m++) { const int i = rg + 4 * m; const double xi = xl[i]; acc = fma(v[m], i < w ? xi : 0.0, acc); }      // (no branch per row; beyond w: 0 * 0)
        part2[rg][t & 63] = acc;
        __syncthreads();
        if (rg == 0) *zat(j) -= (part2[0][t] + part2[1][t]) + (part2[2][t] + part2[3][t]);
        if (stb) P.dbg[27] = clock64();
        return;
    }
    __threadfence();
    __syncthreads();
    const SE3* cams = cur_cams(P);
    SE3* camsT = trial_cams(P);
    for (int cI = t; cI < P.nCams; cI += 256) {
        const int ci = P.camIdx[cI];
        if (ci >= 0) {
            double u[6];
            for (int q = 0; q < 6; q++) u[q] = P.xp[ci * 6 + q];
            camsT[cI] = se3_mul(se3_exp(u), cams[cI]);
        } else {
            camsT[cI] = cams[cI];
        }
    }
    if (t == 0) *solveOk = *P.bigFail ? 0 : 1;
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
        if (threadIdx.x == 0) __hip_atomic_store(&P.lm[4], out2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!decide) return;
    // The accept / reject decision of this trial, by the LAST workgroup of the window to get here (every other one has
    // published its partial sums by then): the next linearisation finds it in the other control block.  Leaving it to the
    // head of that launch instead -- every workgroup re-deriving it from the 2 nL partial sums -- cost each of its 1 100
    // workgroups (25 windows) 2.6 - 4 us of an 8 - 9 us life.
    // Memory model: a workgroup publishes ITS partial sums (wgPart) with agent-scope atomic stores (written through, no L2
    // write-back needed -- a __threadfence() here is a buffer_wbl2 per workgroup and made the launch 3x slower), waits for
    // them (workgroup-scope release = s_waitcnt) and then takes its ticket; the last workgroup reads the partial sums with
    // agent-scope atomic loads (they bypass its CU's L1).  The same goes for the camera part in lm[4].
    __shared__ int s_last;
    __syncthreads();
    block_sum<2, 256>(mine, red, out2);         // this workgroup's 32 landmarks, fixed order
    if (threadIdx.x == 0) {
        __hip_atomic_store(&P.wgPart[2 * bx], out2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&P.wgPart[2 * bx + 1], out2[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        __hip_atomic_store((int*)P.doneCnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

struct LMTraceHost {
    std::vector<double> lambda, chi2;
    std::vector<int> trials;
    float deviceMs = 0;
    int linearizations = 0;
    void clear() { lambda.clear(); chi2.clear(); trials.clear(); deviceMs = 0; linearizations = 0; }
};
thread_local LMTraceHost g_trace;

struct LMContext {  // per-thread device workspace, grow-only
    hipStream_t stream = nullptr;                   // the stream of the call in progress: one of byClass[] (every call ends synchronised, so the next may take another)
    hipStream_t byClass[3] = {nullptr, nullptr, nullptr};      // eao::StreamClass: PoseOptimization / LocalBundleAdjustment / map BundleAdjustment of this host thread
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BAStatus* status = nullptr;   // pinned + mapped
    unsigned char* pin = nullptr; // pinned host mirror of the input part of the arena: ONE H2D copy per upload
    size_t pinCap = 0;
    unsigned char* pinOut = nullptr;   // pinned results, written by k_ba_finish
    size_t pinOutCap = 0;
    eao::DevBuf<unsigned char> bytes;
    std::vector<int> scratch;     // host counters of the structure build (kept to avoid per-call allocation)
    // map-scale path: where every 32-column panel's work records start / the records themselves, as the HOST reads them when it enqueues the panel launches --
    // which, in a batch call, happens after every window has been prepared.  They belong to the window's context (round 5: as thread-local tables of the set-up
    // worker they were overwritten by the next map-scale window the same worker prepared, and the first window ran with the second one's panels --
    // tools/dbg_batch_two_maps.py, tests/test_gpu_lm.py::test_two_map_scale_windows_in_one_batch).
    std::vector<int> bigPanelStart;
    std::vector<int4> bigPanelWork;
    size_t used = 0;
    ~LMContext() {
        if (status) (void)hipHostFree(status);
        if (pin) (void)hipHostFree(pin);
        if (pinOut) (void)hipHostFree(pinOut);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (hipStream_t q : byClass) if (q) (void)hipStreamDestroy(q);
    }
};
thread_local LMContext g_ctx;

eao_status ctx_init(LMContext& c, bool ownStream, eao::StreamClass cls) {
    eao_status st = eao::require_device();
    if (st) return st;
    if (ownStream) {
        hipStream_t& q = c.byClass[(int)cls];
        if (!q) EAO_HIP(eao::create_stream(&q, cls));
        c.stream = q;
        if (!c.ev0) {
            EAO_HIP(hipEventCreate(&c.ev0));
            EAO_HIP(hipEventCreate(&c.ev1));
        }
    }
    if (!c.status) {
        EAO_HIP(hipHostMalloc((void**)&c.status, sizeof(BAStatus), hipHostMallocMapped));
        std::memset(c.status, 0, sizeof(BAStatus));
    }
    return EAO_OK;
}

// bump allocator over one device buffer (256-byte aligned slices)
struct Arena {
    unsigned char* base;
    size_t cap, off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace

extern "C" {

eao_status eao_pose_optimization(const eao_pose_problem* p, eao_pose_result* r) {
    EAO_REQUIRE(p && r, "null argument");
    EAO_REQUIRE(p->n >= 0 && p->Tcw && (p->n == 0 || (p->Xw && p->obs && p->inv_sigma2 && r->outlier)), "bad problem");
    const int M = p->n_planes;
    EAO_REQUIRE(M >= 0 && M <= kPoseMaxPlanes, "at most %d plane edges (got %d)", kPoseMaxPlanes, M);
    EAO_REQUIRE(M == 0 || (p->plane_world && p->plane_obs && p->plane_seen && r->plane_outlier), "plane arrays missing");
    LMContext& c = g_ctx;
    eao_status st = ctx_init(c, true, eao::StreamClass::Latency);
    if (st) return st;
    g_trace.clear();
    const int n = p->n;
    for (int i = 0; i < n; i++) r->outlier[i] = 0;
    r->lm_iterations = 0;
    if (n < 3) {  // "if(nInitialCorrespondences<3) return 0" (src/Optimizer.cc:453-454): pose untouched
        std::memcpy(r->Tcw, p->Tcw, 16 * sizeof(float));
        r->n_inliers = 0;
        return EAO_OK;
    }
    // inputs are staged in the pinned mirror of the arena (float32 -> double exactly as Converter / Eigen would promote
    // them) and sent with ONE copy; the results come back through pinned memory the kernel writes directly
    const size_t need = (size_t)n * (3 + 3 + 1 + 3) * 8 + (size_t)n * 2 + 40 * 256;
    if ((st = c.bytes.reserve(need))) return st;
    Arena a{c.bytes.p, c.bytes.n};
    const size_t off0 = a.off;
    double* dXw = a.take<double>((size_t)n * 3);
    double* dobs = a.take<double>((size_t)n * 3);
    double* dinfo = a.take<double>(n);
    unsigned char* dflags = a.take<unsigned char>(n);
    double* dplanes = a.take<double>((size_t)kPoseMaxPlanes * 10);
    const size_t off1 = (a.off + 255) & ~(size_t)255;
    double* derr = a.take<double>((size_t)n * 3);
    long long* ddbg = a.take<long long>(64);
    if (c.pinCap < off1) {
        if (c.pin) (void)hipHostFree(c.pin);
        c.pin = nullptr; c.pinCap = 0;
        EAO_HIP(hipHostMalloc((void**)&c.pin, off1 + (off1 >> 2), hipHostMallocDefault));
        c.pinCap = off1 + (off1 >> 2);
    }
    const size_t outBytes = sizeof(SE3) + 16 + 192 * 8 + 16 + (size_t)n + 64 + kPoseMaxPlanes;
    if (c.pinOutCap < outBytes) {
        if (c.pinOut) (void)hipHostFree(c.pinOut);
        c.pinOut = nullptr; c.pinOutCap = 0;
        EAO_HIP(hipHostMalloc((void**)&c.pinOut, outBytes + (outBytes >> 2), hipHostMallocMapped));
        c.pinOutCap = outBytes + (outBytes >> 2);
    }
    SE3* oT = (SE3*)c.pinOut;
    double* otrace = (double*)(c.pinOut + ((sizeof(SE3) + 15) & ~(size_t)15));
    int* ores = (int*)(otrace + 192);
    unsigned char* opl = (unsigned char*)(ores + 4);          // kPoseMaxPlanes plane flags, then the point flags
    unsigned char* ooutl = opl + kPoseMaxPlanes;
    std::memset(opl, 0, kPoseMaxPlanes);
    std::memset(ooutl, 0, n);     // the memory variant of the kernel reads the flags before it first writes them
    ores[0] = ores[1] = ores[2] = ores[3] = 0;
    auto hostp = [&](const void* dev) { return c.pin + ((const unsigned char*)dev - a.base); };
    {
        double* hX = (double*)hostp(dXw); double* hO = (double*)hostp(dobs); double* hI = (double*)hostp(dinfo);
        unsigned char* hF = (unsigned char*)hostp(dflags);
        for (int i = 0; i < n; i++) {
            for (int k = 0; k < 3; k++) { hX[3 * i + k] = p->Xw[3 * i + k]; hO[3 * i + k] = p->obs[3 * i + k]; }
            hI[i] = p->inv_sigma2[i];
            hF[i] = (unsigned char)((!(p->obs[3 * i + 2] < 0) ? 1 : 0) | 4);
        }
        // planes: normalised world / measured coefficients and the two information values (src/Optimizer.cc:464-465, 503-516)
        double* hP = (double*)hostp(dplanes);
        const double angleInfo = refc::PLANE_ANGLE_INFO / (1.0 * 1.0), disInfo = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;
        for (int i = 0; i < M; i++) {
            plane_from_f32(p->plane_world + 4 * i, hP + 10 * i);
            plane_from_f32(p->plane_obs + 4 * i, hP + 10 * i + 4);
            const double f = p->plane_seen[i] ? 1.0 : 2.0;
            hP[10 * i + 8] = f * angleInfo; hP[10 * i + 9] = f * disInfo;
        }
    }
    EAO_HIP(hipEventRecord(c.ev0, c.stream));
    // The register variants read every input exactly once: they take it straight from the pinned mirror over PCIe (~60 KB) --
    // an upload in front of the kernel is a copy-engine job plus a hand-over to the compute queue (~10 us).  The memory
    // variant walks the edges in every LM pass and gets its copy.
    static const bool envUpload = getenv("EAO_POSE_UPLOAD") && atoi(getenv("EAO_POSE_UPLOAD")) != 0;      // (A/B switch)
    const bool zeroCopy = n <= 4 * kPoseThreads && M == 0 && !envUpload;      // (plane coefficients are re-read in every pass: uploaded)
    if (!zeroCopy) EAO_HIP(hipMemcpyAsync(a.base + off0, c.pin + off0, off1 - off0, hipMemcpyHostToDevice, c.stream));
    PoseDev P;
    P.nDev = nullptr; P.scatterIdx = nullptr; P.scatterOut = nullptr; P.done = nullptr; P.doneSeq = 0;
    P.n = n; P.Xw = dXw; P.obs = dobs; P.info = dinfo; P.err = derr; P.flags = dflags; P.outlier = ooutl;
    if (zeroCopy) { P.Xw = (const double*)hostp(dXw); P.obs = (const double*)hostp(dobs); P.info = (const double*)hostp(dinfo); P.flags = (unsigned char*)hostp(dflags); }
    P.T0 = se3_from_Tcw_f32(p->Tcw);
    P.cam.fx = p->fx; P.cam.fy = p->fy; P.cam.cx = p->cx; P.cam.cy = p->cy; P.cam.bf = p->bf; P.cam.bf_f = p->bf;
    P.cam.deltaMono = (float)std::sqrt(refc::POSE_HUBER2_MONO); P.cam.deltaStereo = (float)std::sqrt(refc::POSE_HUBER2_STEREO);
    P.Tout = oT; P.result = ores; P.trace = otrace;
    P.nPlanes = M; P.planes = dplanes; P.planeOutlier = opl; P.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2);
    P.dbg = getenv("EAO_DEBUG_STAMPS") ? ddbg : nullptr;
    if (n > 4 * kPoseThreads) hipLaunchKernelGGL(k_pose_optimization_mem, dim3(1), dim3(kPoseThreads), 0, c.stream, P);
    else launch_pose_registers(P, n, M > 0, c.stream);
    EAO_HIP(hipEventRecord(c.ev1, c.stream));
    EAO_HIP(hipStreamSynchronize(c.stream));
    EAO_HIP(hipGetLastError());
    const SE3 Tout = *oT;
    const int* res = ores;
    const double* trace = otrace;
    std::memcpy(r->outlier, ooutl, n);
    if (M) std::memcpy(r->plane_outlier, opl, M);
    if (P.dbg) {
        long long st[64];
        EAO_HIP(hipMemcpy(st, P.dbg, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[eao pose stamps] eval %lld tree + barrier %lld decide + solve + exp %lld clock64 ticks over %d LM iterations, %d passes (%d of them light); of the serial part, wave 0: 6 x 6 LDL^T + scale %lld, exp + compose + publish %lld ticks over %lld solves\n", st[0], st[1], st[2], res[1], res[3] & 0xFFFF, res[3] >> 16, st[3], st[4], st[5]);
        fprintf(stderr, "[eao pose stamps] per wave, heavy evaluation / light evaluation / tree:");
        for (int w = 0; w < (int)st[7] && w < 16; w++) fprintf(stderr, "  w%d %lld / %lld / %lld", w, st[8 + w], st[24 + w], st[40 + w]);
        fprintf(stderr, "\n");
    }
    se3_to_Tcw_f32(Tout, r->Tcw);
    r->n_inliers = n + M - res[0];
    r->lm_iterations = res[1];
    for (int k = 0; k < res[2] && k < 64; k++) {
        g_trace.lambda.push_back(trace[k]); g_trace.chi2.push_back(trace[64 + k]); g_trace.trials.push_back((int)trace[128 + k]);
    }
    g_trace.linearizations = res[1];
    EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, c.ev0, c.ev1));
    return EAO_OK;
}

// Optimizer::PoseOptimization for `nb` frames at once -- the candidate loop of Tracking::Relocalization (src/Tracking.cc:2786-2940: every
// candidate keyframe that passes PnP gets its own PoseOptimization) and offline replays.  One workgroup per frame, ONE upload,
// ONE launch per register variant, ONE synchronisation; each frame's result is what eao_pose_optimization returns for it
// (same kernel body, same reduction order).  Frames the one-workgroup register kernels do not take (more than
// 4 * kPoseThreads correspondences) go through eao_pose_optimization one by one.
// float -> double over one contiguous array (restrict-qualified and on its own, so that the host compiler vectorises it: the
// interleaved per-correspondence loop it replaces ran scalar and was half of a 256-frame call's host time)
static inline void widen(double* __restrict dst, const float* __restrict src, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (double)src[i];
}
eao_status eao_pose_optimization_batch(const eao_pose_problem* ps, int32_t nb, eao_pose_result* rs) {
    EAO_REQUIRE(nb >= 0 && (nb == 0 || (ps && rs)), "bad batch");
    if (nb == 0) return EAO_OK;
    for (int b = 0; b < nb; b++) {
        const eao_pose_problem* p = &ps[b];
        const eao_pose_result* r = &rs[b];
        EAO_REQUIRE(p->n >= 0 && p->Tcw && (p->n == 0 || (p->Xw && p->obs && p->inv_sigma2 && r->outlier)), "bad problem %d", b);
        EAO_REQUIRE(p->n_planes >= 0 && p->n_planes <= kPoseMaxPlanes, "at most %d plane edges (problem %d has %d)", kPoseMaxPlanes, b, p->n_planes);
        EAO_REQUIRE(p->n_planes == 0 || (p->plane_world && p->plane_obs && p->plane_seen && r->plane_outlier), "plane arrays missing (problem %d)", b);
    }
    LMContext& c = g_ctx;
    eao_status st = ctx_init(c, true, eao::StreamClass::Latency);
    if (st) return st;
    std::vector<int> grp[8], single;      // register kernels by geometry class (pose_class) x (plane edges or not)
    for (int b = 0; b < nb; b++) {
        const int n = ps[b].n;
        for (int i = 0; i < n; i++) rs[b].outlier[i] = 0;
        rs[b].lm_iterations = 0;
        if (n < 3) { std::memcpy(rs[b].Tcw, ps[b].Tcw, 16 * sizeof(float)); rs[b].n_inliers = 0; }
        else if (n > 4 * kPoseThreads) single.push_back(b);
        else grp[pose_class(n) + (ps[b].n_planes > 0 ? 4 : 0)].push_back(b);
    }
    int nk = 0;
    for (int g = 0; g < 8; g++) nk += (int)grp[g].size();
    if (nk) {
        // arena: [records | per frame: Xw, obs, info, flags, planes] uploaded, then the per-frame residual scratch
        struct Slot { int b; double *Xw, *obs, *info, *planes, *err; unsigned char* flags; size_t out; };
        std::vector<Slot> slots;
        size_t need = (size_t)nk * sizeof(PoseDev) + 512, outBytes = 0;
        const size_t outFixed = ((sizeof(SE3) + 15) & ~(size_t)15) + 192 * 8 + 16 + kPoseMaxPlanes;
        for (int g = 0; g < 8; g++)
            for (int b : grp[g]) {
                need += (size_t)ps[b].n * (3 + 3 + 1 + 3) * 8 + (size_t)ps[b].n + (size_t)kPoseMaxPlanes * 80 + 6 * 256;
                outBytes += (outFixed + (size_t)ps[b].n + 63) & ~(size_t)63;
            }
        if ((st = c.bytes.reserve(need))) return st;
        Arena a{c.bytes.p, c.bytes.n};
        const size_t off0 = a.off;
        PoseDev* dW = a.take<PoseDev>(nk);
        for (int g = 0; g < 8; g++)
            for (int b : grp[g]) {
                Slot s{};
                const int n = ps[b].n;
                s.b = b;
                s.Xw = a.take<double>((size_t)n * 3); s.obs = a.take<double>((size_t)n * 3); s.info = a.take<double>(n);
                s.flags = a.take<unsigned char>(n); s.planes = a.take<double>((size_t)kPoseMaxPlanes * 10);
                slots.push_back(s);
            }
        const size_t off1 = (a.off + 255) & ~(size_t)255;
        for (Slot& s : slots) s.err = a.take<double>((size_t)ps[s.b].n * 3);
        if (c.pinCap < off1) {
            if (c.pin) (void)hipHostFree(c.pin);
            c.pin = nullptr; c.pinCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pin, off1 + (off1 >> 2), hipHostMallocDefault));
            c.pinCap = off1 + (off1 >> 2);
        }
        if (c.pinOutCap < outBytes) {
            if (c.pinOut) (void)hipHostFree(c.pinOut);
            c.pinOut = nullptr; c.pinOutCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pinOut, outBytes + (outBytes >> 2), hipHostMallocMapped));
            c.pinOutCap = outBytes + (outBytes >> 2);
        }
        auto hostp = [&](const void* dev) { return c.pin + ((const unsigned char*)dev - a.base); };
        PoseDev* hW = (PoseDev*)hostp(dW);
        const bool zeroCopy = false;       // (the single call reads its inputs from the pinned mirror; for a batch it was measured and makes no difference)
        size_t oo = 0;
        const double angleInfo = refc::PLANE_ANGLE_INFO / (1.0 * 1.0), disInfo = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;     // src/Optimizer.cc:464-465
        for (size_t k = 0; k < slots.size(); k++) {
            Slot& s = slots[k];
            const eao_pose_problem* p = &ps[s.b];
            const int n = p->n, M = p->n_planes;
            s.out = oo;
            oo += (outFixed + (size_t)n + 63) & ~(size_t)63;
            unsigned char* o = c.pinOut + s.out;
            double* otrace = (double*)(o + ((sizeof(SE3) + 15) & ~(size_t)15));
            int* ores = (int*)(otrace + 192);
            unsigned char* opl = (unsigned char*)(ores + 4);
            std::memset(opl, 0, kPoseMaxPlanes + (size_t)n);
            ores[0] = ores[1] = ores[2] = ores[3] = 0;
            double* hX = (double*)hostp(s.Xw); double* hO = (double*)hostp(s.obs); double* hI = (double*)hostp(s.info);
            unsigned char* hF = (unsigned char*)hostp(s.flags);
            widen(hX, p->Xw, (size_t)n * 3); widen(hO, p->obs, (size_t)n * 3); widen(hI, p->inv_sigma2, n);
            for (int i = 0; i < n; i++) hF[i] = (unsigned char)((!(p->obs[3 * i + 2] < 0) ? 1 : 0) | 4);
            double* hP = (double*)hostp(s.planes);
            for (int i = 0; i < M; i++) {
                plane_from_f32(p->plane_world + 4 * i, hP + 10 * i);
                plane_from_f32(p->plane_obs + 4 * i, hP + 10 * i + 4);
                const double f = p->plane_seen[i] ? 1.0 : 2.0;
                hP[10 * i + 8] = f * angleInfo; hP[10 * i + 9] = f * disInfo;
            }
            PoseDev& P = hW[k];
            std::memset((void*)&P, 0, sizeof(PoseDev));
            P.nDev = nullptr;
            P.n = n; P.Xw = s.Xw; P.obs = s.obs; P.info = s.info; P.err = s.err; P.flags = s.flags; P.outlier = opl + kPoseMaxPlanes;
            if (zeroCopy) {     // (read once by the register kernels: straight from the pinned mirror, the transfer overlaps other frames' LM)
                P.Xw = (const double*)hostp(s.Xw); P.obs = (const double*)hostp(s.obs); P.info = (const double*)hostp(s.info); P.flags = (unsigned char*)hostp(s.flags);
            }
            P.T0 = se3_from_Tcw_f32(p->Tcw);
            P.cam.fx = p->fx; P.cam.fy = p->fy; P.cam.cx = p->cx; P.cam.cy = p->cy; P.cam.bf = p->bf; P.cam.bf_f = p->bf;
            P.cam.deltaMono = (float)std::sqrt(refc::POSE_HUBER2_MONO); P.cam.deltaStereo = (float)std::sqrt(refc::POSE_HUBER2_STEREO);
            P.Tout = (SE3*)o; P.result = ores; P.trace = otrace;
            P.nPlanes = M; P.planes = s.planes; P.planeOutlier = opl; P.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2);
            P.dbg = nullptr;
        }
        g_trace.clear();
        EAO_HIP(hipEventRecord(c.ev0, c.stream));
        const PoseDev* kW = dW;
        if (zeroCopy) kW = hW;          // (the records too)
        else EAO_HIP(hipMemcpyAsync(a.base + off0, c.pin + off0, off1 - off0, hipMemcpyHostToDevice, c.stream));
        {   // one launch per (geometry class, planes) group; the records lie group after group
            size_t first = 0;
            for (int g = 0; g < 8; g++) {
                if (grp[g].empty()) continue;
                const dim3 gr((unsigned)grp[g].size()), bl(g % 4 == 3 ? kPoseThreads : kPoseWaves4);
                const PoseDev* w = kW + first;
                switch (g) {
                    case 0: hipLaunchKernelGGL((k_pose_optimization_batch<1, false, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    case 1: hipLaunchKernelGGL((k_pose_optimization_batch<2, false, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    case 2: hipLaunchKernelGGL((k_pose_optimization_batch<4, false, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    case 3: hipLaunchKernelGGL((k_pose_optimization_batch<4, false, kPoseThreads>), gr, bl, 0, c.stream, w); break;
                    case 4: hipLaunchKernelGGL((k_pose_optimization_batch<1, true, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    case 5: hipLaunchKernelGGL((k_pose_optimization_batch<2, true, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    case 6: hipLaunchKernelGGL((k_pose_optimization_batch<4, true, kPoseWaves4>), gr, bl, 0, c.stream, w); break;
                    default: hipLaunchKernelGGL((k_pose_optimization_batch<4, true, kPoseThreads>), gr, bl, 0, c.stream, w); break;
                }
                first += grp[g].size();
            }
        }
        EAO_HIP(hipEventRecord(c.ev1, c.stream));
        EAO_HIP(hipStreamSynchronize(c.stream));
        EAO_HIP(hipGetLastError());
        for (const Slot& s : slots) {
            const eao_pose_problem* p = &ps[s.b];
            eao_pose_result* r = &rs[s.b];
            const unsigned char* o = c.pinOut + s.out;
            const SE3 Tout = *(const SE3*)o;
            const double* otrace = (const double*)(o + ((sizeof(SE3) + 15) & ~(size_t)15));
            const int* ores = (const int*)(otrace + 192);
            const unsigned char* opl = (const unsigned char*)(ores + 4);
            std::memcpy(r->outlier, opl + kPoseMaxPlanes, p->n);
            if (p->n_planes) std::memcpy(r->plane_outlier, opl, p->n_planes);
            se3_to_Tcw_f32(Tout, r->Tcw);
            r->n_inliers = p->n + p->n_planes - ores[0];
            r->lm_iterations = ores[1];
        }
        EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, c.ev0, c.ev1));
    }
    for (int b : single)
        if ((st = eao_pose_optimization(&ps[b], &rs[b]))) return st;
    return EAO_OK;
}

// mode 0: Optimizer::LocalBundleAdjustment (two passes with the outlier pass between them, Huber kernels in the first).
// mode 1: Optimizer::BundleAdjustment over keyframes and map points (src/Optimizer.cc:55-323): ONE optimize(its_first) call,
//         Huber kernels only when `robust`, delta_mono = sqrt(5.99) (:94), no outlier pass, no observation is erased.
//         With `pl`: the MapPlane vertices / EdgePlane edges of :203-252 ride along as landmarks nPo.. / edges Ept.. .
}  // extern "C"

namespace {

// Launch geometry of one window -- or, field by field, the largest of a batch (every kernel guards its own window's sizes).
struct BADims {
    int nF = 0, nL = 0, nP = 0, nC = 0, E = 0, nPl = 0;
    bool hasPl = false, bigPath = false, usePairs = false, solveTiles = false, tiles3 = true, wmode = false;
    size_t tileLds = 0;
    int nPairsNZ = 0;          // map-scale path (never batched)
    int nPairsLong = 0, nPairsSlots = 0;        // " : launch slots of the four-wave kernel (first in lpOrder) / of both
    BigStepArgs bigArgs{};     // " : what k_bal_step takes by value (ctl / wa0 / wb0x filled per launch)
    const int* bigCtl0 = nullptr; const int4* bigPanelWork = nullptr;      // " : the control blocks on the device; the host copy of the work records
    double* big = nullptr;     // "
    int bigTiles = 0;          // "
    const int* bigPanelStart = nullptr;      // " : where every 32-column panel's work records start (nbk + 1 entries, thread-local storage that outlives the call's launches;
                                             //     a POINTER: this struct is copied around, a vector member would be re-allocated with every copy)
    BigGeom gB{};
    void merge(const BADims& o) {
        nF = std::max(nF, o.nF); nL = std::max(nL, o.nL); nP = std::max(nP, o.nP); nC = std::max(nC, o.nC); E = std::max(E, o.E);
        tiles3 = tiles3 && o.tiles3; tileLds = std::max(tileLds, o.tileLds);
    }
};

// Where launches go: `nz` windows (device array W) on one stream.  Every kernel of the LM engine is launched from here, with
// the window as grid.z -- a single window is a batch of one.
struct BALaunch {
    const BADev* W = nullptr;
    int nz = 1;
    BADims d;
    hipStream_t s = nullptr;
    int seq = 0;
    eao_status attributes() const {
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
        static std::atomic<int> aT3{0}, aT5{0}, aBack{0};
        eao_status st;
        if (d.solveTiles) {
            if (d.tiles3) { if ((st = raise((const void*)k_ba_solve_tiles<3>, aT3, d.tileLds))) return st; }
            else if ((st = raise((const void*)k_ba_solve_tiles<5>, aT5, d.tileLds))) return st;
        }
        if (d.bigPath) { if ((st = raise((const void*)k_bal_backsolve, aBack, (kBigSB / kBigNB) * kBigNB * kBigNB * sizeof(double)))) return st; }
        return EAO_OK;
    }
    int rot = 0;               // XCD of window 0 (BA_WIN)
    int wp(int par) const { return par | (nz > 1 ? (rot & 7) << 4 | nz << 8 : 0); }      // kernel argument (BA_WIN)
    unsigned gz() const { return nz > 1 ? (unsigned)((nz + 7) & ~7) : 1u; }   // windows are dealt to the XCDs: grid.z padded to 8
    int ptBlocks() const { return eao::cdiv(std::max(d.nL, 1) * 8, 256); }       // eight lanes per landmark
    int linBlocks() const { return eao::cdiv(std::max(d.nL, 1) * 8, kLinThreads); }
    // the pre-scaled-block linearisation: 1024-thread workgroups in a batch (48.9 us per 25 windows against 51.8 with 512), 512 / 256 for one window -- its 24 landmark
    // workgroups of 1024 leave nine tenths of the chip idle (EAO_BA_LIN_THREADS for A/B runs; profiles/r04_ba_pair_ablation.txt)
    template <int NT> void lin_launch(int par, int first, int diagOnly) const {
        const int lb = eao::cdiv(std::max(d.nL, 1) * 8, NT);
        hipLaunchKernelGGL((k_ba_linearize<false, NT, true>), dim3(lb + d.nF, 1, gz()), dim3(NT), 0, s, W, wp(par), lb, first, diagOnly);
    }
    void lin_w(int par, int first, int diagOnly) const {
        static const int envLinT = getenv("EAO_BA_LIN_THREADS") ? atoi(getenv("EAO_BA_LIN_THREADS")) : 0;
        const int nt = envLinT ? envLinT : (nz > 1 ? 1024 : 512);
        if (nt == 256) lin_launch<256>(par, first, diagOnly);
        else if (nt == 512) lin_launch<512>(par, first, diagOnly);
        else lin_launch<kLinThreads>(par, first, diagOnly);
    }
    void setup() const {      // device-side part of the set-up (once per window)
        hipLaunchKernelGGL(k_ba_prepare, dim3(eao::cdiv(std::max(std::max(std::max(std::max(std::max(d.bigPath ? 0 : d.nL * d.nF, d.nL * 8), d.nP * 3), d.nC), d.E), d.nPl * 4), 256), 1, gz()), dim3(256), 0, s, W, wp(0));
        if (d.usePairs) {
            hipLaunchKernelGGL(k_ba_pairs, dim3(d.nF * (d.nF + 1) / 2, 1, gz()), dim3(256), 0, s, W, wp(0));
            hipLaunchKernelGGL(k_ba_tiles_init, dim3(tile_geom(d.nF).nTiles, 1, gz()), dim3(256), 0, s, W, wp(0));
        }
    }
    // one LM trial behind a linearisation: Schur assembly, solve, back substitution + residuals, (decision)
    void trial(int par, int bulk, bool firstTrial, bool withDecide) {
        const int nF = d.nF, nL = d.nL;
        if (nF && d.bigPath) {
            const BigGeom gB = d.gB;
            (void)hipMemsetAsync(d.big, 0, ((size_t)d.bigTiles << 12) * sizeof(double), s);
            if (d.nPairsLong) hipLaunchKernelGGL(k_bal_schur_pairs<256>, dim3(d.nPairsLong), dim3(256), 0, s, W, wp(par), firstTrial ? 1 : 0, 0);
            if (d.nPairsSlots > d.nPairsLong) hipLaunchKernelGGL(k_bal_schur_pairs<64>, dim3(d.nPairsSlots - d.nPairsLong), dim3(64), 0, s, W, wp(par), firstTrial ? 1 : 0, d.nPairsLong);
            const int nbk = gB.N / kBigNB;
            hipLaunchKernelGGL(k_bal_diag0, dim3(1), dim3(64), 0, s, W, wp(par));
            BigStepArgs A = d.bigArgs;
            A.ctl = d.bigCtl0 + 8 * (par & 1);
            for (int kb = 0; kb < nbk; kb++) {
                // the panel's work list: the trailing tiles both of whose tile rows are live in the panel's tile column (a dense system: every trailing tile)
                const int4* rec = d.bigPanelWork + 2 * (size_t)d.bigPanelStart[kb];
                A.wa0 = rec[0]; A.wb0x = rec[1].x;
                hipLaunchKernelGGL(k_bal_step, dim3(d.bigPanelStart[kb + 1] - d.bigPanelStart[kb]), dim3(256), 0, s, A, kb, kb + 1 == nbk ? 1 : 0, d.bigPanelStart[kb]);
            }
            hipLaunchKernelGGL(k_bal_linv, dim3(eao::cdiv(nbk, 2)), dim3(64), 0, s, W, wp(par), nbk);
            for (int J = eao::cdiv(gB.N, kBigSB) - 1; J >= 0; J--)
                hipLaunchKernelGGL(k_bal_backsolve, dim3(std::max(1, J * kBigSB / 64)), dim3(256), (kBigSB / kBigNB) * kBigNB * kBigNB * sizeof(double), s, W, wp(par), J);
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
    void relinearize() {
        if (!d.wmode || !d.nL) return;
        hipLaunchKernelGGL(k_ba_errors<false>, dim3(ptBlocks(), 1, gz()), dim3(256), 0, s, W, wp(0));
        lin_w(0, 0, 0);
    }
    // iterations [from, to) of an optimize() call, one trial each, no host round trip
    void bulk(int from, int to, bool withErrors) {
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
    void classify() const { hipLaunchKernelGGL(k_ba_classify, dim3(eao::cdiv(std::max(d.E, 1), 256), 1, gz()), dim3(256), 0, s, W, wp(0), 1); }
    void finish() const {
        hipLaunchKernelGGL(k_ba_finish, dim3(eao::cdiv(std::max(std::max(std::max(std::max(d.E, d.nP * 3), d.nC), d.nPl * 4), 1), 256), 1, gz()), dim3(256), 0, s, W, wp(0));
    }
    // Both optimize() calls of the reference, the outlier pass between them and the result copy in ONE enqueue.  A rejected LM
    // trial freezes the rest of ITS window's work (halt flag + the takeover check of the outlier pass); the host then
    // continues that window from where the device stopped.
    void chain(int mode, int itsFirst, int itsSecond) {
        eao::Range rg("lm: optimize x2 + outlier pass, enqueue");
        bulk(0, itsFirst, true);
        if (mode == 0) { classify(); bulk(0, itsSecond, true); }
        finish();
    }
};

// One window in flight: LocalBundleAdjustment / BundleAdjustment of one problem on one context (device arena + pinned mirrors).
// The uploads of a batch group: window y of the launch is copied from its pinned host mirror (read over PCIe by the kernel itself) into
// its device arena, 16 bytes per lane.  Both ends are 16-byte aligned (arena offsets are multiples of 256).
struct BAUploadArgs { unsigned char* dst[8]; const unsigned char* src[8]; unsigned long long n16[8]; };
__global__ __launch_bounds__(256) void k_ba_upload(BAUploadArgs A) {
    const int w = blockIdx.y;
    const uint4* __restrict__ s = reinterpret_cast<const uint4*>(A.src[w]);
    uint4* __restrict__ d = reinterpret_cast<uint4*>(A.dst[w]);
    const unsigned long long n = A.n16[w];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) d[i] = s[i];
}

// The host threads of a batch call (window set-up workers, group leaders) are PERSISTENT: a call hands `count` tasks to the crew and
// joins them.  Starting eight std::threads per call cost ~0.3 ms of a 2.9 ms batch (clone + first-touch of the thread's HIP state,
// one after the other) -- which is why more set-up threads used to make a call slower.  The crew is process-wide, grows on
// demand and is never torn down (its threads sleep on a condition variable between calls and die with the process).
thread_local bool t_inCrew = false;      // this thread belongs to the crew: a task must not hand work to the crew itself (one call at a time owns it)
struct HostCrew {
    std::mutex m;
    std::condition_variable wake, finished;
    std::function<void(int)> fn;
    int generation = 0, next = 0, count = 0, running = 0, threads = 0;
    void body() {
        t_inCrew = true;
        int seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            wake.wait(lk, [&] { return generation != seen && next < count; });
            while (next < count) {
                const int i = next++;
                running++;
                lk.unlock();
                fn(i);
                lk.lock();
                running--;
            }
            seen = generation;
            if (running == 0) finished.notify_all();
        }
    }
    // runs fn(0 .. n-1) on the crew (at least n threads, so tasks that wait for each other cannot starve) and fn0() on the caller
    std::mutex callMu;      // one batch call at a time uses the crew (calls from several host threads queue up here)
    void run(int n, const std::function<void(int)>& f, const std::function<void()>& fn0) {
        std::lock_guard<std::mutex> oneCall(callMu);
        {
            std::unique_lock<std::mutex> lk(m);
            while (threads < n) { std::thread(&HostCrew::body, this).detach(); threads++; }
            fn = f; next = 0; count = n; generation++;
        }
        wake.notify_all();
        fn0();
        std::unique_lock<std::mutex> lk(m);
        finished.wait(lk, [&] { return next >= count && running == 0; });
        count = 0;
    }
};
HostCrew& host_crew() {
    // ONE crew per process (round 4; it was one per calling thread: a pool of short-lived caller threads grew the process by ~19 sleeping threads per
    // caller).  Its size is the largest thread count a call ever asked for (the set-up threads + group leaders of eao_local_ba_batch: about
    // nineteen with the defaults); the threads sleep on a condition variable between calls and end with the process (detached: a static destructor
    // that joined them would run after the HIP runtime's own teardown).
    static HostCrew* crew = new HostCrew();
    return *crew;
}

struct BAJob {
    const eao_ba_problem* p = nullptr; const volatile uint8_t* stop = nullptr; eao_ba_result* r = nullptr;
    int mode = 0, robust = 1; const eao_ba_planes* pl = nullptr; float* planes_out = nullptr;
    LMContext* c = nullptr; LMTraceHost* tr = nullptr;
    int nPo = 0, nPl = 0, Ept = 0, Epl = 0, nC = 0, nP = 0, E = 0;
    bool hasPl = false, trivial = false, chained = false, pollStop = false, lazy = false;
    int nPairsLong = 0, nPairsSlots = 0;      // map-scale path: launch slots (lpOrder) of the four-wave assembly kernel / of both kernels
    BADev D; BADev* dW = nullptr;
    BALaunch L;
    int curHost = 0;
    SE3* outCams = nullptr; double* outPts = nullptr; double* outPlanes = nullptr; unsigned char* outCls = nullptr;

    void write_records(BADev* dst) const {      // the two records of this window (see BA_WIN)
        dst[0] = D; dst[0].ctl = D.ctl0; dst[0].lm = D.lm0;
        dst[1] = D; dst[1].ctl = D.ctl0 + 8; dst[1].lm = D.lm0 + 8;
    }
    bool batchable() const { return !trivial && chained && L.d.usePairs && L.d.solveTiles && !hasPl && !L.d.bigPath && D.nFree > 0 && D.nL > 0 && mode == 0; }

    // validation, arena, pinned mirror, upload (two copies on `s`), active structure.  No kernel is launched here.
    // deferUpload (batches): NO call into the HIP runtime at all -- the pinned mirror is filled and [upSrc, upSrc + upBytes) is left
    // for the group's leader, which moves every window of its group with ONE launch of k_ba_upload (the copies' enqueue calls
    // serialise inside the runtime: 50 of them were most of a batch's 0.55 ms of set-up, and more host threads made it worse).
    const unsigned char* upSrc = nullptr; unsigned char* upDst = nullptr; size_t upBytes = 0;
    eao_status prepare(hipStream_t s, bool deferUpload = false) {
        eao::Range rg("lm: window set-up + upload");
        EAO_REQUIRE(p && r && r->cam_Tcw && r->points && (p->n_edges == 0 || r->edge_outlier || mode == 1), "null argument");
        EAO_REQUIRE(p->n_cams > 0 && p->n_points >= 0 && p->n_edges >= 0, "bad sizes");
        if (pl && pl->n_planes <= 0) pl = nullptr;
        EAO_REQUIRE(!pl || (mode == 1 && pl->plane_world && planes_out && pl->n_pedges >= 0 && (pl->n_pedges == 0 || (pl->pedge_plane && pl->pedge_cam && pl->pedge_obs))),
                    "bad plane arguments");
        LMContext& c = *this->c;
        eao_status st;
        tr->clear();
        static const bool hostStamps = getenv("EAO_DEBUG_STAMPS") != nullptr;      // host phases of the set-up, in ms on stderr
        const auto hs0 = std::chrono::steady_clock::now();
        double hsT[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto hs_lap = [&](int k) { if (hostStamps) hsT[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count(); };
        nPo = p->n_points; nPl = pl ? pl->n_planes : 0; Ept = p->n_edges; Epl = pl ? pl->n_pedges : 0;
        nC = p->n_cams; nP = nPo + nPl; E = Ept + Epl;          // landmarks = points then planes, edges = point edges then plane edges
        hasPl = nPl > 0;
        const eao_ba_problem* p = this->p; const eao_ba_planes* pl = this->pl;
        const int nPo = this->nPo, Ept = this->Ept;
        auto edge_cam = [=](int e) { return e < Ept ? p->edge_cam[e] : pl->pedge_cam[e - Ept]; };
        auto edge_lm = [=](int e) { return e < Ept ? p->edge_point[e] : nPo + pl->pedge_plane[e - Ept]; };
        r->iters[0] = r->iters[1] = 0; r->aborted = 0; r->chi2[0] = r->chi2[1] = 0;
        if (stop && *stop) {  // src/Optimizer.cc:961-963: nothing is optimised; poses go through the same SE3 round trip
            r->aborted = 1;
            for (int i = 0; i < nC; i++) se3_to_Tcw_f32(se3_from_Tcw_f32(p->cam_Tcw + 16 * i), r->cam_Tcw + 16 * i);
            for (size_t i = 0; i < (size_t)nPo * 3; i++) r->points[i] = p->points[i];
            for (int i = 0; i < nPl; i++) { double c4[4]; plane_from_f32(pl->plane_world + 4 * i, c4); for (int k = 0; k < 4; k++) planes_out[4 * i + k] = (float)c4[k]; }
            if (Ept && r->edge_outlier) std::memset(r->edge_outlier, 0, Ept);
            trivial = true;
            return EAO_OK;
        }
        bool edgesByLandmark = true;      // the edge list is grouped landmark by landmark, ascending (what the adapters and every generator produce): ptEdges is then the identity
        // ... and, in the same pass over the edges, the edge counts per camera and per landmark the active structure starts from
        std::vector<int>& cnt = c.scratch;
        cnt.assign((size_t)nC + nP, 0);
        int* const camCnt = cnt.data(); int* const ptCnt = camCnt + nC;
        for (int e = 0, prev = 0; e < Ept; e++) {
            const int ec = p->edge_cam[e], ep = p->edge_point[e];
            EAO_REQUIRE(ec >= 0 && ec < nC && ep >= 0 && ep < nPo, "edge %d out of range", e);
            edgesByLandmark = edgesByLandmark && ep >= prev; prev = ep;
            camCnt[ec]++; ptCnt[ep]++;
        }
        for (int e = 0, prev = 0; e < Epl; e++) {
            const int ec = pl->pedge_cam[e], ep = pl->pedge_plane[e];
            EAO_REQUIRE(ec >= 0 && ec < nC && ep >= 0 && ep < nPl, "plane edge %d out of range", e);
            edgesByLandmark = edgesByLandmark && ep >= prev; prev = ep;
            camCnt[ec]++; ptCnt[nPo + ep]++;
        }
        int nFreeIn = 0;
        for (int i = 0; i < nC; i++) nFreeIn += p->cam_fixed[i] ? 0 : 1;
        EAO_REQUIRE(nFreeIn <= kBigMaxFree, "at most %d free keyframes in this build (got %d)", kBigMaxFree, nFreeIn);
        // more free keyframes than the single-workgroup solvers take (or EAO_BA_SOLVER=big, the harness's A/B switch): the
        // map-scale path, dense system in HBM factorised by the whole chip (k_bal_*)
        static const char* solverEnv0 = getenv("EAO_BA_SOLVER");
        // (measured, LocalBundleAdjustment wall time, tools/dbg_ba_sizes.py: the LDS / global-scratch single-workgroup solver with
        //  the slab assembly takes 5.8 ms at 31 free keyframes and 29 ms at 64, the map-scale path 3.6 and 6.9 ms -- so everything
        //  beyond the register-tile solver goes there; that older path was removed in round 5)
        const bool bigPath = nFreeIn > kTileMaxFree || (nFreeIn > 0 && solverEnv0 && !strcmp(solverEnv0, "big"));
        const BigGeom bg = big_geom(std::max(nFreeIn, 1));
        size_t lpEntries = 0, lpPairsMax = 0;
        // ---- round 5: the covisibility structure of the map-scale path, CAMERA-MAJOR.  For every free camera i1 (ascending) the landmarks it observes in ascending
        //      order, and for each of them its observers i2 >= i1: the pairs (i1, i2) of camera i1 are counted in a counter array of nF entries that stays in the
        //      cache, come out sorted, and their entries are later written into ONE contiguous range per camera -- in ascending landmark order, which is the order the
        //      assembly's fixed-order sums need.  (Rounds 3-5 walked the landmarks and scattered every (pair, landmark) entry through a counter per pair of the
        //      whole nF (nF + 1) / 2 triangle, three times -- once into a byte matrix for the tile structure, once to count, once to fill: 9.8 + 2.4 ms of host time in
        //      front of 26.7 ms of device time on the banded 1000-keyframe map.)  Same arrays as before, bit for bit.
        //      Every landmark's observer list is sorted by camera, so a camera's partners i2 >= i1 in a landmark are the SUFFIX behind its own entry: no test per
        //      observer (it failed half the time and mispredicted).  Both walks -- counting and filling -- are split over the host crew by camera ranges of equal
        //      size taken from a shared counter (a camera's pairs and entries are its own: no two workers write the same word).
        static thread_local std::vector<int> fidx, lmOff, lmCam, lmEdge, cmOff, cmLm, cmU, prA, prB, prStart, cmPairStart, pcur;
        // workers for the two walks: the crew unless this thread is one of its own (a map-scale window inside a batch call), or the map is small
        auto crew_for = [&](size_t work, int nChunks, const std::function<void(int)>& chunk) {
            const int hw = (int)std::thread::hardware_concurrency();
            const char* envS = getenv("EAO_BA_SETUP_THREADS");      // (read per call: the tests force the crew onto small maps; 1 = never)
            const int envT = envS ? atoi(envS) : 0;
            const int nT = t_inCrew || (work < 200000 && envT <= 0) ? 1 : std::max(1, std::min(envT > 0 ? envT : std::min(12, hw / 2), nChunks));
            if (nT == 1) { for (int q = 0; q < nChunks; q++) chunk(q); return; }
            std::atomic<int> next(0);
            auto body = [&]() { for (int q; (q = next.fetch_add(1)) < nChunks;) chunk(q); };
            host_crew().run(nT - 1, [&](int) { body(); }, body);
        };
        // ---- round 5: the TILE structure of the map-scale system.  Which 64 x 64 tiles of the lower triangle can ever be non-zero: the tiles a covisible camera pair's
        //      6 x 6 block touches, the diagonal, the tile row of the right-hand side -- and the fill-in of the elimination, worked out here once at tile level
        //      (eliminating tile column k joins every pair of tile rows that are live in it: the block form of the symbolic factorisation a sparse LDL^T starts
        //      with, solvers/linear_solver_eigen.h:95-112; natural keyframe order -- consecutive keyframes are the covisible ones, which is what a fill-reducing
        //      ordering would recover).  Memory and the launches' grids follow this structure.
        static thread_local std::vector<int> tileMap;       // (consumed inside this function: copied into the pinned mirror below)
        std::vector<int>& panelStart = c.bigPanelStart;     // (read by the launches of this window, long after this function has returned: the context's)
        std::vector<int4>& panelWork = c.bigPanelWork;
        int bigT = 0, bigTiles = 0;
        if (bigPath) {
            static thread_local std::vector<unsigned char> live;
            // free cameras with at least one edge, in ascending order (the numbering the active structure below gives them: camIdx)
            fidx.assign((size_t)nC, -1);
            lmOff.assign((size_t)nP + 1, 0);
            int nFa = 0;
            for (int i = 0; i < nC; i++) if (camCnt[i] && !p->cam_fixed[i]) fidx[i] = nFa++;      // (camCnt: the validation pass)
            // per landmark: its free observers and their edges, in edge order (the order of the active structure's ptEdges)
            bool byLandmark = true;        // the edges come landmark by landmark (the adapters and every generator list them so): the observer lists are then a filtered copy
            for (int e = 0, prev = 0; e < E; e++) { const int lmk = edge_lm(e); byLandmark = byLandmark && lmk >= prev; prev = lmk; if (fidx[edge_cam(e)] >= 0) lmOff[lmk + 1]++; }
            for (int i = 0; i < nP; i++) {
                const int m = lmOff[i + 1];
                lpEntries += (size_t)m * (m + 1) / 2;
                lmOff[i + 1] += lmOff[i];
            }
            EAO_REQUIRE(lpEntries < ((size_t)1 << 31), "covisibility structure too large (%zu pair entries)", lpEntries);
            hs_lap(8);
            lmCam.resize((size_t)lmOff[nP] + 1); lmEdge.resize((size_t)lmOff[nP] + 1);      // (+ 1: the branch-free append writes one slot ahead)
            cmOff.assign((size_t)nFa + 1, 0);
            if (byLandmark) {
                // (plain pointers and a branch-free append: the loop is a stream of 2 E loads and at most 2 E stores)
                int* const lc = lmCam.data(); int* const le = lmEdge.data(); int* const co = cmOff.data() + 1; const int* const fi = fidx.data();
                const int* const ecam = p->edge_cam; const int* const pcam = pl ? pl->pedge_cam : nullptr;
                int at = 0;
                for (int e = 0; e < Ept; e++) { const int f = fi[ecam[e]]; lc[at] = f; le[at] = e; const int ok = f >= 0; at += ok; if (ok) co[f]++; }
                for (int e = Ept; e < E; e++) { const int f = fi[pcam[e - Ept]]; lc[at] = f; le[at] = e; const int ok = f >= 0; at += ok; if (ok) co[f]++; }
            } else {
                pcur.assign(lmOff.begin(), lmOff.end() - 1);
                for (int e = 0; e < E; e++) {
                    const int f = fidx[edge_cam(e)];
                    if (f < 0) continue;
                    const int at = pcur[edge_lm(e)]++;
                    lmCam[at] = f; lmEdge[at] = e; cmOff[f + 1]++;
                }
            }
            hs_lap(9);
            for (int i = 0; i < nP; i++)           // observers by camera (insertion sort: a handful per landmark, mostly in order already)
                for (int u = lmOff[i] + 1; u < lmOff[i + 1]; u++) {
                    const int cf = lmCam[u], ce = lmEdge[u];
                    int v = u;
                    for (; v > lmOff[i] && lmCam[v - 1] > cf; v--) { lmCam[v] = lmCam[v - 1]; lmEdge[v] = lmEdge[v - 1]; }
                    lmCam[v] = cf; lmEdge[v] = ce;
                }
            hs_lap(10);
            // per free camera: its landmarks in ascending order (a counting sort over the landmarks, walked in ascending order), each with the position of the
            // camera's own entry in that landmark's list
            for (int f = 0; f < nFa; f++) cmOff[f + 1] += cmOff[f];
            cmLm.resize(cmOff[nFa]); cmU.resize(cmOff[nFa]);
            {   // (camera ranges on the crew: every worker walks all observer lists and files the entries of ITS cameras -- a camera's list is written by one worker, in landmark order)
                const int nRanges = std::max(1, std::min(16, nFa / 32));
                const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const cmOffp = cmOff.data();
                int* const cmLmp = cmLm.data(); int* const cmUp = cmU.data();
                const int nPl_ = nP;
                crew_for(lpEntries, nRanges, [=](int q) {
                    const int f0 = (int)((long long)nFa * q / nRanges), f1 = (int)((long long)nFa * (q + 1) / nRanges);
                    static thread_local std::vector<int> cur;
                    cur.assign(cmOffp + f0, cmOffp + f1);
                    for (int i = 0; i < nPl_; i++)
                        for (int u = lmOffp[i]; u < lmOffp[i + 1]; u++) {
                            const int f = lmCamp[u];
                            if (f < f0 || f >= f1) continue;
                            const int at = cur[f - f0]++;
                            cmLmp[at] = i; cmUp[at] = u;
                        }
                });
            }
            hs_lap(5);
            // the pairs of every camera and their entry counts: chunks of cameras, each into lists of its own, joined in camera order
            constexpr int kChunkCams = 16;
            const int nChunks = (nFa + kChunkCams - 1) / kChunkCams;
            static thread_local std::vector<std::vector<int>> chB, chCnt;
            chB.resize(nChunks); chCnt.resize(nChunks);
            cmPairStart.assign((size_t)nFa + 1, 0);
            {
                int* const pairsOfCam = cmPairStart.data() + 1;
                const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const cmOffp = cmOff.data();
                const int* const cmLmp = cmLm.data(); const int* const cmUp = cmU.data();
                std::vector<int>* const chBp = chB.data(); std::vector<int>* const chCntp = chCnt.data();
                crew_for(lpEntries, nChunks, [=](int q) {
                    static thread_local std::vector<int> cnt2, touched;
                    cnt2.assign((size_t)nFa, 0);
                    std::vector<int>& oB = chBp[q]; std::vector<int>& oC = chCntp[q];
                    oB.clear(); oC.clear();
                    for (int i1 = q * kChunkCams; i1 < std::min(nFa, (q + 1) * kChunkCams); i1++) {
                        touched.clear();
                        for (int k = cmOffp[i1]; k < cmOffp[i1 + 1]; k++)
                            for (int u = cmUp[k], ue = lmOffp[cmLmp[k] + 1]; u < ue; u++) { const int i2 = lmCamp[u]; if (cnt2[i2]++ == 0) touched.push_back(i2); }
                        std::sort(touched.begin(), touched.end());
                        for (int i2 : touched) { oB.push_back(i2); oC.push_back(cnt2[i2]); cnt2[i2] = 0; }
                        pairsOfCam[i1] = (int)touched.size();
                    }
                });
            }
            hs_lap(6);
            prA.clear(); prB.clear(); prStart.clear();
            int run = 0;
            for (int q = 0; q < nChunks; q++) {
                size_t at = 0;
                for (int i1 = q * kChunkCams; i1 < std::min(nFa, (q + 1) * kChunkCams); i1++) {
                    const int np = cmPairStart[i1 + 1];
                    for (int k = 0; k < np; k++, at++) { prA.push_back(i1); prB.push_back(chB[q][at]); prStart.push_back(run); run += chCnt[q][at]; }
                    cmPairStart[i1 + 1] = (int)prA.size();
                }
            }
            prStart.push_back(run);
            EAO_REQUIRE((size_t)run == lpEntries, "internal: covisibility count mismatch (%d entries counted, %zu expected)", run, lpEntries);
            lpPairsMax = prA.size();
            hs_lap(7);
            const BigGeom g = big_geom(std::max(nFa, 1));
            bigT = g.RP / 64;
            const int T = bigT, tN = g.N >> 6;
            live.assign((size_t)T * T, 0);
            // the tiles the pairs' 6 x 6 blocks touch
            for (size_t k = 0; k < prA.size(); k++) {
                const int lo = prA[k], hi = prB[k];
                const int r0 = (6 * hi) >> 6, r1 = (6 * hi + 5) >> 6, c0 = (6 * lo) >> 6, c1 = (6 * lo + 5) >> 6;
                live[(size_t)std::max(r0, c0) * T + std::min(r0, c0)] = 1; live[(size_t)std::max(r0, c1) * T + std::min(r0, c1)] = 1;
                live[(size_t)std::max(r1, c0) * T + std::min(r1, c0)] = 1; live[(size_t)std::max(r1, c1) * T + std::min(r1, c1)] = 1;
            }
            for (int t = 0; t < T; t++) live[(size_t)t * T + t] = 1;
            for (int t = 0; t <= tN; t++) live[(size_t)tN * T + t] = 1;      // the right-hand side row N (and the z it becomes) spans every column
            std::vector<int> rws;
            for (int k = 0; k < T; k++) {      // symbolic elimination
                rws.clear();
                for (int i = k + 1; i < T; i++) if (live[(size_t)i * T + k]) rws.push_back(i);
                for (size_t a1 = 0; a1 < rws.size(); a1++)
                    for (size_t b1 = 0; b1 <= a1; b1++) live[(size_t)rws[a1] * T + rws[b1]] = 1;
            }
            tileMap.assign((size_t)T * T, -1);
            for (int i = 0; i < T; i++) for (int j = 0; j <= i; j++) if (live[(size_t)i * T + j]) tileMap[(size_t)i * T + j] = bigTiles++;
            // per 32-column panel: one record pair per tile its launch updates -- the trailing tiles (a, b) whose two tile rows are live in the panel's tile column;
            // the tile of the next diagonal block first (its workgroup factors that block on the spot, also when the panel itself does not reach it)
            const int nbk = g.N / kBigNB;
            panelStart.assign((size_t)nbk + 1, 0); panelWork.clear();
            for (int kb = 0; kb < nbk; kb++) {
                const int kc = (kb * kBigNB) >> 6, tj0 = (kb * kBigNB + kBigNB) >> 6;
                panelStart[kb] = (int)(panelWork.size() / 2);
                rws.clear();
                rws.push_back(tj0);
                for (int i = tj0 + 1; i < T; i++) if (live[(size_t)i * T + kc]) rws.push_back(i);
                auto pslot = [&](int x) { return tileMap[(size_t)x * T + kc]; };
                for (size_t a1 = 0; a1 < rws.size(); a1++)
                    for (size_t b1 = 0; b1 <= a1; b1++) {
                        const int ta = rws[a1], tb = rws[b1], sc = tileMap[(size_t)ta * T + tb];
                        const bool first = a1 == 0;
                        if (!first && (sc < 0 || pslot(ta) < 0 || pslot(tb) < 0)) continue;      // nothing to subtract from a tile one of whose panel tiles is zero
                        panelWork.push_back(make_int4(ta, tb, sc, pslot(ta)));
                        panelWork.push_back(make_int4(pslot(tb), 0, 0, 0));
                    }
            }
            panelStart[nbk] = (int)(panelWork.size() / 2);
            EAO_REQUIRE(panelWork.size() < ((size_t)1 << 28), "tile structure too large (%zu work records)", panelWork.size() / 2);
        }
        hs_lap(0);
        size_t need = 0;
        need += (size_t)E * (3 * 4 + 4 + 4 + 4 + 1 + 4 + 4 + 4 + 1 + 24 + 18 * 8);
        need += (size_t)nP * (3 + 3 + 9 + 3 + 3 + 1 + 1) * 8 + (size_t)nP * 16 + (bigPath ? 64 : (size_t)nP * nC * 4);
        need += (size_t)nC * (2 * sizeof(SE3) + 36 * 8 + 6 * 8 + 6 * 8 + 16);
        need += (size_t)nP * 8 * sizeof(int4) + 256;
        need += (size_t)nP * 9 * 8 + 2048 + 256;      // Tl, ul, the zero block
        need += 128 * 256 + (size_t)nPl * 4 * 8 * 2 + (size_t)Epl * 4 * 8 + 2 * sizeof(BADev) + (size_t)nP + 1024;     // (+ k_ba_backsub's workgroup sums)
        if (bigPath) {
            need += (2 * ((size_t)bigTiles << 12) + 2 * (size_t)bg.N * kBigNB) * 8;
            need += (3 * lpEntries + 5 * lpPairsMax + 72) * 4 + (tileMap.size() + panelStart.size() + 8) * 4 + panelWork.size() * sizeof(int4) + 2048;
        } else {
            need += 2 * ((size_t)(nC * 6 + 6) * (nC * 6 + 34) + 8) * 8;
            need += (size_t)nC * (nC + 1) / 2 * ((size_t)nP + 64) * (4 + 16);   // landmark lists / item records of the camera pairs
            need += (size_t)nP * 9 * 8 + 1024;
            need += (size_t)nC * (nC + 1) / 2 * 4;
        }
        if ((st = c.bytes.reserve(need))) return st;
        if (bigPath && getenv("EAO_DEBUG_STAMPS"))
            fprintf(stderr, "[eao map-scale arena] %.1f MB for this problem (%d x %d tile grid, %d live tiles = %.1f MB in the two pools, %zu work records), context arena %.1f MB\n",
                    need / 1e6, bigT, bigT, bigTiles, 2.0 * bigTiles * 32768 / 1e6, panelWork.size() / 2, c.bytes.n / 1e6);
        Arena a{c.bytes.p, c.bytes.n};
        std::memset(&D, 0, sizeof(D));
        D.nCams = nC; D.nPts = nP; D.nEdges = E;
        D.cam.fx = p->fx; D.cam.fy = p->fy; D.cam.cx = p->cx; D.cam.cy = p->cy; D.cam.bf = p->bf; D.cam.bf_f = p->bf;
        D.cam.deltaMono = (float)std::sqrt(mode == 1 ? refc::GBA_HUBER2_MONO : refc::LBA_HUBER2_MONO);
        D.cam.deltaStereo = (float)std::sqrt(mode == 1 ? refc::GBA_HUBER2_STEREO : refc::LBA_HUBER2_STEREO);
        // ---- the uploaded part of the arena (problem, initial state, adjacency, zeroed control block, the window record
        //      itself) is mirrored in pinned host memory: filled in place, sent with two copies
        const size_t off0 = a.off;
        float* dobs = a.take<float>((size_t)E * 3); float* dinfo = a.take<float>(E);
        int* decam = a.take<int>(E); int* dept = a.take<int>(E);
        SE3* dcams = a.take<SE3>(nC);
        double* dpts = a.take<double>((size_t)nP * 3);
        unsigned char* dflag = a.take<unsigned char>(E);
        int* dcamIdx = a.take<int>(nC); int* dptIdx = a.take<int>(nP); int* dactCam = a.take<int>(nC); int* dactPt = a.take<int>(nP);
        int* dptStart = a.take<int>(nP + 1); int* dptEdges = a.take<int>(E); int* dcamStart = a.take<int>(nC + 1); int* dcamEdges = a.take<int>(E);
        int* dctl = a.take<int>(16);   // two control blocks: see BADecision
        int* dlpStart = a.take<int>(bigPath ? lpPairsMax + 1 : 1);
        int* dlpPair = a.take<int>(bigPath ? 2 * lpPairsMax : 1);
        int* dlpOrder = a.take<int>(bigPath ? 2 * lpPairsMax + 64 : 1);
        int* dlpPts = a.take<int>(bigPath ? lpEntries : 1);
        int* dlpE1 = a.take<int>(bigPath ? lpEntries : 1);
        int* dlpE2 = a.take<int>(bigPath ? lpEntries : 1);
        int* dbigTile = a.take<int>(bigPath ? tileMap.size() : 1);
        int4* dbigWork = a.take<int4>(bigPath ? std::max<size_t>(panelWork.size(), 1) : 1);
        double* dpl0 = a.take<double>((size_t)nPl * 4 + 1);
        double* dpmeas = a.take<double>((size_t)Epl * 4 + 1);
        dW = a.take<BADev>(2);
        const size_t off1 = (a.off + 255) & ~(size_t)255;
        // ---- device-only part
        int* dtable = bigPath ? nullptr : a.take<int>((size_t)nP * nC);      // (the map-scale path finds a landmark's edges in its pair lists)
        D.slot = a.take<int4>((size_t)std::max(nP, 1) * 8);
        D.camEdgeL = a.take<int>(E);
        const bool pairPath = !bigPath && nFreeIn > 0 && nFreeIn <= kTileMaxFree;
        const int nPairsMax = nFreeIn * (nFreeIn + 1) / 2;
        D.pairCnt = a.take<int>(bigPath ? 1 : std::max(nPairsMax, 1));
        D.pairPts = a.take<int>(pairPath ? (size_t)nPairsMax * std::max(nP, 1) : 1);
        static const bool envNoW = getenv("EAO_BA_WMODE") && !atoi(getenv("EAO_BA_WMODE"));      // (A/B switch: the VALU pair kernels)
        const bool wmode = pairPath && !hasPl && !envNoW;
        D.wmode = wmode ? 1 : 0;
        D.pairItems = a.take<int4>(wmode ? (size_t)nPairsMax * std::max(nP, 1) : 1);
        D.Tl = a.take<double>((size_t)std::max(nP, 1) * 6); D.ul = a.take<double>(((size_t)std::max(nP, 1) + 1) * 3);
        D.cls = a.take<unsigned char>(E);
        SE3* dcamsT = a.take<SE3>(nC);
        double* dptsT = a.take<double>((size_t)nP * 3);
        D.plBuf[0] = dpl0; D.plBuf[1] = a.take<double>((size_t)nPl * 4 + 1); D.pmeas = dpmeas;
        D.nPtsOnly = nPo; D.nEdgesPt = Ept;
        D.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2); D.infoAngle = refc::PLANE_ANGLE_INFO / (1.0 * 1.0); D.infoDist = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;   // src/Optimizer.cc:203-208
        D.err = a.take<double>((size_t)E * 3);
        D.Hpp = a.take<double>((size_t)nC * 36); D.bp = a.take<double>((size_t)nC * 6);
        D.Hll = a.take<double>((size_t)nP * 9); D.bl = a.take<double>((size_t)nP * 3);
        D.Hpl = a.take<double>(((size_t)E + 1) * 18);      // (+ the zero block of k_ba_schur_pairs_mfma)
        D.sys = a.take<double>(bigPath ? 8 : std::max((size_t)(nFreeIn * 6) * (nFreeIn * 6 + 1), (size_t)tile_geom(std::max(nFreeIn, 1)).nTiles * 256) + 8);
        D.big = a.take<double>(bigPath ? ((size_t)bigTiles << 12) : 8);
        D.bigL = a.take<double>(bigPath ? ((size_t)bigTiles << 12) : 8);
        D.bigTile = dbigTile; D.bigT = bigT; D.bigTiles = bigTiles; D.bigWork = dbigWork; D.bigDense = bigPath && bigTiles == bigT * (bigT + 1) / 2 ? 1 : 0;
        D.bigDiag = a.take<double>(bigPath ? (size_t)bg.N * kBigNB : 8);
        D.bigLinv = a.take<double>(bigPath ? (size_t)bg.N * kBigNB : 8);
        D.bigFail = a.take<int>(4);
        D.lpStart = dlpStart; D.lpPair = dlpPair; D.lpOrder = dlpOrder; D.lpPts = dlpPts; D.lpE1 = dlpE1; D.lpE2 = dlpE2;
        D.xp = a.take<double>((size_t)nC * 6); D.xl = a.take<double>((size_t)nP * 3);
        D.partChi = a.take<double>(nP); D.partScale = a.take<double>(nP);
        D.lm0 = a.take<double>(16);
        D.solveOk = a.take<int>(4);
        D.doneCnt = a.take<int>(4);
        D.wgPart = a.take<double>(2 * (size_t)eao::cdiv(std::max(nP, 1) * 8, 256) + 2);
        long long* ddbg = a.take<long long>(32);
        D.dbg = getenv("EAO_DEBUG_STAMPS") ? ddbg : nullptr;
        EAO_REQUIRE(a.off <= a.cap, "internal: arena overflow");
        D.obs = dobs; D.info = dinfo; D.ecam = decam; D.ept = dept; D.eflag = dflag;
        D.camIdx = dcamIdx; D.ptIdx = dptIdx; D.actCam = dactCam; D.actPt = dactPt;
        D.ptStart = dptStart; D.ptEdges = dptEdges; D.camStart = dcamStart; D.camEdges = dcamEdges; D.table = dtable;
        D.camsBuf[0] = dcams; D.camsBuf[1] = dcamsT; D.ptsBuf[0] = dpts; D.ptsBuf[1] = dptsT;
        D.ctl0 = dctl; D.ctl = dctl; D.lm = D.lm0;
        D.status = c.status;
        if (c.pinCap < off1) {
            if (c.pin) (void)hipHostFree(c.pin);
            c.pin = nullptr; c.pinCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pin, off1 + (off1 >> 2), hipHostMallocDefault));
            c.pinCap = off1 + (off1 >> 2);
        }
        const size_t outBytes = (size_t)nC * sizeof(SE3) + (size_t)nP * 24 + (size_t)nPl * 32 + (((size_t)E + 15) & ~(size_t)15) + 64;
        if (c.pinOutCap < outBytes) {
            if (c.pinOut) (void)hipHostFree(c.pinOut);
            c.pinOut = nullptr; c.pinOutCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pinOut, outBytes + (outBytes >> 2), hipHostMallocMapped));
            c.pinOutCap = outBytes + (outBytes >> 2);
        }
        outCams = (SE3*)c.pinOut;
        outPts = (double*)(c.pinOut + (((size_t)nC * sizeof(SE3) + 15) & ~(size_t)15));
        outPlanes = outPts + (size_t)nP * 3;
        outCls = (unsigned char*)(outPlanes + (size_t)nPl * 4);
        D.outCams = outCams; D.outPts = outPts; D.outPlanes = outPlanes; D.outCls = outCls;
        auto hostp = [&](const void* dev) { return c.pin + ((const unsigned char*)dev - a.base); };
        size_t offSplit = off0;
        {
            std::memcpy(hostp(dobs), p->edge_obs, (size_t)Ept * 12);
            std::memcpy(hostp(dinfo), p->edge_inv_sigma2, (size_t)Ept * 4);
            std::memcpy(hostp(decam), p->edge_cam, (size_t)Ept * 4);
            std::memcpy(hostp(dept), p->edge_point, (size_t)Ept * 4);
            if (hasPl) {
                std::memset(hostp(dobs) + (size_t)Ept * 12, 0, (size_t)Epl * 12);
                std::memset(hostp(dinfo) + (size_t)Ept * 4, 0, (size_t)Epl * 4);
                int* hc2 = (int*)hostp(decam); int* hp2 = (int*)hostp(dept);
                for (int e = Ept; e < E; e++) { hc2[e] = edge_cam(e); hp2[e] = edge_lm(e); }
                double* hpl = (double*)hostp(dpl0); double* hpm = (double*)hostp(dpmeas);
                for (int i = 0; i < nPl; i++) plane_from_f32(pl->plane_world + 4 * i, hpl + 4 * i);          // Converter::toPlane3D (:217)
                for (int e = 0; e < Epl; e++) plane_from_f32(pl->pedge_obs + 4 * e, hpm + 4 * e);           // (:239)
            }
            SE3* hc = (SE3*)hostp(dcams);
            for (int i = 0; i < nC; i++) hc[i] = se3_from_Tcw_f32(p->cam_Tcw + 16 * i);
            double* hp = (double*)hostp(dpts);
            for (size_t i = 0; i < (size_t)nPo * 3; i++) hp[i] = p->points[i];
            for (size_t i = (size_t)nPo * 3; i < (size_t)nP * 3; i++) hp[i] = 0;
            // edge flags: bit0 stereo, bit2 robust kernel present (bit1 = level 1 is only ever set on the device)
            unsigned char* hf = (unsigned char*)hostp(dflag);
            for (int e = 0; e < Ept; e++) hf[e] = (unsigned char)((!(p->edge_obs[3 * e + 2] < 0) ? 1 : 0) | (robust ? 4 : 0));
            for (int e = Ept; e < E; e++) hf[e] = 8 | 4;      // EdgePlane: always a Huber kernel (:246-248)
            std::memset(hostp(dctl), 0, 16 * sizeof(int));
            // The problem itself (observations, indices, initial state, flags) is on its way to the device while the host builds
            // the active structure below; the structure follows in a second copy.
            offSplit = (size_t)((unsigned char*)dcamIdx - a.base) & ~(size_t)255;
            hs_lap(1);
            if (!deferUpload) EAO_HIP(hipMemcpyAsync(a.base + off0, c.pin + off0, offSplit - off0, hipMemcpyHostToDevice, s));
            // ---- active structure: SparseOptimizer::initializeOptimization(level 0) + buildIndexMapping
            int* camIdx = (int*)hostp(dcamIdx); int* ptIdx = (int*)hostp(dptIdx);
            int* actCam = (int*)hostp(dactCam); int* actPt = (int*)hostp(dactPt);
            int* ptStart = (int*)hostp(dptStart); int* ptEdges = (int*)hostp(dptEdges);
            int* camStart = (int*)hostp(dcamStart); int* camEdges = (int*)hostp(dcamEdges);
            // (camCnt / ptCnt: counted with the validation pass above)
            int nF = 0, nL = 0;
            for (int i = 0; i < nC; i++) { camIdx[i] = -1; if (camCnt[i] && !p->cam_fixed[i]) { actCam[nF] = i; camIdx[i] = nF++; } }
            ptStart[0] = 0;
            for (int i = 0; i < nP; i++) { ptIdx[i] = -1; if (ptCnt[i]) { actPt[nL] = i; ptIdx[i] = nL; ptStart[nL + 1] = ptStart[nL] + ptCnt[i]; nL++; } }
            camStart[0] = 0;
            for (int i = 0; i < nF; i++) camStart[i + 1] = camStart[i] + camCnt[actCam[i]];
            for (int i = 0; i < nL; i++) ptCnt[actPt[i]] = ptStart[i];         // counters become fill cursors
            for (int i = 0; i < nF; i++) camCnt[actCam[i]] = camStart[i];
            bool dupChecked = false;
            if (edgesByLandmark) {      // (the landmarks' edge lists, concatenated in landmark order, ARE the edge list; the one-edge-per-pair test rides along)
                static thread_local std::vector<int> camLast;
                camLast.assign((size_t)nC, -1);
                for (int e = 0; e < E; e++) {
                    const int cam = edge_cam(e), lmk = edge_lm(e);
                    ptEdges[e] = e;
                    if (camIdx[cam] >= 0) camEdges[camCnt[cam]++] = e;
                    if (camLast[cam] == lmk) { eao::set_error("two edges join camera %d and point %d", cam, lmk); return EAO_ERR_INVALID; }
                    camLast[cam] = lmk;
                }
                dupChecked = true;
            } else {
                for (int e = 0; e < E; e++) {
                    const int cam = edge_cam(e);
                    ptEdges[ptCnt[edge_lm(e)]++] = e;
                    if (camIdx[cam] >= 0) camEdges[camCnt[cam]++] = e;
                }
            }
            // one edge per (camera, point) pair: the device's edge table has one slot per pair
            if (!dupChecked) {
                for (int i = 0; i < nC; i++) camCnt[i] = -1;                        // now: last point seen with this camera
                for (int l = 0; l < nL; l++)
                    for (int k = ptStart[l]; k < ptStart[l + 1]; k++) {
                        const int cam = edge_cam(ptEdges[k]);
                        if (camCnt[cam] == l) { eao::set_error("two edges join camera %d and point %d", cam, actPt[l]); return EAO_ERR_INVALID; }
                        camCnt[cam] = l;
                    }
            }
            D.nFree = nF; D.nL = nL;
            hs_lap(2);
            if (bigPath && nF > 0) {
                // covisibility CSR: for every camera pair (i1 <= i2) sharing a landmark, the landmark blocks in ascending order
                // (counting sort over the landmarks' observer lists; the diagonal pairs carry each camera's own landmarks)
                int* lpStart = (int*)hostp(dlpStart); int* lpPair = (int*)hostp(dlpPair); int* lpPts = (int*)hostp(dlpPts);
                int* lpE1 = (int*)hostp(dlpE1); int* lpE2 = (int*)hostp(dlpE2);
                EAO_REQUIRE(big_geom(nF).RP / 64 == bigT, "internal: tile structure built for another system size");
                std::memcpy(hostp(dbigTile), tileMap.data(), tileMap.size() * sizeof(int));
                if (!panelWork.empty()) std::memcpy(hostp(dbigWork), panelWork.data(), panelWork.size() * sizeof(int4));
                // (the pairs, their entry counts and every camera's landmark list were worked out above, before the arena was sized)
                EAO_REQUIRE((int)cmPairStart.size() == nF + 1, "internal: covisibility structure built for another set of free keyframes");
                const int nz = (int)prA.size();
                for (int k = 0; k < nz; k++) { lpPair[2 * k] = prA[k]; lpPair[2 * k + 1] = prB[k]; }
                std::memcpy(lpStart, prStart.data(), ((size_t)nz + 1) * sizeof(int));
                const size_t nEnt = (size_t)prStart[nz];
                const auto f0 = std::chrono::steady_clock::now();
                {
                    constexpr int kChunkCams = 16;
                    const int nChunks = (nF + kChunkCams - 1) / kChunkCams;
                    const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const lmEdgep = lmEdge.data(); const int* const cmOffp = cmOff.data();
                    const int* const cmLmp = cmLm.data(); const int* const cmUp = cmU.data(); const int* const prBp = prB.data(); const int* const prStartp = prStart.data();
                    const int* const cmPairStartp = cmPairStart.data();
                    crew_for(nEnt, nChunks, [=](int q) {
                        static thread_local std::vector<int> cur;
                        cur.resize((size_t)nF);
                        for (int i1 = q * kChunkCams; i1 < std::min(nF, (q + 1) * kChunkCams); i1++) {
                            for (int k = cmPairStartp[i1]; k < cmPairStartp[i1 + 1]; k++) cur[prBp[k]] = prStartp[k];      // fill cursors of camera i1's pairs
                            for (int k = cmOffp[i1]; k < cmOffp[i1 + 1]; k++) {
                                const int i = cmLmp[k], l = ptIdx[i], u0 = cmUp[k], e1 = lmEdgep[u0];
                                for (int u = u0, ue = lmOffp[i + 1]; u < ue; u++) {
                                    const int at = cur[lmCamp[u]]++;
                                    lpPts[at] = l; lpE1[at] = e1; lpE2[at] = lmEdgep[u];
                                }
                            }
                        }
                    });
                }
                if (hostStamps) fprintf(stderr, "[eao map-scale host set-up] pair CSR fill: %zu entries, %d pairs, %.3f ms\n", nEnt, nz,
                                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count());
                D.nPairsNZ = nz;
                nPairsLong = 0;
                {   // launch order: long pairs first; and inside each class the pairs are dealt to the eight XCDs by camera range -- workgroup b runs on XCD b % 8, a pair
                    // list is sorted by its first camera, and the pairs of neighbouring cameras share their landmarks: dealt round-robin, every landmark's blocks were
                    // pulled into all eight L2s (the assembly re-reads each block once per pair of its landmark: 570 MB per launch on the banded 1000-keyframe map);
                    // with one contiguous camera range per XCD (equal shares of the entries) they stay in one or two.  A slot of -1 is an idle workgroup.
                    int* lpOrder = (int*)hostp(dlpOrder);
                    const int kLong = getenv("EAO_BA_PAIR_LONG") ? atoi(getenv("EAO_BA_PAIR_LONG")) : kBigPairLong;      // (tests: the four-wave kernel on small maps)
                    static thread_local std::vector<int> cls, grp[8];
                    size_t at = 0;
                    const size_t cap = 2 * (size_t)lpPairsMax + 64;
                    bool fits = true;
                    auto deal = [&](bool longOnes) -> int {
                        cls.clear();
                        long long tot = 0;
                        for (int k = 0; k < nz; k++) if ((lpStart[k + 1] - lpStart[k] > kLong) == longOnes) { cls.push_back(k); tot += lpStart[k + 1] - lpStart[k]; }
                        if (cls.empty()) return 0;
                        for (auto& g8 : grp) g8.clear();
                        long long run = 0;
                        int x = 0, lastCam = -1;
                        for (int k : cls) {      // a new XCD only at a camera boundary, once the running share of the entries is reached
                            const int cam = lpPair[2 * k];
                            if (cam != lastCam && x < 7 && run * 8 >= tot * (x + 1)) x++;
                            lastCam = cam;
                            grp[x].push_back(k);
                            run += lpStart[k + 1] - lpStart[k];
                        }
                        size_t len = 0;
                        for (auto& g8 : grp) len = std::max(len, g8.size());
                        if (at + 8 * len > cap) { fits = false; return 0; }
                        for (size_t sl = 0; sl < len; sl++)
                            for (int q = 0; q < 8; q++) lpOrder[at++] = sl < grp[q].size() ? grp[q][sl] : -1;
                        return (int)(8 * len);
                    };
                    nPairsLong = deal(true);
                    nPairsSlots = nPairsLong + deal(false);
                    if (!fits) {      // (one camera holds most of the pairs: plain order)
                        at = 0;
                        for (int k = 0; k < nz; k++) if (lpStart[k + 1] - lpStart[k] > kLong) lpOrder[at++] = k;
                        nPairsLong = (int)at;
                        for (int k = 0; k < nz; k++) if (lpStart[k + 1] - lpStart[k] <= kLong) lpOrder[at++] = k;
                        nPairsSlots = (int)at;
                    }
                }
            }
        }
        hs_lap(3);
        // ---- launch geometry and solver choice of this window
        const int nF = D.nFree, nL = D.nL;
        BADims& d = L.d;
        d = BADims();
        d.nF = nF; d.nL = nL; d.nP = nP; d.nC = nC; d.E = E; d.nPl = nPl; d.hasPl = hasPl; d.bigPath = bigPath;
        d.usePairs = pairPath && nF > 0 && nL > 0;
        d.wmode = D.wmode != 0 && d.usePairs;
        if (!d.wmode) D.wmode = 0;
        // solver choice: register tiles + MFMA up to kTileMaxFree free keyframes, the map-scale path beyond
        d.solveTiles = !bigPath && nF > 0 && nF <= kTileMaxFree;
        d.tileLds = tile_solver_lds(std::max(nF, 1));
        d.tiles3 = tile_geom(std::max(nF, 1)).nTiles <= 3 * (kTileThreads / 64);
        d.gB = big_geom(std::max(nF, 1));
        d.nPairsNZ = D.nPairsNZ; d.nPairsLong = nPairsLong; d.nPairsSlots = nPairsSlots; d.big = D.big; d.bigTiles = bigTiles;
        d.bigPanelStart = bigPath ? panelStart.data() : nullptr;
        d.bigPanelWork = bigPath ? panelWork.data() : nullptr;
        d.bigCtl0 = D.ctl0;
        d.bigArgs = BigStepArgs{D.big, D.bigL, D.bigDiag, D.bigFail, D.bigWork, D.ctl0, D.dbg, d.gB.N, make_int4(0, 0, 0, 0), 0};
        // the window record itself travels with the structure
        write_records((BADev*)hostp(dW));
        if (!deferUpload) EAO_HIP(hipMemcpyAsync(a.base + offSplit, c.pin + offSplit, off1 - offSplit, hipMemcpyHostToDevice, s));
        else { upSrc = c.pin + off0; upDst = a.base + off0; upBytes = (off1 - off0 + 15) & ~(size_t)15; }
        L.W = dW; L.nz = 1; L.s = s; L.seq = c.status->seq;
        c.status->ph[0].touched = c.status->ph[1].touched = 0;
        // map-scale runs (tens of milliseconds) are NOT enqueued speculatively when the caller can abort them: optimize() then
        // submits one LM iteration at a time and reads *stop in between, like g2o's forceStopFlag
        pollStop = bigPath && stop != nullptr;
        // ... and never more than two LM iterations ahead of the device otherwise (`lazy`, see optimize()): a map-scale iteration is ~50 launches, and
        // everything enqueued behind a rejected trial drains as no-ops at the launch rate -- 1.5 of 14 ms on the 200-keyframe benchmark map when all ten
        // iterations were enqueued up front
        lazy = bigPath && !pollStop;
        chained = E > 0 && (nF + nL) > 0 && !pollStop && !lazy;
        hs_lap(4);
        if (hostStamps && bigPath)
            fprintf(stderr, "[eao map-scale host set-up] observer / camera lists %.3f (counts %.3f, observer scatter %.3f, sort %.3f, camera scatter %.3f), pair counts %.3f, pair list %.3f, tiles + symbolic elimination + work records %.3f ms\n",
                    hsT[5], hsT[8], hsT[9] - hsT[8], hsT[10] - hsT[9], hsT[5] - hsT[10], hsT[6] - hsT[5], hsT[7] - hsT[6], hsT[0] - hsT[7]);
        if (hostStamps && bigPath)
            fprintf(stderr, "[eao map-scale host set-up] tile structure + symbolic elimination %.3f, arena + problem pack %.3f, active structure %.3f, pair CSR + launch order %.3f, records + upload enqueue %.3f ms (cumulative %.3f)\n",
                    hsT[0], hsT[1] - hsT[0], hsT[2] - hsT[1], hsT[3] - hsT[2], hsT[4] - hsT[3], hsT[4]);
        return EAO_OK;
    }

    eao_status wait_status(int want) {
        EAO_HIP(hipStreamSynchronize(L.s));
        if (c->status->seq != want) { eao::set_error("LM status hand-off out of sequence"); return EAO_ERR_INTERNAL; }
        return EAO_OK;
    }
    eao_status set_ctl(int halt, int iters, int nBad) {
        const int v[5] = {halt, curHost, iters, kStRunning, nBad};
        EAO_HIP(hipMemcpyAsync(D.ctl0, v, sizeof(v), hipMemcpyHostToDevice, L.s));
        return EAO_OK;
    }
    // ---- SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg::solve per iteration.
    // Iterations are enqueued in bulk (one trial each, no host round trip); the device finishes clean iterations itself
    // and halts the window on anything else, which the host then replays trial by trial like g2o's do/while.
    // resume: the first bulk segment of this call was already enqueued (and has finished) -- start from its outcome
    eao_status optimize(int phase, int iterations, int* itersDone, double* chiOut, const BAPhase* resume) {
        *itersDone = 0;
        LMContext& c = *this->c;
        eao_status st;
        if (D.nFree + D.nL == 0) return EAO_OK;   // "_ivMap.size() == 0": nothing to optimise
        bool needErrors = true, ok = true;
        double currentChi = 0;
        int nBad = 0, done = 0;
        while (done < iterations && !(stop && *stop && !resume) && ok) {
            if (!resume) {
                // ---- bulk segment: every remaining iteration, one trial each.  The control block is clean at the start of
                //      an optimize() call (zeros from the upload / reset by the outlier pass); after a takeover it is rewritten.
                if (done > 0 && (st = set_ctl(0, done, nBad))) return st;
                if (lazy) {
                    // one iteration per enqueue, the next one as soon as the decision of the one before the last has landed in the pinned status block.  The
                    // poll is a pacing hint only: a stale read enqueues later (or one no-op iteration more), never something else -- what the host acts on is
                    // read after the stream synchronisation below, as in the bulk path.
                    volatile const int* pseq = &c.status->seq;
                    volatile const int* pstat = &c.status->status;
                    static thread_local std::vector<int> seqAt;
                    seqAt.assign((size_t)iterations + 1, 0);
                    for (int enq = done; enq < iterations; enq++) {
                        if (enq - done >= 2) {
                            while (*pseq - seqAt[enq - 2] < 0) { if (hipStreamQuery(L.s) != hipErrorNotReady) break; }
                            if (*pstat != kStRunning) break;
                        }
                        L.bulk(enq, enq + 1, needErrors && enq == done);
                        seqAt[enq] = L.seq;
                    }
                } else L.bulk(done, pollStop ? std::min(iterations, done + 1) : iterations, needErrors);
                needErrors = false;
                EAO_HIP(hipStreamSynchronize(L.s));
            }
            const BAPhase S = resume ? *resume : c.status->ph[phase];
            resume = nullptr;
            for (int k = done; k < S.iters && k < 32; k++) {
                tr->lambda.push_back(c.status->trLambda[32 * phase + k]); tr->chi2.push_back(c.status->trChi[32 * phase + k]);
                tr->trials.push_back(c.status->trTrials[32 * phase + k]);
            }
            tr->linearizations += S.iters - done;
            done = S.iters; nBad = S.nBad; curHost = S.cur; currentChi = S.chi;
            if (S.status == kStEmpty) { done = -1; break; }
            if (S.status == kStTerminate) { ok = false; break; }
            if (S.status == kStRunning && done < iterations && pollStop) continue;   // next iteration (after a look at *stop)
            if (S.status != kStTakeover) break;           // all requested iterations done
            // ---- host takeover of iteration `done`: its first trial was rejected (or rho == 0 / NaN)
            tr->linearizations++;
            const double iniChi = S.chi;
            double rho = S.rho;
            int qmax = 1;
            bool accepted = S.accepted != 0;
            while (rho < 0 && qmax < refc::LM_MAX_TRIALS && !(stop && *stop)) {
                if ((st = set_ctl(0, done, nBad))) return st;
                L.relinearize();
                L.trial(0, 0, false, true);
                if ((st = wait_status(L.seq))) return st;
                rho = c.status->rho; accepted = c.status->accepted != 0; curHost = c.status->cur;
                if (accepted) currentChi = c.status->chi;
                qmax++;
            }
            needErrors = !accepted;             // pop(): residuals belong to the rejected state
            tr->lambda.push_back(c.status->lambda); tr->chi2.push_back(currentChi); tr->trials.push_back(qmax);
            done++;
            if (qmax == refc::LM_MAX_TRIALS || rho == 0) { ok = false; break; }
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) ok = false;
        }
        *itersDone = done;
        *chiOut = currentChi;
        return EAO_OK;
    }

    // After the chained enqueue (own or as part of a batch) has finished: takeovers, results.  The abort flag is read before
    // and after: a clean window takes less time than one g2o iteration on the CPU.
    eao_status complete() {
        eao::Range rg("lm: takeovers + results");
        LMContext& c = *this->c;
        eao_status st;
        hipStream_t s = L.s;
        const BAPhase A = c.status->ph[0], B = c.status->ph[1];
        if ((st = optimize(0, p->its_first, &r->iters[0], &r->chi2[0], chained ? &A : nullptr))) return st;
        const bool firstClean = chained && A.status != kStTakeover;
        const bool doMore = mode == 0 && (firstClean || !(stop && *stop));
        bool redo = !chained || (mode == 1 && !firstClean);
        if (doMore && E) {
            // outlier pass (src/Optimizer.cc:978-1008): chi2 of the residual each edge last computed + depth test.  g2o's
            // initializeOptimization(0) would now drop the level-1 edges (and vertices left without edges) from the active
            // set; here they stay in the lists with zero weight, which leaves every sum -- and a vertex without edges --
            // unchanged, and saves the host round trip of rebuilding and re-uploading the structure.
            if (firstClean) {
                if (B.status == kStTakeover) redo = true;
                if ((st = optimize(1, p->its_second, &r->iters[1], &r->chi2[1], &B))) return st;
            } else {
                redo = true;
                if ((st = set_ctl(0, 0, 0))) return st;            // the frozen window left "takeover" in the control block
                L.classify();
                if ((st = optimize(1, p->its_second, &r->iters[1], &r->chi2[1], nullptr))) return st;
            }
        }
        if (redo) {
            L.finish();
            EAO_HIP(hipStreamSynchronize(s));
        }
        EAO_HIP(hipGetLastError());
        if (D.dbg) {
            long long stt[32];
            EAO_HIP(hipMemcpy(stt, D.dbg, sizeof(stt), hipMemcpyDeviceToHost));
            fprintf(stderr, "[eao pair stamps] workgroup 0 (a diagonal pair): loads + accumulation %lld, block sum of 42 values %lld shader-cycles; linearisation: landmark workgroup 0 %lld, camera workgroup 0 %lld\n",
                    stt[13] >> 20, stt[13] & 0xFFFFF, stt[14], stt[15]);
            if (L.d.bigPath) fprintf(stderr, "[eao bal_backsolve stamps] super-block 1, workgroup 0: loads issued %lld, the 256-column triangle (8 blocks) %lld, removal from the columns to the left %lld shader-cycles\n",
                                     stt[25] - stt[24], stt[26] - stt[25], stt[27] - stt[26]);
            if (L.d.bigPath) fprintf(stderr, "[eao bal_step stamps] look-ahead workgroup of panel 2: prologue + loads + barrier %lld, row solves %lld, update %lld, tile to LDS and rows back %lld, 32 x 32 LDL^T %lld, store %lld shader-cycles\n",
                                     stt[17] - stt[16], stt[18] - stt[17], stt[19] - stt[18], stt[20] - stt[19], stt[21] - stt[20], stt[22] - stt[21]);
            fprintf(stderr, "[eao solve stamps] assemble %lld factor %lld (panel %lld trailing %lld / %lld) backsub %lld tail %lld shader-cycles; wall(100MHz) %lld %lld %lld %lld\n",
                    stt[2] - stt[0], stt[4] - stt[2], stt[10], stt[11], stt[12], stt[6] - stt[4], stt[8] - stt[6], stt[3] - stt[1], stt[5] - stt[3], stt[7] - stt[5], stt[9] - stt[7]);
        }
        for (int i = 0; i < nC; i++) se3_to_Tcw_f32(outCams[i], r->cam_Tcw + 16 * i);
        for (size_t i = 0; i < (size_t)nPo * 3; i++) r->points[i] = (float)outPts[i];
        for (size_t i = 0; i < (size_t)nPl * 4; i++) planes_out[i] = (float)outPlanes[i];           // Converter::toCvMat(Plane3D)
        if (Ept && r->edge_outlier) {
            if (mode == 0) std::memcpy(r->edge_outlier, outCls, Ept);
            else std::memset(r->edge_outlier, 0, Ept);
        }
        return EAO_OK;
    }
};

}  // namespace

static eao_status ba_run(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r, int mode, int robust,
                         const eao_ba_planes* pl = nullptr, float* planes_out = nullptr) {
    LMContext& c = g_ctx;
    eao_status st = ctx_init(c, true, mode == 0 ? eao::StreamClass::Background : eao::StreamClass::Bulk);
    if (st) return st;
    BAJob j;
    j.p = p; j.stop = stop; j.r = r; j.mode = mode; j.robust = robust; j.pl = pl; j.planes_out = planes_out; j.c = &c; j.tr = &g_trace;
    static const bool hostStamps = getenv("EAO_DEBUG_STAMPS") != nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
    EAO_HIP(hipEventRecord(c.ev0, c.stream));
    if ((st = j.prepare(c.stream))) return st;
    if (j.trivial) return EAO_OK;
    const double tPrep = ms();
    if ((st = j.L.attributes())) return st;
    j.L.setup();
    if (j.chained) {
        j.L.chain(mode, p->its_first, p->its_second);
        EAO_HIP(hipStreamSynchronize(c.stream));
    }
    const double tSetup = ms();
    if ((st = j.complete())) return st;
    EAO_HIP(hipEventRecord(c.ev1, c.stream));
    EAO_HIP(hipStreamSynchronize(c.stream));
    EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, c.ev0, c.ev1));
    if (hostStamps && j.L.d.bigPath) fprintf(stderr, "[eao map-scale wall] prepare %.3f, set-up launches (+ chain) %.3f, iterations + results %.3f ms; whole call %.3f ms\n", tPrep, tSetup - tPrep, ms() - tSetup, ms());
    return EAO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LocalBundleAdjustment of MANY independent windows (the batched-sequence configuration: 25 windows of 20 keyframes): the
// window is the z dimension of every launch.  Each window owns a context (arena, pinned mirrors, status block); the
// host-side set-up of the windows runs on a few host threads; ONE chained enqueue serves them all; a window whose LM
// rejected a trial freezes by itself (its halt flag) and is finished by the host afterwards exactly like a single call.
namespace {
struct BABatchPool {
    std::vector<std::unique_ptr<LMContext>> ctx;
    std::vector<LMTraceHost> trace;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipStream_t> side;              // streams of the window groups beyond the first
    std::vector<hipEvent_t> sideDone;
    BADev* hW = nullptr; size_t hWCap = 0;      // pinned mirror of the window array
    eao::DevBuf<BADev> dW;
    ~BABatchPool() {
        if (hW) (void)hipHostFree(hW);
        for (hipEvent_t e : sideDone) (void)hipEventDestroy(e);
        for (hipStream_t q : side) (void)hipStreamDestroy(q);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
thread_local BABatchPool g_batch;
constexpr int kBatchGroups = 4, kBatchGroupMin = 4;     // default number of window groups / fewest windows worth a group

}  // namespace

extern "C" {

eao_status eao_local_ba(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r) { return ba_run(p, stop, r, 0, 1); }

eao_status eao_local_ba_batch(const eao_ba_problem* problems, int32_t n, const volatile uint8_t* stop, eao_ba_result* results) {
    EAO_REQUIRE(n >= 0 && (n == 0 || (problems && results)), "null argument");
    if (n == 0) return EAO_OK;
    eao_status st = eao::require_device();
    if (st) return st;
    BABatchPool& B = g_batch;
    if (!B.stream) {
        EAO_HIP(eao::create_stream(&B.stream, eao::StreamClass::Background));
        EAO_HIP(hipEventCreate(&B.ev0));
        EAO_HIP(hipEventCreate(&B.ev1));
    }
    while ((int)B.ctx.size() < n) B.ctx.emplace_back(new LMContext());
    if ((int)B.trace.size() < n) B.trace.resize(n);
    for (int w = 0; w < n; w++)
        if ((st = ctx_init(*B.ctx[w], false, eao::StreamClass::Background))) return st;
    if (B.hWCap < (size_t)n * 2) {
        if (B.hW) (void)hipHostFree(B.hW);
        B.hW = nullptr; B.hWCap = 0;
        EAO_HIP(hipHostMalloc((void**)&B.hW, (size_t)n * 2 * sizeof(BADev), hipHostMallocDefault));
        B.hWCap = (size_t)n * 2;
    }
    if ((st = B.dW.reserve((size_t)n * 2))) return st;
    g_trace.clear();
    std::vector<BAJob> jobs(n);
    for (int w = 0; w < n; w++) {
        BAJob& j = jobs[w];
        j.p = &problems[w]; j.stop = stop; j.r = &results[w]; j.mode = 0; j.robust = 1; j.c = B.ctx[w].get(); j.tr = &B.trace[w];
        j.c->status->seq = 0;        // every window of the batch sees the same hand-off sequence numbers
    }
    static const bool envTiming = getenv("EAO_BA_BATCH_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    EAO_HIP(hipEventRecord(B.ev0, B.stream));
    // ---- The windows are dealt to G groups (contiguous ranges).  A group is a chain of its own on its own stream: its
    //      host-side set-up (validation, pinned mirror, active structure, two uploads per window: ~0.1 ms of memcpy and
    //      counting each, on a few host threads), then ONE batched enqueue for its windows.  The first group's kernels start
    //      while the others are still being set up, and a group's one-workgroup-per-window solver (a tenth of the chip) and
    //      the tails of its other launches overlap the other groups' wide kernels.  More than four streams share hardware
    //      queues on this runtime and serialise (measured: 5+ groups are 40 % slower than one).
    static const int envThreads = getenv("EAO_BA_BATCH_THREADS") ? atoi(getenv("EAO_BA_BATCH_THREADS")) : 0;
    static const int envGroups = getenv("EAO_BA_BATCH_GROUPS") ? atoi(getenv("EAO_BA_BATCH_GROUPS")) : 0;
    const int hw = (int)std::thread::hardware_concurrency();
    // (set-up threads make no HIP call any more -- packing and counting only -- so they scale with the host's cores)
    const int nThreads = std::max(1, std::min(n, envThreads > 0 ? envThreads : std::min(16, std::max(1, hw / 2))));
    // Groups hold WHOLE ROWS of eight windows where the batch has them (BA_WIN pins a window to an XCD row by row, and deals an incomplete row over all eight):
    // 25 windows as 8 + 8 + 9 load every XCD with 3.125 windows, as 6 + 6 + 6 + 7 the fullest one with 3.5 (2.6 against 2.8 ms per call).
    const int fullRows = n / 8, rest = n - 8 * fullRows;
    const bool rowGroups = fullRows >= 1 && !getenv("EAO_BA_BATCH_EVEN");      // (A/B switch: the even split of rounds 2-3)
    const bool restGroup = rowGroups && rest >= kBatchGroupMin;                // an incomplete row large enough to be a group of its own (else it joins the last group)
    const int gWant = std::min(envGroups > 0 ? envGroups : kBatchGroups, nThreads);
    const int G = std::max(1, rowGroups ? std::min(gWant, fullRows + (restGroup ? 1 : 0)) : std::min(gWant, n / kBatchGroupMin));
    std::vector<int> gStart(G + 1, n);
    for (int g = 0; g < G; g++) {
        if (!rowGroups) gStart[g] = (int)((long long)n * g / G);
        else if (restGroup && G > 1) gStart[g] = g == G - 1 ? 8 * fullRows : 8 * (int)((long long)fullRows * g / (G - 1));
        else gStart[g] = 8 * (int)((long long)fullRows * g / G);
    }
    while ((int)B.side.size() < G - 1) {
        hipStream_t q; hipEvent_t e;
        EAO_HIP(eao::create_stream(&q, eao::StreamClass::Background));
        EAO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        B.side.push_back(q); B.sideDone.push_back(e);
    }
    std::vector<eao_status> stw(n, EAO_OK), stg(G, EAO_OK);
    std::vector<std::string> errw(n), errg(G);
    std::vector<BALaunch> LG(G);
    std::vector<int> groupOf(n, -1), nInGroup(G, 0);
    std::vector<char> batched(n, 0);
    std::vector<double> msPrep(G, 0.0), msEnq(G, 0.0), msDone(G, 0.0);
    int dev = 0;
    EAO_HIP(hipGetDevice(&dev));
    // set-up: nThreads workers take the windows in order (window w's uploads go to its group's stream), so the first
    // group is complete after one round; each group's leader then enqueues its chain while the workers carry on
    std::vector<std::atomic<int>> prepared(G);
    for (auto& a : prepared) a.store(0);
    auto groupOfWindow = [&](int w) { int g = 0; while (gStart[g + 1] <= w) g++; return g; };
    auto streamOf = [&](int g) { return g == 0 ? B.stream : B.side[g - 1]; };
    // results: when a group's stream has drained, the set-up workers (idle since the first tenth of the call) take its windows' results out in parallel -- the last
    // group's nine windows were ~0.1 ms of conversions on one thread at the very end of the call
    const bool crewDone = !(nThreads == 1 && G == 1);
    std::mutex doneMu;
    std::condition_variable doneCv;
    std::vector<int> gState(G, 0);      // 0: running, 1: drained, 2: failed (nothing to take out)
    auto finish_window = [&](int w, int g) {
        BAJob& j = jobs[w];
        if (j.trivial || stw[w]) return;
        if (batched[w]) j.L.seq = LG[g].seq;
        stw[w] = j.complete();
        if (stw[w]) errw[w] = eao_last_error();
    };
    auto worker = [&](int t) {
        (void)hipSetDevice(dev);
        for (int w = t; w < n; w += nThreads) {
            const int g = groupOfWindow(w);
            stw[w] = jobs[w].prepare(streamOf(g), true);
            if (stw[w]) errw[w] = eao_last_error();
            prepared[g].fetch_add(1, std::memory_order_release);
        }
        if (!crewDone) return;
        for (int w = t; w < n; w += nThreads) {
            const int g = groupOfWindow(w);
            int state;
            { std::unique_lock<std::mutex> lk(doneMu); doneCv.wait(lk, [&] { return gState[g] != 0; }); state = gState[g]; }
            if (state == 1) finish_window(w, g);
        }
    };
    auto groupWork = [&](int g) {
        (void)hipSetDevice(dev);
        struct Announce {      // whatever way this group ends, the workers waiting for it are told
            std::mutex& mu; std::condition_variable& cv; int& state; int value = 2;
            ~Announce() { { std::lock_guard<std::mutex> lk(mu); state = value; } cv.notify_all(); }
        } announce{doneMu, doneCv, gState[g]};
        const int w0 = gStart[g], w1 = gStart[g + 1];
        hipStream_t sg = streamOf(g);
        while (prepared[g].load(std::memory_order_acquire) < w1 - w0) std::this_thread::yield();    // (the workers above)
        msPrep[g] = since(tp0);
        for (int w = w0; w < w1; w++)
            if (stw[w]) return;
        // the windows that share the batched enqueue (tile-solver path, something to optimise); the others -- windows beyond
        // 30 free keyframes, empty ones -- follow one by one on the same stream
        {   // this group's uploads: one launch per eight windows (see BAJob::prepare)
            BAUploadArgs U;
            int k = 0;
            auto flush = [&]() {
                if (k) hipLaunchKernelGGL(k_ba_upload, dim3(48, k), dim3(256), 0, sg, U);
                k = 0;
            };
            for (int w = w0; w < w1; w++) {
                if (jobs[w].trivial || !jobs[w].upBytes) continue;
                U.dst[k] = jobs[w].upDst; U.src[k] = jobs[w].upSrc; U.n16[k] = jobs[w].upBytes / 16;
                if (++k == 8) flush();
            }
            flush();
        }
        BALaunch& L = LG[g];
        L.s = sg; L.W = B.dW.p + 2 * w0; L.seq = 0; L.rot = w0 & 7;
        int first = -1, cnt = 0;
        for (int w = w0; w < w1; w++) {
            BAJob& j = jobs[w];
            if (!j.batchable()) continue;
            if (first >= 0 && (j.p->its_first != jobs[first].p->its_first || j.p->its_second != jobs[first].p->its_second)) continue;
            if (first < 0) { first = w; L.d = j.L.d; } else L.d.merge(j.L.d);
            j.write_records(B.hW + 2 * (w0 + cnt));
            cnt++;
            groupOf[w] = g; batched[w] = 1;
        }
        nInGroup[g] = cnt;
        auto fail = [&](eao_status e) { stg[g] = e; errg[g] = eao_last_error(); };
        if (cnt) {
            L.nz = cnt;
            if (hipMemcpyAsync(B.dW.p + 2 * w0, B.hW + 2 * w0, (size_t)cnt * 2 * sizeof(BADev), hipMemcpyHostToDevice, sg) != hipSuccess) {
                eao::set_error("hipMemcpyAsync of the window records failed");
                return fail(EAO_ERR_NO_DEVICE);
            }
            eao_status e = L.attributes();
            if (e) return fail(e);
            L.setup();
            L.chain(0, jobs[first].p->its_first, jobs[first].p->its_second);
        }
        for (int w = w0; w < w1; w++) {
            BAJob& j = jobs[w];
            if (batched[w] || j.trivial) continue;
            eao_status e = j.L.attributes();
            if (e) return fail(e);
            j.L.setup();
            if (j.chained) j.L.chain(0, j.p->its_first, j.p->its_second);
        }
        msEnq[g] = since(tp0);
        // results (and, for a window whose device-side run handed over to the host, the rest of its run) as soon as THIS
        // group's stream has drained: only the last group's ~10 us per window are not hidden behind the other groups' kernels
        if (g > 0 && hipEventRecord(B.sideDone[g - 1], sg) != hipSuccess) { eao::set_error("hipEventRecord failed"); return fail(EAO_ERR_NO_DEVICE); }
        if (hipStreamSynchronize(sg) != hipSuccess) { eao::set_error("hipStreamSynchronize: %s", hipGetErrorString(hipGetLastError())); return fail(EAO_ERR_NO_DEVICE); }
        msDone[g] = since(tp0);
        announce.value = 1;
        if (!crewDone)
            for (int w = w0; w < w1; w++) finish_window(w, g);
    };
    if (nThreads == 1 && G == 1) { worker(0); groupWork(0); }          // (a batch of one: no thread is involved)
    else host_crew().run(nThreads + G - 1, [&](int i) { if (i < nThreads) worker(i); else groupWork(i - nThreads + 1); }, [&] { groupWork(0); });
    bool failed = false;
    for (int w = 0; w < n; w++) failed = failed || stw[w];
    for (int g = 0; g < G; g++) failed = failed || stg[g];
    if (failed)
        for (int g = 1; g < G; g++) (void)hipStreamSynchronize(B.side[g - 1]);
    else
        for (int g = 1; g < G; g++) EAO_HIP(hipStreamWaitEvent(B.stream, B.sideDone[g - 1], 0));   // (device time of the call: ev0 .. ev1)
    for (int w = 0; w < n; w++)
        if (stw[w]) { eao::set_error("window %d: %s", w, errw[w].c_str()); (void)hipStreamSynchronize(B.stream); return stw[w]; }
    for (int g = 0; g < G; g++)
        if (stg[g]) { eao::set_error("%s", errg[g].c_str()); (void)hipStreamSynchronize(B.stream); return stg[g]; }
    int nBatched = 0;
    for (int g = 0; g < G; g++) nBatched += nInGroup[g];
    const double msPrepare = *std::max_element(msPrep.begin(), msPrep.end());
    const double msEnqueue = *std::max_element(msEnq.begin(), msEnq.end()), msSync = *std::max_element(msDone.begin(), msDone.end());
    for (int w = 0; w < n; w++)
        if (!jobs[w].trivial) g_trace.linearizations += jobs[w].tr->linearizations;
    EAO_HIP(hipEventRecord(B.ev1, B.stream));
    EAO_HIP(hipStreamSynchronize(B.stream));
    EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, B.ev0, B.ev1));
    if (envTiming)
        fprintf(stderr, "[eao_local_ba_batch] %d windows (%d batched, %d host threads): set-up + uploads enqueued %.3f ms, launches enqueued %.3f, device done %.3f, results out %.3f\n",
                n, nBatched, nThreads, msPrepare, msEnqueue, msSync, since(tp0));
    return EAO_OK;
}

eao_status eao_bundle_adjustment(const eao_ba_problem* p, int32_t robust, const volatile uint8_t* stop, eao_ba_result* r) {
    return ba_run(p, stop, r, 1, robust != 0);
}

eao_status eao_bundle_adjustment_planes(const eao_ba_problem* p, const eao_ba_planes* planes, int32_t robust, const volatile uint8_t* stop,
                                        eao_ba_result* r, float* planes_out) {
    return ba_run(p, stop, r, 1, robust != 0, planes, planes_out);
}

eao_status eao_last_lm_trace(double* lambda, double* chi2, int32_t* trials, int32_t cap, int32_t* n) {
    EAO_REQUIRE(n, "null argument");
    const int m = std::min((int)g_trace.lambda.size(), cap);
    for (int i = 0; i < m; i++) {
        if (lambda) lambda[i] = g_trace.lambda[i];
        if (chi2) chi2[i] = g_trace.chi2[i];
        if (trials) trials[i] = g_trace.trials[i];
    }
    *n = m;
    return EAO_OK;
}

eao_status eao_last_lm_timing(float* device_ms, int32_t* linearizations) {
    if (device_ms) *device_ms = g_trace.deviceMs;
    if (linearizations) *linearizations = g_trace.linearizations;
    return EAO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Hooks for the device-resident tracking chain (csrc/track.hip): PoseOptimization over edges that a previous kernel left
// on the device, their count included.
#include "chain_internal.h"
namespace eao {
namespace lm {
size_t pose_se3_bytes() { return sizeof(SE3); }
void pose_se3_to_Tcw(const void* se3, float* T) { se3_to_Tcw_f32(*(const SE3*)se3, T); }
void pose_plane_records(int n, const float* plane_world, const float* plane_obs, const unsigned char* plane_seen, double* rec) {
    const double angleInfo = refc::PLANE_ANGLE_INFO / (1.0 * 1.0), disInfo = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;      // src/Optimizer.cc:464-465
    for (int i = 0; i < n; i++) {
        plane_from_f32(plane_world + 4 * i, rec + 10 * i);
        plane_from_f32(plane_obs + 4 * i, rec + 10 * i + 4);
        const double f = plane_seen[i] ? 1.0 : 2.0;                                                                                          // :503-516
        rec[10 * i + 8] = f * angleInfo; rec[10 * i + 9] = f * disInfo;
    }
}
eao_status enqueue_pose_device(const PoseChainArgs& a, hipStream_t s) {
    EAO_REQUIRE(a.cap > 0 && a.cap <= 4 * kPoseThreads, "the chained PoseOptimization takes up to %d edges", 4 * kPoseThreads);
    static_assert(kPoseChainMaxPlanes == kPoseMaxPlanes, "one limit");
    EAO_REQUIRE(a.nPlanes >= 0 && a.nPlanes <= kPoseMaxPlanes && (a.nPlanes == 0 || (a.planes && a.planeOutlier)), "at most %d plane edges", kPoseMaxPlanes);
    PoseDev P;
    std::memset(&P, 0, sizeof(P));
    P.n = a.cap; P.nDev = a.nEdges;
    P.Xw = a.Xw; P.obs = a.obs; P.info = a.info; P.err = a.err; P.flags = a.flags; P.outlier = a.outlier;
    P.T0 = se3_from_Tcw_f32(a.Tcw0);
    P.cam.fx = a.fx; P.cam.fy = a.fy; P.cam.cx = a.cx; P.cam.cy = a.cy; P.cam.bf = a.bf; P.cam.bf_f = a.bf;
    P.cam.deltaMono = (float)std::sqrt(refc::POSE_HUBER2_MONO); P.cam.deltaStereo = (float)std::sqrt(refc::POSE_HUBER2_STEREO);
    P.Tout = (SE3*)a.outSE3; P.result = a.outResult; P.trace = a.outTrace;
    P.scatterIdx = a.scatterIdx; P.scatterOut = a.scatterOut;
    P.done = nullptr; P.doneSeq = 0;
    P.pubSrc = (const uint4*)a.pubSrc; P.pubDst = (uint4*)a.pubDst; P.pubN16 = a.pubN16;
    P.nPlanes = a.nPlanes; P.planes = a.planes; P.planeOutlier = a.planeOutlier; P.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2);
    P.dbg = nullptr;
    const bool pl = a.nPlanes > 0;
    // The edge count lives on the device: one launch per geometry class the capacity admits (four waves up to 1024 edges, eight
    // beyond); each returns at its first instruction unless the frame's count falls in its range (a ~3 us launch, against running every
    // frame as eight waves).  Waves without an edge leave a pass at once.
    const bool second = std::min(a.cap, a.maxEdges > 0 ? a.maxEdges : a.cap) > 4 * kPoseWaves4;
    P.devLo = -1; P.devHi = std::min(a.cap, 4 * kPoseWaves4);
    if (!second) { P.done = a.done; P.doneSeq = a.doneSeq; }
    if (pl) hipLaunchKernelGGL((k_pose_optimization<4, true, kPoseWaves4>), dim3(1), dim3(kPoseWaves4), 0, s, P);
    else hipLaunchKernelGGL((k_pose_optimization<4, false, kPoseWaves4>), dim3(1), dim3(kPoseWaves4), 0, s, P);
    if (second) {
        P.done = a.done; P.doneSeq = a.doneSeq;
        P.devLo = 4 * kPoseWaves4; P.devHi = a.cap;
        if (pl) hipLaunchKernelGGL((k_pose_optimization<4, true, kPoseThreads>), dim3(1), dim3(kPoseThreads), 0, s, P);
        else hipLaunchKernelGGL((k_pose_optimization<4, false, kPoseThreads>), dim3(1), dim3(kPoseThreads), 0, s, P);
    }
    return EAO_OK;
}
}  // namespace lm
}  // namespace eao

