// pose.hip -- Optimizer::PoseOptimization on MI355X (gfx950): reference src/Optimizer.cc:325-673 over g2o's unary edges (types/types_six_dof_expmap.cpp:266-364) and
// the plane edges of src/g2oAddition/EdgePlane.h.  ONE persistent workgroup runs all 4 rounds x <= 10 LM iterations x <= 10 trials on the device (no host round trip):
// per-edge 2x6 / 3x6 Jacobians in registers, fixed-order reductions of the 28 accumulators, the 6x6 LDL^T replicated per wave.  Shared pieces: lm_internal.h.
#include "lm_internal.h"

namespace {

// ============================================================================================ PoseOptimization
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
    EAO_HIP(eao::wait_latency(c.stream));
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
        EAO_HIP(eao::wait_latency(c.stream));
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

