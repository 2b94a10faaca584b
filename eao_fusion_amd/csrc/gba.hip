// gba.hip -- the map-scale path of the Levenberg-Marquardt engine (Optimizer::BundleAdjustment / GlobalBundleAdjustemnt, reference src/Optimizer.cc:47-323, and
// oversized local windows).  Shared pieces: lm_internal.h.  (Round 6: split out of csrc/lm.hip.)
#include "lm_internal.h"

namespace {

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
// round 6: the ENTRIES of the pair lists, written on the device.  For pair (i1, i2) the landmarks both cameras observe, in ascending order, each with its edge in
// either camera: the host used to file them pair by pair (0.4 - 1.0 ms of the set-up on its crew, then 4 - 17 MB over PCIe in front of the first launch); it now sends the
// observer lists it builds anyway (per landmark the free observers sorted by camera, per camera its landmarks in ascending order: four ints per edge) and one wave per pair
// walks camera i1's landmarks 64 at a time, looks for i2 behind i1's own entry in each landmark's list and places the hits by their rank in the ballot -- the order the
// host's walk produced, entry for entry (the assembly sums a pair's landmarks in list order).
__global__ __launch_bounds__(256) void k_bal_pair_fill(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    const int lane = threadIdx.x & 63;
    const int pair = (int)bx * 4 + (threadIdx.x >> 6);
    if (pair >= P.nPairsNZ) return;
    const int i1 = P.lpPair[2 * pair], i2 = P.lpPair[2 * pair + 1];
    int at = P.lpStart[pair];
    const int k1 = P.cmOff[i1 + 1];
    for (int kb = P.cmOff[i1]; kb < k1; kb += 64) {
        const int k = kb + lane;
        int found = -1, lmk = 0, u0 = 0;
        if (k < k1) {
            lmk = P.cmLm[k]; u0 = P.cmU[k];
            for (int u = u0, ue = P.lmOff[lmk + 1]; u < ue; u++) {
                const int cam = P.lmCam[u];
                if (cam >= i2) { if (cam == i2) found = u; break; }
            }
        }
        const unsigned long long hits = __ballot(found >= 0);
        if (found >= 0) {
            const int pos = at + __popcll(hits & ((1ull << lane) - 1ull));
            P.lpPts[pos] = P.ptIdx[lmk]; P.lpE1[pos] = P.lmEdge[u0]; P.lpE2[pos] = P.lmEdge[found];
        }
        at += __popcll(hits);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_bal_schur_pairs(const BADev* __restrict__ W, int wpar, int first, int pairOff) {
    BA_WIN(P);
    __shared__ double red[NT / 64][42];
    if (P.ctl[kCtlHalt]) return;
    const int t = threadIdx.x, N = P.bigN;
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
        for (int r = t; r < N; r += NT) if (P.bigRowCam[r] < 0) *big_elem(P, P.big, r, r) = 1.0;     // identity padding (behind every segment and the separator block)
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
    const int R1 = P.bigRow[i1], R2 = P.bigRow[i2];                          // the cameras' rows in the elimination order (round 6)
    if (t < 36) {
        const int r = t / 6, c = t - r * 6;
        if (!diag) {      // the element of the LOWER triangle that holds S(i1,i2)[r][c] = S(i2,i1)[c][r]
            if (R2 > R1) *big_elem(P, P.big, R2 + c, R1 + r) = s;
            else *big_elem(P, P.big, R1 + r, R2 + c) = s;
        } else if (c >= r) *big_elem(P, P.big, R1 + c, R1 + r) = s + P.Hpp[(size_t)i1 * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
    } else if (diag) {
        const int r = t - 36;
        *big_elem(P, P.big, N, R1 + r) = P.bp[i1 * 6 + r] - s;               // right-hand side row
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


// the diagonal blocks of the panels nobody reaches before their launch -- the head of every segment -- straight from the assembled tiles (the others are
// factored by the look-ahead workgroup of the last panel that updates them)
__global__ __launch_bounds__(64) void k_bal_diag(const BADev* __restrict__ W, int wpar, int listOff) {
    BA_WIN(P);
    if (P.ctl[kCtlHalt]) return;
    const int kb = P.bigDiagList[listOff + bx], tk = kb >> 1, oq = (kb & 1) * kBigNB;
    const double* T0 = big_tile(P.big, big_slot(P, tk, tk)) + oq * 64 + oq;
    double X0[16], X1[kBigNB];
    bal_load_rows_dpp(T0, 64, X0, X1);
    const bool bad = bal_factor_diag_dpp(X0, X1);
    bal_store_diag_dpp(P.bigDiag, P.bigFail, kb, X0, X1, bad);
}

// One launch per 32-column panel: every 64 x 64 tile of the trailing lower triangle first solves the panel rows it needs ITSELF
// -- 64 rows of its tile row (w = a L_kk^-T, kept as W) and 64 of its tile column (l = w D^-1) on 128 threads, 496 FMAs each on
// registers against the factored diagonal block in LDS; a row is solved by every tile that uses it, which costs no wall time --
// then C -= W L^T (4 x 4 outputs per thread from transposed LDS tiles).  The first tile column archives l in `bigL` (the
// factor proper: the working matrix keeps its stale panel columns, nobody reads them again, so there is no race between a
// tile that still reads a and one that would overwrite it with l).  The tile that holds the NEXT diagonal block factors it on
// the spot (one wavefront, see above), so the next launch starts from a finished L_kk.  The first version ran the row solves
// as a launch of their own: two dependent launches per panel instead of one (12 + 13 us at 40 free keyframes).
__global__ __launch_bounds__(256) void k_bal_step(BigStepArgs A, int workOff) {
    const unsigned bx = blockIdx.x;
    __shared__ __attribute__((aligned(16))) double Wt[kBigNB][68];
    __shared__ __attribute__((aligned(16))) double Lt[kBigNB][68];
    __shared__ double dv[kBigNB];                      // 1 / d of the panel's diagonal block
    __shared__ double da[kBigNB][kBigNB + 1];
    constexpr size_t ld = 64;                          // (inside a tile)
    const int t = threadIdx.x;
    // the workgroup's tile: record bx of the LAUNCH's work list (round 6: a launch holds the panels the schedule put side by side -- GbaPlan; the record names its panel).
    // The halt flag travels with the record: one round trip to memory for both, not one behind the other (every launch of a trial starts with this chain)
    const int halted = A.ctl[kCtlHalt];
    int4 wa, wb;
    if ((int)bx < A.nByValue) { wa = A.rec[2 * bx]; wb = A.rec[2 * bx + 1]; }      // (scalar loads from the kernel-argument segment: no round trip to the work list)
    else { wa = A.bigWork[2 * (size_t)(workOff + bx)]; wb = A.bigWork[2 * (size_t)(workOff + bx) + 1]; }
    const int wbx = wb.x, kb = wb.y, flags = wb.z, nextKb = wb.w;
    const int k0 = kb * kBigNB;
    if (halted) return;
    const bool stp = A.dbg && bx == 0 && t == 0 && (flags & 1) && kb == 2;      // phase stamps of the look-ahead workgroup of panel 2 (EAO_DEBUG_STAMPS)
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
    const bool crit = flags & 1;                   // a look-ahead workgroup (see below): it walks its tile quadrant by quadrant
    double c[4][4];
    double* const Ct = big_tile(A.big, wa.z);      // (the host's symbolic elimination made sure the tile exists)
    double* C = Ct + (size_t)(ty * 4) * ld + tx * 4;
    const int oq = nextKb * kBigNB - c0;           // (look-ahead) 0 or 32: offset inside this tile of the diagonal block of panel nextKb, which this workgroup factors
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
            if ((flags & 2) && act) {   // ONE workgroup per (panel, tile row) archives the row's panel entries: the diagonal target's, or the z row's first listed one
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
    if (flags & 4) return;      // (the right-hand side row's own tile: its record archives the panel's z entries, nothing is subtracted from scratch)
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
            bal_store_diag_dpp(A.bigDiag, A.bigFail, nextKb, X0, X1, bad);
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
// to the left: z_j -= sum_i L(i, j) x_i.  L is read once, by as many workgroups as there are column chunks.
// Round 6: a launch takes a LIST of super-blocks (blockIdx.y: the separator block's one by one, then one of every segment per launch -- GbaPlan::sb), each with
// the range of column chunks to its left that hear from it (a segment's: its own columns only); exp(dx) * T moved into k_bal_apply behind the last launch.
__global__ __launch_bounds__(256) void k_bal_backsolve(const BADev* __restrict__ W, int wpar, int sbOff) {
    BA_WIN(P);
    __shared__ double xl[kBigSB];      // z of the super-block on entry, x on exit
    extern __shared__ double sdiag[];  // (kBigSB / 32) x 32 x 32: the inverses of the super-block's unit-lower diagonal blocks
    __shared__ double part2[4][64];
    if (P.ctl[kCtlHalt]) return;
    const int N = P.bigN, t = threadIdx.x;
    double* S = P.bigL;            // the archived factor (rows below each panel's diagonal block, z = row N), as tiles
    const int4 sbd = P.bigSB[sbOff + blockIdx.y];
    const int J0 = sbd.x, w = sbd.y, chunk = sbd.z + (int)bx;      // this workgroup's column chunk of 64 (workgroup 0 also delivers x)
    const bool hasLeft = chunk < sbd.w;
    auto zat = [&](int col) { return big_elem(P, S, N, col); };      // z = row N: its tile row is dense by construction
    if (bx > 0) {      // a column chunk in which none of the super-block's tile rows holds a tile has nothing to remove (workgroup-uniform)
        if (!hasLeft) return;
        bool any = false;
        for (int r = J0 >> 6; r <= (J0 + w - 1) >> 6; r++) any = any || big_slot(P, r, chunk) >= 0;
        if (!any) return;
    }
    const bool stb = P.dbg && bx == 0 && t == 0 && blockIdx.y == 0 && sbOff == 1;      // phase stamps (EAO_DEBUG_STAMPS): the second launch's first super-block, workgroup 0
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
        for (int i = t; i < w; i += 256) { const int nat = P.bigRowCam[J0 + i]; if (nat >= 0) P.xp[nat] = xl[i]; }      // (x in the cameras' natural order: what the landmarks' back substitution reads)
    if (hasLeft) {
        // this workgroup's 64 columns to the left of the super-block
        // (rows rg, rg + 4, ... of the super-block, ALL 64 loads of a thread in flight at once behind four tile look-ups: in batches of eight, each behind its own
        //  look-ups, this loop was 21 k cycles of a 60 k-cycle launch)
        const int j = chunk * 64 + (t & 63), rg = t >> 6;
        // (a tile that does not exist reads the z tile instead and is masked afterwards: unconditional loads, no branch per load)
        const double* const safe = big_tile(S, big_slot(P, N >> 6, chunk)) + (t & 63);
        const double* tl[kBigSB / 64];
        bool ok[kBigSB / 64];
#pragma unroll
        for (int r = 0; r < kBigSB / 64; r++) {
            const int sl = (r << 6) < w ? big_slot(P, (J0 >> 6) + r, chunk) : -1;
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
    }
}

// exp(dx) * T of every camera from the solved step (behind the last launch of the back substitution), and the solver's verdict
__global__ __launch_bounds__(256) void k_bal_apply(const BADev* __restrict__ W, int wpar) {
    BA_WIN(P);
    if (P.ctl[kCtlHalt]) return;
    const int t = threadIdx.x;
    const SE3* cams = cur_cams(P);
    SE3* camsT = trial_cams(P);
    for (int cI = t + 256 * (int)bx; cI < P.nCams; cI += 256 * (int)gridDim.x) {
        const int ci = P.camIdx[cI];
        if (ci >= 0) {
            double u[6];
            for (int q = 0; q < 6; q++) u[q] = P.xp[ci * 6 + q];
            camsT[cI] = se3_mul(se3_exp(u), cams[cI]);
        } else {
            camsT[cI] = cams[cI];
        }
    }
    if (t == 0 && bx == 0) *P.solveOk = *P.bigFail ? 0 : 1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the ELIMINATION ORDER and LAUNCH SCHEDULE of the map-scale path.  The reference factors the reduced camera system with Eigen's
// SimplicialLDLT behind an approximate-minimum-degree ordering (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:95-112): its elimination tree is what lets independent
// parts of the map factor independently.  Rounds 3-5 kept the natural keyframe order and ran ONE launch per 32-column panel, each waiting for the one before:
// 188 dependent launches per LM trial on a 1000-keyframe trajectory (72 % of the call).  Here:
//   order     one level of nested dissection on the covisibility graph.  The keyframes are laid out on a line (their natural order, or reverse Cuthill-McKee when that
//             has the smaller bandwidth); P - 1 cuts of the line give P segments; a keyframe with a neighbour across a cut joins the SEPARATOR set (the smaller of the
//             two sides' choices), so no covisibility edge links two segments.  Segments first, separators last; every segment and the separator block start on a
//             64-row tile boundary (identity padding in between).  P minimises the estimated chain: longest segment's panels + the separator block's panels.
//   symbolic  the tile-level fill-in of that order (as before: the block form of a sparse LDL^T's symbolic phase).
//   schedule  panel k depends on panel j < k when j's panel rows reach k's tile row; two panels that share a tile row update a common tile.  List scheduling:
//             a panel runs in the first launch behind all its dependencies in which no panel shares a tile row with it -- every launch takes one panel of every
//             independent segment, so the chain is ~ N / (32 P) + (separator panels) launches long instead of N / 32.  All dependencies of a panel share its tile row,
//             hence sit in DIFFERENT launches: the last one is unique, and its workgroup for the panel's diagonal tile factors the panel's 32 x 32 diagonal block on
//             the spot (the look-ahead of round 4, now by the schedule's rule); panels without a dependency get theirs from one k_bal_diag launch.
//   solve     back substitution bottom-up: the separator block in super-blocks of 256 columns (one launch each, every column to the left), then the segments'
//             super-blocks level by level -- one launch per level for ALL segments (a segment's columns only hear from the separators and from itself).
// A different order changes the rounding of the factorisation, nothing else (DESIGN.md section 2 already lists the Schur solver as "same factorisation up to rounding order").
// Everything here is a pure function of the covisibility pattern: the plan is cached in the thread's context and reused while the pattern's hash stays the same.
namespace {

struct Csr { std::vector<int> off, adj; };

void plan_adjacency(int n, const std::vector<int>& prA, const std::vector<int>& prB, Csr& g) {
    g.off.assign((size_t)n + 1, 0);
    for (size_t k = 0; k < prA.size(); k++) if (prA[k] != prB[k]) { g.off[prA[k] + 1]++; g.off[prB[k] + 1]++; }
    for (int i = 0; i < n; i++) g.off[i + 1] += g.off[i];
    g.adj.resize(g.off[n]);
    std::vector<int> cur(g.off.begin(), g.off.end() - 1);
    for (size_t k = 0; k < prA.size(); k++) if (prA[k] != prB[k]) { g.adj[cur[prA[k]]++] = prB[k]; g.adj[cur[prB[k]]++] = prA[k]; }
}

// reverse Cuthill-McKee over every component (start: a pseudo-peripheral vertex found by two sweeps from the component's vertex of smallest degree)
void plan_rcm(int n, const Csr& g, std::vector<int>& order) {
    order.clear();
    std::vector<char> seen((size_t)n, 0);
    std::vector<int> q, deg((size_t)n);
    for (int i = 0; i < n; i++) deg[i] = g.off[i + 1] - g.off[i];
    auto bfs = [&](int s, std::vector<int>& out, std::vector<char>& mark) {
        out.clear(); out.push_back(s); mark[s] = 1;
        std::vector<int> nb;
        for (size_t h = 0; h < out.size(); h++) {
            const int v = out[h];
            nb.clear();
            for (int e = g.off[v]; e < g.off[v + 1]; e++) if (!mark[g.adj[e]]) { mark[g.adj[e]] = 1; nb.push_back(g.adj[e]); }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
            out.insert(out.end(), nb.begin(), nb.end());
        }
    };
    std::vector<char> tmp((size_t)n, 0);
    for (int s0 = 0; s0 < n; s0++) {
        if (seen[s0]) continue;
        // the component of s0, its vertex of smallest degree, two sweeps to the far end
        std::vector<int> comp;
        bfs(s0, comp, tmp);
        int s = comp[0];
        for (int v : comp) if (deg[v] < deg[s] || (deg[v] == deg[s] && v < s)) s = v;
        for (int sweep = 0; sweep < 2; sweep++) {
            for (int v : comp) tmp[v] = 0;
            bfs(s, q, tmp);
            s = q.back();
        }
        bfs(s, q, seen);
        order.insert(order.end(), q.rbegin(), q.rend());
    }
}

int plan_bandwidth(int n, const Csr& g, const std::vector<int>& pos) {
    int bw = 0;
    for (int v = 0; v < n; v++) for (int e = g.off[v]; e < g.off[v + 1]; e++) bw = std::max(bw, std::abs(pos[v] - pos[g.adj[e]]));
    return bw;
}

inline int up64(int x) { return (x + 63) & ~63; }

}  // namespace

namespace eao {
namespace lm {

uint64_t gba_pattern_hash(int nFa, const std::vector<int>& prA, const std::vector<int>& prB) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)nFa;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
    mix(prA.size());
    for (size_t k = 0; k < prA.size(); k++) mix(((uint64_t)(unsigned)prA[k] << 32) | (unsigned)prB[k]);
    return h;
}

namespace {
struct PlanGraph {      // what every candidate order of one pattern shares
    int nFa = 0, rcm = 0, bandwidth = 0;
    Csr g;
    std::vector<int> line, pos, lo, hi;
};
void plan_for(const PlanGraph& G, const std::vector<int>& prA, const std::vector<int>& prB, int wantP, GbaPlan& pl);
}  // namespace

// forceP: 0 = choose, 1 = natural order, p > 1 = that many segments (the callers read EAO_BA_ND for it)
void gba_build_plan(int nFa, const std::vector<int>& prA, const std::vector<int>& prB, int forceP, GbaPlan& pl) {
    // ---- the lines the keyframes can be laid out on: their natural order, and reverse Cuthill-McKee when that narrows the band.  A loop that closes on itself (the
    //      last keyframes see the first) is the case where the wider line wins: on the natural line only the first few keyframes reach across every cut (they join the
    //      separators once), while RCM folds the ring and doubles the band at every cut -- so both lines are tried and the plans decide.
    PlanGraph G[2];
    int nLines = 1;
    auto reach = [&](PlanGraph& g) {      // per vertex the reach of its neighbourhood along the line
        g.lo.resize((size_t)nFa); g.hi.resize((size_t)nFa);
        for (int v = 0; v < nFa; v++) {
            int a = g.pos[v], b = g.pos[v];
            for (int e = g.g.off[v]; e < g.g.off[v + 1]; e++) { a = std::min(a, g.pos[g.g.adj[e]]); b = std::max(b, g.pos[g.g.adj[e]]); }
            g.lo[v] = a; g.hi[v] = b;
        }
    };
    G[0].nFa = nFa;
    plan_adjacency(nFa, prA, prB, G[0].g);
    G[0].line.resize((size_t)nFa); G[0].pos.resize((size_t)nFa);
    for (int i = 0; i < nFa; i++) G[0].line[i] = G[0].pos[i] = i;
    G[0].bandwidth = plan_bandwidth(nFa, G[0].g, G[0].pos);
    reach(G[0]);
    if (nFa >= 64 && G[0].bandwidth > 24) {
        G[1].nFa = nFa; G[1].g = G[0].g; G[1].rcm = 1;
        plan_rcm(nFa, G[1].g, G[1].line);
        G[1].pos.resize((size_t)nFa);
        for (int i = 0; i < nFa; i++) G[1].pos[G[1].line[i]] = i;
        G[1].bandwidth = plan_bandwidth(nFa, G[1].g, G[1].pos);
        if (G[1].bandwidth * 5 < G[0].bandwidth * 4) { reach(G[1]); nLines = 2; }
    }
    plan_for(G[0], prA, prB, 1, pl);
    const int natural = (int)pl.launches.size();
    auto cost = [](const GbaPlan& q) { return (long long)(q.launches.size() + q.sbLaunches.size()) * 100000 + (long long)(q.work.size() / 2); };
    GbaPlan cand;
    if (forceP > 1) {
        const int P = std::min(forceP, std::max(1, nFa / 2));
        plan_for(G[0], prA, prB, P, pl);
        if (nLines == 2) { plan_for(G[1], prA, prB, P, cand); if (cost(cand) < cost(pl)) std::swap(pl, cand); }
    } else if (forceP == 0 && nFa >= 96) {
        // the schedule itself is the estimate: a plan costs a fraction of a millisecond (0.23 ms at 1000 keyframes), so a handful of segment counts are built and
        // the one with the fewest dependent launches -- factorisation + back substitution -- is kept (ties: fewer work records)
        for (int li = 0; li < nLines; li++)
            for (int P : {4, 6, 8, 12, 16, 24, 32, 48}) {
                if (nFa / P < 8) break;
                plan_for(G[li], prA, prB, P, cand);
                if (cost(cand) < cost(pl)) std::swap(pl, cand);
            }
        if (pl.P > 1 && (pl.launches.size() + pl.sbLaunches.size()) * 10 > (size_t)(natural + (pl.N + kBigSB - 1) / kBigSB) * 7) plan_for(G[0], prA, prB, 1, pl);      // (less than 30 % shorter: not worth the padding and the fill)
    } else if (nLines == 2) {      // natural order asked for, or a small map: the narrower line as before
        plan_for(G[1], prA, prB, 1, cand);
        if (forceP != 1 && cost(cand) < cost(pl)) std::swap(pl, cand);
    }
    pl.chainNatural = natural;
}

namespace {
void plan_for(const PlanGraph& G, const std::vector<int>& prA, const std::vector<int>& prB, int wantP, GbaPlan& pl) {
    pl = GbaPlan();
    const int nFa = G.nFa;
    pl.nFa = nFa; pl.rcm = G.rcm; pl.bandwidth = G.bandwidth;
    const std::vector<int>& line = G.line;
    const std::vector<int>& pos = G.pos;
    const std::vector<int>& lo = G.lo;
    const std::vector<int>& hi = G.hi;
    // (sep[v] = 1 + the cut v's separator group belongs to, 0 for a segment's keyframe.  Every group starts on a tile boundary of its own: two groups in one tile
    //  row would make the segments to their left and right share that tile row -- the schedule could then never run them side by side)
    std::vector<int> sepA, sepB, sep;
    auto dissect = [&](int P, std::vector<int>& out, int& nSep) {
        // cut i sits in front of line position cut[i]; side A: the vertex left of a cut with a neighbour at or behind it; side B: the mirror image
        std::vector<int> cut;
        for (int i = 1; i < P; i++) { const int c = (int)((long long)nFa * i / P); if (c > 0 && c < nFa && (cut.empty() || c > cut.back())) cut.push_back(c); }
        sepA.assign((size_t)nFa, 0); sepB.assign((size_t)nFa, 0);
        int nA = 0, nB = 0;
        for (int v = 0; v < nFa; v++) {
            const int p = pos[v];
            const auto nx = std::upper_bound(cut.begin(), cut.end(), p);            // first cut behind p; the cuts <= p are [begin, nx)
            if (nx != cut.end() && *nx <= hi[v]) { sepA[v] = 1 + (int)(nx - cut.begin()); nA++; }
            if (nx != cut.begin() && *(nx - 1) > lo[v]) { sepB[v] = 1 + (int)(nx - 1 - cut.begin()); nB++; }
        }
        out = nA <= nB ? sepA : sepB;
        nSep = std::min(nA, nB);
        return cut;
    };
    std::vector<int> cut;
    int nSep = 0;
    if (wantP > 1) cut = dissect(wantP, sep, nSep);
    else sep.assign((size_t)nFa, 0);
    pl.P = (int)cut.size() + 1; pl.nSep = nSep;
    // ---- rows: segments (64-aligned), then the separator block (64-aligned), N a multiple of 64 so that the right-hand side row has a tile row of its own
    pl.rowOf.assign((size_t)nFa, -1);
    pl.segStart.clear();
    int row = 0;
    {
        size_t ci = 0;
        bool open = false;
        for (int p = 0; p < nFa; p++) {
            while (ci < cut.size() && cut[ci] == p) { open = false; ci++; }
            const int v = line[p];
            if (sep[v]) continue;
            if (!open) { row = up64(row); pl.segStart.push_back(row); open = true; }
            pl.rowOf[v] = row; row += 6;
        }
    }
    row = up64(row);
    pl.sepStart = row;
    // the separator groups, each from a tile boundary, in the order of a cyclic reduction of the chain they form once the segments are gone (every group is coupled
    // to its two neighbours through the segments between them): the odd cuts first -- they do not see each other --, then every second of the rest, and so on.  With
    // the groups in the order of their cuts the block is a chain of 2 (P - 1) dependent panels; this way its depth is logarithmic in P.
    std::vector<int> gorder;
    for (int step = 1; step <= (int)cut.size(); step *= 2)
        for (int gq = step; gq <= (int)cut.size(); gq += 2 * step) gorder.push_back(gq);
    static const bool envSepChain = getenv("EAO_BA_ND_SEP_CHAIN") != nullptr;      // (A/B: the separators in the order of their cuts)
    if (envSepChain) { gorder.clear(); for (int gq = 1; gq <= (int)cut.size(); gq++) gorder.push_back(gq); }
    for (int gq : gorder) {
        row = up64(row);
        for (int p = 0; p < nFa; p++) { const int v = line[p]; if (sep[v] == gq) { pl.rowOf[v] = row; row += 6; } }
    }
    pl.N = std::max(64, up64(row));
    pl.segStart.push_back(pl.sepStart);      // (sentinel: segment s spans rows [segStart[s], segStart[s + 1]))
    pl.rowCam.assign((size_t)pl.N, -1);
    for (int v = 0; v < nFa; v++) for (int q = 0; q < 6; q++) pl.rowCam[pl.rowOf[v] + q] = v * 6 + q;
    const int N = pl.N, tN = N >> 6, T = tN + 1;
    pl.T = T; pl.RP = T * 64;
    // ---- live tiles + fill-in
    std::vector<unsigned char> live((size_t)T * T, 0);
    auto mark = [&](int r, int c) { const int a = r >> 6, b = c >> 6; live[(size_t)std::max(a, b) * T + std::min(a, b)] = 1; };
    for (size_t k = 0; k < prA.size(); k++) {
        const int ra = pl.rowOf[prA[k]], rb = pl.rowOf[prB[k]];
        mark(ra, rb); mark(ra + 5, rb); mark(ra, rb + 5); mark(ra + 5, rb + 5);
    }
    for (int t = 0; t < T; t++) live[(size_t)t * T + t] = 1;
    for (int t = 0; t <= tN; t++) live[(size_t)tN * T + t] = 1;
    std::vector<int> rws;
    for (int k = 0; k < T; k++) {
        rws.clear();
        for (int i = k + 1; i < T; i++) if (live[(size_t)i * T + k]) rws.push_back(i);
        for (size_t a1 = 0; a1 < rws.size(); a1++)
            for (size_t b1 = 0; b1 <= a1; b1++) live[(size_t)rws[a1] * T + rws[b1]] = 1;
    }
    pl.tileMap.assign((size_t)T * T, -1);
    pl.bigTiles = 0;
    for (int i = 0; i < T; i++) for (int j = 0; j <= i; j++) if (live[(size_t)i * T + j]) pl.tileMap[(size_t)i * T + j] = pl.bigTiles++;
    // ---- panels: the tile rows below each (rowsBelow), dependencies, launch indices
    const int nbk = N / kBigNB;
    pl.nbk = nbk;
    std::vector<std::vector<int>> below((size_t)nbk);
    for (int kb = 0; kb < nbk; kb++) {
        const int kc = kb >> 1;
        std::vector<int>& b = below[kb];
        if (!(kb & 1)) b.push_back(kc);                                  // the second half of the panel's own tile
        for (int i = kc + 1; i < T; i++) if (live[(size_t)i * T + kc]) b.push_back(i);
    }
    std::vector<int> launchOf((size_t)nbk, 0), provider((size_t)nbk, -1);
    std::vector<std::vector<unsigned char>> used;                      // used[launch][tile row]
    std::vector<int> lastAt((size_t)T, -1), lastWho((size_t)T, -1);    // per tile row: the latest launch that touches it, and its panel
    for (int kb = 0; kb < nbk; kb++) {
        const int tk = kb >> 1;
        // every earlier panel that reaches tile row tk is a dependency; the latest of them is unique (they all share tk)
        int t = lastAt[tk] + 1;
        provider[kb] = lastWho[tk];
        for (;; t++) {
            if ((int)used.size() <= t) used.resize((size_t)t + 1, std::vector<unsigned char>((size_t)T, 0));
            bool clash = false;
            for (int r : below[kb]) if (r != tN && used[t][r]) { clash = true; break; }
            if (!clash) break;
        }
        launchOf[kb] = t;
        for (int r : below[kb]) if (r != tN) { used[t][r] = 1; if (t > lastAt[r]) { lastAt[r] = t; lastWho[r] = kb; } }
    }
    // (a panel whose tile row nobody reached before starts from the assembled block: its diagonal block comes from a k_bal_diag launch)
    const int nLaunch = (int)used.size();
    std::vector<std::vector<int>> byLaunch((size_t)nLaunch);
    for (int kb = 0; kb < nbk; kb++) byLaunch[launchOf[kb]].push_back(kb);
    // which (provider panel, tile) carries a look-ahead: panel kb's diagonal block is factored by its provider's workgroup for tile (tk, tk)
    std::vector<std::vector<int>> looks((size_t)nbk);                  // looks[j] = panels whose diagonal block j's launch factors
    for (int kb = 0; kb < nbk; kb++) if (provider[kb] >= 0) looks[provider[kb]].push_back(kb);
    pl.work.clear(); pl.launches.clear(); pl.diagList.clear();
    for (int t = 0; t < nLaunch; t++) {
        GbaPlan::Launch L;
        L.diagOff = (int)pl.diagList.size();
        for (int kb : byLaunch[t]) if (provider[kb] < 0) pl.diagList.push_back(kb);
        L.diagCnt = (int)pl.diagList.size() - L.diagOff;
        L.off = (int)(pl.work.size() / 2);
        // two passes: every panel's look-ahead records first (they are the chain's critical path: dispatched first), then the rest
        for (int pass = 0; pass < 2; pass++)
            for (int kb : byLaunch[t]) {
                const int kc = kb >> 1;
                const std::vector<int>& b = below[kb];
                auto pslot = [&](int x) { return pl.tileMap[(size_t)x * T + kc]; };
                for (size_t a1 = 0; a1 < b.size(); a1++)
                    for (size_t b1 = 0; b1 <= a1; b1++) {
                        const int ta = b[a1], tb = b[b1], sc = pl.tileMap[(size_t)ta * T + tb];
                        if (sc < 0 || pslot(ta) < 0 || pslot(tb) < 0) continue;
                        int next = -1;
                        if (ta == tb) for (int k2 : looks[kb]) if ((k2 >> 1) == ta) next = k2;
                        // the right-hand side row's own tile (tN, tN) is scratch: its record only exists to archive the panel's z entries, and only when no other
                        // record of this panel has tile row tN as its row role
                        const bool zOnly = ta == tN && tb == tN;
                        if (zOnly && b.size() > 1) continue;
                        if ((next >= 0) != (pass == 0)) continue;
                        // archive: ONE record per (panel, tile row) stores the row's l entries -- the diagonal-target record, or for the z row the first listed one
                        const bool archive = ta == tb ? true : (ta == tN && b1 == 0);
                        const int flags = (next >= 0 ? 1 : 0) | (archive ? 2 : 0) | (zOnly ? 4 : 0);
                        pl.work.push_back(make_int4(ta, tb, sc, pslot(ta)));
                        pl.work.push_back(make_int4(pslot(tb), kb, flags, next));
                    }
            }
        L.cnt = (int)(pl.work.size() / 2) - L.off;
        pl.launches.push_back(L);
    }
    // ---- back substitution: separator super-blocks bottom-up (one launch each), then the segments' super-blocks level by level
    pl.sb.clear(); pl.sbLaunches.clear();
    auto add_launch = [&](int off) {
        GbaPlan::SbLaunch s; s.off = off; s.cnt = (int)pl.sb.size() - off; s.maxChunks = 1;
        for (int i = off; i < (int)pl.sb.size(); i++) s.maxChunks = std::max(s.maxChunks, pl.sb[i].w - pl.sb[i].z);
        if (s.cnt) pl.sbLaunches.push_back(s);
    };
    if (N > pl.sepStart)
        for (int J0 = pl.sepStart + (N - pl.sepStart - 1) / kBigSB * kBigSB; J0 >= pl.sepStart; J0 -= kBigSB) {
            const int off = (int)pl.sb.size();
            pl.sb.push_back(make_int4(J0, std::min(kBigSB, N - J0), 0, J0 >> 6));
            add_launch(off);
        }
    const int nSeg = (int)pl.segStart.size() - 1;
    int maxSb = 0;
    for (int s = 0; s < nSeg; s++) maxSb = std::max(maxSb, (pl.segStart[s + 1] - pl.segStart[s] + kBigSB - 1) / kBigSB);
    for (int lev = 0; lev < maxSb; lev++) {
        const int off = (int)pl.sb.size();
        for (int s = 0; s < nSeg; s++) {
            const int r0 = pl.segStart[s], r1 = pl.segStart[s + 1] > r0 ? pl.segStart[s + 1] : r0;
            // (the rows of a segment end at its padded length: [r0, next segment's start) may hold nothing but the alignment gap, which up64 already closed)
            const int len = r1 - r0, nSb = (len + kBigSB - 1) / kBigSB, j = nSb - 1 - lev;
            if (j < 0) continue;
            const int J0 = r0 + j * kBigSB;
            pl.sb.push_back(make_int4(J0, std::min(kBigSB, r1 - J0), r0 >> 6, J0 >> 6));
        }
        add_launch(off);
    }
    pl.chainEstimate = (int)(pl.launches.size() + pl.sbLaunches.size());
}
}  // namespace

}  // namespace lm
}  // namespace eao

namespace eao {
namespace lm {
eao_status gba_attributes() {
    // (per-function, process-wide state: only ever raised -- see BALaunch::attributes)
    static std::atomic<int> aBack{0};
    const int want = (int)((kBigSB / kBigNB) * kBigNB * kBigNB * sizeof(double));
    int have = aBack.load();
    while (want > have) {
        EAO_HIP(hipFuncSetAttribute((const void*)k_bal_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize, want));
        if (aBack.compare_exchange_weak(have, want)) break;
    }
    return EAO_OK;
}

// one LM trial of a map-scale window behind its linearisation: memset of the working tiles, pair assembly, the panel factorisation, back substitution
void gba_enqueue_pair_fill(const BALaunch& L) {
    if (L.d.nPairsNZ > 0) hipLaunchKernelGGL(k_bal_pair_fill, dim3(eao::cdiv(L.d.nPairsNZ, 4), 1, 1), dim3(256), 0, L.s, L.W, L.wp(0));
}
void gba_enqueue_trial(const BALaunch& L, int par, bool firstTrial) {
    const BADims& d = L.d;
    const BADev* W = L.W;
    hipStream_t s = L.s;
    auto wp = [&](int p) { return L.wp(p); };
    const GbaPlan& pl = *d.plan;
    (void)hipMemsetAsync(d.big, 0, ((size_t)d.bigTiles << 12) * sizeof(double), s);
    if (d.nPairsLong) hipLaunchKernelGGL(k_bal_schur_pairs<256>, dim3(d.nPairsLong), dim3(256), 0, s, W, wp(par), firstTrial ? 1 : 0, 0);
    if (d.nPairsSlots > d.nPairsLong) hipLaunchKernelGGL(k_bal_schur_pairs<64>, dim3(d.nPairsSlots - d.nPairsLong), dim3(64), 0, s, W, wp(par), firstTrial ? 1 : 0, d.nPairsLong);
    BigStepArgs A = d.bigArgs;
    A.ctl = d.bigCtl0 + 8 * (par & 1);
    // the factorisation: one launch per level of the schedule (GbaPlan::launches) -- the panels of independent segments side by side
    for (const GbaPlan::Launch& Lq : pl.launches) {
        if (Lq.diagCnt) hipLaunchKernelGGL(k_bal_diag, dim3(Lq.diagCnt), dim3(64), 0, s, W, wp(par), Lq.diagOff);
        if (Lq.cnt) {
            A.nByValue = std::min(Lq.cnt, kBigByValue);
            std::memcpy(A.rec, pl.work.data() + 2 * (size_t)Lq.off, (size_t)A.nByValue * 2 * sizeof(int4));
            hipLaunchKernelGGL(k_bal_step, dim3(Lq.cnt), dim3(256), 0, s, A, Lq.off);
        }
    }
    hipLaunchKernelGGL(k_bal_linv, dim3(eao::cdiv(pl.nbk, 2)), dim3(64), 0, s, W, wp(par), pl.nbk);
    for (const GbaPlan::SbLaunch& Sq : pl.sbLaunches)
        hipLaunchKernelGGL(k_bal_backsolve, dim3(Sq.maxChunks, Sq.cnt), dim3(256), (kBigSB / kBigNB) * kBigNB * kBigNB * sizeof(double), s, W, wp(par), Sq.off);
    hipLaunchKernelGGL(k_bal_apply, dim3(eao::cdiv(std::max(d.nC, 1), 256)), dim3(256), 0, s, W, wp(par));
}
}  // namespace lm
}  // namespace eao

extern "C" {

eao_status eao_bundle_adjustment_plan(int32_t n_free, int32_t n_pairs, const int32_t* pair_a, const int32_t* pair_b, int32_t force_segments, eao_gba_plan_info* info,
                                      int32_t* row_of, int32_t* tile_map, int32_t cap_tile_map, int32_t* work, int32_t cap_work, int32_t* launches, int32_t cap_launches,
                                      int32_t* diag_list, int32_t cap_diag, int32_t* sb, int32_t cap_sb, int32_t* sb_launches, int32_t cap_sb_launches) {
    EAO_REQUIRE(n_free > 0 && n_free <= kBigMaxFree && n_pairs >= 0 && (n_pairs == 0 || (pair_a && pair_b)) && info, "bad argument");
    std::vector<int> A(pair_a, pair_a + n_pairs), B(pair_b, pair_b + n_pairs);
    for (int k = 0; k < n_pairs; k++) EAO_REQUIRE(A[k] >= 0 && A[k] <= B[k] && B[k] < n_free, "pair %d: (%d, %d) is not 0 <= a <= b < n_free", k, A[k], B[k]);
    GbaPlan pl;
    gba_build_plan(n_free, A, B, force_segments, pl);
    *info = eao_gba_plan_info{pl.nFa, pl.N, pl.T, pl.nbk, pl.bigTiles, pl.P, pl.nSep, pl.sepStart, pl.rcm, pl.bandwidth, pl.chainNatural, pl.chainEstimate,
                              (int32_t)pl.launches.size(), (int32_t)(pl.work.size() / 2), (int32_t)pl.diagList.size(), (int32_t)pl.sb.size(), (int32_t)pl.sbLaunches.size()};
    if (row_of) std::memcpy(row_of, pl.rowOf.data(), pl.rowOf.size() * 4);
    if (tile_map && cap_tile_map >= (int)pl.tileMap.size()) std::memcpy(tile_map, pl.tileMap.data(), pl.tileMap.size() * 4);
    if (work && cap_work >= (int)(pl.work.size() / 2)) std::memcpy(work, pl.work.data(), pl.work.size() * sizeof(int4));
    if (launches && cap_launches >= (int)pl.launches.size()) std::memcpy(launches, pl.launches.data(), pl.launches.size() * sizeof(GbaPlan::Launch));
    if (diag_list && cap_diag >= (int)pl.diagList.size()) std::memcpy(diag_list, pl.diagList.data(), pl.diagList.size() * 4);
    if (sb && cap_sb >= (int)pl.sb.size()) std::memcpy(sb, pl.sb.data(), pl.sb.size() * sizeof(int4));
    if (sb_launches && cap_sb_launches >= (int)pl.sbLaunches.size()) std::memcpy(sb_launches, pl.sbLaunches.data(), pl.sbLaunches.size() * sizeof(GbaPlan::SbLaunch));
    return EAO_OK;
}

}  // extern "C"
