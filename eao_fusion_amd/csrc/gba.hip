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

}  // namespace

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
void gba_enqueue_trial(const BALaunch& L, int par, bool firstTrial) {
    const BADims& d = L.d;
    const BADev* W = L.W;
    hipStream_t s = L.s;
    auto wp = [&](int p) { return L.wp(p); };
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
}
}  // namespace lm
}  // namespace eao
