// hamming.hip -- 256-bit Hamming distances for MI355X (gfx950).
//
// Stands behind ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1649-1665) and the candidate loops of
// the Search* routines (e.g. :83-115, :1402-1426): the distances are exact integers, v_xor + v_bcnt per 32-bit
// word.  Two products:
//   k_hamming_matrix  full na x nb uint16 matrix; write-bound (2 B per pair): each lane keeps two B descriptors in
//                     registers and streams 32-row A tiles from LDS, storing packed u32 (256 B per wave store)
//   k_hamming_best2   per A row the two smallest (distance, column) under lexicographic order == the result of
//                     upstream's sequential "if d<best ... else if d<second" scan; one wavefront per row, lane-
//                     local two-smallest then a 6-step xor-shuffle merge
#include "common.h"

namespace {

__device__ __forceinline__ int dist8(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

constexpr int kRowsPerBlock = 32;

__global__ __launch_bounds__(128) void k_hamming_matrix(const uint4* __restrict__ A, int na, const uint4* __restrict__ B, int nb,
                                                        unsigned short* __restrict__ D) {
    __shared__ uint4 sa[kRowsPerBlock * 2];
    const int pair = blockIdx.z;
    A += (long long)pair * na * 2;
    B += (long long)pair * nb * 2;
    D += (long long)pair * na * nb;
    const int i0 = blockIdx.y * kRowsPerBlock;
    const int j0 = (blockIdx.x * 128 + threadIdx.x) * 2;
    if (threadIdx.x < kRowsPerBlock * 2) {
        const int r = i0 + (threadIdx.x >> 1);
        sa[threadIdx.x] = r < na ? A[(long long)r * 2 + (threadIdx.x & 1)] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (j0 >= nb) return;
    const bool two = j0 + 1 < nb;
    const uint4 b00 = B[(long long)j0 * 2], b01 = B[(long long)j0 * 2 + 1];
    const uint4 b10 = two ? B[(long long)j0 * 2 + 2] : b00, b11 = two ? B[(long long)j0 * 2 + 3] : b01;
    const int rows = min(kRowsPerBlock, na - i0);
    const bool aligned = two && ((nb & 1) == 0);
    for (int r = 0; r < rows; r++) {
        const uint4 a0 = sa[2 * r], a1 = sa[2 * r + 1];
        const unsigned d0 = dist8(a0, a1, b00, b01), d1 = dist8(a0, a1, b10, b11);
        unsigned short* o = D + (long long)(i0 + r) * nb + j0;
        if (aligned) *reinterpret_cast<unsigned*>(o) = d0 | (d1 << 16);
        else { o[0] = (unsigned short)d0; if (two) o[1] = (unsigned short)d1; }
    }
}

// Eight B descriptors per lane (64 VGPRs) and ONE 16-byte store per row and lane: a wave writes 1 KB per store instruction
// instead of 256 B -- the matrix is write-bound, and the wider stores are what the memory system wants (1000 x 1000 x 64
// pairs: 52 -> 47 us, 2.5 -> 2.8 TB/s; the 16 v_xor / v_bcnt per distance are then about as long as the writes).  Needs nb % 8 == 0 and a 16-byte aligned D; everything else takes k_hamming_matrix.
constexpr int kRowsPerBlock8 = 16;
__global__ __launch_bounds__(128) void k_hamming_matrix8(const uint4* __restrict__ A, int na, const uint4* __restrict__ B, int nb,
                                                         unsigned short* __restrict__ D) {
    __shared__ uint4 sa[kRowsPerBlock8 * 2];
    const int pair = blockIdx.z;
    A += (long long)pair * na * 2;
    B += (long long)pair * nb * 2;
    D += (long long)pair * na * nb;
    const int i0 = blockIdx.y * kRowsPerBlock8;
    const int j0 = (blockIdx.x * 128 + threadIdx.x) * 8;
    if (threadIdx.x < kRowsPerBlock8 * 2) {
        const int r = i0 + (threadIdx.x >> 1);
        sa[threadIdx.x] = r < na ? A[(long long)r * 2 + (threadIdx.x & 1)] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (j0 >= nb) return;
    uint4 b[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = B[(long long)j0 * 2 + k];
    const int rows = min(kRowsPerBlock8, na - i0);
    for (int r = 0; r < rows; r++) {
        const uint4 a0 = sa[2 * r], a1 = sa[2 * r + 1];
        unsigned d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = dist8(a0, a1, b[2 * k], b[2 * k + 1]);
        uint4 o;
        o.x = d[0] | (d[1] << 16); o.y = d[2] | (d[3] << 16); o.z = d[4] | (d[5] << 16); o.w = d[6] | (d[7] << 16);
        *reinterpret_cast<uint4*>(D + (long long)(i0 + r) * nb + j0) = o;
    }
}

__global__ __launch_bounds__(256) void k_hamming_best2(const uint4* __restrict__ A, int na, const uint4* __restrict__ B, int nb,
                                                       const uint8_t* __restrict__ mask, eao_best2* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.y;
    const int i = blockIdx.x * 4 + wv;
    if (i >= na) return;
    A += (long long)pair * na * 2;
    B += (long long)pair * nb * 2;
    const uint4 a0 = A[(long long)i * 2], a1 = A[(long long)i * 2 + 1];
    const uint8_t* m = mask ? mask + ((long long)pair * na + i) * nb : nullptr;
    const unsigned none = (256u << 20) | 0xFFFFFu;
    unsigned k1 = none, k2 = none;
    for (int j = lane; j < nb; j += 64) {
        if (m && !m[j]) continue;
        const unsigned d = dist8(a0, a1, B[(long long)j * 2], B[(long long)j * 2 + 1]);
        const unsigned key = (d << 20) | (unsigned)j;
        if (key < k1) { k2 = k1; k1 = key; }
        else if (key < k2) k2 = key;
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        const unsigned o1 = __shfl_xor(k1, dlt), o2 = __shfl_xor(k2, dlt);
        const unsigned n1 = min(k1, o1);
        const unsigned n2 = min(max(k1, o1), min(k2, o2));
        k1 = n1; k2 = n2;
    }
    if (lane == 0) {
        eao_best2 r;
        r.best = (int)(k1 >> 20);
        r.second = (int)(k2 >> 20);
        r.idx = k1 == none ? -1 : (int)(k1 & 0xFFFFF);
        r.idx2 = k2 == none ? -1 : (int)(k2 & 0xFFFFF);
        out[(long long)pair * na + i] = r;
    }
}

// The same result for SIXTEEN A rows per wavefront (no mask): a lane loads each of its B descriptors once and tests it
// against the sixteen rows (LDS broadcast reads), keeping sixteen lane-local (best, second) pairs in registers.  One row per
// wave made every wave pull all of B through L1 / L2 -- 32 bytes per distance, 2 GB per 1000 x 1000 x 64 launch; this is 2 bytes.
constexpr int kB2Rows = 16;
__global__ __launch_bounds__(256) void k_hamming_best2_rows(const uint4* __restrict__ A, int na, const uint4* __restrict__ B, int nb,
                                                            eao_best2* __restrict__ out) {
    __shared__ uint4 sa[4][kB2Rows * 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.y;
    const int i0 = (blockIdx.x * 4 + wv) * kB2Rows;
    if (i0 >= na) return;
    A += (long long)pair * na * 2;
    B += (long long)pair * nb * 2;
    if (lane < kB2Rows * 2) {
        const int r = i0 + (lane >> 1);
        sa[wv][lane] = r < na ? A[(long long)r * 2 + (lane & 1)] : make_uint4(0, 0, 0, 0);
    }
    eao::wave_sync();
    const unsigned none = (256u << 20) | 0xFFFFFu;
    unsigned k1[kB2Rows], k2[kB2Rows];
#pragma unroll
    for (int r = 0; r < kB2Rows; r++) { k1[r] = none; k2[r] = none; }
    for (int j = lane; j < nb; j += 64) {
        const uint4 b0 = B[(long long)j * 2], b1 = B[(long long)j * 2 + 1];
#pragma unroll
        for (int r = 0; r < kB2Rows; r++) {
            const unsigned d = dist8(sa[wv][2 * r], sa[wv][2 * r + 1], b0, b1);
            const unsigned key = (d << 20) | (unsigned)j;
            const unsigned lo = min(key, k1[r]);
            k2[r] = min(max(key, k1[r]), k2[r]);
            k1[r] = lo;
        }
    }
#pragma unroll
    for (int r = 0; r < kB2Rows; r++) {
        unsigned a1 = k1[r], a2 = k2[r];
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {
            const unsigned o1 = __shfl_xor(a1, dlt), o2 = __shfl_xor(a2, dlt);
            const unsigned n1 = min(a1, o1);
            const unsigned n2 = min(max(a1, o1), min(a2, o2));
            a1 = n1; a2 = n2;
        }
        if (lane == r && i0 + r < na) {
            eao_best2 res;
            res.best = (int)(a1 >> 20);
            res.second = (int)(a2 >> 20);
            res.idx = a1 == none ? -1 : (int)(a1 & 0xFFFFF);
            res.idx2 = a2 == none ? -1 : (int)(a2 & 0xFFFFF);
            out[(long long)pair * na + i0 + r] = res;
        }
    }
}

// Consecutive-frame matcher of a device-resident batch (the batched-sequence configuration): pair f = (frame f - 1, frame f)
// of one descriptor block [batch][cap][32], keypoint counts read ON THE DEVICE (the extractor's d_n) -- no host round trip
// between extraction and matching.  Pair 0 takes the halo frame (the last frame of the previous shard) as its left side, or
// is skipped when there is none.  Same sixteen-rows-per-wavefront scheme as k_hamming_best2_rows.
__global__ __launch_bounds__(256) void k_hamming_best2_seq(const uint4* __restrict__ desc, int cap, const int* __restrict__ counts,
                                                           const uint4* __restrict__ halo, int haloN, eao_best2* __restrict__ out) {
    __shared__ uint4 sa[4][kB2Rows * 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = blockIdx.y;
    const uint4* A = f ? desc + (long long)(f - 1) * cap * 2 : halo;
    const int na = f ? min(counts[f - 1], cap) : haloN, nb = min(counts[f], cap);
    const int i0 = (blockIdx.x * 4 + wv) * kB2Rows;
    if (!A || i0 >= na) return;
    const uint4* B = desc + (long long)f * cap * 2;
    if (lane < kB2Rows * 2) {
        const int r = i0 + (lane >> 1);
        sa[wv][lane] = r < na ? A[(long long)r * 2 + (lane & 1)] : make_uint4(0, 0, 0, 0);
    }
    eao::wave_sync();
    const unsigned none = (256u << 20) | 0xFFFFFu;
    unsigned k1[kB2Rows], k2[kB2Rows];
#pragma unroll
    for (int r = 0; r < kB2Rows; r++) { k1[r] = none; k2[r] = none; }
    for (int j = lane; j < nb; j += 64) {
        const uint4 b0 = B[(long long)j * 2], b1 = B[(long long)j * 2 + 1];
#pragma unroll
        for (int r = 0; r < kB2Rows; r++) {
            const unsigned d = dist8(sa[wv][2 * r], sa[wv][2 * r + 1], b0, b1);
            const unsigned key = (d << 20) | (unsigned)j;
            const unsigned lo = min(key, k1[r]);
            k2[r] = min(max(key, k1[r]), k2[r]);
            k1[r] = lo;
        }
    }
#pragma unroll
    for (int r = 0; r < kB2Rows; r++) {
        unsigned a1 = k1[r], a2 = k2[r];
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {
            const unsigned o1 = __shfl_xor(a1, dlt), o2 = __shfl_xor(a2, dlt);
            const unsigned n1 = min(a1, o1);
            const unsigned n2 = min(max(a1, o1), min(a2, o2));
            a1 = n1; a2 = n2;
        }
        if (lane == r && i0 + r < na) {
            eao_best2 res;
            res.best = (int)(a1 >> 20);
            res.second = (int)(a2 >> 20);
            res.idx = a1 == none ? -1 : (int)(a1 & 0xFFFFF);
            res.idx2 = a2 == none ? -1 : (int)(a2 & 0xFFFFF);
            out[(long long)f * cap + i0 + r] = res;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the distances on the MATRIX CORES.  popcount(a ^ b) = popcount(a) + popcount(b) - 2 popcount(a & b), and popcount(a & b) is
// the dot product of the descriptors' bits spread to 0 / 1 bytes: v_mfma_i32_32x32x32_i8 takes 32 x 32 of those products over 32 bit positions
// per instruction (eight instructions per 32 x 32 tile of distances, exact in i32) where the VALU path spends 16 v_xor / v_bcnt per distance.
// A workgroup of four waves owns a block of 128 A rows x 128 B columns:
//   B side   the block's 128 B descriptors are spread ONCE into LDS (byte k of a row = bit k, 272-byte row pitch: the lanes' 16-byte
//            fragment reads fall on distinct banks) together with their popcounts;
//   A side   wave w owns A rows 32 w .. 32 w + 31; lane (r = lane & 31, h = lane >> 5) spreads bits [32 kc + 16 h, + 16) of row r into the
//            sixteen bytes of its operand fragment for step kc -- in registers, once per block (any bijection of the 256 bit positions onto the
//            (kc, h, j) slots gives the same dot product as long as both sides use the same one; this one is the identity);
//   product  D'[j][i] = sum_k Bspread[j][k] Aspread[i][k]: the B rows are the instruction's first operand, so an accumulator register
//            holds four CONSECUTIVE columns j of ONE matrix row i = lane & 31 (reg 4 q + e <-> j = 8 q + 4 h + e): a lane packs them to
//            four uint16 without any lane exchange.
// Matrix mode stages the wave's 32 x 128 uint16 tile through LDS so that the global stores are 16 bytes per lane and 256 contiguous bytes per row
// (the matrix is write-bound: 2 MB per 1000 x 1000 pair); best-2 mode never stores distances: the A side is spread to 0 / -1 bytes, so the
// accumulator holds -popcount(a & b) and a key (distance << 20 | column) is ONE v_lshl_add on top of a per-column LDS word
// (popcount(b_j) << 20 | j) -- two more instructions (v_med3_u32, v_min_u32) keep the lane's two smallest keys, i.e. upstream's
// "if(d < best) ... else if(d < second)" scan in column order (src/ORBmatcher.cc:102-114: the first of equal distances wins).
typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));
constexpr int kMB = 128;                 // block edge (A rows and B columns)
constexpr int kMPitch = 272;             // bytes per spread row
constexpr int kOPitch = 144;             // bytes per staged output row: 64 columns x 2 bytes + 16 (16-byte aligned rows; 136 misaligns the 16-byte reads, 160 changes nothing)
constexpr unsigned kKeyNone = (256u << 20) | 0xFFFFFu;

__device__ __forceinline__ unsigned spread4(unsigned nib) { return __umul24(nib, 0x204081u) & 0x01010101u; }      // bit e of the nibble -> byte e (0 / 1)
template <bool NEG>
__device__ __forceinline__ hm_v4i spread16(unsigned half) {      // sixteen bits -> sixteen bytes (0 / 1, or 0 / -1)
    unsigned w[4] = {spread4(half & 15u), spread4((half >> 4) & 15u), spread4((half >> 8) & 15u), spread4((half >> 12) & 15u)};
    if (NEG) for (int q = 0; q < 4; q++) w[q] = (w[q] << 8) - w[q];      // 1 -> 0xFF per byte (no carries between bytes)
    return hm_v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
}
// thread t of 256 spreads half a descriptor (16 bytes = 128 bits) of B row (t >> 1) into the LDS image and leaves the row's popcount (or key base) behind
__device__ __forceinline__ void spread_b_rows(unsigned char* sB, const uint4 v, int t) {
    const unsigned dw[4] = {v.x, v.y, v.z, v.w};
    unsigned char* dst = sB + (t >> 1) * kMPitch + (t & 1) * 128;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        *reinterpret_cast<hm_v4i*>(dst + 32 * q) = spread16<false>(dw[q] & 0xFFFFu);
        *reinterpret_cast<hm_v4i*>(dst + 32 * q + 16) = spread16<false>(dw[q] >> 16);
    }
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) { unsigned d; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// LDS hand-over between the lanes of ONE wave (its LDS operations execute in order: nothing to wait for).  The empty asm statements with a memory clobber keep the
// COMPILER from moving LDS accesses across the point: the two sides park and fetch through different types (8-byte pairs in, 16-byte rows out), which do not alias in
// its view, and a wavefront-scope fence lowers to no instruction at all (round 4: a hand-over of this shape in csrc/lm.hip was reordered by the scheduler).
__device__ __forceinline__ void hm_wave_sync() { eao::wave_sync(); }

// full matrix: grid (column blocks, row blocks, pairs); nb % 8 == 0 and a 16-byte aligned D (else the popcount kernels above)
__global__ __launch_bounds__(256, 3) void k_hamming_matrix_mfma(const uint4* __restrict__ A, int na, const uint4* __restrict__ B, int nb,
                                                                unsigned short* __restrict__ D) {
    __shared__ __attribute__((aligned(16))) unsigned char sB[kMB * kMPitch];
    __shared__ __attribute__((aligned(16))) unsigned char sOut[4][32 * kOPitch];      // 32 rows x 64 columns of uint16 per wave: two tiles at a time
    __shared__ __attribute__((aligned(16))) unsigned short sPb[kMB];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, h = lane >> 5;
    const int pair = blockIdx.z;
    A += (long long)pair * na * 2;
    B += (long long)pair * nb * 2;
    D += (long long)pair * na * nb;
    const int i0 = blockIdx.y * kMB + 32 * wv, j0 = blockIdx.x * kMB;
    {   // B side: 128 rows, two threads per row
        const int j = j0 + (t >> 1);
        const uint4 v = j < nb ? B[(long long)j * 2 + (t & 1)] : make_uint4(0, 0, 0, 0);
        spread_b_rows(sB, v, t);
        int pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        pc += __shfl_xor(pc, 1);
        if (!(t & 1)) sPb[t >> 1] = (unsigned short)pc;
    }
    // A side: row i0 + r, the half-wave's sixteen bits of every dword
    const int i = i0 + r;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (i < na) { a0 = A[(long long)i * 2]; a1 = A[(long long)i * 2 + 1]; }
    const unsigned adw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    unsigned pa = 0;
    hm_v4i af[8];
#pragma unroll
    for (int kc = 0; kc < 8; kc++) { pa += __popc(adw[kc]); af[kc] = spread16<false>((adw[kc] >> (16 * h)) & 0xFFFFu); }
    const unsigned paPk = pa | (pa << 16);
    __syncthreads();
    unsigned char* so = sOut[wv];
#pragma unroll
    for (int jt = 0; jt < 4; jt++) {
        hm_v16i acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0;
        const unsigned char* bsrc = sB + (jt * 32 + r) * kMPitch + 16 * h;
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            const hm_v4i bf = *reinterpret_cast<const hm_v4i*>(bsrc + 32 * kc);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af[kc], acc, 0, 0, 0);
        }
        // reg 4 q + e: column jt * 32 + 8 q + 4 h + e of row r.  Pairs of uint16 in one dword: every intermediate stays below 2^16 and no half
        // goes negative, so plain 32-bit arithmetic on the packed pairs is exact.
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int jc = jt * 32 + 8 * q + 4 * h;
            const uint2 pb = *reinterpret_cast<const uint2*>(&sPb[jc]);
            const unsigned lo = (paPk + pb.x) - (((unsigned)acc[4 * q] | ((unsigned)acc[4 * q + 1] << 16)) << 1);
            const unsigned hi = (paPk + pb.y) - (((unsigned)acc[4 * q + 2] | ((unsigned)acc[4 * q + 3] << 16)) << 1);
            *reinterpret_cast<uint2*>(so + r * kOPitch + ((jt & 1) * 32 + 8 * q + 4 * h) * 2) = make_uint2(lo, hi);
        }
        if (jt & 1) {
            // two tiles = 32 rows x 128 bytes: eight lanes per row, 16 bytes each -- whole cache lines, 1 KB per store instruction.  (Staging all four
            // tiles first cost 35 KB of LDS per workgroup on top of the 35 KB of spread rows: two workgroups per CU instead of three.)
            hm_wave_sync();
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int row = p * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(so + row * kOPitch + ch * 16);
                const int ii = i0 + row, jj = j0 + (jt - 1) * 32 + ch * 8;
                if (ii < na && jj < nb) *reinterpret_cast<uint4*>(D + (long long)ii * nb + jj) = v;
            }
            hm_wave_sync();
        }
    }
}

// best / second-best per A row.  SEQ = false: `pairs` independent (A, B) sets as eao_hamming_best2_device lays them out; SEQ = true: the
// consecutive frames of a device-resident batch (see k_hamming_best2_seq).  grid (row blocks, pairs or frames).
struct Best2Src { const uint4* A; const uint4* B; int na, nb; const int* counts; const uint4* halo; int haloN; int cap; };
template <bool SEQ>
__global__ __launch_bounds__(256, 2) void k_hamming_best2_mfma(Best2Src S, eao_best2* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char sB[2][kMB * kMPitch];
    __shared__ __attribute__((aligned(16))) unsigned sKb[2][kMB];      // per column: popcount << 20 | column (a huge word beyond nb)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, h = lane >> 5;
    const uint4* A; const uint4* B; int na, nb; long long outBase;
    if (SEQ) {
        const int f = blockIdx.y;
        A = f ? S.A + (long long)(f - 1) * S.cap * 2 : S.halo;
        na = f ? min(S.counts[f - 1], S.cap) : S.haloN; nb = min(S.counts[f], S.cap);
        B = S.A + (long long)f * S.cap * 2;
        outBase = (long long)f * S.cap;
        if (!A) return;
    } else {
        const int pair = blockIdx.y;
        na = S.na; nb = S.nb;
        A = S.A + (long long)pair * na * 2; B = S.B + (long long)pair * nb * 2;
        outBase = (long long)pair * na;
    }
    const int ib = blockIdx.x * kMB;
    if (ib >= na) return;                 // (workgroup-uniform)
    const int i = ib + 32 * wv + r;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (i < na) { a0 = A[(long long)i * 2]; a1 = A[(long long)i * 2 + 1]; }
    const unsigned adw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    unsigned pa = 0;
    hm_v4i af[8];
#pragma unroll
    for (int kc = 0; kc < 8; kc++) { pa += __popc(adw[kc]); af[kc] = spread16<true>((adw[kc] >> (16 * h)) & 0xFFFFu); }
    const unsigned paS = pa << 20;
    const int nChunks = (nb + kMB - 1) / kMB;
    auto fetch = [&](int c) { const int j = c * kMB + (t >> 1); return j < nb ? B[(long long)j * 2 + (t & 1)] : make_uint4(0, 0, 0, 0); };
    auto stage = [&](int c, const uint4 v) {
        spread_b_rows(sB[c & 1], v, t);
        int pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        pc += __shfl_xor(pc, 1);
        const int j = c * kMB + (t >> 1);
        if (!(t & 1)) sKb[c & 1][t >> 1] = j < nb ? ((unsigned)pc << 20) | (unsigned)j : 0x20000000u;
    };
    unsigned k1 = 0x7FFFFFFFu, k2 = 0x7FFFFFFFu;
    if (nChunks > 0) stage(0, fetch(0));
    __syncthreads();
    for (int c = 0; c < nChunks; c++) {
        const bool more = c + 1 < nChunks;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (more) nxt = fetch(c + 1);
        const unsigned char* sb = sB[c & 1];
        const unsigned* kb = sKb[c & 1];
#pragma unroll
        for (int jt = 0; jt < 4; jt++) {
            hm_v16i acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0;
            const unsigned char* bsrc = sb + (jt * 32 + r) * kMPitch + 16 * h;
#pragma unroll
            for (int kc = 0; kc < 8; kc++) {
                const hm_v4i bf = *reinterpret_cast<const hm_v4i*>(bsrc + 32 * kc);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af[kc], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 kq = *reinterpret_cast<const uint4*>(&kb[jt * 32 + 8 * q + 4 * h]);
                const unsigned kbv[4] = {kq.x, kq.y, kq.z, kq.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    // acc = -popcount(a & b):  key = (pa + pb - 2 popcount(a & b)) << 20 | j
                    const unsigned key = ((unsigned)acc[4 * q + e] << 21) + (kbv[e] + paS);
                    k2 = umed3(key, k1, k2);      // (k1 <= k2 always: the median of the three is the new second smallest)
                    k1 = min(k1, key);
                }
            }
        }
        if (more) stage(c + 1, nxt);
        __syncthreads();
    }
    // the two half-waves hold the same rows: merge, then lanes 0..31 write
    {
        const unsigned o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32);
        const unsigned n1 = min(k1, o1);
        const unsigned n2 = min(max(k1, o1), min(k2, o2));
        k1 = min(n1, kKeyNone); k2 = min(n2, kKeyNone);
    }
    if (h == 0 && i < na) {
        eao_best2 res;
        res.best = (int)(k1 >> 20);
        res.second = (int)(k2 >> 20);
        res.idx = k1 == kKeyNone ? -1 : (int)(k1 & 0xFFFFF);
        res.idx2 = k2 == kKeyNone ? -1 : (int)(k2 & 0xFFFFF);
        out[outBase + i] = res;
    }
}

// EAO_HAMMING_MFMA: 0 = popcount kernels only, 1 (default) = matrix cores when the call holds at least 64 blocks of 128 rows, 2 = matrix
// cores whenever the shapes allow.  Read on every call (the parity tests switch it inside one process).
inline int hamming_mfma_mode() { const char* e = getenv("EAO_HAMMING_MFMA"); return e ? atoi(e) : 1; }

struct Scratch {
    eao::DevBuf<uint8_t> a, b, mask;
    eao::DevBuf<unsigned short> d;
    eao::DevBuf<eao_best2> best;
};
thread_local Scratch g_scr;

}  // namespace

namespace {

// One wavefront per (set, row): the row's Hamming distances to every member of its set (itself included) go into a
// 257-bin LDS histogram (LDS atomics: counts do not depend on the order), the median is the bin where the running count
// passes (int)(0.5 (N - 1)), and the set's winner is an atomicMin over (median << 20 | row): least median, first row.
__global__ __launch_bounds__(256) void k_distinct_rows(const uint4* __restrict__ desc, const int* __restrict__ setStart,
                                                       const int* __restrict__ rowSet, int total, unsigned* __restrict__ key) {
    __shared__ unsigned hist[4][264];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    unsigned* h = hist[wv];
    for (int b = lane; b < 264; b += 64) h[b] = 0;
    __syncthreads();
    const bool live = row < total;
    const int s = live ? rowSet[row] : 0;
    const int beg = setStart[s], end = setStart[s + 1], N = end - beg;
    if (live) {
        const uint4 a0 = desc[2 * row], a1 = desc[2 * row + 1];
        for (int j = beg + lane; j < end; j += 64) {
            const uint4 b0 = desc[2 * j], b1 = desc[2 * j + 1];
            const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                          __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    if (live) {
        const int k = (int)(0.5 * (N - 1));      // index of the median in the sorted row (src/MapPoint.cc:294)
        // exclusive prefix over the 257 bins, five bins per lane (lanes own consecutive bins)
        unsigned mine[5], run = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) { const int b = lane * 5 + q; mine[q] = b < 257 ? h[b] : 0; run += mine[q]; }
        unsigned incl = run;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) { const unsigned o = __shfl_up(incl, dlt); if (lane >= dlt) incl += o; }
        unsigned before = incl - run;
        int med = 0x7FFFFFFF;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (med == 0x7FFFFFFF && mine[q] && before + mine[q] > (unsigned)k) med = lane * 5 + q;
            before += mine[q];
        }
        // the lane holding the smallest qualifying bin
        for (int dlt = 32; dlt >= 1; dlt >>= 1) med = min(med, __shfl_down(med, dlt));
        if (lane == 0) atomicMin(&key[s], ((unsigned)med << 20) | (unsigned)(row - beg));
    }
}

struct DistinctScratch {
    hipStream_t stream = nullptr;
    eao::DevBuf<int> start, rowset;
    eao::DevBuf<unsigned char> desc;
    eao::DevBuf<unsigned> key;
    ~DistinctScratch() { if (stream) (void)hipStreamDestroy(stream); }
};
thread_local DistinctScratch g_distinct;

}  // namespace

extern "C" {

eao_status eao_hamming_matrix_device(const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb, int32_t pairs,
                                     uint16_t* d_D, void* stream) {
    EAO_REQUIRE(d_A && d_B && d_D && na > 0 && nb > 0 && pairs > 0, "bad argument");
    EAO_REQUIRE(((uintptr_t)d_A & 15) == 0 && ((uintptr_t)d_B & 15) == 0 && ((uintptr_t)d_D & 3) == 0, "descriptor arrays must be 16-byte aligned");
    eao_status st = eao::require_device();
    if (st) return st;
    // matrix cores from 64 blocks of 128 x 128 on (a quarter of the chip); EAO_HAMMING_MFMA=0 keeps the popcount kernels (A/B runs)
    const int envMfma = hamming_mfma_mode();
    const long long blocks = (long long)eao::cdiv(nb, kMB) * eao::cdiv(na, kMB) * pairs;
    if ((nb & 7) == 0 && ((uintptr_t)d_D & 15) == 0 && envMfma && (blocks >= 64 || envMfma > 1)) {
        hipLaunchKernelGGL(k_hamming_matrix_mfma, dim3(eao::cdiv(nb, kMB), eao::cdiv(na, kMB), pairs), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_A, na,
                           (const uint4*)d_B, nb, d_D);
    } else if ((nb & 7) == 0 && ((uintptr_t)d_D & 15) == 0 && !getenv("EAO_HAMMING_NARROW")) {
        dim3 grid8(eao::cdiv(nb, 1024), eao::cdiv(na, kRowsPerBlock8), pairs);
        hipLaunchKernelGGL(k_hamming_matrix8, grid8, dim3(128), 0, (hipStream_t)stream, (const uint4*)d_A, na, (const uint4*)d_B, nb, d_D);
    } else {
        dim3 grid(eao::cdiv(nb, 256), eao::cdiv(na, kRowsPerBlock), pairs);
        hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(128), 0, (hipStream_t)stream, (const uint4*)d_A, na, (const uint4*)d_B, nb, d_D);
    }
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

eao_status eao_hamming_best2_device(const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb, int32_t pairs,
                                    const uint8_t* d_mask, eao_best2* d_out, void* stream) {
    EAO_REQUIRE(d_A && d_B && d_out && na > 0 && nb > 0 && pairs > 0, "bad argument");
    EAO_REQUIRE(nb < (1 << 20), "nb must be below 2^20");
    EAO_REQUIRE(((uintptr_t)d_A & 15) == 0 && ((uintptr_t)d_B & 15) == 0, "descriptor arrays must be 16-byte aligned");
    eao_status st = eao::require_device();
    if (st) return st;
    const int envMfma = hamming_mfma_mode();
    if (!d_mask && envMfma && ((long long)eao::cdiv(na, kMB) * pairs >= 64 || envMfma > 1)) {      // matrix cores: enough row blocks to spread over the chip
        Best2Src S{(const uint4*)d_A, (const uint4*)d_B, na, nb, nullptr, nullptr, 0, 0};
        hipLaunchKernelGGL(k_hamming_best2_mfma<false>, dim3(eao::cdiv(na, kMB), pairs), dim3(256), 0, (hipStream_t)stream, S, d_out);
    } else if (!d_mask && na >= 256 && !getenv("EAO_HAMMING_NARROW"))     // (few rows: one row per wave fills the chip better)
        hipLaunchKernelGGL(k_hamming_best2_rows, dim3(eao::cdiv(na, 4 * kB2Rows), pairs), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_A, na,
                           (const uint4*)d_B, nb, d_out);
    else
        hipLaunchKernelGGL(k_hamming_best2, dim3(eao::cdiv(na, 4), pairs), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_A, na,
                           (const uint4*)d_B, nb, d_mask, d_out);
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

eao_status eao_hamming_best2_sequence_device(const uint8_t* d_desc, int32_t cap, const int32_t* d_counts, int32_t batch,
                                             const uint8_t* d_halo_desc, int32_t halo_n, eao_best2* d_out, void* stream) {
    EAO_REQUIRE(d_desc && d_counts && d_out && cap > 0 && batch > 0 && halo_n >= 0 && (halo_n == 0 || d_halo_desc), "bad argument");
    EAO_REQUIRE(cap < (1 << 20), "cap must be below 2^20");
    EAO_REQUIRE(((uintptr_t)d_desc & 15) == 0 && ((uintptr_t)d_halo_desc & 15) == 0, "descriptor arrays must be 16-byte aligned");
    eao_status st = eao::require_device();
    if (st) return st;
    const int envMfma = hamming_mfma_mode();
    if (envMfma && ((long long)eao::cdiv(cap, kMB) * batch >= 64 || envMfma > 1)) {
        Best2Src S{(const uint4*)d_desc, nullptr, 0, 0, d_counts, halo_n ? (const uint4*)d_halo_desc : nullptr, halo_n, cap};
        hipLaunchKernelGGL(k_hamming_best2_mfma<true>, dim3(eao::cdiv(cap, kMB), batch), dim3(256), 0, (hipStream_t)stream, S, d_out);
    } else
        hipLaunchKernelGGL(k_hamming_best2_seq, dim3(eao::cdiv(cap, 4 * kB2Rows), batch), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_desc, cap,
                           d_counts, halo_n ? (const uint4*)d_halo_desc : nullptr, halo_n, d_out);
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

eao_status eao_hamming_matrix(const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, uint16_t* D) {
    EAO_REQUIRE(A && B && D && na > 0 && nb > 0, "bad argument");
    eao_status st = eao::require_device();
    if (st) return st;
    Scratch& s = g_scr;
    if ((st = s.a.reserve((size_t)na * 32))) return st;
    if ((st = s.b.reserve((size_t)nb * 32))) return st;
    if ((st = s.d.reserve((size_t)na * nb))) return st;
    EAO_HIP(hipMemcpy(s.a.p, A, (size_t)na * 32, hipMemcpyHostToDevice));
    EAO_HIP(hipMemcpy(s.b.p, B, (size_t)nb * 32, hipMemcpyHostToDevice));
    if ((st = eao_hamming_matrix_device(s.a.p, na, s.b.p, nb, 1, s.d.p, nullptr))) return st;
    EAO_HIP(hipMemcpy(D, s.d.p, (size_t)na * nb * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return EAO_OK;
}

eao_status eao_hamming_best2(const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, const uint8_t* mask, eao_best2* out) {
    EAO_REQUIRE(A && B && out && na > 0 && nb > 0, "bad argument");
    eao_status st = eao::require_device();
    if (st) return st;
    Scratch& s = g_scr;
    if ((st = s.a.reserve((size_t)na * 32))) return st;
    if ((st = s.b.reserve((size_t)nb * 32))) return st;
    if ((st = s.best.reserve((size_t)na))) return st;
    EAO_HIP(hipMemcpy(s.a.p, A, (size_t)na * 32, hipMemcpyHostToDevice));
    EAO_HIP(hipMemcpy(s.b.p, B, (size_t)nb * 32, hipMemcpyHostToDevice));
    const uint8_t* dm = nullptr;
    if (mask) {
        if ((st = s.mask.reserve((size_t)na * nb))) return st;
        EAO_HIP(hipMemcpy(s.mask.p, mask, (size_t)na * nb, hipMemcpyHostToDevice));
        dm = s.mask.p;
    }
    if ((st = eao_hamming_best2_device(s.a.p, na, s.b.p, nb, 1, dm, s.best.p, nullptr))) return st;
    EAO_HIP(hipMemcpy(out, s.best.p, (size_t)na * sizeof(eao_best2), hipMemcpyDeviceToHost));
    return EAO_OK;
}


/* MapPoint::ComputeDistinctiveDescriptors, batched (see include/eao_fusion.h) */
eao_status eao_distinctive_descriptors(int32_t n_sets, const int32_t* set_start, const uint8_t* desc, int32_t* best) {
    EAO_REQUIRE(n_sets >= 0 && (n_sets == 0 || (set_start && best)), "null argument");
    eao_status st = eao::require_device();
    if (st) return st;
    if (n_sets == 0) return EAO_OK;
    const int total = set_start[n_sets];
    EAO_REQUIRE(set_start[0] == 0 && total >= 0 && (total == 0 || desc), "bad set table");
    for (int s2 = 0; s2 < n_sets; s2++) EAO_REQUIRE(set_start[s2 + 1] >= set_start[s2], "set table not ascending");
    DistinctScratch& c = g_distinct;
    if (!c.stream) EAO_HIP(eao::create_stream(&c.stream, eao::StreamClass::Background));
    if ((st = c.start.reserve((size_t)n_sets + 1))) return st;
    if ((st = c.desc.reserve(std::max((size_t)total, (size_t)1) * 32))) return st;
    if ((st = c.rowset.reserve(std::max(total, 1)))) return st;
    if ((st = c.key.reserve(n_sets))) return st;
    std::vector<int> rowset(std::max(total, 1));
    for (int s2 = 0; s2 < n_sets; s2++)
        for (int k = set_start[s2]; k < set_start[s2 + 1]; k++) rowset[k] = s2;
    EAO_HIP(hipMemcpyAsync(c.start.p, set_start, ((size_t)n_sets + 1) * sizeof(int), hipMemcpyHostToDevice, c.stream));
    if (total) {
        EAO_HIP(hipMemcpyAsync(c.desc.p, desc, (size_t)total * 32, hipMemcpyHostToDevice, c.stream));
        EAO_HIP(hipMemcpyAsync(c.rowset.p, rowset.data(), (size_t)total * sizeof(int), hipMemcpyHostToDevice, c.stream));
    }
    EAO_HIP(hipMemsetAsync(c.key.p, 0xFF, (size_t)n_sets * sizeof(unsigned), c.stream));
    if (total) hipLaunchKernelGGL(k_distinct_rows, dim3(eao::cdiv(total, 4)), dim3(256), 0, c.stream, (const uint4*)c.desc.p, c.start.p, c.rowset.p, total, c.key.p);
    std::vector<unsigned> key(n_sets);
    EAO_HIP(hipMemcpyAsync(key.data(), c.key.p, (size_t)n_sets * sizeof(unsigned), hipMemcpyDeviceToHost, c.stream));
    EAO_HIP(hipStreamSynchronize(c.stream));
    EAO_HIP(hipGetLastError());
    for (int s2 = 0; s2 < n_sets; s2++) best[s2] = key[s2] == 0xFFFFFFFFu ? -1 : (int)(key[s2] & 0xFFFFF);
    return EAO_OK;
}

}  // extern "C"
