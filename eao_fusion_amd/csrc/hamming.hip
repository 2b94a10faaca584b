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
    __builtin_amdgcn_wave_barrier();
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
    __builtin_amdgcn_wave_barrier();
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
    if ((nb & 7) == 0 && ((uintptr_t)d_D & 15) == 0 && !getenv("EAO_HAMMING_NARROW")) {
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
    if (!d_mask && na >= 256 && !getenv("EAO_HAMMING_NARROW"))     // (few rows: one row per wave fills the chip better)
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
    if (!c.stream) EAO_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
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
