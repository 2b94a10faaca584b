// micro-benchmark (round 4): global_load_lds_dwordx4 (gfx950: 16 bytes per lane straight from global memory into LDS, lane L's data at M0 base + 16 L) against
// global_load_dwordx4 + ds_write_b128 for the fetch of k_ba_schur_pairs_mfma (lane 9 j + c: chunk c of one of seven 144-byte blocks; blocks scattered in a 2 MB array).
//   hipcc --offload-arch=gfx950 -O3 -o gload_lds gload_lds.hip && ./gload_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const unsigned char* __restrict__ src, const int* __restrict__ blk, int nblk, double* out, long long* t0s, long long* t1s, int iters, int* bad) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16][2 * 1024];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fl = lane < 63 ? lane : 62, fj = fl / 9, fc = fl - 9 * fj;
    double acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        const int b = blk[(blockIdx.x * 977 + wave * 131 + i * 7 + fj) % nblk];
        const unsigned char* g = src + (size_t)b * 144 + fc * 16;
        unsigned char* dst = &lds[wave][(i & 1) * 1024];
        if (MODE == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(g);
            *reinterpret_cast<uint4*>(dst + lane * 16) = v;
        } else {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
        const double* p = reinterpret_cast<const double*>(dst + ((lane * 5) & 63) * 16);
        acc += p[0];
        if (i == iters - 1 && bad) {      // layout check: LDS [16 L, 16 L + 16) of the last round must hold lane L's chunk
            const uint4 want = *reinterpret_cast<const uint4*>(g), got = *reinterpret_cast<const uint4*>(dst + lane * 16);
            if (want.x != got.x || want.y != got.y || want.z != got.z || want.w != got.w) atomicAdd(bad, 1);
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    out[blockIdx.x * 1024 + t] = acc;
    if (lane == 0 && blockIdx.x == 0) { t0s[wave] = t0; t1s[wave] = t1; }
}
int main() {
    const int nblk = 15000;
    std::vector<unsigned char> h((size_t)nblk * 144 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 2654435761u >> 13);
    std::vector<int> hb(nblk);
    for (int i = 0; i < nblk; i++) hb[i] = (int)(((long long)i * 7919) % nblk);
    unsigned char* src; int* blk; double* out; long long *t0s, *t1s; int* bad;
    hipMalloc(&src, h.size()); hipMalloc(&blk, nblk * 4); hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&t0s, 128); hipMalloc(&t1s, 128); hipMalloc(&bad, 4);
    hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice); hipMemcpy(blk, hb.data(), nblk * 4, hipMemcpyHostToDevice);
    const int iters = 400;
    for (int mode = 0; mode < 2; mode++) {
        for (int grid : {1, 256}) {
            hipMemset(bad, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(1024), 0, 0, src, blk, nblk, out, t0s, t1s, iters, bad);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(1024), 0, 0, src, blk, nblk, out, t0s, t1s, iters, bad);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long a[16], b[16]; int nb = 0;
            hipMemcpy(a, t0s, sizeof(a), hipMemcpyDeviceToHost); hipMemcpy(b, t1s, sizeof(b), hipMemcpyDeviceToHost); hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
            const long long lo = *std::min_element(a, a + 16), hi = *std::max_element(b, b + 16);
            printf("%-46s %3d workgroups of 16 waves: %7.1f cycles per wave-fetch on a CU (workgroup 0), launch %.3f ms, layout mismatches %d\n",
                   mode == 0 ? "global_load_dwordx4 + ds_write_b128" : "global_load_lds_dwordx4", grid, (double)(hi - lo) / (iters * 16.0), ms, nb / 2);
        }
    }
    return 0;
}
