// micro-benchmark: cost of a grid-wide barrier (atomic counter + agent-scope fences) across all XCDs, with a
// cross-block data exchange check, vs back-to-back dependent kernel launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ inline void grid_sync(unsigned* bar, unsigned nblocks, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        gen++;
        __atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE);   // agent scope by default for global atomics
        const unsigned target = gen * nblocks;
        int spins = 0;
        while (__atomic_load_n(bar, __ATOMIC_ACQUIRE) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) break; }
    }
    __syncthreads();
}
__global__ void k_persist(unsigned* bar, int* data, int iters, int* errs, long long* cyc) {
    unsigned gen = 0;
    const unsigned nb = gridDim.x;
    long long t0 = wall_clock64();
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x == 0) data[blockIdx.x] = it * 1000 + blockIdx.x;
        grid_sync(bar, nb, gen);
        const int nbr = (blockIdx.x + 37) % nb;
        if (threadIdx.x == 0) { const int v = __atomic_load_n(&data[nbr], __ATOMIC_RELAXED); if (v != it * 1000 + nbr) bad++; }
        grid_sync(bar, nb, gen);
    }
    if (threadIdx.x == 0) { if (bad) atomicAdd(errs, bad); if (blockIdx.x == 0) cyc[0] = wall_clock64() - t0; }
}
__global__ void k_tiny(int* data) { if (threadIdx.x == 0) data[blockIdx.x] += 1; }
int main() {
    unsigned* bar; int* data; int* errs; long long* cyc;
    CK(hipMalloc(&bar, 4)); CK(hipMalloc(&data, 4096)); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&cyc, 8));
    for (int nb : {8, 64, 128, 256}) {
        for (int threads : {256, 1024}) {
            CK(hipMemset(bar, 0, 4)); CK(hipMemset(errs, 0, 4));
            const int iters = 200;
            hipLaunchKernelGGL(k_persist, dim3(nb), dim3(threads), 0, 0, bar, data, iters, errs, cyc);
            CK(hipDeviceSynchronize());
            long long h; int e;
            CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost));
            printf("blocks %3d x %4d threads: %.2f us per grid barrier, exchange errors %d\n", nb, threads, h / 100.0 / (2.0 * iters), e);
        }
    }
    // dependent tiny kernels back to back
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, s, data);
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("1000 dependent tiny kernels: %.2f us each\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 1000.0);
    }
    return 0;
}
