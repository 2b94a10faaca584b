// micro-benchmark: fp64 FMA issue rate / dependent latency and LDS read latency on one CU (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_indep(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double m = 1.0000001, c = 0.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
        a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
    }
    long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x;
    const double m = 1.0000001, c = 0.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
        a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
    }
    long long t1 = clock64();
    out[threadIdx.x] = a0;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp(double* out, long long* cyc, int iters) {
    double a0 = threadIdx.x + 1.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) a0 = __builtin_amdgcn_rcp(a0) + 1.0;
    }
    long long t1 = clock64();
    out[threadIdx.x] = a0;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* cyc, int iters) {
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = (double)((i * 7 + 8) & 4088);
    __syncthreads();
    int idx = threadIdx.x & 4095;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) idx = (int)sm[idx] & 4095;   // dependent LDS reads (+ cvt)
    }
    long long t1 = clock64();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_barrier(double* out, long long* cyc, int iters) {
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    const int iters = 1000;
    for (int threads : {64, 256, 512, 1024}) {
        long long h[4];
        hipLaunchKernelGGL(k_indep, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_dep, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(&h[1], cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_rcp, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(&h[2], cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(&h[3], cyc, 8, hipMemcpyDeviceToHost);
        long long hb;
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, out, cyc, iters); hipMemcpy(&hb, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %4d: indep fma %.2f cyc/instr  dep fma %.2f  rcp+add %.2f  dep lds read(+cvt,and) %.2f  barrier %.2f\n", threads,
               h[0] / (8.0 * iters), h[1] / (8.0 * iters), h[2] / (8.0 * iters), h[3] / (8.0 * iters), hb / (4.0 * iters));
    }
    return 0;
}
