// micro-benchmark: v_fmac_f64_dpp with row_newbcast (the only DPP control 64-bit ALU operations take on gfx90a+): does gfx950 execute it, with what result, at what rate?
// One instruction = acc += (-x of lane K of my 16-lane row) * y: a broadcast FMA without LDS and without v_readlane (2 per double) -- the candidate for the row solves and
// the in-register LDL^T of the map-scale factorisation (csrc/lm.hip).
//   hipcc -O3 --offload-arch=gfx950 -o dpp_f64 dpp_f64.hip && ./dpp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int K> __device__ __forceinline__ void fmac_nb(double& acc, double src, double y) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(y), "n"(K));
}
template <int K> __device__ __forceinline__ double mov_nb(double src) {
    double d;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(K));
    return d;
}
__global__ void k_check(double* out, const double* in) {
    const int t = threadIdx.x;
    const double x = in[t], y = in[64 + t];
    double a0 = in[128 + t], a1 = a0, a2 = a0;
    fmac_nb<0>(a0, x, y); fmac_nb<5>(a1, x, y); fmac_nb<15>(a2, x, y);
    out[t] = a0; out[64 + t] = a1; out[128 + t] = a2; out[192 + t] = mov_nb<9>(x);
}
template <int MODE> __global__ void k_rate(double* out, const double* in, int iters, long long* cyc) {
    const int t = threadIdx.x;
    double x = in[t], y = in[64 + t];
    double a[8];
    for (int q = 0; q < 8; q++) a[q] = in[128 + t] + q;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {      // eight independent accumulators
            fmac_nb<0>(a[0], x, y); fmac_nb<1>(a[1], x, y); fmac_nb<2>(a[2], x, y); fmac_nb<3>(a[3], x, y);
            fmac_nb<4>(a[4], x, y); fmac_nb<5>(a[5], x, y); fmac_nb<6>(a[6], x, y); fmac_nb<7>(a[7], x, y);
        } else if (MODE == 1) {   // one dependent chain
            fmac_nb<0>(a[0], x, y); fmac_nb<1>(a[0], x, y); fmac_nb<2>(a[0], x, y); fmac_nb<3>(a[0], x, y);
            fmac_nb<4>(a[0], x, y); fmac_nb<5>(a[0], x, y); fmac_nb<6>(a[0], x, y); fmac_nb<7>(a[0], x, y);
        } else {                  // the same with plain fma (reference rate)
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = fma(-x, y, a[q]);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < 8; q++) s += a[q];
    out[t] = s;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double h[192], o[256];
    for (int i = 0; i < 64; i++) { h[i] = i + 1.0; h[64 + i] = 2.0 + i * 0.25; h[128 + i] = 100.0 + i; }
    double *din, *dout; long long* dc;
    CK(hipMalloc(&din, sizeof(h))); CK(hipMalloc(&dout, sizeof(o))); CK(hipMalloc(&dc, 64));
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dout, din);
    CK(hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost));
    int bad = 0;
    const int ks[3] = {0, 5, 15};
    for (int v = 0; v < 3; v++)
        for (int i = 0; i < 64; i++) {
            const double src = h[(i & ~15) + ks[v]];
            const double want = fma(-src, h[64 + i], h[128 + i]);
            if (o[64 * v + i] != want) { if (bad < 5) printf("fmac row_newbcast:%d lane %d: got %.17g want %.17g\n", ks[v], i, o[64 * v + i], want); bad++; }
        }
    for (int i = 0; i < 64; i++) if (o[192 + i] != h[(i & ~15) + 9]) { if (bad < 8) printf("mov row_newbcast:9 lane %d: got %g want %g\n", i, o[192 + i], h[(i & ~15) + 9]); bad++; }
    printf("v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast on this device: %s (%d mismatches of 256)\n", bad ? "WRONG" : "correct", bad);
    const int iters = 4000;
    long long c[4];
    const char* names[3] = {"v_fmac_f64_dpp row_newbcast, 8 independent accumulators", "v_fmac_f64_dpp row_newbcast, one dependent chain", "v_fma_f64 (plain), 8 independent accumulators"};
    for (int mode = 0; mode < 3; mode++) {
        for (int threads : {64, 256}) {
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(threads), 0, 0, dout, din, iters, dc);
            else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(threads), 0, 0, dout, din, iters, dc);
            else hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(threads), 0, 0, dout, din, iters, dc);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c, dc, 8, hipMemcpyDeviceToHost));
            printf("%-58s %4d thr: %.2f cycles per wave-instruction\n", names[mode], threads, (double)c[0] / (8.0 * iters));
        }
    }
    return 0;
}
