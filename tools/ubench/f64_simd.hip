// micro-benchmark (round 3): what ONE CU's fp64 pipes sustain as a function of the waves per SIMD -- the whole workgroup's span
// (first wave's start to last wave's end), not thread 0's own time as tools/ubench/f64.hip reports.
//   hipcc --offload-arch=gfx950 -O3 -o f64_simd f64_simd.hip && ./f64_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
// the loop body is 64 instructions long (REP8): a taken branch costs about as much as seven fp64 instructions, and the first version of
// this file (8 instructions per iteration) charged it to them -- "8.5 cycles per instruction for a lone wave" was 4.5 + the branch
#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
template <int MODE>   // 0: 8 independent fma chains, 1: one dependent fma chain, 2: independent mul + add pairs (no contraction), 3: 32-bit v_add chains
                      // 4: fma with THREE VGPR operands, 5: mul with two VGPR operands, 6: fma with two VGPR operands + one SGPR
__global__ void k(double* out, long long* t0s, long long* t1s, int iters, const double* in = nullptr, double sc = 0) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (MODE >= 4) { const double* q = in + threadIdx.x * 8; b0 = q[0]; b1 = q[1]; b2 = q[2]; b3 = q[3]; c0 = q[4]; c1 = q[5]; c2 = q[6]; c3 = q[7]; }
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    const double m = 1.0000001, c = 0.5;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);) }
        if (MODE == 1) { REP8(a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);) }
        if (MODE == 2) { REP8(a0 = a0 * m; a1 = a1 + c; a2 = a2 * m; a3 = a3 + c; a4 = a4 * m; a5 = a5 + c; a6 = a6 * m; a7 = a7 + c;) }
        if (MODE == 4) { REP8(a0 = fma(a0, b0, c0); a1 = fma(a1, b1, c1); a2 = fma(a2, b2, c2); a3 = fma(a3, b3, c3); a4 = fma(a4, b0, c1); a5 = fma(a5, b1, c2); a6 = fma(a6, b2, c3); a7 = fma(a7, b3, c0);) }
        if (MODE == 5) { REP8(a0 = a0 * b0; a1 = a1 * b1; a2 = a2 * b2; a3 = a3 * b3; a4 = a4 * c0; a5 = a5 * c1; a6 = a6 * c2; a7 = a7 * c3;) }
        if (MODE == 6) { REP8(a0 = fma(a0, b0, sc); a1 = fma(a1, b1, sc); a2 = fma(a2, b2, sc); a3 = fma(a3, b3, sc); a4 = fma(a4, c0, sc); a5 = fma(a5, c1, sc); a6 = fma(a6, c2, sc); a7 = fma(a7, c3, sc);) }
        if (MODE == 3) { REP8(i0 += i1; i1 += i2; i2 += i3; i3 += i4; i4 += i5; i5 += i6; i6 += i7; i7 += i0;) }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
    if ((threadIdx.x & 63) == 0) { t0s[threadIdx.x >> 6] = t0; t1s[threadIdx.x >> 6] = t1; }
}
static double* g_in = nullptr;
template <int MODE> void run(const char* name, double* out, long long* t0s, long long* t1s) {
    const int iters = 500;
    printf("%-34s", name);
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, t0s, t1s, iters, g_in, 0.25);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, t0s, t1s, iters, g_in, 0.25);
        long long a[16], b[16];
        hipMemcpy(a, t0s, sizeof(a), hipMemcpyDeviceToHost); hipMemcpy(b, t1s, sizeof(b), hipMemcpyDeviceToHost);
        const int nw = threads / 64;
        const long long lo = *std::min_element(a, a + nw), hi = *std::max_element(b, b + nw);
        // cycles of the workgroup's span per instruction of ONE wave, and per instruction issued on a SIMD
        printf("  %4d thr: %5.2f cyc/wave-instr (%4.2f per SIMD-instr)", threads, (hi - lo) / (64.0 * iters), (hi - lo) / (64.0 * iters) / std::max(1, nw / 4));
    }
    printf("\n");
}
int main() {
    double* out; long long *t0s, *t1s;
    hipMalloc(&out, 8 * 1024); hipMalloc(&t0s, 128); hipMalloc(&t1s, 128);
    {
        hipMalloc(&g_in, 8 * 1024 * 8);
        double* h = new double[8 * 1024];
        for (int i = 0; i < 8 * 1024; i++) h[i] = (i % 8 < 4) ? 1.0 + 1e-9 * (i % 97) : 1e-3 * (i % 13);
        hipMemcpy(g_in, h, 8 * 1024 * 8, hipMemcpyHostToDevice);
    }
    run<0>("v_fma_f64, 8 independent chains", out, t0s, t1s);
    run<1>("v_fma_f64, one dependent chain", out, t0s, t1s);
    run<2>("v_mul_f64 / v_add_f64 independent", out, t0s, t1s);
    run<3>("v_add_u32 dependent ring", out, t0s, t1s);
    run<4>("v_fma_f64, three VGPR operands", out, t0s, t1s);
    run<5>("v_mul_f64, two VGPR operands", out, t0s, t1s);
    run<6>("v_fma_f64, two VGPR + one SGPR", out, t0s, t1s);
    return 0;
}
