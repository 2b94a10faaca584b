// micro-benchmark (round 3, behind bench.py's VALU_PEAK_WAVE_INSTS): chip-wide issue rate of the integer / packed VALU instructions the
// ORB kernels lean on, at 1 / 2 / 4 / 8 resident waves per SIMD (grid = 256 CUs x occ workgroups of 256 threads; HIP events around the
// launch).  Output: wave-instructions per second of the whole chip and the cycles per wave-instruction per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o intops_chip intops_chip.hip && ./intops_chip > profiles/r04_ubench_intops.txt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define BENCH(NAME, EXPR)                                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, unsigned k) {                \
        unsigned a[8];                                                                                 \
        for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j + blockIdx.x;                 \
        unsigned b = k * 3 + 1;                                                                        \
        for (int i = 0; i < iters; i++) {                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; j++) { unsigned x = a[j]; a[j] = (EXPR); }        \
        }                                                                                              \
        unsigned s = 0;                                                                                \
        for (int j = 0; j < 8; j++) s += a[j];                                                         \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                    \
    }
BENCH(k_add, x + b)
BENCH(k_mul_lo, x * b)
BENCH(k_dot4, __builtin_amdgcn_udot4(x, b, k, false))
BENCH(k_dot2, __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, b), k, false))
BENCH(k_perm, __builtin_amdgcn_perm(x, b, 0x07020500u))
BENCH(k_align, __builtin_amdgcn_alignbyte(x, b, 1))
BENCH(k_pkmin, __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, b))))
BENCH(k_pksub, __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, x) - __builtin_bit_cast(u16x2, b)))
BENCH(k_min3, min(min(x, b), k))
BENCH(k_max3, max(max(x, b), k))
BENCH(k_sad, __builtin_amdgcn_sad_u8(x, b, k))
BENCH(k_xor_bcnt, __builtin_popcount(x ^ b) + k)
// round 4 (VERDICT r3 next #8c): the fp32 instructions beside the integer ones -- the guide's 2-cycle figure is v_pk_fma_f32's (two FMAs per lane and
// instruction), the scalar v_fma_f32 issues like the integer instructions above
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define BENCHF(NAME, EXPR)                                                                             \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, unsigned k) {                \
        float a[8];                                                                                    \
        for (int j = 0; j < 8; j++) a[j] = 1.0f + 1e-6f * (float)(threadIdx.x + j + blockIdx.x);       \
        const float b = 1.0f + 1e-7f * (float)k, c = 1e-9f * (float)k;                                 \
        for (int i = 0; i < iters; i++) {                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; j++) { float x = a[j]; a[j] = (EXPR); }           \
        }                                                                                              \
        float s = 0;                                                                                   \
        for (int j = 0; j < 8; j++) s += a[j];                                                         \
        if (s == 12345.678f) out[threadIdx.x] = (unsigned)s;                                           \
    }
BENCHF(k_fma_f32, __builtin_fmaf(x, b, c))
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned* out, int iters, unsigned k) {
    f32x2 a[8];
    for (int j = 0; j < 8; j++) { a[j].x = 1.0f + 1e-6f * (float)(threadIdx.x + j + blockIdx.x); a[j].y = a[j].x + 0.5f; }
    const f32x2 b = {1.0f + 1e-7f * (float)k, 1.0f - 1e-7f * (float)k}, c = {1e-9f * (float)k, 2e-9f * (float)k};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = __builtin_elementwise_fma(a[j], b, c);
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;
    if (s == 12345.678f) out[threadIdx.x] = (unsigned)s;
}
int main() {
    unsigned* out;
    hipMalloc(&out, 4 * 1024);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 20000;
    printf("# %s: %d CUs, %d SIMDs; 8 independent chains per lane, %d iterations; peak assumed by bench.py: 1024 SIMDs x 2.4e9 / 4 = 6.144e11 wave-inst/s\n",
           prop.name, cus, 4 * cus, iters);
    printf("%-18s %10s %16s %22s %20s\n", "instruction", "waves/SIMD", "wave-inst/s", "cycles/inst/SIMD@2.4GHz", "fraction of 6.144e11");
    typedef void (*kern)(unsigned*, int, unsigned);
    struct { const char* name; kern f; int perIter; } ks[] = {{"v_add_u32", k_add, 8}, {"v_mul_lo_u32", k_mul_lo, 8}, {"v_dot4_u32_u8", k_dot4, 8}, {"v_dot2_u32_u16", k_dot2, 8},
        {"v_perm_b32", k_perm, 8}, {"v_alignbyte_b32", k_align, 8}, {"v_pk_min_u16", k_pkmin, 8}, {"v_pk_sub_u16", k_pksub, 8}, {"v_min3_u32", k_min3, 8},
        {"v_max3_u32", k_max3, 8}, {"v_sad_u8", k_sad, 8}, {"v_xor+v_bcnt(+add)", k_xor_bcnt, 16}, {"v_fma_f32", k_fma_f32, 8}, {"v_pk_fma_f32", k_pk_fma_f32, 8}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& kk : ks)
        for (int occ : {1, 2, 4, 8}) {
            const dim3 grid(cus * occ), block(256);
            hipLaunchKernelGGL(kk.f, grid, block, 0, 0, out, 200, 3u);       // warm
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kk.f, grid, block, 0, 0, out, iters, 3u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double winst = (double)cus * occ * 4 * (double)iters * kk.perIter;
            const double rate = winst / (ms * 1e-3);
            printf("%-18s %10d %16.4e %22.2f %20.3f\n", kk.name, occ, rate, 4.0 * cus * 2.4e9 / rate, rate / 6.144e11);
        }
    return 0;
}
