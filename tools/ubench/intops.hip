// micro-benchmark: issue rate of the integer VALU instructions the ORB kernels lean on (gfx950), 8 independent chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define BENCH(NAME, EXPR)                                                                              \
    __global__ void NAME(unsigned* out, long long* cyc, int iters, unsigned k) {                       \
        unsigned a[8];                                                                                 \
        for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j;                              \
        unsigned b = k * 3 + 1;                                                                        \
        long long t0 = clock64();                                                                      \
        for (int i = 0; i < iters; i++) {                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; j++) { unsigned x = a[j]; a[j] = (EXPR); }        \
        }                                                                                              \
        long long t1 = clock64();                                                                      \
        unsigned s = 0;                                                                                \
        for (int j = 0; j < 8; j++) s += a[j];                                                         \
        out[threadIdx.x] = s;                                                                          \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                        \
    }
BENCH(k_add, x + b)
BENCH(k_mul_lo, x * b)
BENCH(k_mul24, __umul24(x, b))
BENCH(k_mad24, __umul24(x, b) + k)
BENCH(k_dot4, __builtin_amdgcn_udot4(x, b, k, false))
BENCH(k_dot2, __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, b), k, false))
BENCH(k_perm, __builtin_amdgcn_perm(x, b, 0x07020500u))
BENCH(k_align, __builtin_amdgcn_alignbyte(x, b, 1))
BENCH(k_pkmin, __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, b))))
BENCH(k_min3, min(min(x, b), k))
BENCH(k_sad, __builtin_amdgcn_sad_u8(x, b, k))
BENCH(k_bfe, __builtin_amdgcn_ubfe(x, 8, 8) + b)
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 4 * 1024); hipMalloc(&cyc, 64);
    const int iters = 1000;
    typedef void (*kern)(unsigned*, long long*, int, unsigned);
    struct { const char* name; kern f; } ks[] = {{"v_add_u32", k_add}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
        {"v_dot4_u32_u8", k_dot4}, {"v_dot2_u32_u16", k_dot2}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_align}, {"v_pk_min_u16", k_pkmin},
        {"v_min3_u32", k_min3}, {"v_sad_u8", k_sad}, {"v_bfe_u32+add", k_bfe}};
    for (int threads : {64, 256, 512}) {
        printf("threads %d (waves per SIMD %d): cycles per instruction per wave\n", threads, threads / 256 ? threads / 256 : 1);
        for (auto& k : ks) {
            long long h;
            hipLaunchKernelGGL(k.f, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 5u);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("  %-18s %.2f\n", k.name, h / (8.0 * iters));
        }
    }
    return 0;
}
