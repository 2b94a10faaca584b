// Is DivBy<true> (csrc/lm.hip: one refined reciprocal shared by several quotients) the same double as the compiler's IEEE
// division?  2^26 random (numerator, denominator) pairs per range on the device, bit patterns compared.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/chk_divby.hip -o /tmp/chk_divby && /tmp/chk_divby
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__device__ __forceinline__ double rnd(uint64_t k, int emin, int emax) {      // random mantissa and sign, exponent in [emin, emax]
    const uint64_t h = mix(k);
    const int e = emin + (int)((h >> 52) % (uint64_t)(emax - emin + 1));
    const double m = 1.0 + (double)(h & ((1ull << 52) - 1)) * 0x1p-52;
    return ldexp((h >> 63) ? -m : m, e);
}
__global__ void k(unsigned long long n, int e0, int e1, int d0, int d1, unsigned long long* bad, double* ex) {
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = rnd(2 * i + (unsigned long long)e0 * 7919u, e0, e1), d = rnd(2 * i + 1 + (unsigned long long)d0 * 104729u, d0, d1);
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    const double q = a * r;
    const double fast = fma(fma(-d, q, a), r, q);
    const double ieee = a / d;
    if (__double_as_longlong(fast) != __double_as_longlong(ieee)) {
        if (atomicAdd(bad, 1ull) == 0) { ex[0] = a; ex[1] = d; ex[2] = fast; ex[3] = ieee; }
    }
}
int main() {
    unsigned long long* bad; double* ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 32);
    const int ranges[][4] = {{-20, 20, -20, 20}, {-60, 60, -60, 60}, {-300, 300, -200, 200}, {-300, 300, -400, 400}, {-900, 900, -400, 400}};
    int rc = 0;
    for (auto& r : ranges) {
        hipMemset(bad, 0, 8);
        const unsigned long long n = 1ull << 26;
        hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, r[0], r[1], r[2], r[3], bad, ex);
        unsigned long long h; double e[4];
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 32, hipMemcpyDeviceToHost);
        printf("numerator 2^[%d, %d], denominator 2^[%d, %d]: %llu of %llu quotients differ", r[0], r[1], r[2], r[3], h, n);
        if (h) printf("   e.g. %a / %a = %a (shared) vs %a (ieee)", e[0], e[1], e[2], e[3]);
        printf("\n");
        if (h && r[1] <= 300) rc = 1;
    }
    return rc;
}
