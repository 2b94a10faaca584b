// micro-benchmark (round 4): what ONE CU's LDS pipe sustains for the access shapes of k_ba_schur_pairs_mfma (eao_fusion_amd/csrc/lm.hip) -- the
// operand reads in the 4x4x4 instruction's layout with and without the padding lanes, the 16-byte parking writes, ds_bpermute -- and the rate of the two
// fp64 matrix instructions.  One workgroup of 1024 threads (4 waves per SIMD); the figure is the workgroup's span per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o lds_ops lds_ops.hip && ./lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, long long* t0s, long long* t1s, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16][4096];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 16 * 4096 / 8; i += 1024) reinterpret_cast<double*>(&lds[0][0])[i] = i;
    const int blk = (lane >> 2) & 3, kk = lane >> 4;
    const int aRow = 4 * (blk >> 1) + (lane & 3);
    const bool aOn = aRow < 6 && kk < 3;
    const bool half = ((lane >> 2) & 1) == 0;
    unsigned base = (unsigned)(size_t)(&lds[wave][0]);      // LDS byte address (the low 32 bits of the generic pointer's offset are what ds_* take)
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(reinterpret_cast<size_t>(&lds[wave][0]) & 0xFFFFFFFFu)));
    unsigned addr = 0;
    if (MODE == 0 || MODE == 4) addr = base + lane * (MODE == 4 ? 16 : 8);
    if (MODE == 11 || MODE == 12 || MODE == 14) addr = base + ((aRow < 6 ? aRow : 5) * 3 + (kk < 3 ? kk : 2)) * 8;
    if (MODE == 13) addr = base + lane * 16;
    if (MODE == 1 || MODE == 2 || MODE == 5) addr = base + (aOn ? (aRow * 3 + kk) * 8 : 144);
    if (MODE == 3) addr = base + ((4 * (blk >> 1) + (lane & 3)) * 3 + (kk < 3 ? kk : 0)) * 8;
    if (MODE == 6) { const int fl = std::min(lane, 62), fj = fl / 9, fc = fl - 9 * fj; addr = base + fj * 160 + fc * 16; }
    if (MODE == 7) addr = (unsigned)(((lane * 9) & 63) * 4);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d big = {0, 0, 0, 0};
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
    double x = lane, y = wave;
    int bp = lane;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 1 || MODE == 3) {
            REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:160\n ds_read_b64 %2, %8 offset:320\n ds_read_b64 %3, %8 offset:480\n"
                              "ds_read_b64 %4, %8 offset:640\n ds_read_b64 %5, %8 offset:800\n ds_read_b64 %6, %8 offset:960\n ds_read_b64 %7, %8 offset:1120\n s_waitcnt lgkmcnt(0)"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
        }
        if (MODE == 2) {      // the same under an execution mask: only the useful lanes
            if (aOn) {
                REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:160\n ds_read_b64 %2, %8 offset:320\n ds_read_b64 %3, %8 offset:480\n"
                                  "ds_read_b64 %4, %8 offset:640\n ds_read_b64 %5, %8 offset:800\n ds_read_b64 %6, %8 offset:960\n ds_read_b64 %7, %8 offset:1120\n s_waitcnt lgkmcnt(0)"
                                  : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
            }
        }
        if (MODE == 8) {      // ... only the lanes of the column quadrant 0 (the other quadrant's copy would come by DPP)
            if (aOn && half) {
                REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:160\n ds_read_b64 %2, %8 offset:320\n ds_read_b64 %3, %8 offset:480\n"
                                  "ds_read_b64 %4, %8 offset:640\n ds_read_b64 %5, %8 offset:800\n ds_read_b64 %6, %8 offset:960\n ds_read_b64 %7, %8 offset:1120\n s_waitcnt lgkmcnt(0)"
                                  : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
            }
        }
        if (MODE == 5) {      // ds_read2_b64: two landmarks per instruction (what the compiler emits for the A operand)
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d p0, p1, p2, p3;
            REP8(asm volatile("ds_read2_b64 %0, %4 offset1:20\n ds_read2_b64 %1, %4 offset0:40 offset1:60\n ds_read2_b64 %2, %4 offset0:80 offset1:100\n ds_read2_b64 %3, %4 offset0:120 offset1:140\n s_waitcnt lgkmcnt(0)"
                              : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr)); a0 += p0.x; a1 += p1.y; a2 += p2.x; a3 += p3.y;)
        }
        if (MODE == 11) {
            REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:144\n ds_read_b64 %2, %8 offset:288\n ds_read_b64 %3, %8 offset:432\n"
                              "ds_read_b64 %4, %8 offset:576\n ds_read_b64 %5, %8 offset:720\n ds_read_b64 %6, %8 offset:864\n ds_read_b64 %7, %8 offset:1008\n s_waitcnt lgkmcnt(0)"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
        }
        if (MODE == 12 || MODE == 14) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d p0, p1, p2, p3;
            if (MODE == 12) { REP8(asm volatile("ds_read2_b64 %0, %4 offset1:18\n ds_read2_b64 %1, %4 offset0:36 offset1:54\n ds_read2_b64 %2, %4 offset0:72 offset1:90\n ds_read2_b64 %3, %4 offset0:108 offset1:126\n s_waitcnt lgkmcnt(0)"
                              : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr)); a0 += p0.x; a1 += p1.y; a2 += p2.x; a3 += p3.y;) }
            else { REP8(asm volatile("ds_read2_b64 %0, %4 offset1:64\n ds_read2_b64 %1, %4 offset0:128 offset1:192\n ds_read2_b64 %2, %4 offset0:32 offset1:96\n ds_read2_b64 %3, %4 offset0:160 offset1:224\n s_waitcnt lgkmcnt(0)"
                              : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr)); a0 += p0.x; a1 += p1.y; a2 += p2.x; a3 += p3.y;) }
        }
        if (MODE == 13) {
            REP8(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1008\n ds_write_b128 %0, %1 offset:2016\n ds_write_b128 %0, %1 offset:1008\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(q0) : "memory");)
        }
        if (MODE == 4) {
            REP8(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(addr)); ) a0 += q0.x + q1.y;
        }
        if (MODE == 6) {
            REP8(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1120\n ds_write_b128 %0, %1 offset:2240\n ds_write_b128 %0, %1 offset:1120\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(q0) : "memory");)
        }
        if (MODE == 7) {
            REP8(asm volatile("ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(bp) : "v"(addr));)
        }
        if (MODE == 9) { REP8(a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a3, 0, 0, 0);) }
        if (MODE == 10) { REP8(big = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, big, 0, 0, 0);) }
    }
    const long long t1 = clock64();
    out[t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + big.x + big.y + big.z + big.w + bp;
    if (lane == 0) { t0s[wave] = t0; t1s[wave] = t1; }
}
template <int MODE> void run(const char* name, int perIter, double bytes, double* out, long long* t0s, long long* t1s) {
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(1024), 0, 0, out, t0s, t1s, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(1024), 0, 0, out, t0s, t1s, iters);
    long long a[16], b[16];
    hipMemcpy(a, t0s, sizeof(a), hipMemcpyDeviceToHost); hipMemcpy(b, t1s, sizeof(b), hipMemcpyDeviceToHost);
    const long long lo = *std::min_element(a, a + 16), hi = *std::max_element(b, b + 16);
    const double per = (double)(hi - lo) / ((double)iters * perIter * 16);      // span per wave-instruction, all 16 waves of the CU counted
    printf("%-78s %6.2f cycles per wave-instruction on the CU", name, per);
    if (bytes > 0) printf("  (%5.1f useful bytes / cycle)", bytes / per);
    printf("\n");
}
int main() {
    double* out; long long *t0s, *t1s;
    hipMalloc(&out, 8 * 1024); hipMalloc(&t0s, 128); hipMalloc(&t1s, 128);
    printf("# one workgroup of 16 waves on one CU; clock64() (s_memtime) cycles\n");
    run<0>("ds_read_b64, 64 lanes, consecutive addresses", 64, 512, out, t0s, t1s);
    run<1>("ds_read_b64, operand layout of the 4x4x4 instruction: 36 lanes + 28 on a zero slot", 64, 288, out, t0s, t1s);
    run<2>("ds_read_b64, the 36 useful lanes only (execution mask)", 64, 288, out, t0s, t1s);
    run<8>("ds_read_b64, the 18 lanes of one column quadrant only (execution mask)", 64, 144, out, t0s, t1s);
    run<3>("ds_read_b64, all 64 lanes on operand addresses (48 distinct)", 64, 384, out, t0s, t1s);
    run<5>("ds_read2_b64, operand layout, two landmarks per instruction", 32, 576, out, t0s, t1s);
    run<4>("ds_read_b128, 64 lanes, consecutive addresses", 16, 1024, out, t0s, t1s);
    run<6>("ds_write_b128, 63 lanes, 16-byte chunks of seven 144-byte blocks at pitch 160", 32, 1008, out, t0s, t1s);
    run<11>("ds_read_b64, operand layout at pitch 144 (rows 6, 7 and k = 3 repeat a neighbour)", 64, 288, out, t0s, t1s);
    run<12>("ds_read2_b64, the same, two landmarks per instruction", 32, 576, out, t0s, t1s);
    run<14>("ds_read2_b64, the same addresses, the two halves 512 bytes apart", 32, 576, out, t0s, t1s);
    run<13>("ds_write_b128, 64 lanes, consecutive 16-byte chunks", 32, 1024, out, t0s, t1s);
    run<7>("ds_bpermute_b32", 32, 0, out, t0s, t1s);
    run<9>("v_mfma_f64_4x4x4_4b (per SIMD: x 4 waves)", 32, 0, out, t0s, t1s);
    run<10>("v_mfma_f64_16x16x4 (per SIMD: x 4 waves)", 8, 0, out, t0s, t1s);
    return 0;
}
