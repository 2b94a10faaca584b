// micro-benchmark (round 3): how fast does the chip START waves?  Empty kernels (one store per workgroup so that nothing is optimised away) over grids
// of the ORB kernels' shapes: workgroups of 64 / 256 threads, with and without a static LDS allocation.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch dispatch.hip && ./dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSW> __global__ void k(int* out) {
    __shared__ int lds[LDSW > 0 ? LDSW : 1];
    if (LDSW > 0) lds[threadIdx.x % LDSW] = threadIdx.x;
    if (threadIdx.x == 0 && blockIdx.x == 0x7FFFFFF) out[0] = lds[0];
}
template <int LDSW> void run(const char* name, int wgs, int threads, int* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<LDSW>, dim3(wgs), dim3(threads), 0, 0, out);
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<LDSW>, dim3(wgs), dim3(threads), 0, 0, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, waves = (double)wgs * threads / 64;
    printf("%-44s %7d workgroups x %4d threads: %7.1f us per launch = %5.2f waves / ns, %5.2f workgroups / ns\n", name, wgs, threads, us, waves / us / 1e3, wgs / us / 1e3);
}
int main() {
    int* out; hipMalloc(&out, 64);
    run<0>("empty, no LDS", 16576, 256, out);
    run<2596>("empty, 10 KB LDS (orientation + description)", 16576, 256, out);
    run<0>("empty, no LDS", 66304, 64, out);
    run<1536>("empty, 6 KB LDS (FAST cells)", 57344, 64, out);
    run<0>("empty, no LDS", 4144, 1024, out);
    run<0>("empty, no LDS", 1024, 256, out);
    return 0;
}
