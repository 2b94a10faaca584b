// What does FETCH_SIZE count per byte actually read, by load width?  MI355X_MICROARCH.md prescribes 2 x FETCH_SIZE for streaming reads on gfx950 (measured with
// 16-byte-per-lane loads); the ORB kernels read 4 and 12 bytes per lane.  Each kernel streams the same buffer once with one load width; run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib        (and once more with WRITE_SIZE for the store kernels)
// and divide the counter (KiB x 1024) by the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <typename T> __device__ unsigned fold(const T& v);
template <> __device__ unsigned fold<unsigned>(const unsigned& v) { return v; }
template <> __device__ unsigned fold<uint2>(const uint2& v) { return v.x ^ v.y; }
template <> __device__ unsigned fold<uint4>(const uint4& v) { return v.x ^ v.y ^ v.z ^ v.w; }
struct u3 { unsigned x, y, z; };
template <> __device__ unsigned fold<u3>(const u3& v) { return v.x ^ v.y ^ v.z; }
template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= fold(src[i]);
    if (acc == 0x12345678u) out[0] = acc;
}
// 12 bytes per lane as the blur reads them: an aligned dwordx3 per lane at a 16-byte stride is NOT what it does -- it loads 12 of every 16... keep the packed form
__global__ __launch_bounds__(256) void k_read_u8(const unsigned char* __restrict__ src, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n, T v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
int main() {
    const size_t bytes = 768ull << 20;      // beyond the 256 MB Infinity Cache
    void* buf; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k_read<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_read<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_read<u3>, dim3(grid), dim3(256), 0, 0, (const u3*)buf, bytes / 12, out);
    hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_read_u8, dim3(grid), dim3(256), 0, 0, (const unsigned char*)buf, bytes / 4, out);      // (a quarter of the buffer, byte loads)
    hipLaunchKernelGGL(k_write<unsigned>, dim3(grid), dim3(256), 0, 0, (unsigned*)buf, bytes / 4, 7u);
    hipLaunchKernelGGL(k_write<uint4>, dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16, make_uint4(1, 2, 3, 4));
    CK(hipDeviceSynchronize());
    printf("bytes per kernel: k_read<unsigned> %zu, k_read<uint2> %zu, k_read<u3> %zu, k_read<uint4> %zu, k_read_u8 %zu, k_write<unsigned> %zu, k_write<uint4> %zu\n",
           bytes, bytes, bytes / 12 * 12, bytes, bytes / 4, bytes, bytes);
    return 0;
}
