// frame.hip -- the Frame glue either side of the matcher, MI355X (gfx950):
//   k_is_in_frustum      Frame::isInFrustum for every local map point of Tracking::SearchLocalPoints: one thread per point
//   k_grid_assign        Frame::AssignFeaturesToGrid: (cell << 16 | index) keys, bitonic sort in LDS by ONE workgroup,
//                        cell offsets from the sorted keys -- mGrid as a CSR in push_back order
//   k_stereo_from_rgbd   Frame::ComputeStereoFromRGBD: one thread per keypoint
// Behind eao_frame_is_in_frustum / eao_assign_features_to_grid / eao_compute_stereo_from_rgbd (include/eao_fusion.h).
// Float semantics of the cv::Mat expressions: A*x + b accumulates in double and rounds once to float
// (cv::gemm); cv::norm / Mat::dot accumulate in double.  This file is compiled with -ffp-contract=off.
#include <cmath>
#include <cstring>

#include "common.h"
#include "chain_internal.h"

namespace {

using eao::frame::FrustumArgs;

// reference src/Frame.cc:638-695: one thread per map point (the test itself: frustum_point, chain_internal.h)
__global__ __launch_bounds__(256) void k_is_in_frustum(FrustumArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    eao::frame::frustum_point(A, i);
}

// Frame::UndistortKeyPoints, src/Frame.cc:773-806: one thread per keypoint (the arithmetic: undistort_point, chain_internal.h)
__global__ __launch_bounds__(256) void k_undistort(eao::frame::Distortion D, int n, const float* __restrict__ x, const float* __restrict__ y,
                                                   float* __restrict__ ox, float* __restrict__ oy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u = x[i], v = y[i];
    if (D.on) eao::frame::undistort_point(D, u, v, u, v);
    ox[i] = u; oy[i] = v;
}

// One workgroup.  keys[] (LDS) = cell << 16 | index for keypoints that fall into the grid, 0xFFFFFFFF for the others and
// for the padding up to the next power of two; after the sort the items of a cell are in ascending index (= push_back)
// order; cell_start[c] = first position whose cell is >= c (binary search over the sorted keys).
constexpr int kGridThreads = 1024;
__global__ __launch_bounds__(kGridThreads) void k_grid_assign(int n, const float* __restrict__ kx, const float* __restrict__ ky,
                                                              float minX, float minY, float invW, float invH, int cols, int rows,
                                                              int npow2, int* __restrict__ cellStart, int* __restrict__ items) {
    extern __shared__ unsigned gkeys[];
    const int t = threadIdx.x;
    for (int i = t; i < npow2; i += kGridThreads) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n) {
            const int px = (int)roundf((kx[i] - minX) * invW);      // PosInGrid, src/Frame.cc:753-757
            const int py = (int)roundf((ky[i] - minY) * invH);
            if (px >= 0 && px < cols && py >= 0 && py < rows) key = ((unsigned)(px * rows + py) << 16) | (unsigned)i;
        }
        gkeys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < npow2; i += kGridThreads) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned a = gkeys[i], b = gkeys[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { gkeys[i] = b; gkeys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    // number of keypoints inside the grid = first padding key
    const int nCells = cols * rows;
    for (int c = t; c <= nCells; c += kGridThreads) {
        const unsigned lim = c == nCells ? 0xFFFFFFFFu : ((unsigned)c << 16);   // first key with cell >= c
        int lo = 0, hi = npow2;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (gkeys[mid] < lim) lo = mid + 1; else hi = mid;
        }
        cellStart[c] = lo;
    }
    for (int i = t; i < n; i += kGridThreads) {
        const unsigned key = gkeys[i];
        if (key != 0xFFFFFFFFu) items[i] = (int)(key & 0xFFFFu);
    }
}

// reference src/Frame.cc:1016-1037
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(int n, const float* __restrict__ kx, const float* __restrict__ ky,
                                                          const float* __restrict__ kux, const float* __restrict__ depth, int pitch,
                                                          float mbf, float* __restrict__ uRight, float* __restrict__ outDepth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ur = -1.0f, dz = -1.0f;
    const float d = depth[(size_t)(int)ky[i] * pitch + (int)kx[i]];   // Mat::at<float>(float v, float u): truncation
    if (d > 0) {
        dz = d;
        ur = kux[i] - mbf / d;
    }
    uRight[i] = ur;
    outDepth[i] = dz;
}

struct FCtx {   // per-thread workspace, grow-only
    hipStream_t stream = nullptr;
    eao::DevBuf<unsigned char> dev;
    unsigned char* host = nullptr;   // pinned staging
    size_t hostCap = 0;
    eao_status pin(size_t need) {
        if (need <= hostCap) return EAO_OK;
        if (host) (void)hipHostFree(host);
        host = nullptr; hostCap = 0;
        const size_t cap = need + (need >> 2) + 4096;
        EAO_HIP(hipHostMalloc((void**)&host, cap, hipHostMallocDefault));
        hostCap = cap;
        return EAO_OK;
    }
    eao_status ready() {
        eao_status st = eao::require_device();
        if (st) return st;
        if (!stream) EAO_HIP(eao::create_stream(&stream, eao::StreamClass::Latency));
        return EAO_OK;
    }
    ~FCtx() {
        if (host) (void)hipHostFree(host);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
thread_local FCtx g_fctx;

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

eao_status eao_frame_is_in_frustum(const eao_frustum_frame* Fr, const eao_map_points* pts, float viewing_cos_limit, uint8_t* in_view,
                                   float* proj_x, float* proj_y, float* proj_xr, float* view_cos, int32_t* pred_level) {
    EAO_REQUIRE(Fr && pts && pts->n >= 0, "null argument");
    const int n = pts->n;
    if (n == 0) return EAO_OK;
    EAO_REQUIRE(in_view && proj_x && proj_y && proj_xr && view_cos && pred_level, "null output array");
    EAO_REQUIRE(pts->Xw && pts->normal && pts->min_dist_inv && pts->max_dist_inv && pts->max_dist, "incomplete map-point arrays");
    FCtx& c = g_fctx;
    eao_status st = c.ready();
    if (st) return st;
    // staging: Xw normal (3n floats each) | min max num (n floats each)  ->  outputs: u v ur cos (n floats) | level (n int) | in_view (n)
    const size_t fN = al256(4 * (size_t)n), f3 = al256(12 * (size_t)n);
    const size_t oX = 0, oN = f3, oMin = 2 * f3, oMax = oMin + fN, oNum = oMax + fN, inBytes = oNum + fN;
    const size_t oU = inBytes, oV = oU + fN, oUr = oV + fN, oCos = oUr + fN, oLvl = oCos + fN, oIn = oLvl + fN, total = oIn + al256(n);
    if ((st = c.pin(total))) return st;
    if ((st = c.dev.reserve(total))) return st;
    std::memcpy(c.host + oX, pts->Xw, 12 * (size_t)n);
    std::memcpy(c.host + oN, pts->normal, 12 * (size_t)n);
    std::memcpy(c.host + oMin, pts->min_dist_inv, 4 * (size_t)n);
    std::memcpy(c.host + oMax, pts->max_dist_inv, 4 * (size_t)n);
    std::memcpy(c.host + oNum, pts->max_dist, 4 * (size_t)n);
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, c.host, inBytes, hipMemcpyHostToDevice, s));
    // outputs of points that are not in view stay as the caller left them: seed the device copies with the caller's values
    std::memcpy(c.host + oU, proj_x, 4 * (size_t)n); std::memcpy(c.host + oV, proj_y, 4 * (size_t)n);
    std::memcpy(c.host + oUr, proj_xr, 4 * (size_t)n); std::memcpy(c.host + oCos, view_cos, 4 * (size_t)n);
    std::memcpy(c.host + oLvl, pred_level, 4 * (size_t)n);
    EAO_HIP(hipMemcpyAsync(c.dev.p + oU, c.host + oU, oIn - oU, hipMemcpyHostToDevice, s));
    FrustumArgs A;
    A.n = n;
    A.Xw = (const float*)(c.dev.p + oX); A.normal = (const float*)(c.dev.p + oN); A.minDist = (const float*)(c.dev.p + oMin);
    A.maxDist = (const float*)(c.dev.p + oMax); A.maxDistNum = (const float*)(c.dev.p + oNum);
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) A.R[3 * r + k] = Fr->Tcw[4 * r + k];
        A.t[r] = Fr->Tcw[4 * r + 3];
        A.Ow[r] = Fr->Ow[r];
    }
    A.fx = Fr->fx; A.fy = Fr->fy; A.cx = Fr->cx; A.cy = Fr->cy; A.mbf = Fr->mbf;
    A.minX = Fr->min_x; A.maxX = Fr->max_x; A.minY = Fr->min_y; A.maxY = Fr->max_y;
    A.logScale = Fr->log_scale_factor; A.cosLimit = viewing_cos_limit;
    A.projX = (float*)(c.dev.p + oU); A.projY = (float*)(c.dev.p + oV); A.projXR = (float*)(c.dev.p + oUr);
    A.viewCos = (float*)(c.dev.p + oCos); A.level = (int*)(c.dev.p + oLvl); A.inView = c.dev.p + oIn;
    hipLaunchKernelGGL(k_is_in_frustum, dim3(eao::cdiv(n, 256)), dim3(256), 0, s, A);
    EAO_HIP(hipMemcpyAsync(c.host + oU, c.dev.p + oU, total - oU, hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    std::memcpy(proj_x, c.host + oU, 4 * (size_t)n); std::memcpy(proj_y, c.host + oV, 4 * (size_t)n);
    std::memcpy(proj_xr, c.host + oUr, 4 * (size_t)n); std::memcpy(view_cos, c.host + oCos, 4 * (size_t)n);
    std::memcpy(pred_level, c.host + oLvl, 4 * (size_t)n); std::memcpy(in_view, c.host + oIn, (size_t)n);
    return EAO_OK;
}

eao_status eao_assign_features_to_grid(int32_t n, const float* kp_x, const float* kp_y, float min_x, float min_y, float grid_inv_w,
                                       float grid_inv_h, int32_t cols, int32_t rows, int32_t* cell_start, int32_t* items) {
    EAO_REQUIRE(n >= 0 && cols > 0 && rows > 0 && cell_start, "bad argument");
    EAO_REQUIRE((long long)cols * rows < 65535, "at most 65534 grid cells (cell ids are packed in 16 bits)");
    EAO_REQUIRE(n <= 32768, "at most 32768 keypoints per frame (the grid sort runs in the LDS of one workgroup)");
    const int nCells = cols * rows;
    if (n == 0) { for (int c = 0; c <= nCells; c++) cell_start[c] = 0; return EAO_OK; }
    EAO_REQUIRE(kp_x && kp_y && items, "null keypoint arrays");
    FCtx& c = g_fctx;
    eao_status st = c.ready();
    if (st) return st;
    int npow2 = 64;
    while (npow2 < n) npow2 <<= 1;
    const size_t fN = al256(4 * (size_t)n), oX = 0, oY = fN, oS = 2 * fN, oI = oS + al256(4 * (size_t)(nCells + 1)), total = oI + fN;
    if ((st = c.pin(total))) return st;
    if ((st = c.dev.reserve(total))) return st;
    std::memcpy(c.host + oX, kp_x, 4 * (size_t)n); std::memcpy(c.host + oY, kp_y, 4 * (size_t)n);
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, c.host, oS, hipMemcpyHostToDevice, s));
    const size_t lds = (size_t)npow2 * sizeof(unsigned);
    EAO_HIP(hipFuncSetAttribute((const void*)k_grid_assign, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_grid_assign, dim3(1), dim3(kGridThreads), lds, s, n, (const float*)(c.dev.p + oX), (const float*)(c.dev.p + oY),
                       min_x, min_y, grid_inv_w, grid_inv_h, cols, rows, npow2, (int*)(c.dev.p + oS), (int*)(c.dev.p + oI));
    EAO_HIP(hipMemcpyAsync(c.host + oS, c.dev.p + oS, total - oS, hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    std::memcpy(cell_start, c.host + oS, 4 * (size_t)(nCells + 1));
    std::memcpy(items, c.host + oI, 4 * (size_t)cell_start[nCells]);
    return EAO_OK;
}

eao_status eao_compute_stereo_from_rgbd(int32_t n, const float* kp_x, const float* kp_y, const float* kpu_x, const float* depth,
                                        int32_t width, int32_t height, int32_t pitch, int32_t depth_on_device, float mbf,
                                        float* u_right, float* out_depth) {
    EAO_REQUIRE(n >= 0 && width > 0 && height > 0 && pitch >= width, "bad argument");
    if (n == 0) return EAO_OK;
    EAO_REQUIRE(kp_x && kp_y && kpu_x && depth && u_right && out_depth, "null argument");
    for (int i = 0; i < n; i++)   // cv::Mat::at does not check either: a keypoint outside the image is a caller bug, refuse it
        EAO_REQUIRE((int)kp_x[i] >= 0 && (int)kp_x[i] < width && (int)kp_y[i] >= 0 && (int)kp_y[i] < height, "keypoint %d outside the depth image", i);
    FCtx& c = g_fctx;
    eao_status st = c.ready();
    if (st) return st;
    const size_t fN = al256(4 * (size_t)n), img = depth_on_device ? 0 : al256(4 * (size_t)pitch * height);
    const size_t oX = 0, oY = fN, oU = 2 * fN, oD = 3 * fN, inBytes = oD + img, oUr = inBytes, oZ = oUr + fN, total = oZ + fN;
    if ((st = c.pin(total))) return st;
    if ((st = c.dev.reserve(total))) return st;
    std::memcpy(c.host + oX, kp_x, 4 * (size_t)n); std::memcpy(c.host + oY, kp_y, 4 * (size_t)n); std::memcpy(c.host + oU, kpu_x, 4 * (size_t)n);
    if (!depth_on_device) std::memcpy(c.host + oD, depth, 4 * (size_t)pitch * height);
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, c.host, inBytes, hipMemcpyHostToDevice, s));
    const float* dDepth = depth_on_device ? depth : (const float*)(c.dev.p + oD);
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3(eao::cdiv(n, 256)), dim3(256), 0, s, n, (const float*)(c.dev.p + oX), (const float*)(c.dev.p + oY),
                       (const float*)(c.dev.p + oU), dDepth, pitch, mbf, (float*)(c.dev.p + oUr), (float*)(c.dev.p + oZ));
    EAO_HIP(hipMemcpyAsync(c.host + oUr, c.dev.p + oUr, total - oUr, hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    std::memcpy(u_right, c.host + oUr, 4 * (size_t)n); std::memcpy(out_depth, c.host + oZ, 4 * (size_t)n);
    return EAO_OK;
}

eao_status eao_undistort_keypoints(int32_t n, const float* kp_x, const float* kp_y, float fx, float fy, float cx, float cy,
                                   const float* dist_coef, int32_t n_coef, float* out_x, float* out_y) {
    EAO_REQUIRE(n >= 0 && n_coef >= 0 && n_coef <= 5 && (n_coef == 0 || dist_coef) && fx != 0 && fy != 0, "bad argument");
    if (n == 0) return EAO_OK;
    EAO_REQUIRE(kp_x && kp_y && out_x && out_y, "null argument");
    eao::frame::Distortion D;
    eao::frame::fill_distortion(D, fx, fy, cx, cy, dist_coef, n_coef);
    FCtx& c = g_fctx;
    eao_status st = c.ready();
    if (st) return st;
    const size_t fN = al256(4 * (size_t)n), total = 4 * fN;
    if ((st = c.pin(total))) return st;
    if ((st = c.dev.reserve(total))) return st;
    std::memcpy(c.host, kp_x, 4 * (size_t)n); std::memcpy(c.host + fN, kp_y, 4 * (size_t)n);
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, c.host, 2 * fN, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_undistort, dim3(eao::cdiv(n, 256)), dim3(256), 0, s, D, n, (const float*)c.dev.p, (const float*)(c.dev.p + fN), (float*)(c.dev.p + 2 * fN),
                       (float*)(c.dev.p + 3 * fN));
    EAO_HIP(hipMemcpyAsync(c.host + 2 * fN, c.dev.p + 2 * fN, 2 * fN, hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    std::memcpy(out_x, c.host + 2 * fN, 4 * (size_t)n); std::memcpy(out_y, c.host + 3 * fN, 4 * (size_t)n);
    return EAO_OK;
}

eao_status eao_compute_image_bounds(int32_t cols, int32_t rows, float fx, float fy, float cx, float cy, const float* dist_coef, int32_t n_coef,
                                    float* bounds) {
    EAO_REQUIRE(cols > 0 && rows > 0 && bounds && n_coef >= 0 && n_coef <= 5 && (n_coef == 0 || dist_coef), "bad argument");
    if (!(n_coef >= 1 && dist_coef[0] != 0.0f)) {      // src/Frame.cc:836-841
        bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
        return EAO_OK;
    }
    // the four corners (0, 0), (cols, 0), (0, rows), (cols, rows) through the same kernel as the keypoints (:812-821)
    const float x[4] = {0.0f, (float)cols, 0.0f, (float)cols}, y[4] = {0.0f, 0.0f, (float)rows, (float)rows};
    float ux[4], uy[4];
    eao_status st = eao_undistort_keypoints(4, x, y, fx, fy, cx, cy, dist_coef, n_coef, ux, uy);
    if (st) return st;
    float minX = std::min(ux[0], ux[2]), maxX = std::max(ux[1], ux[3]), minY = std::min(uy[0], uy[1]), maxY = std::max(uy[2], uy[3]);      // :823-826
    bounds[0] = std::max(minX, 0.0f); bounds[1] = std::min(maxX, (float)cols); bounds[2] = std::max(minY, 0.0f); bounds[3] = std::min(maxY, (float)rows);   // :829-832
    return EAO_OK;
}

}  // extern "C"

eao_status eao::frame::enqueue_undistort_device(const Distortion& D, int n, const float* dx, const float* dy, float* ox, float* oy, hipStream_t s) {
    if (n <= 0) return EAO_OK;
    hipLaunchKernelGGL(k_undistort, dim3(eao::cdiv(n, 256)), dim3(256), 0, s, D, n, dx, dy, ox, oy);
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

// hook for the device-resident tracking chain (csrc/track.hip): the same kernel over arrays that already live on the device
#include "chain_internal.h"
eao_status eao::frame::enqueue_frustum_device(const FrustumDevArgs& a, hipStream_t s) {
    if (a.n <= 0) return EAO_OK;
    FrustumArgs A;
    fill_frustum_args(a, A);
    hipLaunchKernelGGL(k_is_in_frustum, dim3(eao::cdiv(a.n, 256)), dim3(256), 0, s, A);
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

