// chain_internal.h -- hooks between the translation units for the device-resident tracking chain (csrc/track.hip); not part
// of the C-ABI.
#pragma once
#include <cstddef>

#include "common.h"
#include "match_internal.h"

namespace eao {

namespace lm {
struct PoseChainArgs {            // every pointer is a device address
    const int* nEdges; int cap;   // number of edges (device) and the capacity of the arrays
    int maxEdges = 0;             // what the host knows about the count (0: nothing beyond cap): launches for larger frames are left out
    int* done = nullptr; int doneSeq = 0;   // mapped host word that receives doneSeq when the chain's last launch has made its results visible
    const void* pubSrc = nullptr; void* pubDst = nullptr; int pubN16 = 0;   // ... after copying pubN16 x 16 bytes from the device block pubSrc (which every
                                                                            // kernel of the chain writes) to its mapped host twin pubDst: one writer of host memory
    const double* Xw; const double* obs; const double* info; unsigned char* flags; double* err; unsigned char* outlier;
    float Tcw0[16]; float fx, fy, cx, cy, bf;
    void* outSE3; int* outResult; double* outTrace;
    const int* scatterIdx; unsigned char* scatterOut;   // optional: the final outlier flag of edge e also goes to scatterOut[scatterIdx[e]] (mvbOutlier by keypoint)
    // plane edges of the frame (src/Optimizer.cc:456-535), round 5: nPlanes records of 10 doubles as pose_plane_records writes them (device-visible memory),
    // their outlier flags (pFrame->mvbPlaneOutlier) into planeOutlier (a slice of the chain's device result block)
    int nPlanes = 0; const double* planes = nullptr; unsigned char* planeOutlier = nullptr;
};
// world / measured coefficients normalised as Converter::toPlane3D does and the two information values of every plane edge (10 doubles per plane)
void pose_plane_records(int n, const float* plane_world, const float* plane_obs, const unsigned char* plane_seen, double* rec);
constexpr int kPoseChainMaxPlanes = 32;
size_t pose_se3_bytes();
void pose_se3_to_Tcw(const void* se3, float* T);       // Converter::toCvMat(SE3Quat)
eao_status enqueue_pose_device(const PoseChainArgs& a, hipStream_t s);
}  // namespace lm

namespace frame {
struct FrustumDevArgs {           // Frame::isInFrustum over device-resident map points; outputs device-resident
    int n;
    const float* Xw; const float* normal; const float* minDist; const float* maxDist; const float* maxDistNum;
    float Tcw[16], Ow[3];
    float fx, fy, cx, cy, mbf, minX, maxX, minY, maxY, logScale, cosLimit;
    unsigned char* inView; float* projX; float* projY; float* projXR; float* viewCos; int* level;
};
eao_status enqueue_frustum_device(const FrustumDevArgs& a, hipStream_t s);

// Frame::UndistortKeyPoints (src/Frame.cc:773-806) = cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) of OpenCV 3.x
// (modules/imgproc/src/undistort.cpp, cvUndistortPoints; the reference asks for "OpenCV 3.0" without pinning a release): every point is
// normalised with K in double, the distortion model (k1 k2 p1 p2 [k3]) is inverted by FIVE fixed-point iterations, and the result is
// projected with P = K and rounded to float.  The zero tilt / identity R / P = K factors of the library multiply by exact ones and
// zeros and are left out.  Shared by k_undistort (csrc/frame.hip) and the tracker's frame set-up (csrc/track.hip).
struct Distortion {
    int on;                       // 0: mDistCoef.at<float>(0) == 0.0 -- upstream copies mvKeys (src/Frame.cc:775-779)
    double fx, fy, cx, cy, ifx, ify;
    double k1, k2, p1, p2, k3;
};
inline void fill_distortion(Distortion& D, float fx, float fy, float cx, float cy, const float* dist, int nCoef) {
    D.on = dist && nCoef >= 1 && dist[0] != 0.0f ? 1 : 0;      // (coefficients the caller did not pass read as zero below -- a one- to three-coefficient camera is distorted too)
    D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy; D.ifx = 1. / (double)fx; D.ify = 1. / (double)fy;
    D.k1 = dist && nCoef > 0 ? dist[0] : 0; D.k2 = dist && nCoef > 1 ? dist[1] : 0; D.p1 = dist && nCoef > 2 ? dist[2] : 0; D.p2 = dist && nCoef > 3 ? dist[3] : 0;
    D.k3 = dist && nCoef > 4 ? dist[4] : 0;
}
__host__ __device__ __forceinline__ void undistort_point(const Distortion& D, float u, float v, float& uo, float& vo) {
    double x = ((double)u - D.cx) * D.ifx, y = ((double)v - D.cy) * D.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((D.k3 * r2 + D.k2) * r2 + D.k1) * r2);      // (k4 = k5 = k6 = 0: the rational model's numerator 1 + ((k6 r2 + k5) r2 + k4) r2 is exactly 1)
        const double deltaX = 2 * D.p1 * x * y + D.p2 * (r2 + 2 * x * x);      // (+ s1 r2 + s2 r2^2 with s = 0: adds exact zeros)
        const double deltaY = D.p1 * (r2 + 2 * y * y) + 2 * D.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    uo = (float)(D.fx * x + D.cx);
    vo = (float)(D.fy * y + D.cy);
}
eao_status enqueue_undistort_device(const Distortion& D, int n, const float* dx, const float* dy, float* ox, float* oy, hipStream_t s);

// The kernel-side view of the same test, shared by k_is_in_frustum (csrc/frame.hip) and by the tracker's first launch
// (csrc/track.hip), whose extra workgroups run it beside the single-workgroup frame set-up.
struct FrustumArgs {
    int n;
    const float* Xw; const float* normal; const float* minDist; const float* maxDist; const float* maxDistNum;
    float R[9], t[3], Ow[3];
    float fx, fy, cx, cy, mbf, minX, maxX, minY, maxY, logScale, cosLimit;
    unsigned char* inView; float* projX; float* projY; float* projXR; float* viewCos; int* level;
};
inline void fill_frustum_args(const FrustumDevArgs& a, FrustumArgs& A) {
    A.n = a.n;
    A.Xw = a.Xw; A.normal = a.normal; A.minDist = a.minDist; A.maxDist = a.maxDist; A.maxDistNum = a.maxDistNum;
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) A.R[3 * r + k] = a.Tcw[4 * r + k];
        A.t[r] = a.Tcw[4 * r + 3];
        A.Ow[r] = a.Ow[r];
    }
    A.fx = a.fx; A.fy = a.fy; A.cx = a.cx; A.cy = a.cy; A.mbf = a.mbf;
    A.minX = a.minX; A.maxX = a.maxX; A.minY = a.minY; A.maxY = a.maxY; A.logScale = a.logScale; A.cosLimit = a.cosLimit;
    A.inView = a.inView; A.projX = a.projX; A.projY = a.projY; A.projXR = a.projXR; A.viewCos = a.viewCos; A.level = a.level;
}
// reference src/Frame.cc:638-695 for map point i; the statements keep upstream's order (each early return of upstream is a `return` here)
__device__ __forceinline__ void frustum_point(const FrustumArgs& A, int i) {
    A.inView[i] = 0;
    const float P0 = A.Xw[3 * i], P1 = A.Xw[3 * i + 1], P2 = A.Xw[3 * i + 2];
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double acc = (double)A.R[3 * r] * (double)P0 + (double)A.R[3 * r + 1] * (double)P1 + (double)A.R[3 * r + 2] * (double)P2;
        Pc[r] = (float)(acc + (double)A.t[r]);
    }
    if (Pc[2] < 0.0f) return;
    const float invz = 1.0f / Pc[2];
    const float u = A.fx * Pc[0] * invz + A.cx;
    const float v = A.fy * Pc[1] * invz + A.cy;
    if (u < A.minX || u > A.maxX) return;
    if (v < A.minY || v > A.maxY) return;
    const float PO0 = P0 - A.Ow[0], PO1 = P1 - A.Ow[1], PO2 = P2 - A.Ow[2];
    const float dist = (float)sqrt((double)PO0 * PO0 + (double)PO1 * PO1 + (double)PO2 * PO2);
    if (dist < A.minDist[i] || dist > A.maxDist[i]) return;
    const double dot = (double)PO0 * A.normal[3 * i] + (double)PO1 * A.normal[3 * i + 1] + (double)PO2 * A.normal[3 * i + 2];
    const float viewCos = (float)(dot / (double)dist);
    if (viewCos < A.cosLimit) return;
    // MapPoint::PredictScale (src/MapPoint.cc:385-394): float log, float division, ceil.  The float logarithm is taken
    // as the double logarithm rounded to float (correctly rounded but for ~2^-29 of the inputs; a host libm's logf may
    // differ from that by one ulp on rare inputs, which only matters when the quotient sits on an integer).
    const float ratio = A.maxDistNum[i] / dist;
    const float lg = (float)log((double)ratio);
    const int level = (int)ceilf(lg / A.logScale);
    A.inView[i] = 1;
    A.projX[i] = u;
    A.projXR[i] = u - A.mbf * invz;
    A.projY[i] = v;
    A.level[i] = level;
    A.viewCos[i] = viewCos;
}
}  // namespace frame

namespace match {
struct FrameDevArgs {             // a frame in the layout k_match_candidates walks, device-resident
    int cap;                      // capacity of the per-keypoint arrays
    const int* nOrdered;          // device: number of keypoints inside the grid (= entries of order / cellx / celly)
    const float* kx; const float* ky; const int* oct; const float* ur; const uint8_t* desc;
    const int* order; const unsigned short* cellx; const unsigned short* celly;
    float minX, minY, invW, invH; int cols, rows;
    const int* colStart;          // optional, cols + 1 entries: first ordered entry of every grid column (a query then walks its columns only)
};
// When handed to enqueue_candidates_device, the wave of query m BUILDS the query itself -- the search window of local map point m
// (Tracking::SearchLocalPoints / ORBmatcher::SearchByProjection, src/ORBmatcher.cc:45-137: RadiusByViewingCos x th x scale factor
// of the predicted level) from the tracker's per-point arrays -- and stores it in qOut[m] for the assignment step; q is then unused.
struct QueryBuild {
    const unsigned char* active; const unsigned char* skip; const unsigned char* inView;
    const float* projX; const float* projY; const float* projXR; const float* viewCos; const int* level; const float* scale;
    int nlevels; float th;
    int* errFlags;            // bit 0: a point in view has a predicted level outside the pyramid
    Query* qOut;
};
// the search windows of SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1339-1393) for every last-frame keypoint; q: n_last entries
struct FrameQueryArgs {
    const float* Tcw; const float* Tlw; int n_last; const uint8_t* valid; const float* Xw; const int32_t* last_octave;
    float fx, fy, cx, cy, mbf, mb, th; int mono;
    float min_x, max_x, min_y, max_y; const float* scale_factors; int nlevels;
};
eao_status build_frame_queries(const FrameQueryArgs& A, Query* q);
// candidate lists of nq device-resident queries: out (packed distance << 16 | keypoint), segStart / segCount per query,
// cursor (zeroed here)
eao_status enqueue_candidates_device(const FrameDevArgs& F, const Query* q, const uint8_t* qdesc, int nq, unsigned* out, int outCap,
                                     int* segStart, int* segCount, int* cursor, hipStream_t s, bool cursorIsZero = false,
                                     const QueryBuild* build = nullptr);
}  // namespace match

}  // namespace eao
