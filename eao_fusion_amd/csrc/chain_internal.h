// chain_internal.h -- hooks between the translation units for the device-resident tracking chain (csrc/track.hip); not part
// of the C-ABI.
#pragma once
#include <cstddef>

#include "common.h"
#include "match_internal.h"

namespace eao {

namespace lm {
struct PoseChainArgs {            // every pointer is a device address (outSE3 / outResult / outTrace may be mapped host memory)
    const int* nEdges; int cap;   // number of edges (device) and the capacity of the arrays
    const double* Xw; const double* obs; const double* info; unsigned char* flags; double* err; unsigned char* outlier;
    float Tcw0[16]; float fx, fy, cx, cy, bf;
    void* outSE3; int* outResult; double* outTrace;
};
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
}  // namespace frame

namespace match {
struct FrameDevArgs {             // a frame in the layout k_match_candidates walks, device-resident
    int cap;                      // capacity of the per-keypoint arrays
    const int* nOrdered;          // device: number of keypoints inside the grid (= entries of order / cellx / celly)
    const float* kx; const float* ky; const int* oct; const float* ur; const uint8_t* desc;
    const int* order; const unsigned short* cellx; const unsigned short* celly;
    float minX, minY, invW, invH; int cols, rows;
};
// candidate lists of nq device-resident queries: out (packed distance << 16 | keypoint), segStart / segCount per query,
// cursor (zeroed here)
eao_status enqueue_candidates_device(const FrameDevArgs& F, const Query* q, const uint8_t* qdesc, int nq, unsigned* out, int outCap,
                                     int* segStart, int* segCount, int* cursor, hipStream_t s, bool cursorIsZero = false);
}  // namespace match

}  // namespace eao
