// Stand-in for the part of the reference's include/Tracking.h that Tracking::SearchLocalPoints touches (src/Tracking.cc:2587-2641).
#ifndef TRACKING_H
#define TRACKING_H

#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBmatcher.h"
#include "Optimizer.h"

#include <eaofusion/DeviceTracker.h>

namespace ORB_SLAM2 {
struct System { enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 }; };
class Tracking {
public:
    void SearchLocalPoints();      // body = upstream's first loop (unchanged) + the INTEGRATION.md fragment
    // the body is the fragment src/Tracking_TrackLocalMap.inc; the arguments stand for the device
    // buffers a maintainer keeps beside the extractor
    void TrackLocalMapOnDevice(const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, const float* d_depth, int depthPitch, int width,
                               int height, void* stream);
    // ... and src/Tracking_TrackWithMotionModel.inc (round 4): the search + pose optimisation + outlier discard of Tracking::TrackWithMotionModel
    bool TrackWithMotionModelOnDevice(const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, const float* d_depth, int depthPitch, int width,
                                      int height, void* stream, const cv::Mat& predictedPose);
    // ... and src/Tracking_TrackReferenceKeyFrame.inc (round 4): everything of Tracking::TrackReferenceKeyFrame behind mCurrentFrame.ComputeBoW()
    bool TrackReferenceKeyFrameOnDevice(const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, const float* d_depth, int depthPitch, int width,
                                        int height, void* stream);
    Frame mLastFrame;
    Map* mpMap = nullptr;
    KeyFrame* mpReferenceKF = nullptr;
    eaofusion::DeviceTracker* mpDeviceTracker = nullptr;
    int mSensor = System::RGBD;
    Frame mCurrentFrame;
    std::vector<MapPoint*> mvpLocalMapPoints;
    unsigned int mnLastRelocFrameId = 0;
    int nToMatchSeen = -1;         // test bookkeeping
};
}  // namespace ORB_SLAM2
#endif  // TRACKING_H
