// Stand-in for the reference's include/MapPoint.h (same include guard, same namespace, the members the drop-in snippets of
// INTEGRATION.md and the adapters of include/eaofusion/ touch).  Test infrastructure: written for tests/test_integration_snippets.py,
// not part of the product and not a copy of the reference class -- bodies are the simplest thing that keeps the data.
#ifndef MAPPOINT_H
#define MAPPOINT_H

#include <map>
#include <mutex>
#include <vector>

#include <eaofusion/cv_compat.h>

namespace ORB_SLAM2 {

class KeyFrame;
class Map;
class Frame;

class MapPoint {
public:
    MapPoint() {}
    MapPoint(const MapPoint& o)   // mutexes are not copyable; the tests copy scenes
        : mnId(o.mnId), mnBALocalForKF(o.mnBALocalForKF), mnBAGlobalForKF(o.mnBAGlobalForKF), mPosGBA(o.mPosGBA), mTrackProjX(o.mTrackProjX),
          mTrackProjY(o.mTrackProjY), mTrackProjXR(o.mTrackProjXR), mbTrackInView(o.mbTrackInView), mnTrackScaleLevel(o.mnTrackScaleLevel),
          mTrackViewCos(o.mTrackViewCos), mnLastFrameSeen(o.mnLastFrameSeen), nVisible(o.nVisible), normalUpdates(o.normalUpdates),
          replacedBy(o.replacedBy), mWorldPos(o.mWorldPos), mNormalVector(o.mNormalVector), mDescriptor(o.mDescriptor),
          mObservations(o.mObservations), mbBad(o.mbBad), mfMinDistance(o.mfMinDistance), mfMaxDistance(o.mfMaxDistance) {}
    MapPoint& operator=(const MapPoint&) = delete;

    void SetWorldPos(const cv::Mat& Pos) { mWorldPos = Pos.clone(); }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    std::map<KeyFrame*, size_t> GetObservations() { return mObservations; }
    int Observations() { return (int)mObservations.size(); }
    void AddObservation(KeyFrame* pKF, size_t idx) { mObservations[pKF] = idx; }
    void EraseObservation(KeyFrame* pKF) { mObservations.erase(pKF); }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    bool isBad() { return mbBad; }
    void Replace(MapPoint* pMP) { mbBad = true; replacedBy = pMP; }
    void IncreaseVisible(int n = 1) { nVisible += n; }
    void ComputeDistinctiveDescriptors();      // defined by the INTEGRATION.md snippet src/MapPoint_hip.cc
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    void UpdateNormalAndDepth() { normalUpdates++; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    // the two allocation-free accessors, extracted verbatim from the snippet when the scratch checkout holds it
#if defined(__has_include)
#if __has_include("MapPoint_accessors.inc")
#include "MapPoint_accessors.inc"
#define EAO_TEST_MAPPOINT_HAS_ACCESSORS 1
#endif
#endif

    long unsigned int mnId = 0;
    long unsigned int mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0;
    cv::Mat mPosGBA;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    long unsigned int mnLastFrameSeen = ~0ul;
    static std::mutex mGlobalMutex;

    // test bookkeeping
    int nVisible = 0, normalUpdates = 0;
    MapPoint* replacedBy = nullptr;
    void TestSet(const cv::Mat& pos, const cv::Mat& normal, const cv::Mat& desc, float dmin, float dmax) {
        mWorldPos = pos.clone(); mNormalVector = normal.clone(); mDescriptor = desc.clone(); mfMinDistance = dmin; mfMaxDistance = dmax;
    }
    void TestSetBad(bool b) { mbBad = b; }

protected:
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame*, size_t> mObservations;
    bool mbBad = false;
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::mutex mMutexPos, mMutexFeatures;
};

}  // namespace ORB_SLAM2
#endif  // MAPPOINT_H
