// Stand-in for the reference's include/MapPlane.h (test infrastructure, see MapPoint.h here).
#ifndef MAPPLANE_H
#define MAPPLANE_H

#include <map>
#include <mutex>

#include <eaofusion/cv_compat.h>

namespace ORB_SLAM2 {
class KeyFrame;
class MapPlane {
public:
    MapPlane() {}
    MapPlane(const MapPlane& o) : mnId(o.mnId), mnBAGlobalForKF(o.mnBAGlobalForKF), mPosGBA(o.mPosGBA), mbSeen(o.mbSeen), world(o.world), obs(o.obs), bad(o.bad) {}
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return world.clone(); }
    void SetWorldPos(const cv::Mat& p) { world = p.clone(); }
    std::map<KeyFrame*, int> GetObservations() { return obs; }
    long unsigned int mnId = 0, mnBAGlobalForKF = 0;
    cv::Mat mPosGBA;
    bool mbSeen = false;
    static std::mutex mGlobalMutex;
    cv::Mat world;
    std::map<KeyFrame*, int> obs;
    bool bad = false;
};
}  // namespace ORB_SLAM2
#endif  // MAPPLANE_H
