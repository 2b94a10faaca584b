// Stand-in for the reference's include/Map.h (test infrastructure, see MapPoint.h here).
#ifndef MAP_H
#define MAP_H

#include <mutex>
#include <vector>

#include "KeyFrame.h"
#include "MapPlane.h"
#include "MapPoint.h"

namespace ORB_SLAM2 {
class Frame;
class Map {
public:
    void AssociatePlanesByBoundary(Frame &pF, bool out=false);      // reference include/Map.h:93 (defined by the test: the association itself is outside the path)
    std::vector<KeyFrame*> GetAllKeyFrames() { return kfs; }
    std::vector<MapPoint*> GetAllMapPoints() { return mps; }
    std::vector<MapPlane*> GetAllMapPlanes() { return planes; }
    std::mutex mMutexMapUpdate;
    std::vector<KeyFrame*> kfs;
    std::vector<MapPoint*> mps;
    std::vector<MapPlane*> planes;
};
}  // namespace ORB_SLAM2
#endif  // MAP_H
