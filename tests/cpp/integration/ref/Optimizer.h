// Stand-in for the reference's include/Optimizer.h: same guard, namespace and the declarations of the statics the snippets
// define (reference include/Optimizer.h:50-56).  The g2o-bound statics (OptimizeEssentialGraph, OptimizeSim3) are out of scope
// and left out.  Test infrastructure only.
#ifndef OPTIMIZER_H
#define OPTIMIZER_H

#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"

namespace ORB_SLAM2 {
class Optimizer {
public:
    void static BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, const std::vector<MapPlane*>& vpMPl,
                                 int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true);
    void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);
    int static PoseOptimization(Frame* pFrame);
};
}  // namespace ORB_SLAM2
#endif  // OPTIMIZER_H
