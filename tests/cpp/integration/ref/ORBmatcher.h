// Test infrastructure only: what a translation unit that includes the reference's matcher header sees of class ORB_SLAM2::ORBmatcher -- its include
// guard, its namespace and the C++ signatures of the members the INTEGRATION.md snippet src/ORBmatcher_hip.cc defines (written from the call sites in
// src/Tracking.cc, src/LocalMapping.cc, src/LoopClosing.cc and the definitions in src/ORBmatcher.cc:41-43, 45, 159, 290, 405, 522, 657, 825, 977, 1102, 1328,
// 1474, 1649; parameter names left out, container types through aliases).  The snippet includes this header AND <eaofusion/ORBmatcher.h>, which is the
// situation it meets in a checkout.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H
#include <set>
#include <utility>
#include <vector>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class ORBmatcher {
    typedef std::vector<MapPoint*> Points;
    typedef std::vector<std::pair<size_t, size_t> > IndexPairs;

public:
    // thresholds on the descriptor distance and the rotation histogram's size (values in the .cc)
    static const int TH_HIGH, TH_LOW, HISTO_LENGTH;
    static int DescriptorDistance(const cv::Mat&, const cv::Mat&);
    ORBmatcher(float = 0.6, bool = true);

    // duplicate fusion (local mapping / loop correction)
    int Fuse(KeyFrame*, cv::Mat, const Points&, float, Points&);
    int Fuse(KeyFrame*, const Points&, const float = 3.0);
    // loop closing, new map points, monocular initialisation
    int SearchBySim3(KeyFrame*, KeyFrame*, Points&, const float&, const cv::Mat&, const cv::Mat&, const float);
    int SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat, IndexPairs&, const bool);
    int SearchForInitialization(Frame&, Frame&, std::vector<cv::Point2f>&, std::vector<int>&, int = 10);
    // vocabulary-guided matching
    int SearchByBoW(KeyFrame*, KeyFrame*, Points&);
    int SearchByBoW(KeyFrame*, Frame&, Points&);
    // projection-guided matching: loop detection, relocalisation, frame to frame, local map
    int SearchByProjection(KeyFrame*, cv::Mat, const Points&, Points&, int);
    int SearchByProjection(Frame&, KeyFrame*, const std::set<MapPoint*>&, const float, const int);
    int SearchByProjection(Frame&, const Frame&, const float, const bool);
    int SearchByProjection(Frame&, const Points&, const float = 3);

protected:
    float mfNNratio;            // the two constructor arguments
    bool mbCheckOrientation;
};
}
#endif
