// Stand-in for the reference's include/Frame.h (test infrastructure, see MapPoint.h here).  The member functions whose bodies
// INTEGRATION.md replaces are declared here and DEFINED by the snippets.
#ifndef FRAME_H
#define FRAME_H

#include <vector>

#include "KeyFrame.h"
#include "MapPlane.h"
#include "MapPoint.h"
#include "ORBextractor.h"      // edit 1 of INTEGRATION.md: this file is the one-line include of <eaofusion/ORBextractor.h>

namespace ORB_SLAM2 {
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

class Frame {
public:
    void SetPose(cv::Mat Tcw) {
        mTcw = Tcw.clone();
        mRcw = cv::Mat(3, 3, CV_32F); mtcw = cv::Mat(3, 1, CV_32F); mOw = cv::Mat(3, 1, CV_32F);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) mRcw.at<float>(r, c) = mTcw.at<float>(r, c); mtcw.at<float>(r) = mTcw.at<float>(r, 3); }
        for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s -= (double)mRcw.at<float>(k, r) * (double)mtcw.at<float>(k); mOw.at<float>(r) = (float)s; }
    }
    void ComputeStereoMatches();                        // src/Frame.cc:841  -> snippet src/Frame_hip.cc
    void ComputeStereoFromRGBD(const cv::Mat& imDepth); // src/Frame.cc:1016 -> snippet
    void TestAssignFeaturesToGrid() { AssignFeaturesToGrid(); }   // upstream's constructors call the private member

    ORBextractor *mpORBextractorLeft = nullptr, *mpORBextractorRight = nullptr;
    static float fx, fy, cx, cy, invfx, invfy;
    float mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    long unsigned int mnId = 0;
    int mnScaleLevels = 8;
    float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    // planes (this fork)
    int mnPlaneNum = 0;
    std::vector<MapPlane*> mvpMapPlanes;
    std::vector<cv::Mat> mvPlaneCoefficients;
    std::vector<bool> mvbPlaneOutlier;
    inline cv::Mat GetCameraCenter() { return mOw.clone(); }

private:
    void AssignFeaturesToGrid();                        // src/Frame.cc:599  -> snippet
    cv::Mat mRcw, mtcw, mOw;       // private upstream too (include/Frame.h:322-339): the snippets must not need them
};

}  // namespace ORB_SLAM2
#endif  // FRAME_H
