// Stand-in for the reference's include/KeyFrame.h (test infrastructure, see MapPoint.h here).  DBoW2::FeatureVector is a
// std::map<unsigned, std::vector<unsigned>> upstream (Thirdparty/DBoW2/DBoW2/FeatureVector.h) and is declared as that here.
#ifndef KEYFRAME_H
#define KEYFRAME_H

#include <map>
#include <set>
#include <vector>

#include "MapPoint.h"

namespace DBoW2 { typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector; }

namespace ORB_SLAM2 {

class KeyFrame {
public:
    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { Tcw = T.clone(); }
    cv::Mat GetRotation() { return sub(0, 0, 3, 3); }
    cv::Mat GetTranslation() { return sub(0, 3, 3, 1); }
    cv::Mat GetCameraCenter() {
        cv::Mat Ow(3, 1, CV_32F);
        for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s -= (double)Tcw.at<float>(k, r) * (double)Tcw.at<float>(k, 3); Ow.at<float>(r) = (float)s; }
        return Ow;
    }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(MapPoint* pMP) { for (MapPoint*& p : mvpMapPoints) if (p == pMP) { p = nullptr; erased++; } }
    bool isBad() { return false; }

    long unsigned int mnId = 0;
    long unsigned int mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    cv::Mat mTcwGBA;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 640, mnMaxY = 480, mnGridCols = 64, mnGridRows = 48;
    float mfGridElementWidthInv = 64.f / 640.f, mfGridElementHeightInv = 48.f / 480.f;
    std::vector<cv::Mat> mvPlaneCoefficients;

    // test bookkeeping
    cv::Mat Tcw;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<KeyFrame*> covisible;
    int erased = 0;

private:
    cv::Mat sub(int r0, int c0, int nr, int nc) {
        cv::Mat m(nr, nc, CV_32F);
        for (int r = 0; r < nr; r++) for (int c = 0; c < nc; c++) m.at<float>(r, c) = Tcw.at<float>(r0 + r, c0 + c);
        return m;
    }
};

}  // namespace ORB_SLAM2
#endif  // KEYFRAME_H
