// include/eaofusion/Frame.h -- drop-in bodies for the Frame glue either side of the matcher, over
// libeaofusion_hip.so (include/eao_fusion.h).  Each template replaces the body of one reference member function:
//
//   Tracking::SearchLocalPoints, projection loop      src/Tracking.cc:2612-2626   ->  eaofusion::IsInFrustum(frame, points, 0.5f, skip)
//     (Frame::isInFrustum                             src/Frame.cc:638-695)
//   Frame::AssignFeaturesToGrid                       src/Frame.cc:597-614        ->  eaofusion::AssignFeaturesToGrid(*this)
//   Frame::ComputeStereoFromRGBD                      src/Frame.cc:1016-1037      ->  eaofusion::ComputeStereoFromRGBD(*this, imDepth)
//
// The templates only name members the reference's Frame / MapPoint already have (mTcw, GetCameraCenter(), fx .. mbf, mnMinX ..,
// mfLogScaleFactor, mGrid, mvKeys, mvKeysUn, mvuRight, mvDepth; GetWorldPos, GetNormal, Get{Min,Max}DistanceInvariance,
// mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos).  Nothing here is copied from the reference.
#ifndef EAOFUSION_FRAME_H
#define EAOFUSION_FRAME_H

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../eao_fusion.h"
#include "cv_compat.h"

namespace eaofusion {
namespace detail {
// mfMaxDistance is what MapPoint::PredictScale divides (src/MapPoint.cc:385-394); upstream exposes it only through
// GetMaxDistanceInvariance() = 1.2f * mfMaxDistance, which does not round-trip in float.  A derived class may name the
// protected member of its base and the resulting pointer-to-member applies to any MapPoint: an exact read without touching
// MapPoint.h.
template <class MapPointT>
struct MaxDistanceOf : MapPointT {
    static float get(MapPointT* p) { return p->*(&MaxDistanceOf::mfMaxDistance); }
};
inline void check(eao_status st, const char* what) {
    if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
}
// mDistCoef of the frame as the C-ABI takes it (k1, k2, p1, p2[, k3]: src/Tracking.cc:90-101); a Frame class without the member (a stand-in) is distortion-free.
template <class FrameT, class = void>
struct DistCoefOf { static int get(const FrameT&, float*) { return 0; } };
template <class FrameT>
struct DistCoefOf<FrameT, decltype(void(std::declval<const FrameT&>().mDistCoef))> {
    static int get(const FrameT& F, float* out) {
        const cv::Mat& d = F.mDistCoef;
        if (d.empty()) return 0;
        const int n = std::min(5, d.rows * d.cols);
        for (int i = 0; i < n; i++) out[i] = d.template at<float>(i);
        return n;
    }
};
}  // namespace detail

// Frame::UndistortKeyPoints() (src/Frame.cc:773-806) without OpenCV's undistortPoints: mvKeysUn = mvKeys with the undistorted coordinates
// (eao_undistort_keypoints: one thread per keypoint on the device).
template <class FrameT>
void UndistortKeyPoints(FrameT& F) {
    float d[5] = {0, 0, 0, 0, 0};
    const int nc = detail::DistCoefOf<FrameT>::get(F, d);
    F.mvKeysUn = F.mvKeys;
    if (nc == 0 || d[0] == 0.0f) return;                               // :775-779
    const size_t n = F.mvKeys.size();
    std::vector<float> x(n), y(n), ux(n), uy(n);
    for (size_t i = 0; i < n; i++) { x[i] = F.mvKeys[i].pt.x; y[i] = F.mvKeys[i].pt.y; }
    detail::check(eao_undistort_keypoints((int)n, x.data(), y.data(), F.fx, F.fy, F.cx, F.cy, d, nc, ux.data(), uy.data()), "eao_undistort_keypoints");
    for (size_t i = 0; i < n; i++) { F.mvKeysUn[i].pt.x = ux[i]; F.mvKeysUn[i].pt.y = uy[i]; }
}
// Frame::ComputeImageBounds(imLeft) (src/Frame.cc:808-842): mnMinX .. mnMaxY from the undistorted image corners.
template <class FrameT>
void ComputeImageBounds(FrameT& F, int cols, int rows) {
    float d[5] = {0, 0, 0, 0, 0}, b[4];
    const int nc = detail::DistCoefOf<FrameT>::get(F, d);
    detail::check(eao_compute_image_bounds(cols, rows, F.fx, F.fy, F.cx, F.cy, d, nc, b), "eao_compute_image_bounds");
    F.mnMinX = b[0]; F.mnMaxX = b[1]; F.mnMinY = b[2]; F.mnMaxY = b[3];
}

// for (pMP : vpMPs) if (!skip(pMP)) if (F.isInFrustum(pMP, viewingCosLimit)) nToMatch++;   returns nToMatch.
// skip(pMP) holds the caller's own `continue`s (src/Tracking.cc:2615-2618: mnLastFrameSeen == mCurrentFrame.mnId, isBad()).
// Every visited point gets mbTrackInView; points in view also get mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos, and
// inView(pMP) is called for them in list order -- the place of upstream's pMP->IncreaseVisible() (src/Tracking.cc:2623).
template <class FrameT, class MapPointT, class Skip, class InView>
int IsInFrustum(FrameT& F, const std::vector<MapPointT*>& vpMPs, float viewingCosLimit, Skip&& skip, InView&& inView) {
    std::vector<int> idx;
    idx.reserve(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); i++)
        if (!skip(vpMPs[i])) idx.push_back((int)i);
    const int n = (int)idx.size();
    if (n == 0) return 0;
    std::vector<float> Xw(3 * (size_t)n), nrm(3 * (size_t)n), dmin(n), dmax(n), draw(n);
    for (int k = 0; k < n; k++) {
        MapPointT* p = vpMPs[idx[k]];
        const cv::Mat P = p->GetWorldPos(), Pn = p->GetNormal();
        for (int a = 0; a < 3; a++) { Xw[3 * (size_t)k + a] = P.template at<float>(a); nrm[3 * (size_t)k + a] = Pn.template at<float>(a); }
        dmin[k] = p->GetMinDistanceInvariance(); dmax[k] = p->GetMaxDistanceInvariance(); draw[k] = detail::MaxDistanceOf<MapPointT>::get(p);
    }
    eao_frustum_frame fr;
    // upstream's isInFrustum reads the private mRcw / mtcw / mOw; they are copies of blocks of the public mTcw
    // (Frame::UpdatePoseMatrices, src/Frame.cc:624-636) and mOw is what the public GetCameraCenter() returns
    const cv::Mat Ow = F.GetCameraCenter();
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 4; c++) fr.Tcw[4 * r + c] = F.mTcw.template at<float>(r, c);
        fr.Tcw[12 + r] = 0.f;
        fr.Ow[r] = Ow.template at<float>(r);
    }
    fr.Tcw[15] = 1.f;
    fr.fx = FrameT::fx; fr.fy = FrameT::fy; fr.cx = FrameT::cx; fr.cy = FrameT::cy; fr.mbf = F.mbf;
    fr.min_x = FrameT::mnMinX; fr.max_x = FrameT::mnMaxX; fr.min_y = FrameT::mnMinY; fr.max_y = FrameT::mnMaxY;
    fr.log_scale_factor = F.mfLogScaleFactor;
    eao_map_points mp;
    std::memset(&mp, 0, sizeof(mp));
    mp.n = n; mp.Xw = Xw.data(); mp.normal = nrm.data(); mp.min_dist_inv = dmin.data(); mp.max_dist_inv = dmax.data(); mp.max_dist = draw.data();
    std::vector<uint8_t> in(n, 0);
    std::vector<float> u(n, 0.f), v(n, 0.f), ur(n, 0.f), vc(n, 0.f);
    std::vector<int32_t> lvl(n, 0);
    detail::check(eao_frame_is_in_frustum(&fr, &mp, viewingCosLimit, in.data(), u.data(), v.data(), ur.data(), vc.data(), lvl.data()),
                  "eao_frame_is_in_frustum");
    int nToMatch = 0;
    for (int k = 0; k < n; k++) {
        MapPointT* p = vpMPs[idx[k]];
        p->mbTrackInView = in[k] != 0;
        if (!in[k]) continue;
        p->mTrackProjX = u[k]; p->mTrackProjXR = ur[k]; p->mTrackProjY = v[k];
        p->mnTrackScaleLevel = lvl[k]; p->mTrackViewCos = vc[k];
        inView(p);
        nToMatch++;
    }
    return nToMatch;
}
template <class FrameT, class MapPointT, class Skip>
int IsInFrustum(FrameT& F, const std::vector<MapPointT*>& vpMPs, float viewingCosLimit, Skip&& skip) {
    return IsInFrustum(F, vpMPs, viewingCosLimit, skip, [](MapPointT*) {});
}

// fills F.mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] from F.mvKeysUn (the reserve() of upstream is an allocation hint only)
template <class FrameT>
void AssignFeaturesToGrid(FrameT& F) {
    typedef typename std::remove_reference<decltype(F.mGrid)>::type GridT;
    const int cols = (int)std::extent<GridT, 0>::value, rows = (int)std::extent<GridT, 1>::value;
    const int N = F.N;
    std::vector<float> x(N), y(N);
    for (int i = 0; i < N; i++) { x[i] = F.mvKeysUn[i].pt.x; y[i] = F.mvKeysUn[i].pt.y; }
    std::vector<int32_t> start((size_t)cols * rows + 1, 0), items(N > 0 ? N : 1, 0);
    detail::check(eao_assign_features_to_grid(N, x.data(), y.data(), FrameT::mnMinX, FrameT::mnMinY, FrameT::mfGridElementWidthInv,
                                              FrameT::mfGridElementHeightInv, cols, rows, start.data(), items.data()),
                  "eao_assign_features_to_grid");
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {
            const size_t c = (size_t)i * rows + j;
            F.mGrid[i][j].assign(items.begin() + start[c], items.begin() + start[c + 1]);
        }
}

// imDepth: CV_32F, as Tracking::GrabImageRGBD hands it over after the depth-factor conversion
template <class FrameT>
void ComputeStereoFromRGBD(FrameT& F, const cv::Mat& imDepth) {
    const int N = F.N;
    F.mvuRight = std::vector<float>(N, -1.0f);
    F.mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return;
    std::vector<float> x(N), y(N), xu(N);
    for (int i = 0; i < N; i++) { x[i] = F.mvKeys[i].pt.x; y[i] = F.mvKeys[i].pt.y; xu[i] = F.mvKeysUn[i].pt.x; }
    detail::check(eao_compute_stereo_from_rgbd(N, x.data(), y.data(), xu.data(), reinterpret_cast<const float*>(imDepth.data), imDepth.cols,
                                               imDepth.rows, (int)(imDepth.step / sizeof(float)), 0, F.mbf, F.mvuRight.data(), F.mvDepth.data()),
                  "eao_compute_stereo_from_rgbd");
}

}  // namespace eaofusion

#endif  // EAOFUSION_FRAME_H
