// include/eaofusion/DeviceTracker.h -- the device-resident tracked frame (second half) behind the reference's
// own objects: Tracking::TrackLocalMap's data path (reference src/Tracking.cc:1717-2231, 2587-2641) with nothing returning to
// the host between ORBextractor::operator() and the optimised pose.  Over libeaofusion_hip.so (eao_tracker_*, include/eao_fusion.h).
//
//   eaofusion::DeviceTracker trk(frame, maxKeypoints, maxMapPoints);      // once per camera (Frame statics: fx .. mbf, bounds, scale tables)
//   trk.SetLocalMap(mvpLocalMapPoints);                                    // when Tracking::UpdateLocalMap changed the local map
//   int nm = trk.TrackWithMotionModel(mCurrentFrame, mLastFrame, d_kps, d_desc, d_n, d_depth, depthPitch, w, h, th, bMono, stream, &nmatchesMap);   // round 4
//   int nMap = trk.TrackReferenceKeyFrame(mCurrentFrame, mpReferenceKF, mLastFrame.mTcw, d_kps, d_desc, d_n, d_depth, depthPitch, w, h, stream, &nSearch);      // round 4
//   int nInliers = trk.TrackLocalMap(mCurrentFrame, d_kps, d_desc, d_n, d_depth, depthPitch, th, stream);
//
// TrackLocalMap replaces, for an RGB-D / monocular camera (with or without lens distortion: the constructor hands mDistCoef to the handle and the frame
// set-up then undistorts the keypoints on the device, Frame::UndistortKeyPoints), the sequence Frame::ComputeStereoFromRGBD +
// AssignFeaturesToGrid (the part of the Frame constructor after the extractor) -> Tracking::SearchLocalPoints ->
// Optimizer::PoseOptimization(&mCurrentFrame): it fills mvuRight, mvDepth, mvpMapPoints (new matches added to the ones the
// frame already has), mvbOutlier and the pose exactly as those calls would, and applies SearchLocalPoints' bookkeeping on the
// map points (bad prior matches dropped, IncreaseVisible, mnLastFrameSeen, mbTrackInView).  Only members the reference's classes
// already have are named.  Nothing here is copied from the reference.
#ifndef EAOFUSION_DEVICE_TRACKER_H
#define EAOFUSION_DEVICE_TRACKER_H

#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../eao_fusion.h"
#include "Frame.h"
#include "OptimizerImpl.h"      // collect_planes / store_planes: the frame's associated map planes as PoseOptimization's adapter reads them

namespace eaofusion {

class DeviceTracker {
public:
    template <class FrameT>
    DeviceTracker(const FrameT& F, int maxKeypoints, int maxMapPoints) : cap_(maxKeypoints), capMp_(maxMapPoints) {
        eao_tracker_cfg c;
        c.fx = F.fx; c.fy = F.fy; c.cx = F.cx; c.cy = F.cy; c.mbf = F.mbf;
        c.min_x = F.mnMinX; c.max_x = F.mnMaxX; c.min_y = F.mnMinY; c.max_y = F.mnMaxY;
        c.grid_cols = 64; c.grid_rows = 48;                       // FRAME_GRID_COLS / ROWS, include/Frame.h:89-90
        c.nlevels = F.mnScaleLevels;
        c.scale_factors = F.mvScaleFactors.data(); c.inv_level_sigma2 = F.mvInvLevelSigma2.data();
        c.log_scale_factor = F.mfLogScaleFactor;
        c.max_keypoints = maxKeypoints; c.max_map_points = maxMapPoints;
        detail::check(eao_tracker_create(&c, &h_), "eao_tracker_create");
        float d[5] = {0, 0, 0, 0, 0};
        const int nc = detail::DistCoefOf<FrameT>::get(F, d);      // (mnMinX .. mnMaxY above are upstream's bounds of the UNDISTORTED image, src/Frame.cc:808-842)
        if (nc) detail::check(eao_tracker_set_distortion(h_, d, nc), "eao_tracker_set_distortion");
    }
    ~DeviceTracker() { eao_tracker_destroy(h_); }
    DeviceTracker(const DeviceTracker&) = delete;
    DeviceTracker& operator=(const DeviceTracker&) = delete;

    // The handle's limits (4096 keypoints per frame, 16384 local map points at most, whatever the constructor asked for below that): a frame beyond them
    // goes through the host-hop calls -- upstream's own SearchLocalPoints / PoseOptimization over the drop-in adapters -- as the INTEGRATION.md fragments do.
    // A camera WITH LENS DISTORTION (TUM1 / TUM2: k1 = 0.26) is served since round 5: the chain undistorts the extractor's keypoints itself (eao_tracker_set_distortion)
    // and reads mvKeysUn wherever upstream does (rounds 3-4 read mvKeys, which is the same thing only while mDistCoef is zero, and refused such frames in early round 5).
    template <class FrameT> bool Fits(const FrameT& F) const { return F.N <= cap_; }
    int MaxKeypoints() const { return cap_; }
    int MaxMapPoints() const { return capMp_; }

    // Tracking::mvpLocalMapPoints -> HBM (kept until the next call).  Bad points stay in the arrays with active = 0.
    // Returns false -- and leaves the previous map in place -- when the local map is larger than maxMapPoints (the handle's
    // capacity, at most 16384): the caller then runs upstream's own SearchLocalPoints + PoseOptimization for this frame.
    template <class MapPointT>
    bool SetLocalMap(const std::vector<MapPointT*>& vpMPs) {
        const size_t n = vpMPs.size();
        if ((int)n > capMp_) return false;
        map_.assign(vpMPs.begin(), vpMPs.end());
        index_.clear();
        index_.reserve(2 * n);
        std::vector<unsigned char> active(n);
        std::vector<float> Xw(3 * n), nrm(3 * n), dmin(n), dmax(n), draw(n);
        std::vector<unsigned char> desc(32 * n);
        for (size_t i = 0; i < n; i++) {
            MapPointT* p = vpMPs[i];
            if (p) index_.emplace((void*)p, (int)i);          // (a point listed twice keeps its first slot, like upstream's first visit)
            active[i] = (p && !p->isBad()) ? 1 : 0;
            if (!active[i]) continue;
            const cv::Mat P = p->GetWorldPos(), Pn = p->GetNormal(), D = p->GetDescriptor();
            for (int a = 0; a < 3; a++) { Xw[3 * i + a] = P.template at<float>(a); nrm[3 * i + a] = Pn.template at<float>(a); }
            dmin[i] = p->GetMinDistanceInvariance(); dmax[i] = p->GetMaxDistanceInvariance(); draw[i] = detail::MaxDistanceOf<MapPointT>::get(p);
            std::memcpy(&desc[32 * i], D.template ptr<unsigned char>(0), 32);
        }
        eao_map_points mp;
        mp.n = (int)n; mp.active = active.data(); mp.Xw = Xw.data(); mp.normal = nrm.data(); mp.min_dist_inv = dmin.data();
        mp.max_dist_inv = dmax.data(); mp.max_dist = draw.data(); mp.desc = desc.data();
        detail::check(eao_tracker_set_local_map(h_, &mp), "eao_tracker_set_local_map");
        return true;
    }

    // d_kps / d_desc / d_n: this frame's slice of eao_orb_extract_batch_device's outputs (F.mvKeys / mDescriptors are their host
    // copies, F.N = the count); d_depth: imDepth on the device or nullptr.  Returns Optimizer::PoseOptimization's return value.
    //
    // Bookkeeping of Tracking::SearchLocalPoints (src/Tracking.cc:2590-2627), applied here exactly as upstream orders it:
    //   * a prior match that isBad() is set to NULL first (:2596-2599) -- its keypoint is free for the search;
    //   * every remaining prior match: IncreaseVisible(), mnLastFrameSeen = F.mnId, mbTrackInView = false (:2603-2607).  A prior
    //     match that is not in the local map (a temporal RGB-D point, a point the local map dropped) stays in mvpMapPoints, keeps
    //     its keypoint occupied and is an edge of the pose optimisation from its own GetWorldPos(), as upstream takes every
    //     mvpMapPoints entry (src/Optimizer.cc:361-447);
    //   * every other non-bad local point in the frustum: mbTrackInView = true, IncreaseVisible() (:2621-2625); out of it:
    //     mbTrackInView = false.  (mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos stay on the device: only the matcher
    //     reads them, and the matcher ran there.)
    template <class FrameT>
    int TrackLocalMap(FrameT& F, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, const float* d_depth, int depthPitch,
                      int width, int height, float th, float nnratio, void* stream) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(F.mvpMapPoints[0])>::type>::type;
        std::vector<int32_t> prior(cap_, -1), kpMp(cap_, -1);
        std::vector<uint8_t> outl(cap_, 0), inView(map_.size() ? map_.size() : 1, 0);
        std::vector<float> ur(cap_), dz(cap_), priorXw;
        if (F.N > cap_) throw std::runtime_error("DeviceTracker: the frame has more keypoints than maxKeypoints");
        for (int k = 0; k < F.N; k++) {
            MapPointT* pMP = F.mvpMapPoints[k];
            if (!pMP) continue;
            if (pMP->isBad()) { F.mvpMapPoints[k] = static_cast<MapPointT*>(NULL); continue; }
            pMP->IncreaseVisible();
            pMP->mnLastFrameSeen = F.mnId;
            pMP->mbTrackInView = false;
            const auto it = index_.find((void*)pMP);
            if (it != index_.end()) { prior[k] = it->second; continue; }
            prior[k] = -2;
            if (priorXw.empty()) priorXw.assign(3 * (size_t)cap_, 0.f);
            const cv::Mat P = pMP->GetWorldPos();
            for (int a = 0; a < 3; a++) priorXw[3 * (size_t)k + a] = P.template at<float>(a);
        }
        float T[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) T[4 * r + c] = F.mTcw.template at<float>(r, c);
        eao_track_result R = {};      // (zero first: a member this header does not set reads as "not wanted", include/eao_fusion.h EAO_ABI_VERSION)
        R.kp_map_point = kpMp.data(); R.kp_outlier = outl.data(); R.kp_u_right = ur.data(); R.kp_depth = dz.data(); R.map_in_view = inView.data();
        PlaneEdges pe;
        std::vector<uint8_t> pout;
        set_options(F, 0, pe, pout);
        detail::check(eao_tracker_track_local_map(h_, d_kps, d_desc, d_n, d_depth, depthPitch, width, height, T, prior.data(),
                                                  priorXw.empty() ? nullptr : priorXw.data(), th, nnratio, &R, stream),
                      "eao_tracker_track_local_map");
        for (size_t m = 0; m < map_.size(); m++) {
            MapPointT* pMP = static_cast<MapPointT*>(map_[m]);
            if (!pMP || pMP->mnLastFrameSeen == F.mnId || pMP->isBad()) continue;      // upstream's two `continue`s, :2615-2618
            pMP->mbTrackInView = inView[m] != 0;
            if (inView[m]) pMP->IncreaseVisible();
        }
        F.mvuRight.assign(ur.begin(), ur.begin() + R.n_keypoints);
        F.mvDepth.assign(dz.begin(), dz.begin() + R.n_keypoints);
        for (int k = 0; k < R.n_keypoints; k++) {
            if (kpMp[k] >= 0 && !F.mvpMapPoints[k]) F.mvpMapPoints[k] = static_cast<MapPointT*>(map_[kpMp[k]]);
            F.mvbOutlier[k] = outl[k] != 0;
        }
        cv::Mat pose(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = R.Tcw[4 * r + c];
        F.SetPose(pose);
        if (R.n_edges >= 3) store_planes(&F, pe, pout);      // mvbPlaneOutlier (upstream resets / sets them behind the "< 3 correspondences" test only)
        return R.n_inliers;
    }

    // Tracking::TrackWithMotionModel's data path (src/Tracking.cc:1717-2231) for the frame the extractor just left on the device: the part of the Frame
    // constructor after the extractor (mvuRight / mvDepth, the grid), ORBmatcher(0.9, true).SearchByProjection(Cur, Last, th, bMono), Optimizer::
    // PoseOptimization(&Cur) and the "Discard outliers" loop, one copy back.  The caller has set Cur's predicted pose (SetPose(mVelocity * mLastFrame.mTcw),
    // :1726) and cleared Cur.mvpMapPoints (:1729); it repeats the call with 2 * th when fewer than 20 matches come back (:1756-1760), exactly as upstream
    // repeats the search.  Returns the search's nmatches minus the discarded outliers (upstream's `nmatches` at :2231); nmatchesMap counts the kept matches
    // whose map point has observations (:2202-2203).  What upstream does between the search and the optimisation (object association, plane association) reads
    // Cur.mvpMapPoints, which this call has filled by then only at its END -- a caller that needs them runs this stage with the host-hop calls instead.
    // Round 5: (a) a search that returns fewer than minMatches (upstream's 20, :1756 / :1762) is NOT followed by the pose optimisation and the discard -- the call
    // returns the search's count right away and has touched no map point and no pose, exactly as upstream at that point; the caller repeats it with 2 * th or
    // returns false.  (b) The planes the caller associated with the predicted pose (Map::AssociatePlanesByBoundary, :2181 -- it reads the pose, not the matches)
    // are edges of the chained optimisation; mvbPlaneOutlier comes back, the plane discard loop (:2210-2221) stays with the caller.
    template <class FrameT>
    int TrackWithMotionModel(FrameT& Cur, const FrameT& Last, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, const float* d_depth,
                             int depthPitch, int width, int height, float th, bool bMono, void* stream, int* nmatchesMap = nullptr, int* nSearch = nullptr,
                             int minMatches = 20) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(Cur.mvpMapPoints[0])>::type>::type;
        if (Cur.N > cap_ || Last.N > cap_) throw std::runtime_error("DeviceTracker: a frame has more keypoints than maxKeypoints");
        const size_t nl = (size_t)Last.N;
        std::vector<uint8_t> valid(nl ? nl : 1, 0), desc(32 * (nl ? nl : 1), 0);
        std::vector<float> Xw(3 * (nl ? nl : 1), 0.f), ang(nl ? nl : 1, 0.f);
        std::vector<int32_t> oct(nl ? nl : 1, 0);
        for (size_t i = 0; i < nl; i++) {
            MapPointT* pMP = Last.mvpMapPoints[i];
            oct[i] = Last.mvKeys[i].octave; ang[i] = Last.mvKeysUn[i].angle;
            if (!pMP || Last.mvbOutlier[i]) continue;
            valid[i] = 1;
            const cv::Mat P = pMP->GetWorldPos(), D = pMP->GetDescriptor();
            for (int a = 0; a < 3; a++) Xw[3 * i + a] = P.template at<float>(a);
            std::memcpy(&desc[32 * i], D.template ptr<unsigned char>(0), 32);
        }
        float Tc[16], Tl[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) { Tc[4 * r + c] = Cur.mTcw.template at<float>(r, c); Tl[4 * r + c] = Last.mTcw.template at<float>(r, c); }
        std::vector<int32_t> kpMp(cap_, -1);
        std::vector<uint8_t> outl(cap_, 0);
        std::vector<float> ur(cap_), dz(cap_);
        eao_track_result R = {};
        R.kp_map_point = kpMp.data(); R.kp_outlier = outl.data(); R.kp_u_right = ur.data(); R.kp_depth = dz.data();
        PlaneEdges pe;
        std::vector<uint8_t> pout;
        set_options(Cur, minMatches, pe, pout);
        detail::check(eao_tracker_track_with_motion_model(h_, d_kps, d_desc, d_n, d_depth, depthPitch, width, height, Tc, Tl, (int)nl, valid.data(), Xw.data(),
                                                          desc.data(), oct.data(), ang.data(), th, bMono ? 1 : 0, 1, /* discard on this side */ 0, &R, stream),
                      "eao_tracker_track_with_motion_model");
        if (nSearch) *nSearch = R.n_matches;
        Cur.mvuRight.assign(ur.begin(), ur.begin() + R.n_keypoints);
        Cur.mvDepth.assign(dz.begin(), dz.begin() + R.n_keypoints);
        if (R.n_matches < minMatches) {      // upstream has only searched at this point: no pose, no outlier flag, no map point touched
            if (nmatchesMap) *nmatchesMap = 0;
            return R.n_matches;
        }
        cv::Mat pose(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = R.Tcw[4 * r + c];
        Cur.SetPose(pose);
        if (R.n_edges >= 3) store_planes(&Cur, pe, pout);
        int nmatches = R.n_matches, nMap = 0;
        for (int k = 0; k < R.n_keypoints; k++) {      // "Discard outliers", src/Tracking.cc:2188-2207
            Cur.mvbOutlier[k] = false;
            if (kpMp[k] < 0) continue;
            MapPointT* pMP = Last.mvpMapPoints[kpMp[k]];
            if (outl[k]) {
                pMP->mbTrackInView = false;
                pMP->mnLastFrameSeen = Cur.mnId;
                nmatches--;
            } else {
                Cur.mvpMapPoints[k] = pMP;
                if (pMP->Observations() > 0) nMap++;
            }
        }
        if (nmatchesMap) *nmatchesMap = nMap;
        return nmatches;
    }

    // Tracking::TrackReferenceKeyFrame's data path (src/Tracking.cc:1568-1631) for the frame the extractor just left on the device: the part of the Frame
    // constructor after the extractor, ORBmatcher(0.7, true).SearchByBoW(mpReferenceKF, Cur, vpMapPointMatches), Cur.SetPose(mLastFrame.mTcw),
    // Optimizer::PoseOptimization(&Cur) and the "Discard outliers" loop, one copy back.  The caller has run Cur.ComputeBoW() (:1571: DBoW2 works on the host
    // copy of the descriptors) and passes the last frame's pose.  Returns nmatchesMap (:1630 compares it with 10); nSearch = SearchByBoW's return value (:1580
    // compares it with 10 -- below it the caller takes upstream's "return false" and ignores what this call filled in).  As with TrackWithMotionModel, what this
    // fork does between the search and the optimisation (AssociatePlanesByBoundary, :1587) sees Cur.mvpMapPoints only after the call.
    template <class KeyFrameT, class FrameT>
    int TrackReferenceKeyFrame(FrameT& Cur, KeyFrameT* pRefKF, const cv::Mat& lastTcw, const eao_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                               const float* d_depth, int depthPitch, int width, int height, void* stream, int* nSearch = nullptr, float nnratio = 0.7f,
                               int minMatches = 10) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(Cur.mvpMapPoints[0])>::type>::type;
        const std::vector<MapPointT*> vpKF = pRefKF->GetMapPointMatches();
        const size_t nk = vpKF.size();
        if (Cur.N > cap_ || (int)nk > cap_) throw std::runtime_error("DeviceTracker: more keypoints than maxKeypoints");
        std::vector<uint8_t> valid(nk ? nk : 1, 0), desc(32 * (nk ? nk : 1), 0);
        std::vector<float> Xw(3 * (nk ? nk : 1), 0.f), ang(nk ? nk : 1, 0.f);
        for (size_t i = 0; i < nk; i++) {
            ang[i] = pRefKF->mvKeysUn[i].angle;
            std::memcpy(&desc[32 * i], pRefKF->mDescriptors.ptr(i), 32);
            MapPointT* pMP = vpKF[i];
            if (!pMP || pMP->isBad()) continue;
            valid[i] = 1;
            const cv::Mat P = pMP->GetWorldPos();
            for (int a = 0; a < 3; a++) Xw[3 * i + a] = P.template at<float>(a);
        }
        FeatVec fk, fc;
        const eao_feature_vector fvK = flatten(pRefKF->mFeatVec, fk), fvC = flatten(Cur.mFeatVec, fc);
        float Tl[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) Tl[4 * r + c] = lastTcw.template at<float>(r, c);
        std::vector<int32_t> kpMp(cap_, -1);
        std::vector<uint8_t> outl(cap_, 0);
        std::vector<float> ur(cap_), dz(cap_);
        eao_track_result R = {};
        R.kp_map_point = kpMp.data(); R.kp_outlier = outl.data(); R.kp_u_right = ur.data(); R.kp_depth = dz.data();
        PlaneEdges pe;
        std::vector<uint8_t> pout;
        set_options(Cur, minMatches, pe, pout);      // (the planes were associated with mLastFrame.mTcw set as the frame's pose, :1584-1587)
        detail::check(eao_tracker_track_reference_keyframe(h_, d_kps, d_desc, d_n, d_depth, depthPitch, width, height, Tl, (int)nk, valid.data(), Xw.data(), desc.data(),
                                                           ang.data(), &fvK, &fvC, nnratio, 1, /* discard on this side */ 0, &R, stream),
                      "eao_tracker_track_reference_keyframe");
        if (nSearch) *nSearch = R.n_matches;
        Cur.mvuRight.assign(ur.begin(), ur.begin() + R.n_keypoints);
        Cur.mvDepth.assign(dz.begin(), dz.begin() + R.n_keypoints);
        if (R.n_matches < minMatches) return 0;      // "if (nmatches < 10) return false" (:1580-1581): nothing optimised, nothing touched
        cv::Mat pose(4, 4, CV_32F);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = R.Tcw[4 * r + c];
        Cur.SetPose(pose);
        if (R.n_edges >= 3) store_planes(&Cur, pe, pout);
        int nMap = 0;
        for (int k = 0; k < R.n_keypoints; k++) {      // "Discard outliers", src/Tracking.cc:1593-1612
            Cur.mvbOutlier[k] = false;
            Cur.mvpMapPoints[k] = static_cast<MapPointT*>(NULL);
            if (kpMp[k] < 0) continue;
            MapPointT* pMP = vpKF[kpMp[k]];
            if (outl[k]) {
                pMP->mbTrackInView = false;
                pMP->mnLastFrameSeen = Cur.mnId;
            } else {
                Cur.mvpMapPoints[k] = pMP;
                if (pMP->Observations() > 0) nMap++;
            }
        }
        return nMap;
    }

private:
    // The one-shot options of the next chain call (eao_tracker_set_options, round 5): the frame's associated map planes -- in this fork
    // Map::AssociatePlanesByBoundary runs BEFORE Optimizer::PoseOptimization in every stage (src/Tracking.cc:1587, :2181, TrackLocalMap), so they are edges of
    // the optimisation (src/Optimizer.cc:456-535); the caller has associated them with the pose the stage starts from -- and the stage's match threshold.
    template <class FrameT>
    void set_options(FrameT& F, int minMatches, PlaneEdges& pe, std::vector<uint8_t>& pout) {
        collect_planes(&F, pe);
        const int M = (int)pe.slot.size();
        if (M > 32) throw std::runtime_error("DeviceTracker: more than 32 associated planes");
        pout.assign(M ? M : 1, 0);
        eao_track_options O;
        O.min_matches = minMatches; O.n_planes = M;
        O.plane_world = M ? pe.world.data() : nullptr; O.plane_obs = M ? pe.obs.data() : nullptr; O.plane_seen = M ? pe.seen.data() : nullptr; O.plane_outlier = pout.data();
        detail::check(eao_tracker_set_options(h_, &O), "eao_tracker_set_options");
    }
    struct FeatVec { std::vector<uint32_t> id, index; std::vector<int32_t> start; };
    template <class FeatVecT>
    static eao_feature_vector flatten(const FeatVecT& fv, FeatVec& a) {      // DBoW2::FeatureVector = std::map<node id, std::vector<keypoint index>>
        a.id.clear(); a.index.clear(); a.start.assign(1, 0);
        for (typename FeatVecT::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            a.id.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) a.index.push_back((uint32_t)it->second[k]);
            a.start.push_back((int32_t)a.index.size());
        }
        eao_feature_vector f;
        f.n_nodes = (int32_t)a.id.size(); f.node_id = a.id.data(); f.node_start = a.start.data(); f.index = a.index.data();
        return f;
    }
    eao_tracker* h_ = nullptr;
    int cap_, capMp_;
    std::vector<void*> map_;
    std::unordered_map<void*, int> index_;      // map point -> slot of the uploaded local map (the prior-match lookup, O(1) per keypoint)
};

}  // namespace eaofusion
#endif
