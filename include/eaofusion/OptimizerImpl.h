// OptimizerImpl.h -- Optimizer::PoseOptimization / Optimizer::LocalBundleAdjustment on top of the C-ABI.
//
// The reference functions (src/Optimizer.cc:325-673, 675-1138) walk the SLAM object graph, build a g2o graph,
// optimise, and write back.  Here the walk and the write-back stay on the host, restated over the SAME member
// names of the reference's Frame / KeyFrame / MapPoint / Map classes (so the templates instantiate against the
// real classes in a checkout -- see INTEGRATION.md), and the optimisation is one call into libeaofusion_hip.so.
//
//   // src/Optimizer_hip.cc in an EAO-Fusion checkout (replaces the two function bodies in src/Optimizer.cc):
//   #include "Optimizer.h"            // reference header, unchanged
//   #include <eaofusion/OptimizerImpl.h>
//   int  ORB_SLAM2::Optimizer::PoseOptimization(Frame* f) { return eaofusion::PoseOptimization<MapPoint>(f); }
//   void ORB_SLAM2::Optimizer::LocalBundleAdjustment(KeyFrame* kf, bool* stop, Map* m) { eaofusion::LocalBundleAdjustment<MapPoint>(kf, stop, m); }
//
// The plane edges PoseOptimization adds when PEAC planes are associated (src/Optimizer.cc:456-535, 626-658) are part
// of the call: the template reads Frame::mnPlaneNum / mvpMapPlanes / mvPlaneCoefficients / mvbPlaneOutlier and
// MapPlane::GetWorldPos() / mbSeen / mGlobalMutex when the Frame class has them (upstream's does).
#pragma once

#include <algorithm>
#include <list>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../eao_fusion.h"
#include "cv_compat.h"

namespace eaofusion {

inline void check(eao_status st, const char* what) {
    if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
}

// Frame classes with associated map planes (upstream's include/Frame.h: mnPlaneNum, mvpMapPlanes, ...)
template <class F, class = void> struct has_planes : std::false_type {};
template <class F> struct has_planes<F, decltype(void(std::declval<F&>().mnPlaneNum), void(std::declval<F&>().mvpMapPlanes))> : std::true_type {};

struct PlaneEdges { std::vector<int> slot; std::vector<float> world, obs; std::vector<uint8_t> seen; };
// src/Optimizer.cc:456-535: one edge per associated map plane, in index order (under MapPlane::mGlobalMutex)
template <class FrameT>
typename std::enable_if<has_planes<FrameT>::value>::type collect_planes(FrameT* pFrame, PlaneEdges& pe) {
    typedef typename std::remove_pointer<typename std::decay<decltype(pFrame->mvpMapPlanes[0])>::type>::type MapPlaneT;
    std::unique_lock<std::mutex> lock(MapPlaneT::mGlobalMutex);
    const int M = pFrame->mnPlaneNum;
    for (int i = 0; i < M; ++i) {
        MapPlaneT* pMP = pFrame->mvpMapPlanes[i];
        if (!pMP) continue;
        pFrame->mvbPlaneOutlier[i] = false;
        const cv::Mat w = pMP->GetWorldPos();
        const cv::Mat& c = pFrame->mvPlaneCoefficients[i];
        pe.slot.push_back(i);
        for (int k = 0; k < 4; k++) { pe.world.push_back(w.template at<float>(k, 0)); pe.obs.push_back(c.template at<float>(k, 0)); }
        pe.seen.push_back(pMP->mbSeen ? 1 : 0);
    }
}
template <class FrameT>
typename std::enable_if<!has_planes<FrameT>::value>::type collect_planes(FrameT*, PlaneEdges&) {}
template <class FrameT>
typename std::enable_if<has_planes<FrameT>::value>::type store_planes(FrameT* pFrame, const PlaneEdges& pe, const std::vector<uint8_t>& out) {
    for (size_t k = 0; k < pe.slot.size(); k++) pFrame->mvbPlaneOutlier[pe.slot[k]] = out[k] != 0;
}
template <class FrameT>
typename std::enable_if<!has_planes<FrameT>::value>::type store_planes(FrameT*, const PlaneEdges&, const std::vector<uint8_t>&) {}

// the world position of a map point without a cv::Mat clone when the checkout carries the optional accessor of INTEGRATION.md row 2c (GetWorldPos(float*)), upstream's otherwise
namespace detail {
template <class MP, class = void> struct HasWorldPosOut : std::false_type {};
template <class MP> struct HasWorldPosOut<MP, decltype(std::declval<MP&>().GetWorldPos(static_cast<float*>(nullptr)), void())> : std::true_type {};
template <class MP> void world_pos(MP* mp, float* xyz, std::true_type) { mp->GetWorldPos(xyz); }
template <class MP> void world_pos(MP* mp, float* xyz, std::false_type) {
    const cv::Mat P = mp->GetWorldPos();
    for (int k = 0; k < 3; k++) xyz[k] = P.template at<float>(k);
}
}  // namespace detail

// ---- Optimizer::PoseOptimization(Frame*) --------------------------------------------------------------------
// The flattened problem of one frame and what is needed to write its result back.
struct PoseJob {
    std::vector<int> slot;            // frame index of every correspondence, ascending
    std::vector<float> Xw, obs, inv;
    float T[16];
    PlaneEdges pe;
    std::vector<uint8_t> outl, pout;
    eao_pose_problem P;
    eao_pose_result R;
    int n = 0;
};
// src/Optimizer.cc:349-453 (vertices and edges) -- returns false when upstream would return 0 without optimising
template <class MapPointT, class FrameT>
bool gather_pose(FrameT* pFrame, PoseJob& j) {
    const int N = pFrame->N;
    j.slot.reserve(N); j.Xw.reserve(3 * (size_t)N); j.obs.reserve(3 * (size_t)N); j.inv.reserve(N);
    {
        std::unique_lock<std::mutex> lock(MapPointT::mGlobalMutex);
        for (int i = 0; i < N; i++) {
            MapPointT* pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            pFrame->mvbOutlier[i] = false;
            const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
            float xyz[3];
            detail::world_pos(pMP, xyz, detail::HasWorldPosOut<MapPointT>());
            j.slot.push_back(i);
            j.Xw.push_back(xyz[0]); j.Xw.push_back(xyz[1]); j.Xw.push_back(xyz[2]);
            j.obs.push_back(kpUn.pt.x); j.obs.push_back(kpUn.pt.y); j.obs.push_back(pFrame->mvuRight[i]);   // < 0 => monocular edge
            j.inv.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    j.n = (int)j.slot.size();
    if (j.n < 3) return false;   // pose untouched, as upstream
    cv::Mat Tcw = pFrame->mTcw;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) j.T[r * 4 + c] = Tcw.template at<float>(r, c);
    collect_planes(pFrame, j.pe);     // added after the "< 3 correspondences" test, as upstream (:453-456)
    const int M = (int)j.pe.slot.size();
    j.P = eao_pose_problem{j.n, j.T, j.Xw.data(), j.obs.data(), j.inv.data(), pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf,
                           M, j.pe.world.data(), j.pe.obs.data(), j.pe.seen.data()};
    j.outl.assign(j.n, 0);
    j.pout.assign(M ? M : 1, 0);
    j.R.outlier = j.outl.data();
    j.R.plane_outlier = j.pout.data();
    return true;
}
// src/Optimizer.cc:660-672 (outlier flags were written round by round upstream; the last round's are what remains)
template <class FrameT>
int store_pose(FrameT* pFrame, const PoseJob& j) {
    for (int k = 0; k < j.n; k++) pFrame->mvbOutlier[j.slot[k]] = j.outl[k] != 0;
    store_planes(pFrame, j.pe, j.pout);
    cv::Mat pose(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = j.R.Tcw[r * 4 + c];
    pFrame->SetPose(pose);
    return j.R.n_inliers;
}

template <class MapPointT, class FrameT>
int PoseOptimization(FrameT* pFrame) {
    static thread_local PoseJob j;      // (its vectors keep their capacity: nothing is allocated per frame after the first)
    j.slot.clear(); j.Xw.clear(); j.obs.clear(); j.inv.clear(); j.pe.slot.clear(); j.pe.world.clear(); j.pe.obs.clear(); j.pe.seen.clear();
    if (!gather_pose<MapPointT>(pFrame, j)) return 0;
    check(eao_pose_optimization(&j.P, &j.R), "eao_pose_optimization");
    return store_pose(pFrame, j);
}

// Several frames at once: the candidate loop of Tracking::Relocalization (src/Tracking.cc:2786-2940) keeps one Frame copy per
// candidate keyframe and calls this once instead of PoseOptimization per candidate.  Element k of the result is what
// PoseOptimization(frames[k]) returns; every frame is written back exactly as by that call.
template <class MapPointT, class FrameT>
std::vector<int> PoseOptimizationBatch(const std::vector<FrameT*>& frames) {
    std::vector<PoseJob> jobs(frames.size());
    std::vector<int> live, ret(frames.size(), 0);
    for (size_t k = 0; k < frames.size(); k++)
        if (gather_pose<MapPointT>(frames[k], jobs[k])) live.push_back((int)k);
    std::vector<eao_pose_problem> P(live.size());
    std::vector<eao_pose_result> R(live.size());
    for (size_t q = 0; q < live.size(); q++) { P[q] = jobs[live[q]].P; R[q] = jobs[live[q]].R; }
    check(eao_pose_optimization_batch(P.data(), (int32_t)live.size(), R.data()), "eao_pose_optimization_batch");
    for (size_t q = 0; q < live.size(); q++) {
        jobs[live[q]].R = R[q];
        ret[live[q]] = store_pose(frames[live[q]], jobs[live[q]]);
    }
    return ret;
}

// ---- Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) ------------------------------------------------
// Round 5 (the class surface timed -- the first version of this walk cost 2.2 ms around a 1.04 ms library call on the 20 + 4 keyframe x
// 3000 point window): every map point's observations are read ONCE (upstream copies the std::map twice per point, src/Optimizer.cc:722 and :836) into flat
// per-thread arrays that also serve the fixed-camera pass, the edge pass and the write-back; the point -> index map is the rank of the point's mnId (one sort
// of 3000 pairs, no tree with 15 000 look-ups); nothing is allocated per call after the first.  What remains is what the reference's own accessors cost -- a
// mutex and a cv::Mat clone per GetWorldPos / GetPose, a std::map copy per GetObservations, SetWorldPos -- and tests/cpp/adapter_bench.cpp reports that floor.
// Round 6 (OPTIONAL): half of this call's wall time at the class surface is the reference's object model -- GetObservations()
// returns a COPY of a std::map (five nodes allocated and freed per map point), GetWorldPos() a cv::Mat clone.  A checkout that adds the two allocation-free accessors
//     template <class F> void ForEachObservation(F&& f) { unique_lock<mutex> lock(mMutexFeatures); for (auto& o : mObservations) f(o.first, o.second); }
//     void GetWorldPos(float* xyz) { unique_lock<mutex> lock(mMutexPos); for (int i = 0; i < 3; i++) xyz[i] = mWorldPos.at<float>(i); }
// to include/MapPoint.h (beside :43 / src/MapPoint.cc:139) is detected here and the walk uses them; an unedited MapPoint takes upstream's accessors as before.  The
// callback runs under the point's mMutexFeatures and only appends to a flat array (no other lock is taken inside it).
namespace detail {
template <class MP, class KF, class = void> struct HasForEachObservation : std::false_type {};
template <class MP, class KF>
struct HasForEachObservation<MP, KF, decltype(std::declval<MP&>().ForEachObservation(std::declval<void (*)(KF*, size_t)>()), void())> : std::true_type {};
template <class KF, class MP, class F> void for_each_observation(MP* mp, F& f, std::true_type) { mp->ForEachObservation(f); }
template <class KF, class MP, class F> void for_each_observation(MP* mp, F& f, std::false_type) {
    const std::map<KF*, size_t> seenBy = mp->GetObservations();      // the ONE copy per point (upstream: two, src/Optimizer.cc:725, :836)
    for (typename std::map<KF*, size_t>::const_iterator it = seenBy.begin(); it != seenBy.end(); ++it) f(it->first, it->second);
}
}  // namespace detail

template <class KeyFrameT, class MapPointT>
struct LbaScratch {
    struct Obs { KeyFrameT* kf; size_t idx; };
    struct CamRec { KeyFrameT* kf; bool fixed; bool local; int32_t slot; };
    std::vector<KeyFrameT*> localKFs, fixedKFs, slotKF;
    std::vector<MapPointT*> localMPs;
    std::vector<Obs> obs;                 // all observations, point after point in localMPs order
    std::vector<int32_t> obsSlot;         // ... and the slot of each observation's keyframe (-1: a bad keyframe)
    std::vector<int32_t> obsStart;        // localMPs.size() + 1
    std::vector<CamRec> cams;
    std::vector<std::pair<unsigned long, int32_t> > order, orderTmp;
    std::vector<int32_t> rank, eCam, ePt, slotCam;
    std::vector<float> camT, xyz, eObs, eInv, camOut, xyzOut;
    std::vector<uint8_t> camFixed, camBad, erase;
    std::vector<MapPointT*> pts;
    // keyframe -> slot: the window holds a few dozen keyframes, every observation asks once.  Open addressing over a power of two, rebuilt per call.
    std::vector<std::pair<KeyFrameT*, int32_t> > table;
    size_t tableMask = 0, tableUsed = 0;
    static size_t hash_of(const void* p) { return (size_t)(((uintptr_t)p >> 4) * (uintptr_t)0x9E3779B97F4A7C15ull >> 40); }
    void table_reset(size_t cap) {
        table.assign(cap, std::make_pair((KeyFrameT*)nullptr, (int32_t)0));
        tableMask = cap - 1; tableUsed = 0;
    }
    std::pair<KeyFrameT*, int32_t>* table_find(KeyFrameT* kf) {      // the keyframe's entry, or the empty one it would take
        size_t h = hash_of(kf) & tableMask;
        while (table[h].first && table[h].first != kf) h = (h + 1) & tableMask;
        return &table[h];
    }
    const std::pair<KeyFrameT*, int32_t>* table_find(KeyFrameT* kf) const { return const_cast<LbaScratch*>(this)->table_find(kf); }
    void table_insert(KeyFrameT* kf, int32_t slot) {
        if (2 * (tableUsed + 1) > table.size()) {
            std::vector<std::pair<KeyFrameT*, int32_t> > old;
            old.swap(table);
            table_reset(old.size() * 2);
            for (size_t i = 0; i < old.size(); i++) if (old[i].first) { *table_find(old[i].first) = old[i]; tableUsed++; }
        }
        std::pair<KeyFrameT*, int32_t>* e = table_find(kf);
        if (!e->first) tableUsed++;
        e->first = kf; e->second = slot;
    }
};

namespace detail {
// order[i] = (id, k) sorted by id: least-significant-digit radix sort over the id RANGE of the window (map point ids of one window lie close together: two passes of
// eleven bits; a comparison sort of 3000 pairs was a tenth of the adapter's walk)
inline void sort_by_id(std::vector<std::pair<unsigned long, int32_t> >& v, std::vector<std::pair<unsigned long, int32_t> >& tmp) {
    const size_t n = v.size();
    if (n < 64) { std::sort(v.begin(), v.end()); return; }
    unsigned long lo = v[0].first, hi = v[0].first;
    for (size_t i = 1; i < n; i++) { lo = std::min(lo, v[i].first); hi = std::max(hi, v[i].first); }
    tmp.resize(n);
    size_t count[2048];
    for (int shift = 0; shift < 64 && ((hi - lo) >> shift) != 0; shift += 11) {
        std::memset(count, 0, sizeof(count));
        for (size_t i = 0; i < n; i++) count[((v[i].first - lo) >> shift) & 2047]++;
        size_t run = 0;
        for (int d = 0; d < 2048; d++) { const size_t c = count[d]; count[d] = run; run += c; }
        for (size_t i = 0; i < n; i++) tmp[count[((v[i].first - lo) >> shift) & 2047]++] = v[i];
        v.swap(tmp);
    }
    // (equal ids cannot occur -- one MapPoint, one id; the sort is stable in k anyway)
}
}  // namespace detail

template <class MapPointT, class KeyFrameT, class MapT>
void LocalBundleAdjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap) {
    typedef LbaScratch<KeyFrameT, MapPointT> Scratch;
    static thread_local Scratch S;
    const unsigned long me = pKF->mnId;
    // the local window: this keyframe + its covisible ones; every map point they see; every other observer is fixed (src/Optimizer.cc:680-738).
    // Every keyframe the walk meets gets a SLOT (local ones first, fixed ones as their first observation turns up; -1 = bad): the observations carry the slot, and
    // the edge pass below turns it into the camera index with one table look-up (it was a binary search over the window's cameras per edge).
    S.localKFs.clear(); S.fixedKFs.clear(); S.slotKF.clear(); S.localMPs.clear(); S.obs.clear(); S.obsSlot.clear(); S.obsStart.assign(1, 0);
    S.table_reset(256);
    S.localKFs.push_back(pKF);
    pKF->mnBALocalForKF = me;
    S.table_insert(pKF, 0); S.slotKF.push_back(pKF);
    for (KeyFrameT* n : pKF->GetVectorCovisibleKeyFrames()) {
        n->mnBALocalForKF = me;
        if (!n->isBad()) { S.localKFs.push_back(n); S.table_insert(n, (int32_t)S.slotKF.size()); S.slotKF.push_back(n); }
        else S.table_insert(n, -1);
    }
    const size_t nLocal = S.localKFs.size();
    for (KeyFrameT* kf : S.localKFs)
        for (MapPointT* mp : kf->GetMapPointMatches())
            if (mp && mp->mnBALocalForKF != me && !mp->isBad()) {      // (the mark first: isBad() takes two mutexes, and four of five visits find the point marked already)
                S.localMPs.push_back(mp);
                mp->mnBALocalForKF = me;
            }
    const size_t nP = S.localMPs.size();
    // (the callback runs under the point's mutex: it looks the keyframe's slot up and appends -- a keyframe it has not met yet is entered after the accessor has returned,
    //  because isBad() takes the keyframe's own mutex)
    struct Append {
        Scratch* S;
        bool unknown;
        void operator()(KeyFrameT* kf, size_t idx) {
            typename Scratch::Obs o = {kf, idx};
            S->obs.push_back(o);
            const std::pair<KeyFrameT*, int32_t>* e = S->table_find(kf);
            S->obsSlot.push_back(e->first ? e->second : -2);
            unknown = unknown || !e->first;
        }
    } append = {&S, false};
    for (MapPointT* mp : S.localMPs) {
        const size_t first = S.obs.size();
        detail::for_each_observation<KeyFrameT>(mp, append, detail::HasForEachObservation<MapPointT, KeyFrameT>());
        if (append.unknown) {
            append.unknown = false;
            for (size_t o = first; o < S.obs.size(); o++) {
                if (S.obsSlot[o] != -2) continue;
                KeyFrameT* kf = S.obs[o].kf;
                const std::pair<KeyFrameT*, int32_t>* e = S.table_find(kf);
                if (e->first) { S.obsSlot[o] = e->second; continue; }      // (twice in one point's map: cannot happen, but costs nothing)
                kf->mnBAFixedForKF = me;
                int32_t slot = -1;
                if (!kf->isBad()) { slot = (int32_t)S.slotKF.size(); S.slotKF.push_back(kf); S.fixedKFs.push_back(kf); }
                S.table_insert(kf, slot);
                S.obsSlot[o] = slot;
            }
        }
        S.obsStart.push_back((int32_t)S.obs.size());
    }
    // flatten: cameras / points in ascending mnId (= g2o's vertex order), edges in the reference's insertion order
    S.cams.clear();
    for (size_t i = 0; i < S.slotKF.size(); i++) { typename Scratch::CamRec c = {S.slotKF[i], i >= nLocal || S.slotKF[i]->mnId == 0, i < nLocal, (int32_t)i}; S.cams.push_back(c); }
    std::sort(S.cams.begin(), S.cams.end(), [](const typename Scratch::CamRec& a, const typename Scratch::CamRec& b) { return a.kf->mnId < b.kf->mnId; });
    const size_t nC = S.cams.size();
    S.slotCam.resize(nC);
    for (size_t i = 0; i < nC; i++) S.slotCam[S.cams[i].slot] = (int32_t)i;
    // point index = rank of the point's mnId among the window's points
    S.order.resize(nP);
    for (size_t k = 0; k < nP; k++) S.order[k] = std::make_pair((unsigned long)S.localMPs[k]->mnId, (int32_t)k);
    detail::sort_by_id(S.order, S.orderTmp);
    S.rank.resize(nP); S.pts.resize(nP);
    for (size_t i = 0; i < nP; i++) { S.rank[S.order[i].second] = (int32_t)i; S.pts[i] = S.localMPs[S.order[i].second]; }

    S.camT.resize(nC * 16); S.camFixed.resize(nC); S.camBad.resize(nC); S.xyz.resize(nP * 3);
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy, bf = pKF->mbf;
    // isBad() of an observing keyframe (upstream asks it per EDGE, src/Optimizer.cc:843: 15 000 mutex round trips on this window) is asked once per CAMERA here
    for (size_t i = 0; i < nC; i++) {
        KeyFrameT* kf = S.cams[i].kf;
        S.camBad[i] = kf->isBad() ? 1 : 0;
        const cv::Mat T = kf->GetPose();
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) S.camT[i * 16 + r * 4 + c] = T.template at<float>(r, c);
        S.camFixed[i] = S.cams[i].fixed ? 1 : 0;
        // upstream copies the intrinsics into every edge from ITS keyframe (e->fx = pKFi->fx ..., src/Optimizer.cc:858-898); the
        // C-ABI carries one set per window, which is what this fork's single-camera Frame / KeyFrame statics amount to --
        // a window that mixes cameras is refused here rather than optimised with the wrong projection
        if (!S.camBad[i] && (kf->fx != fx || kf->fy != fy || kf->cx != cx || kf->cy != cy || kf->mbf != bf))
            throw std::runtime_error("eaofusion::LocalBundleAdjustment: keyframes of the window do not share fx, fy, cx, cy, mbf");
    }
    for (size_t i = 0; i < nP; i++) detail::world_pos(S.pts[i], &S.xyz[i * 3], detail::HasWorldPosOut<MapPointT>());
    const size_t nObs = S.obs.size();
    S.eCam.resize(nObs); S.ePt.resize(nObs); S.eObs.resize(nObs * 3); S.eInv.resize(nObs);
    size_t nE = 0;
    for (size_t k = 0; k < nP; k++) {
        const int32_t pt = S.rank[k];
        for (int32_t o = S.obsStart[k]; o < S.obsStart[k + 1]; o++) {
            const int32_t slot = S.obsSlot[o];
            if (slot < 0) continue;               // a bad keyframe (upstream: `continue`)
            const int32_t ci = S.slotCam[slot];
            if (S.camBad[ci]) continue;
            KeyFrameT* kf = S.obs[o].kf;
            const size_t idx = S.obs[o].idx;
            const cv::KeyPoint& kpUn = kf->mvKeysUn[idx];
            S.eCam[nE] = ci; S.ePt[nE] = pt;
            S.eObs[3 * nE] = kpUn.pt.x; S.eObs[3 * nE + 1] = kpUn.pt.y; S.eObs[3 * nE + 2] = kf->mvuRight[idx];
            S.eInv[nE] = kf->mvInvLevelSigma2[kpUn.octave];
            nE++;
        }
    }
    if (pbStopFlag && *pbStopFlag) return;
    eao_ba_problem P;
    P.n_cams = (int)nC; P.n_points = (int)nP; P.n_edges = (int)nE;
    P.cam_Tcw = S.camT.data(); P.cam_fixed = S.camFixed.data(); P.points = S.xyz.data();
    P.edge_cam = S.eCam.data(); P.edge_point = S.ePt.data(); P.edge_obs = S.eObs.data(); P.edge_inv_sigma2 = S.eInv.data();
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf; P.its_first = 5; P.its_second = 10;
    S.camOut.resize(nC * 16); S.xyzOut.resize(nP * 3); S.erase.assign(std::max<size_t>(nE, 1), 0);
    eao_ba_result R;
    R.cam_Tcw = S.camOut.data(); R.points = S.xyzOut.data(); R.edge_outlier = S.erase.data();
    static_assert(sizeof(bool) == 1, "bool* abort flag is polled as a byte");
    check(eao_local_ba(&P, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), &R), "eao_local_ba");
    if (R.aborted) return;

    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t e = 0; e < nE; e++) {
        if (!S.erase[e]) continue;
        MapPointT* mp = S.pts[S.ePt[e]];
        if (mp->isBad()) continue;
        KeyFrameT* kf = S.cams[S.eCam[e]].kf;
        kf->EraseMapPointMatch(mp);
        mp->EraseObservation(kf);
    }
    cv::Mat pose(4, 4, CV_32F), pos(3, 1, CV_32F);      // (SetPose / SetWorldPos copy their argument, src/KeyFrame.cc:74-86, src/MapPoint.cc:68-73)
    for (size_t i = 0; i < nC; i++) {
        if (!S.cams[i].local) continue;   // only local keyframes are written back
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose.template at<float>(r, c) = S.camOut[i * 16 + r * 4 + c];
        S.cams[i].kf->SetPose(pose);
    }
    for (size_t i = 0; i < nP; i++) {
        for (int k = 0; k < 3; k++) pos.template at<float>(k) = S.xyzOut[i * 3 + k];
        S.pts[i]->SetWorldPos(pos);
        S.pts[i]->UpdateNormalAndDepth();
    }
}

// ---- Optimizer::BundleAdjustment(vpKFs, vpMP, vpMPl, nIterations, pbStopFlag, nLoopKF, bRobust) ------------------
// reference src/Optimizer.cc:55-323 over keyframes, map points and map planes: every non-bad MapPlane becomes a plane
// landmark, every observation of it by a (non-bad) keyframe of vpKFs an EdgePlane (:203-252, eao_bundle_adjustment_planes).
// (Upstream looks the keyframe of a plane observation up by id without the `mnId > maxKFid` guard of the point edges: an
// observation by a keyframe outside vpKFs could there land on a map-point vertex -- undefined behaviour; here it is skipped.)
template <class MapPointT, class KeyFrameT, class MapPlaneT>
void BundleAdjustment(const std::vector<KeyFrameT*>& vpKFs, const std::vector<MapPointT*>& vpMP, const std::vector<MapPlaneT*>& vpMPl,
                      int nIterations = 5, bool* pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true) {
    // cameras / points in ascending mnId (g2o's vertex order).  Edges: point after point in ascending mnId, a point's observations in its map's order -- upstream walks
    // vpMP, which Map::GetAllMapPoints() fills from a std::set<MapPoint*>: the order of the allocator's addresses, nothing a result may depend on; the ascending order
    // is what lets the library's map-scale set-up run its parallel passes (csrc/lm_host.hip).
    // Round 6: this walk was two std::map look-ups per edge, a map copy per point and three allocations per vertex -- 89 ms in front of a 12 ms library call on a
    // 1000-keyframe map (101 ms in all; 27 ms now); it now shares LocalBundleAdjustment's pieces (pointer table, id sort, the optional accessors of INTEGRATION.md row 2c).
    typedef LbaScratch<KeyFrameT, MapPointT> Scratch;
    static thread_local Scratch S;
    std::vector<KeyFrameT*> cams;
    for (KeyFrameT* kf : vpKFs) if (!kf->isBad()) cams.push_back(kf);
    std::sort(cams.begin(), cams.end(), [](KeyFrameT* a, KeyFrameT* b) { return a->mnId < b->mnId; });
    if (cams.empty()) return;
    size_t cap = 256;
    while (cap < 4 * cams.size()) cap *= 2;
    S.table_reset(cap);
    for (size_t i = 0; i < cams.size(); i++) S.table_insert(cams[i], (int32_t)i);
    S.order.clear();
    std::vector<MapPointT*> pts;
    for (MapPointT* mp : vpMP) if (!mp->isBad()) { S.order.push_back(std::make_pair((unsigned long)mp->mnId, (int32_t)pts.size())); pts.push_back(mp); }
    detail::sort_by_id(S.order, S.orderTmp);
    {
        std::vector<MapPointT*> sorted(pts.size());
        for (size_t i = 0; i < pts.size(); i++) sorted[i] = pts[S.order[i].second];
        pts.swap(sorted);
    }
    std::vector<float> camT(cams.size() * 16), xyz(pts.size() * 3), obs, inv;
    std::vector<uint8_t> camFixed(cams.size()), included(pts.size(), 0);
    std::vector<int32_t> eCam, ePt;
    for (size_t i = 0; i < cams.size(); i++) {
        const cv::Mat T = cams[i]->GetPose();
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) camT[i * 16 + r * 4 + c] = T.template at<float>(r, c);
        camFixed[i] = cams[i]->mnId == 0 ? 1 : 0;                       // :91
    }
    // The read side of the walk -- a lock, a position and a std::map of observations per map point, a keypoint per observation: pointer chasing through the whole map,
    // 30 ms on a 1000-keyframe map where the device call takes 9 -- runs on a few threads, each over a contiguous range of the (sorted) points: every accessor takes the
    // object's own mutex exactly as on one thread, the keyframe table is read-only by now, and the ranges' edge lists are joined in order.
    struct Append {
        std::vector<typename Scratch::Obs>* obs;
        void operator()(KeyFrameT* kf, size_t idx) { typename Scratch::Obs o = {kf, idx}; obs->push_back(o); }
    };
    struct Part { std::vector<int32_t> eCam, ePt; std::vector<float> obs, inv; std::vector<typename Scratch::Obs> seen; };
    const size_t nPts = pts.size();
    const int nWalk = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, std::thread::hardware_concurrency() / 2), nPts / 4096));      // (a thread per 4096 points: starting one costs ~50 us)
    std::vector<Part> parts((size_t)nWalk);
    const Scratch* const table = &S;      // (S is thread_local: a helper thread must look the keyframes up in the CALLER's table, not in its own empty one)
    auto walk = [&, table](int t) {
        Part& Q = parts[(size_t)t];
        const size_t i0 = nPts * (size_t)t / (size_t)nWalk, i1 = nPts * (size_t)(t + 1) / (size_t)nWalk;
        Q.eCam.reserve((i1 - i0) * 7); Q.ePt.reserve((i1 - i0) * 7); Q.obs.reserve((i1 - i0) * 21); Q.inv.reserve((i1 - i0) * 7);
        Append append = {&Q.seen};
        for (size_t i = i0; i < i1; i++) {
            detail::world_pos(pts[i], &xyz[i * 3], detail::HasWorldPosOut<MapPointT>());
            Q.seen.clear();
            detail::for_each_observation<KeyFrameT>(pts[i], append, detail::HasForEachObservation<MapPointT, KeyFrameT>());
            for (size_t o = 0; o < Q.seen.size(); o++) {      // (outside the accessor: isBad() takes the keyframe's own mutex)
                KeyFrameT* kf = Q.seen[o].kf;
                const std::pair<KeyFrameT*, int32_t>* e = table->table_find(kf);
                if (!e->first) continue;                                    // a keyframe outside vpKFs, or a bad one (:124; asked once per keyframe above, not once per
                                                                            // edge: 325 000 round trips through a thousand mutexes, from several threads at once)
                const size_t idx = Q.seen[o].idx;
                const cv::KeyPoint& kpUn = kf->mvKeysUn[idx];
                Q.eCam.push_back(e->second); Q.ePt.push_back((int32_t)i);
                Q.obs.push_back(kpUn.pt.x); Q.obs.push_back(kpUn.pt.y); Q.obs.push_back(kf->mvuRight[idx]);
                Q.inv.push_back(kf->mvInvLevelSigma2[kpUn.octave]);
                included[i] = 1;                                            // nEdges != 0, :193-201
            }
        }
    };
    if (nWalk == 1) walk(0);
    else {
        std::vector<std::thread> helpers;
        int started = 1;
        for (; started < nWalk; started++) {
            try { helpers.emplace_back(walk, started); } catch (const std::system_error&) { break; }      // (no thread to be had: the caller walks the remaining ranges itself)
        }
        walk(0);
        for (int t = started; t < nWalk; t++) walk(t);
        for (std::thread& h : helpers) h.join();
    }
    if (nWalk == 1) { eCam.swap(parts[0].eCam); ePt.swap(parts[0].ePt); obs.swap(parts[0].obs); inv.swap(parts[0].inv); }
    else {
        size_t nE = 0;
        for (const Part& Q : parts) nE += Q.eCam.size();
        eCam.reserve(nE); ePt.reserve(nE); obs.reserve(nE * 3); inv.reserve(nE);
        for (const Part& Q : parts) {
            eCam.insert(eCam.end(), Q.eCam.begin(), Q.eCam.end()); ePt.insert(ePt.end(), Q.ePt.begin(), Q.ePt.end());
            obs.insert(obs.end(), Q.obs.begin(), Q.obs.end()); inv.insert(inv.end(), Q.inv.begin(), Q.inv.end());
        }
    }
    auto cam_index = [&](KeyFrameT* kf) -> int32_t { const std::pair<KeyFrameT*, int32_t>* e = S.table_find(kf); return e->first ? e->second : -1; };
    // map planes (:210-252), in ascending mnId like their vertex ids
    std::vector<MapPlaneT*> planes;
    for (MapPlaneT* pl : vpMPl) if (pl && !pl->isBad()) planes.push_back(pl);
    std::sort(planes.begin(), planes.end(), [](MapPlaneT* a, MapPlaneT* b) { return a->mnId < b->mnId; });
    std::vector<float> plWorld(planes.size() * 4), plObs;
    std::vector<int32_t> plEdgePlane, plEdgeCam;
    for (size_t i = 0; i < planes.size(); i++) {
        const cv::Mat W = planes[i]->GetWorldPos();
        for (int k = 0; k < 4; k++) plWorld[i * 4 + k] = W.template at<float>(k);
        const auto seenBy = planes[i]->GetObservations();           // map<KeyFrame*, int>
        for (const auto& ob : seenBy) {
            KeyFrameT* kf = ob.first;
            if (kf->isBad()) continue;                                  // :231
            const int32_t ci = cam_index(kf);
            if (ci < 0) continue;                                       // optimizer.vertex(pKFi->mnId) == NULL, :234-235
            const cv::Mat& c = kf->mvPlaneCoefficients[ob.second];
            plEdgePlane.push_back((int32_t)i); plEdgeCam.push_back(ci);
            for (int k = 0; k < 4; k++) plObs.push_back(c.template at<float>(k));
        }
    }
    eao_ba_problem P;
    P.n_cams = (int)cams.size(); P.n_points = (int)pts.size(); P.n_edges = (int)eCam.size();
    P.cam_Tcw = camT.data(); P.cam_fixed = camFixed.data(); P.points = xyz.data();
    P.edge_cam = eCam.data(); P.edge_point = ePt.data(); P.edge_obs = obs.data(); P.edge_inv_sigma2 = inv.data();
    P.fx = cams[0]->fx; P.fy = cams[0]->fy; P.cx = cams[0]->cx; P.cy = cams[0]->cy; P.bf = cams[0]->mbf;
    P.its_first = nIterations; P.its_second = 0;
    eao_ba_planes PL;
    PL.n_planes = (int)planes.size(); PL.plane_world = plWorld.data(); PL.n_pedges = (int)plEdgeCam.size();
    PL.pedge_plane = plEdgePlane.data(); PL.pedge_cam = plEdgeCam.data(); PL.pedge_obs = plObs.data();
    std::vector<float> camOut(camT.size()), xyzOut(xyz.size()), plOut(plWorld.size());
    eao_ba_result R;
    R.cam_Tcw = camOut.data(); R.points = xyzOut.data(); R.edge_outlier = nullptr;
    static_assert(sizeof(bool) == 1, "bool* abort flag is polled as a byte");
    check(eao_bundle_adjustment_planes(&P, planes.empty() ? nullptr : &PL, bRobust ? 1 : 0, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), &R,
                                       plOut.data()), "eao_bundle_adjustment_planes");
    cv::Mat T(4, 4, CV_32F), X(3, 1, CV_32F);                           // (SetPose / SetWorldPos / copyTo copy their argument)
    for (size_t i = 0; i < cams.size(); i++) {                          // :258-275
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.template at<float>(r, c) = camOut[i * 16 + r * 4 + c];
        if (nLoopKF == 0) {
            cams[i]->SetPose(T);
        } else {
            cams[i]->mTcwGBA.create(4, 4, CV_32F);
            T.copyTo(cams[i]->mTcwGBA);
            cams[i]->mnBAGlobalForKF = nLoopKF;
        }
    }
    for (size_t i = 0; i < pts.size(); i++) {                           // :277-300
        if (!included[i]) continue;
        for (int k = 0; k < 3; k++) X.template at<float>(k) = xyzOut[i * 3 + k];
        if (nLoopKF == 0) {
            pts[i]->SetWorldPos(X);
            pts[i]->UpdateNormalAndDepth();
        } else {
            pts[i]->mPosGBA.create(3, 1, CV_32F);
            X.copyTo(pts[i]->mPosGBA);
            pts[i]->mnBAGlobalForKF = nLoopKF;
        }
    }
    for (size_t i = 0; i < planes.size(); i++) {                        // :303-322
        cv::Mat C4(4, 1, CV_32F);
        for (int k = 0; k < 4; k++) C4.template at<float>(k) = plOut[i * 4 + k];
        if (nLoopKF == 0) {
            planes[i]->SetWorldPos(C4);
        } else {
            planes[i]->mPosGBA.create(4, 1, CV_32F);
            C4.copyTo(planes[i]->mPosGBA);
            planes[i]->mnBAGlobalForKF = nLoopKF;
        }
    }
}

// Optimizer::GlobalBundleAdjustemnt(Map*, nIterations, pbStopFlag, nLoopKF, bRobust) -- src/Optimizer.cc:47-53
template <class MapPointT, class MapT>
void GlobalBundleAdjustemnt(MapT* pMap, int nIterations = 5, bool* pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true) {
    BundleAdjustment<MapPointT>(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), pMap->GetAllMapPlanes(), nIterations, pbStopFlag, nLoopKF, bRobust);
}

}  // namespace eaofusion
