// Drives eaofusion::DeviceTracker (include/eaofusion/DeviceTracker.h) against stand-ins of the reference's Frame / MapPoint
// and compares what it leaves in the frame -- mvuRight, mvDepth, mvpMapPoints, mvbOutlier, the pose -- with a direct call of
// the C-ABI on the same arrays.  Device buffers come from the HIP runtime (this test is built with hipcc).
// Exit code 0 = everything agrees.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include <eaofusion/DeviceTracker.h>

struct MapPoint {
    cv::Mat pos, normal, desc;
    bool bad = false, mbTrackInView = false;
    int nVisible = 0;
    long unsigned int mnLastFrameSeen = ~0ul;
    int nObs = 1;
    int Observations() { return nObs; }
    void IncreaseVisible(int n = 1) { nVisible += n; }
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    cv::Mat GetDescriptor() { return desc.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    void SetDistances(float mn, float mx) { mfMinDistance = mn; mfMaxDistance = mx; }
protected:
    float mfMinDistance = 0, mfMaxDistance = 0;
};

struct Frame {
    int N = 0;
    long unsigned int mnId = 9;
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mbf = 40.f, mfLogScaleFactor = std::log(1.2f);
    int mnScaleLevels = 8;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2;
    cv::Mat mTcw;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;      // DBoW2::FeatureVector
    void SetPose(cv::Mat T) { mTcw = T.clone(); }
    cv::Mat mDistCoef;                                        // upstream include/Frame.h:216 (empty / all zero: no distortion)
};

struct KeyFrame {      // what TrackReferenceKeyFrame reads of the reference keyframe
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
};
float Frame::fx = 535.4f, Frame::fy = 539.2f, Frame::cx = 320.1f, Frame::cy = 247.6f;
float Frame::mnMinX = 0.f, Frame::mnMaxX = 640.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 480.f;

static unsigned long long g_s = 88172645463325252ull;
static double rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) / 9007199254740992.0; }
#define HIPCHK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s\n", #x); return 2; } } while (0)

int main() {
    const int M = 700, cap = 1024, W = 640, H = 480;
    Frame F;
    for (int l = 0; l < 8; l++) { F.mvScaleFactors.push_back(l ? F.mvScaleFactors[l - 1] * 1.2f : 1.0f); F.mvInvLevelSigma2.push_back(1.0f / (F.mvScaleFactors[l] * F.mvScaleFactors[l])); }
    F.mTcw = cv::Mat::eye(4, 4, CV_32F);
    F.mTcw.at<float>(0, 3) = 0.02f; F.mTcw.at<float>(2, 3) = -0.03f;
    std::vector<MapPoint> pts(M);
    std::vector<MapPoint*> local;
    std::vector<float> depth((size_t)W * H, 0.f);
    std::vector<unsigned char> desc;
    std::vector<int> seen;                 // keypoint k observes map point seen[k]
    for (int m = 0; m < M; m++) {
        MapPoint& p = pts[m];
        const float z = 2.f + 4.f * (float)rnd(), x = (float)(rnd() - 0.5) * z * 0.9f, y = (float)(rnd() - 0.5) * z * 0.7f;
        p.pos = cv::Mat(3, 1, CV_32F); p.pos.at<float>(0) = x; p.pos.at<float>(1) = y; p.pos.at<float>(2) = z;
        const float nrm = std::sqrt(x * x + y * y + z * z);
        p.normal = cv::Mat(3, 1, CV_32F); p.normal.at<float>(0) = x / nrm; p.normal.at<float>(1) = y / nrm; p.normal.at<float>(2) = z / nrm;
        const int oct = (int)(rnd() * 7.99);
        p.SetDistances(0.7f * nrm, nrm * std::pow(1.2f, oct - 0.5f));
        p.desc = cv::Mat(1, 32, CV_8U);
        for (int b = 0; b < 32; b++) p.desc.at<unsigned char>(0, b) = (unsigned char)(rnd() * 256);
        p.bad = rnd() < 0.03;
        local.push_back(&p);
        // the keypoint that observes it: projection + noise, the same descriptor with a few bits flipped
        const float xc = x + 0.02f, zc = z - 0.03f;
        const float u = Frame::fx * xc / zc + Frame::cx + (float)(rnd() - 0.5) * 2.f, v = Frame::fy * y / zc + Frame::cy + (float)(rnd() - 0.5) * 2.f;
        if (u < 2 || u > W - 3 || v < 2 || v > H - 3) continue;
        cv::KeyPoint kp(u, v, 31.f, (float)(rnd() * 360.0), 50.f, oct);
        F.mvKeys.push_back(kp);
        seen.push_back(m);
        for (int b = 0; b < 32; b++) desc.push_back((unsigned char)(p.desc.at<unsigned char>(0, b) ^ (rnd() < 0.1 ? 1 << (int)(rnd() * 8) : 0)));
        depth[(size_t)(int)v * W + (int)u] = rnd() < 0.2 ? 0.f : zc;
    }
    F.N = (int)F.mvKeys.size();
    F.mvpMapPoints.assign(F.N, nullptr);
    F.mvbOutlier.assign(F.N, false);
    F.mvpMapPoints[3] = local[3 < M ? 3 : 0];        // a match the frame already carries
    //  a prior match that has gone bad: upstream's SearchLocalPoints sets it to NULL and the keypoint is free again;
    // and a prior match that is NOT in the local map: it stays, keeps its keypoint and is an edge of the pose optimisation
    int badIdx = -1;
    for (int m = 0; m < M && badIdx < 0; m++) if (pts[m].bad && m != 3) badIdx = m;
    if (badIdx < 0) { fprintf(stderr, "no bad point in the scene\n"); return 2; }
    F.mvpMapPoints[5] = local[badIdx];
    MapPoint outsider;
    {
        const cv::KeyPoint& kp = F.mvKeys[7];
        const float z = 3.5f, xc = (kp.pt.x - Frame::cx) * z / Frame::fx, yc = (kp.pt.y - Frame::cy) * z / Frame::fy;
        outsider.pos = cv::Mat(3, 1, CV_32F); outsider.pos.at<float>(0) = xc - 0.02f; outsider.pos.at<float>(1) = yc; outsider.pos.at<float>(2) = z + 0.03f;
        outsider.normal = outsider.pos.clone(); outsider.desc = cv::Mat(1, 32, CV_8U);
    }
    F.mvpMapPoints[7] = &outsider;
    eao_keypoint* d_kps; uint8_t* d_desc; int32_t* d_n; float* d_depth;
    HIPCHK(hipMalloc((void**)&d_kps, cap * sizeof(eao_keypoint))); HIPCHK(hipMalloc((void**)&d_desc, cap * 32)); HIPCHK(hipMalloc((void**)&d_n, 4));
    HIPCHK(hipMalloc((void**)&d_depth, depth.size() * 4));
    HIPCHK(hipMemset(d_kps, 0, cap * sizeof(eao_keypoint))); HIPCHK(hipMemset(d_desc, 0, cap * 32));
    HIPCHK(hipMemcpy(d_kps, F.mvKeys.data(), (size_t)F.N * sizeof(eao_keypoint), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_desc, desc.data(), desc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_n, &F.N, 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
    Frame F2 = F;                                      // for the direct C-ABI call below
    eaofusion::DeviceTracker trk(F, cap, 1024);
    if (!trk.SetLocalMap(local)) { fprintf(stderr, "SetLocalMap refused %d points\n", M); return 2; }
    {
        std::vector<MapPoint*> tooMany(1025, local[0]);
        if (trk.SetLocalMap(tooMany)) { fprintf(stderr, "SetLocalMap accepted a map beyond the handle's capacity\n"); return 2; }
    }
    const int nIn = trk.TrackLocalMap(F, d_kps, d_desc, d_n, d_depth, W, W, H, 3.0f, 0.8f, nullptr);
    // ---- the same through the C-ABI, arrays built by hand
    eao_tracker_cfg c = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, F2.mbf, 0.f, 640.f, 0.f, 480.f, 64, 48, 8, F2.mvScaleFactors.data(), F2.mvInvLevelSigma2.data(),
                         F2.mfLogScaleFactor, cap, 1024};
    eao_tracker* h = nullptr;
    if (eao_tracker_create(&c, &h) != EAO_OK) { fprintf(stderr, "%s\n", eao_last_error()); return 2; }
    std::vector<unsigned char> act(M), dsc(32 * (size_t)M);
    std::vector<float> Xw(3 * (size_t)M), nr(3 * (size_t)M), dmin(M), dmax(M), draw(M);
    for (int m = 0; m < M; m++) {
        act[m] = pts[m].bad ? 0 : 1;
        for (int a = 0; a < 3; a++) { Xw[3 * m + a] = pts[m].pos.at<float>(a); nr[3 * m + a] = pts[m].normal.at<float>(a); }
        dmin[m] = pts[m].GetMinDistanceInvariance(); dmax[m] = pts[m].GetMaxDistanceInvariance(); draw[m] = dmax[m] / 1.2f;
        memcpy(&dsc[32 * (size_t)m], pts[m].desc.ptr<unsigned char>(0), 32);
    }
    // (mfMaxDistance itself: GetMaxDistanceInvariance() / 1.2f does not round-trip -- read it the way the adapter does)
    for (int m = 0; m < M; m++) draw[m] = eaofusion::detail::MaxDistanceOf<MapPoint>::get(&pts[m]);
    eao_map_points mp = {M, act.data(), Xw.data(), nr.data(), dmin.data(), dmax.data(), draw.data(), dsc.data()};
    if (eao_tracker_set_local_map(h, &mp) != EAO_OK) { fprintf(stderr, "%s\n", eao_last_error()); return 2; }
    std::vector<int32_t> prior(cap, -1), kpMp(cap, -1);
    prior[3] = 3;
    prior[5] = badIdx;            // the library itself drops a prior on an inactive point
    prior[7] = -2;                // outside the local map: position handed over
    std::vector<float> priorXw(3 * (size_t)cap, 0.f);
    for (int a = 0; a < 3; a++) priorXw[3 * 7 + a] = outsider.pos.at<float>(a);
    std::vector<uint8_t> inView(1024, 0);
    std::vector<uint8_t> outl(cap);
    std::vector<float> ur(cap), dz(cap);
    eao_track_result R = {};
    R.kp_map_point = kpMp.data(); R.kp_outlier = outl.data(); R.kp_u_right = ur.data(); R.kp_depth = dz.data(); R.map_in_view = inView.data();
    float T[16];
    for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) T[4 * r + k] = F2.mTcw.at<float>(r, k);
    if (eao_tracker_track_local_map(h, d_kps, d_desc, d_n, d_depth, W, W, H, T, prior.data(), priorXw.data(), 3.0f, 0.8f, &R, nullptr) != EAO_OK) { fprintf(stderr, "%s\n", eao_last_error()); return 2; }
    {   // a stale prior table (an index beyond the uploaded map) is refused before any kernel runs
        std::vector<int32_t> stale(prior);
        stale[11] = M;
        eao_track_result R2 = R;
        std::vector<int32_t> k2(cap); std::vector<uint8_t> o2(cap);
        R2.kp_map_point = k2.data(); R2.kp_outlier = o2.data(); R2.kp_u_right = nullptr; R2.kp_depth = nullptr; R2.map_in_view = nullptr;
        if (eao_tracker_track_local_map(h, d_kps, d_desc, d_n, d_depth, W, W, H, T, stale.data(), priorXw.data(), 3.0f, 0.8f, &R2, nullptr) != EAO_ERR_INVALID) {
            fprintf(stderr, "a prior index beyond the local map was accepted\n");
            return 2;
        }
    }
    int bad = 0, matched = 0;
    if (nIn != R.n_inliers) { fprintf(stderr, "inliers %d vs %d\n", nIn, R.n_inliers); bad++; }
    // SearchLocalPoints' bookkeeping (src/Tracking.cc:2590-2627)
    if (kpMp[5] == badIdx || F.mvpMapPoints[5] == local[badIdx]) { fprintf(stderr, "the bad prior match survived\n"); bad++; }
    if (kpMp[7] != -2 || F.mvpMapPoints[7] != &outsider) { fprintf(stderr, "the prior match outside the local map was lost\n"); bad++; }
    if (outsider.nVisible != 1 || outsider.mnLastFrameSeen != F.mnId || local[3]->nVisible != 1 || local[3]->mnLastFrameSeen != F.mnId) { fprintf(stderr, "prior-match bookkeeping\n"); bad++; }
    if (local[badIdx]->nVisible != 0) { fprintf(stderr, "a bad point was counted visible\n"); bad++; }
    int nView = 0;
    for (int m = 0; m < M; m++) {
        if (m == 3 || pts[m].bad) continue;
        nView += inView[m];
        if (pts[m].nVisible != (int)inView[m] || pts[m].mbTrackInView != (inView[m] != 0)) bad++;
    }
    if (nView < 100) { fprintf(stderr, "only %d points in view\n", nView); bad++; }
    for (int k = 0; k < F.N; k++) {
        MapPoint* want = kpMp[k] >= 0 ? &pts[kpMp[k]] : (kpMp[k] == -2 ? &outsider : nullptr);
        if (F.mvpMapPoints[k] != want) bad++;
        if (F.mvbOutlier[k] != (outl[k] != 0)) bad++;
        if (F.mvuRight[k] != ur[k] || F.mvDepth[k] != dz[k]) bad++;
        matched += want ? 1 : 0;
    }
    for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) if (F.mTcw.at<float>(r, k) != R.Tcw[4 * r + k]) bad++;
    // ---- TrackWithMotionModel (round 4): the LAST frame saw the same keypoints (every one with its map point, a few flagged as outliers, one a
    //      temporal point without observations); the current frame enters with the predicted pose and empty matches.  Adapter vs a direct C-ABI call.
    {
        Frame Last = F2, Cur = F2;
        Last.mvKeysUn = Last.mvKeys; Cur.mvKeysUn = Cur.mvKeys;
        Last.mnId = 8; Cur.mnId = 9;
        for (int k = 0; k < Last.N; k++) { Last.mvpMapPoints[k] = &pts[seen[k]]; Last.mvbOutlier[k] = (k % 17) == 0; }
        Last.mvpMapPoints[2] = nullptr;
        pts[seen[4]].nObs = 0;
        Cur.mvpMapPoints.assign(Cur.N, nullptr); Cur.mvbOutlier.assign(Cur.N, false);
        Cur.mTcw = F2.mTcw.clone(); Cur.mTcw.at<float>(0, 3) += 0.004f;
        for (int k = 0; k < Cur.N; k++) Cur.mvKeysUn[k].angle = Cur.mvKeys[k].angle = std::fmod(Last.mvKeys[k].angle + 9.f, 360.f);
        std::vector<eao_keypoint> ck(Cur.N);
        for (int k = 0; k < Cur.N; k++) std::memcpy(&ck[k], &Cur.mvKeys[k], sizeof(eao_keypoint));
        HIPCHK(hipMemcpy(d_kps, ck.data(), (size_t)Cur.N * sizeof(eao_keypoint), hipMemcpyHostToDevice));
        Frame Cur2 = Cur;
        int nMap = -1, nSearch = -1;
        const int nm = trk.TrackWithMotionModel(Cur, Last, d_kps, d_desc, d_n, d_depth, W, W, H, 15.f, false, nullptr, &nMap, &nSearch);
        // direct call, arrays by hand, the library's own discard
        const int nl = Last.N;
        std::vector<uint8_t> valid(nl, 0), ldesc(32 * (size_t)nl, 0), o2(cap, 0);
        std::vector<float> lX(3 * (size_t)nl, 0.f), lang(nl);
        std::vector<int32_t> loct(nl), k2(cap, -1);
        for (int i = 0; i < nl; i++) {
            loct[i] = Last.mvKeys[i].octave; lang[i] = Last.mvKeysUn[i].angle;
            MapPoint* q = Last.mvpMapPoints[i];
            if (!q || Last.mvbOutlier[i]) continue;
            valid[i] = 1;
            for (int a = 0; a < 3; a++) lX[3 * i + a] = q->pos.at<float>(a);
            memcpy(&ldesc[32 * (size_t)i], q->desc.ptr<unsigned char>(0), 32);
        }
        float Tc[16], Tl[16];
        for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) { Tc[4 * r + k] = Cur2.mTcw.at<float>(r, k); Tl[4 * r + k] = Last.mTcw.at<float>(r, k); }
        eao_track_result R3 = {};
        std::vector<float> ur3(cap), dz3(cap);
        R3.kp_map_point = k2.data(); R3.kp_outlier = o2.data(); R3.kp_u_right = ur3.data(); R3.kp_depth = dz3.data();
        if (eao_tracker_track_with_motion_model(h, d_kps, d_desc, d_n, d_depth, W, W, H, Tc, Tl, nl, valid.data(), lX.data(), ldesc.data(), loct.data(), lang.data(),
                                                15.f, 0, 1, 1, &R3, nullptr) != EAO_OK) { fprintf(stderr, "%s\n", eao_last_error()); return 2; }
        int kept = 0, withObs = 0, mmBad = 0;
        for (int k = 0; k < Cur.N; k++) {
            MapPoint* want = k2[k] >= 0 ? Last.mvpMapPoints[k2[k]] : nullptr;
            if (Cur.mvpMapPoints[k] != want) mmBad++;
            if (Cur.mvbOutlier[k]) mmBad++;
            if (Cur.mvuRight[k] != ur3[k] || Cur.mvDepth[k] != dz3[k]) mmBad++;
            if (want) { kept++; withObs += want->Observations() > 0 ? 1 : 0; }
        }
        for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) if (Cur.mTcw.at<float>(r, k) != R3.Tcw[4 * r + k]) mmBad++;
        if (nm != kept || nm != R3.n_inliers || nMap != withObs || nSearch != R3.n_matches) { fprintf(stderr, "motion model counts: %d kept %d / %d, map %d / %d, search %d / %d\n", nm, kept, R3.n_inliers, nMap, withObs, nSearch, R3.n_matches); mmBad++; }
        if (kept < 100) { fprintf(stderr, "motion model: only %d matches kept\n", kept); mmBad++; }
        fprintf(stderr, "motion model: %d matches after the search, %d kept, %d with observations, %d disagreements\n", nSearch, kept, withObs, mmBad);
        bad += mmBad;
        pts[seen[4]].nObs = 1;
    }
    // ---- TrackReferenceKeyFrame (round 4): the reference keyframe saw the same points (descriptors a few bits off, its own keypoint order, one point bad, one
    //      keypoint without a point); both feature vectors file a keypoint under (map point mod 37), a tenth of them elsewhere.  Adapter vs a direct C-ABI call.
    {
        Frame Cur = F2;
        Cur.mvKeysUn = Cur.mvKeys; Cur.mnId = 11;
        Cur.mvpMapPoints.assign(Cur.N, nullptr); Cur.mvbOutlier.assign(Cur.N, false);
        std::vector<eao_keypoint> ck(Cur.N);
        for (int k = 0; k < Cur.N; k++) std::memcpy(&ck[k], &Cur.mvKeys[k], sizeof(eao_keypoint));
        HIPCHK(hipMemcpy(d_kps, ck.data(), (size_t)Cur.N * sizeof(eao_keypoint), hipMemcpyHostToDevice));
        KeyFrame KF;
        const int nk = Cur.N;
        KF.mDescriptors = cv::Mat(nk, 32, CV_8U);
        for (int i = 0; i < nk; i++) {
            const int k = (int)(((long long)i * 11 + 3) % nk);      // keyframe keypoint i observes what frame keypoint k observes (its own keypoint order)
            MapPoint* q = &pts[seen[k]];
            KF.mvpMapPoints.push_back(i == 6 ? nullptr : q);
            cv::KeyPoint kp = Cur.mvKeys[k]; kp.angle = std::fmod(kp.angle + 351.f, 360.f);
            KF.mvKeysUn.push_back(kp);
            for (int b = 0; b < 32; b++) KF.mDescriptors.ptr(i)[b] = (unsigned char)(q->desc.at<unsigned char>(0, b) ^ (rnd() < 0.08 ? 1 << (int)(rnd() * 8) : 0));
            KF.mFeatVec[(unsigned)(rnd() < 0.1 ? (int)(rnd() * 37) : seen[k] % 37)].push_back((unsigned)i);
        }
        for (int k = 0; k < Cur.N; k++) Cur.mFeatVec[(unsigned)(rnd() < 0.1 ? (int)(rnd() * 37) : seen[k] % 37)].push_back((unsigned)k);
        pts[seen[10]].nObs = 0;
        cv::Mat lastT = F2.mTcw.clone(); lastT.at<float>(1, 3) += 0.006f;
        int nSearch = -1;
        const int nMap = trk.TrackReferenceKeyFrame(Cur, &KF, lastT, d_kps, d_desc, d_n, d_depth, W, W, H, nullptr, &nSearch);
        // direct call, arrays by hand, the library's own discard
        std::vector<uint8_t> valid(nk, 0), kdesc(32 * (size_t)nk, 0), o2(cap, 0);
        std::vector<float> kX(3 * (size_t)nk, 0.f), kang(nk);
        std::vector<int32_t> k2(cap, -1);
        for (int i = 0; i < nk; i++) {
            kang[i] = KF.mvKeysUn[i].angle;
            memcpy(&kdesc[32 * (size_t)i], KF.mDescriptors.ptr(i), 32);
            MapPoint* q = KF.mvpMapPoints[i];
            if (!q || q->bad) continue;
            valid[i] = 1;
            for (int a = 0; a < 3; a++) kX[3 * i + a] = q->pos.at<float>(a);
        }
        auto flat = [](const std::map<unsigned, std::vector<unsigned> >& fv, std::vector<uint32_t>& id, std::vector<int32_t>& st, std::vector<uint32_t>& ix) {
            st.assign(1, 0);
            for (const auto& kv : fv) { id.push_back(kv.first); for (unsigned v : kv.second) ix.push_back(v); st.push_back((int32_t)ix.size()); }
            eao_feature_vector f; f.n_nodes = (int32_t)id.size(); f.node_id = id.data(); f.node_start = st.data(); f.index = ix.data();
            return f;
        };
        std::vector<uint32_t> idK, ixK, idC, ixC; std::vector<int32_t> stK, stC;
        const eao_feature_vector fvK = flat(KF.mFeatVec, idK, stK, ixK), fvC = flat(Cur.mFeatVec, idC, stC, ixC);
        float Tl[16];
        for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) Tl[4 * r + k] = lastT.at<float>(r, k);
        eao_track_result R4 = {};
        std::vector<float> ur4(cap), dz4(cap);
        R4.kp_map_point = k2.data(); R4.kp_outlier = o2.data(); R4.kp_u_right = ur4.data(); R4.kp_depth = dz4.data();
        if (eao_tracker_track_reference_keyframe(h, d_kps, d_desc, d_n, d_depth, W, W, H, Tl, nk, valid.data(), kX.data(), kdesc.data(), kang.data(), &fvK, &fvC, 0.7f, 1, 1, &R4,
                                                 nullptr) != EAO_OK) { fprintf(stderr, "%s\n", eao_last_error()); return 2; }
        int kept = 0, withObs = 0, rkBad = 0;
        for (int k = 0; k < Cur.N; k++) {
            MapPoint* want = k2[k] >= 0 ? KF.mvpMapPoints[k2[k]] : nullptr;
            if (Cur.mvpMapPoints[k] != want) rkBad++;
            if (Cur.mvbOutlier[k]) rkBad++;
            if (Cur.mvuRight[k] != ur4[k] || Cur.mvDepth[k] != dz4[k]) rkBad++;
            if (want) { kept++; withObs += want->Observations() > 0 ? 1 : 0; }
        }
        for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) if (Cur.mTcw.at<float>(r, k) != R4.Tcw[4 * r + k]) rkBad++;
        if (kept != R4.n_inliers || nMap != withObs || nSearch != R4.n_matches) { fprintf(stderr, "reference keyframe counts: kept %d / %d, map %d / %d, search %d / %d\n", kept, R4.n_inliers, nMap, withObs, nSearch, R4.n_matches); rkBad++; }
        if (kept < 100) { fprintf(stderr, "reference keyframe: only %d matches kept\n", kept); rkBad++; }
        fprintf(stderr, "reference keyframe: %d matches after the search, %d kept, %d with observations, %d disagreements\n", nSearch, kept, withObs, rkBad);
        bad += rkBad;
        pts[seen[10]].nObs = 1;
    }
    {   // a camera with lens distortion (TUM1: ros_test/config/TUM1.yaml:13-17): the adapter hands mDistCoef to the handle, the chain undistorts the keypoints itself and
        // reads mvKeysUn where upstream does -- mvuRight must be (undistorted column) - mbf / depth(at the DISTORTED pixel), with the undistortion of eaofusion::UndistortKeyPoints
        Frame D = F;
        D.mDistCoef = cv::Mat(5, 1, CV_32F);
        const float tum1[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
        for (int i = 0; i < 5; i++) D.mDistCoef.at<float>(i) = tum1[i];
        eaofusion::UndistortKeyPoints(D);
        float moved = 0.f;
        for (int k = 0; k < D.N; k++) moved = std::max(moved, std::fabs(D.mvKeysUn[k].pt.x - D.mvKeys[k].pt.x));
        if (!(moved > 1.0f)) { fprintf(stderr, "UndistortKeyPoints moved no keypoint (max %.3f px)\n", moved); bad++; }
        eaofusion::DeviceTracker trkD(D, cap, 1024);
        if (!trkD.Fits(D)) { fprintf(stderr, "Fits() is false for a distorted camera\n"); bad++; }
        trkD.SetLocalMap(local);
        for (int k = 0; k < D.N; k++) D.mvpMapPoints[k] = nullptr;
        trkD.TrackLocalMap(D, d_kps, d_desc, d_n, d_depth, W, W, H, 3.0f, 0.8f, nullptr);
        int dBad = 0, stereo = 0;
        for (int k = 0; k < D.N; k++) {
            const float dpt = depth[(size_t)(int)D.mvKeys[k].pt.y * W + (int)D.mvKeys[k].pt.x];
            const float want = dpt > 0 ? D.mvKeysUn[k].pt.x - D.mbf / dpt : -1.0f;
            if (D.mvuRight[k] != want) dBad++;
            stereo += dpt > 0;
        }
        if (dBad || stereo < 100) { fprintf(stderr, "distorted camera: %d of %d mvuRight entries differ from the undistorted column - mbf / depth\n", dBad, D.N); bad++; }
        Frame Z = F;                                              // k1 == 0: upstream copies mvKeys whatever the other coefficients say (src/Frame.cc:775-779)
        Z.mDistCoef = cv::Mat(4, 1, CV_32F);
        for (int i = 0; i < 4; i++) Z.mDistCoef.at<float>(i) = 0.f;
        Z.mDistCoef.at<float>(3) = -0.001f;
        eaofusion::UndistortKeyPoints(Z);
        for (int k = 0; k < Z.N; k++) if (Z.mvKeysUn[k].pt.x != Z.mvKeys[k].pt.x || Z.mvKeysUn[k].pt.y != Z.mvKeys[k].pt.y) { bad++; break; }
        fprintf(stderr, "distorted camera: keypoints moved by up to %.2f px, %d stereo entries checked, %d disagreements\n", moved, stereo, dBad);
    }
    eao_tracker_destroy(h);
    fprintf(stderr, "%d keypoints, %d with a map point, %d inliers, %d disagreements\n", F.N, matched, nIn, bad);
    return (bad == 0 && matched > 100 && nIn > 50) ? 0 : 1;
}
