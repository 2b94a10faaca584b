// adapter_test.cpp -- drives the header-only drop-in adapters (include/eaofusion/*.h) the way Tracking.cc /
// LocalMapping.cc drive the reference classes, against small stand-ins of the reference's Frame / KeyFrame /
// MapPoint / Map (same member names the adapters touch).  Built and run by tests/test_gpu_adapters.py.
//   adapter_test <problem.bin> <result.bin>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <vector>

#include <eaofusion/ORBextractor.h>
#include <eaofusion/ORBmatcher.h>
#include <eaofusion/OptimizerImpl.h>

struct KeyFrame;
struct MapPoint {
    static std::mutex mGlobalMutex;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul;
    // tracking members read by ORBmatcher::SearchByProjection
    bool mbTrackInView = true;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    cv::Mat descriptor;
    int nObs = 1;
    cv::Mat GetDescriptor() { return descriptor.clone(); }
    int Observations() { return nObs; }
    cv::Mat pos;
    std::map<KeyFrame*, size_t> observations;
    int normalUpdates = 0;
    cv::Mat mPosGBA;                       // Optimizer::BundleAdjustment with nLoopKF != 0
    long unsigned int mnBAGlobalForKF = 0;
    bool isBad() { return false; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    void SetWorldPos(const cv::Mat& p) { pos = p.clone(); }
    std::map<KeyFrame*, size_t> GetObservations() { return observations; }
    void EraseObservation(KeyFrame* kf) { observations.erase(kf); }
    void UpdateNormalAndDepth() { normalUpdates++; }
};
std::mutex MapPoint::mGlobalMutex;

struct KeyFrame {
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul;
    float fx, fy, cx, cy, mbf;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<KeyFrame*> covisible;
    cv::Mat Tcw, mTcwGBA;
    std::vector<cv::Mat> mvPlaneCoefficients;   // upstream include/KeyFrame.h:261
    long unsigned int mnBAGlobalForKF = 0;
    int erased = 0;
    bool isBad() { return false; }
    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { Tcw = T.clone(); }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    void EraseMapPointMatch(MapPoint* mp) {
        for (auto& p : mvpMapPoints) if (p == mp) { p = nullptr; erased++; }
    }
};
struct Map { std::mutex mMutexMapUpdate; };
struct MapPlane {
    static std::mutex mGlobalMutex;
    cv::Mat world;
    bool mbSeen = false, bad = false;
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return world.clone(); }
    // Optimizer::BundleAdjustment (upstream include/MapPlane.h:32-66)
    long unsigned int mnId = 0, mnBAGlobalForKF = 0;
    std::map<KeyFrame*, int> observations;
    cv::Mat mPosGBA;
    std::map<KeyFrame*, int> GetObservations() { return observations; }
    void SetWorldPos(const cv::Mat& p) { world = p.clone(); }
};
std::mutex MapPlane::mGlobalMutex;
struct Frame {
    int N = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    std::vector<float> mvScaleFactors;
    float mb = 0;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    float fx, fy, cx, cy, mbf;
    void SetPose(cv::Mat T) { mTcw = T.clone(); }
    // associated map planes (upstream include/Frame.h)
    int mnPlaneNum = 0;
    std::vector<MapPlane*> mvpMapPlanes;
    std::vector<cv::Mat> mvPlaneCoefficients;
    std::vector<bool> mvbPlaneOutlier;
};

float Frame::mnMinX = 0, Frame::mnMaxX = 640, Frame::mnMinY = 0, Frame::mnMaxY = 480;
float Frame::mfGridElementWidthInv = 64.f / 640.f, Frame::mfGridElementHeightInv = 48.f / 480.f;

template <typename T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), n * sizeof(T)); }
template <typename T> static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), n * sizeof(T)); }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    // ---------------------------------------------------------------- ORBextractor, as Frame::ExtractORB calls it
    int32_t H, W;
    rd(in, &H, 1); rd(in, &W, 1);
    cv::Mat img(H, W, CV_8UC1);
    rd(in, img.data, (size_t)H * W);
    ORB_SLAM2::ORBextractor extractor(1000, 1.2f, 8, 20, 7);
    extractor.keepPyramid = false;      // the on-demand form first (the default, keepPyramid = true, is the `eager` extractor below)
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    extractor(img, cv::Mat(), keys, descriptors);
    int32_t nk = (int32_t)keys.size();
    wr(out, &nk, 1);
    wr(out, keys.data(), keys.size());
    for (int i = 0; i < nk; i++) wr(out, descriptors.ptr(i), 32);
    // mvImagePyramid on demand: every level a w x h view with the 19 px BORDER_REFLECT_101 frame physically around it (src/ORBextractor.cc:1113-1128) --
    // checked pixel by pixel against the tight level images (eao_orb_level) reflected here
    int32_t lazyEmpty = extractor.mvImagePyramid[0].empty() ? 1 : 0, borderMismatch = 0;
    std::vector<cv::Mat>& pyrd = extractor.ImagePyramid();
    for (int l = 0; l < 8; l++) {
        int w = 0, h = 0;
        eao_orb_level(extractor.handle(), 0, l, 0, &w, &h, nullptr);
        std::vector<unsigned char> tight((size_t)w * h);
        eao_orb_level(extractor.handle(), 0, l, 0, nullptr, nullptr, tight.data());
        if (pyrd[l].cols != w || pyrd[l].rows != h) { borderMismatch++; continue; }
        for (int y = -19; y < h + 19; y++) {
            const int sy = y < 0 ? -y : (y >= h ? 2 * h - 2 - y : y);
            const unsigned char* row = pyrd[l].data + (long)y * (long)pyrd[l].step;
            for (int x = -19; x < w + 19; x++) {
                const int sx = x < 0 ? -x : (x >= w ? 2 * w - 2 - x : x);
                if (row[x] != tight[(size_t)sy * w + sx]) borderMismatch++;
            }
        }
    }
    int32_t pyr0[5] = {extractor.mvImagePyramid[0].rows, extractor.mvImagePyramid[0].cols, (int32_t)extractor.mvImagePyramid[7].cols, lazyEmpty, borderMismatch};
    wr(out, pyr0, 5);
    {   // keepPyramid = true, the default: the member is filled by operator() itself, as upstream's is
        ORB_SLAM2::ORBextractor eager(1000, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> k2;
        cv::Mat d2;
        eager(img, cv::Mat(), k2, d2);
        int32_t same = (k2.size() == keys.size() && !eager.mvImagePyramid[3].empty() && eager.mvImagePyramid[3].cols == pyrd[3].cols) ? 1 : 0;
        for (int y = -19; same && y < pyrd[3].rows + 19; y++)
            if (std::memcmp(eager.mvImagePyramid[3].data + (long)y * (long)eager.mvImagePyramid[3].step - 19, pyrd[3].data + (long)y * (long)pyrd[3].step - 19, pyrd[3].cols + 38)) same = 0;
        wr(out, &same, 1);
    }
    int32_t dd = eaofusion::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1));
    wr(out, &dd, 1);
    cv::Mat empty;
    std::vector<cv::KeyPoint> untouched(3);
    extractor(empty, cv::Mat(), untouched, descriptors);   // empty image: outputs untouched
    int32_t stillThree = (int32_t)untouched.size();
    wr(out, &stillThree, 1);
    // ---------------------------------------------------------------- ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)
    {
        Frame TF;                                   // the extracted keypoints ARE the frame; each becomes a map point to re-find
        TF.N = nk; TF.mvKeysUn = keys; TF.mvKeys = keys; TF.mDescriptors = descriptors.clone();
        TF.mvuRight.assign(nk, -1.f); TF.mvpMapPoints.assign(nk, nullptr); TF.mvbOutlier.assign(nk, false);
        TF.mvScaleFactors = extractor.GetScaleFactors();
        std::vector<MapPoint> tmps(nk);
        std::vector<MapPoint*> vp;
        for (int i = 0; i < nk; i++) {
            tmps[i].mTrackProjX = keys[i].pt.x + 0.5f; tmps[i].mTrackProjY = keys[i].pt.y - 0.5f; tmps[i].mTrackProjXR = -1.f;
            tmps[i].mnTrackScaleLevel = keys[i].octave; tmps[i].mTrackViewCos = 0.9999f;
            tmps[i].descriptor = descriptors.row(i).clone();
            vp.push_back(&tmps[i]);
        }
        eaofusion::ORBmatcher matcher(0.8f, true);
        int32_t nm = matcher.SearchByProjection(TF, vp, 1.0f);
        int32_t selfHits = 0;
        for (int i = 0; i < nk; i++) selfHits += (TF.mvpMapPoints[i] == &tmps[i]);
        wr(out, &nm, 1); wr(out, &selfHits, 1);
    }
    // ---------------------------------------------------------------- Optimizer::PoseOptimization(Frame*)
    int32_t n;
    rd(in, &n, 1);
    Frame F;
    F.N = n + 5;   // a few keypoints without a map point
    F.mTcw = cv::Mat(4, 4, CV_32F);
    rd(in, F.mTcw.ptr<float>(0), 16);
    std::vector<float> Xw(n * 3), obs(n * 3), inv(n);
    rd(in, Xw.data(), Xw.size()); rd(in, obs.data(), obs.size()); rd(in, inv.data(), inv.size());
    float K[5];
    rd(in, K, 5);
    F.fx = K[0]; F.fy = K[1]; F.cx = K[2]; F.cy = K[3]; F.mbf = K[4];
    std::vector<MapPoint> poseMPs(n);
    F.mvpMapPoints.assign(F.N, nullptr); F.mvKeysUn.resize(F.N); F.mvuRight.assign(F.N, -1.f); F.mvbOutlier.assign(F.N, true);
    F.mvInvLevelSigma2.assign(n, 0.f);   // one table entry per correspondence keeps the exact float of the problem file
    for (int i = 0; i < n; i++) {
        poseMPs[i].pos = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) poseMPs[i].pos.at<float>(k) = Xw[3 * i + k];
        F.mvpMapPoints[i] = &poseMPs[i];
        F.mvKeysUn[i].pt.x = obs[3 * i]; F.mvKeysUn[i].pt.y = obs[3 * i + 1]; F.mvuRight[i] = obs[3 * i + 2];
        F.mvKeysUn[i].octave = i;
        F.mvInvLevelSigma2[i] = inv[i];
    }
    // planes: m associated + one empty slot in front
    int32_t m;
    rd(in, &m, 1);
    std::vector<float> pworld(m * 4), pobs(m * 4);
    std::vector<uint8_t> pseen(m);
    rd(in, pworld.data(), pworld.size()); rd(in, pobs.data(), pobs.size()); rd(in, pseen.data(), m);
    std::vector<MapPlane> planes(m);
    F.mnPlaneNum = m + 1;
    F.mvpMapPlanes.assign(m + 1, nullptr); F.mvPlaneCoefficients.assign(m + 1, cv::Mat(4, 1, CV_32F)); F.mvbPlaneOutlier.assign(m + 1, true);
    for (int i = 0; i < m; i++) {
        planes[i].world = cv::Mat(4, 1, CV_32F);
        cv::Mat c(4, 1, CV_32F);
        for (int k = 0; k < 4; k++) { planes[i].world.at<float>(k, 0) = pworld[4 * i + k]; c.at<float>(k, 0) = pobs[4 * i + k]; }
        planes[i].mbSeen = pseen[i] != 0;
        F.mvpMapPlanes[i + 1] = &planes[i];
        F.mvPlaneCoefficients[i + 1] = c;
    }
    Frame Fa = F, Fb = F;              // two more candidates with the same correspondences: PoseOptimizationBatch (Relocalization)
    Fa.mTcw = F.mTcw.clone(); Fb.mTcw = F.mTcw.clone();
    int32_t inliers = eaofusion::PoseOptimization<MapPoint>(&F);
    {
        std::vector<Frame*> cands = {&Fa, &Fb};
        const std::vector<int> got = eaofusion::PoseOptimizationBatch<MapPoint>(cands);
        for (int q = 0; q < 2; q++) {
            const Frame& G = *cands[q];
            bool same = got[q] == inliers && G.mvbOutlier == F.mvbOutlier && G.mvbPlaneOutlier == F.mvbPlaneOutlier;
            for (int k = 0; k < 16; k++) same = same && G.mTcw.ptr<float>(0)[k] == F.mTcw.ptr<float>(0)[k];
            if (!same) { std::fprintf(stderr, "PoseOptimizationBatch: candidate %d differs from PoseOptimization\n", q); return 3; }
        }
    }
    wr(out, &inliers, 1);
    std::vector<uint8_t> pfl(m + 1);
    for (int i = 0; i <= m; i++) pfl[i] = F.mvbPlaneOutlier[i];
    wr(out, pfl.data(), m + 1);
    wr(out, F.mTcw.ptr<float>(0), 16);
    std::vector<uint8_t> ofl(n);
    for (int i = 0; i < n; i++) ofl[i] = F.mvbOutlier[i];
    wr(out, ofl.data(), n);
    // ---------------------------------------------------------------- Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*)
    int32_t dims[3];
    rd(in, dims, 3);
    const int nc = dims[0], np = dims[1], ne = dims[2];
    std::vector<float> camT(nc * 16), pts(np * 3), eobs(ne * 3), einv(ne);
    std::vector<uint8_t> fixed(nc);
    std::vector<int32_t> ecam(ne), ept(ne);
    rd(in, camT.data(), camT.size()); rd(in, fixed.data(), nc); rd(in, pts.data(), pts.size());
    rd(in, ecam.data(), ne); rd(in, ept.data(), ne); rd(in, eobs.data(), eobs.size()); rd(in, einv.data(), ne);
    rd(in, K, 5);
    std::vector<KeyFrame> kfs(nc);     // contiguous: pointer order == camera order, like the generator's edge order
    std::vector<MapPoint> mps(np);
    int firstFree = -1;
    for (int c = 0; c < nc; c++) {
        kfs[c].mnId = c;               // camera 0 is fixed by the mnId == 0 rule, other fixed ones by not being covisible
        kfs[c].fx = K[0]; kfs[c].fy = K[1]; kfs[c].cx = K[2]; kfs[c].cy = K[3]; kfs[c].mbf = K[4];
        kfs[c].Tcw = cv::Mat(4, 4, CV_32F);
        for (int k = 0; k < 16; k++) kfs[c].Tcw.ptr<float>(0)[k] = camT[c * 16 + k];
        kfs[c].mvInvLevelSigma2.assign(ne ? 1 : 1, 0.f);
        if (!fixed[c] && firstFree < 0) firstFree = c;
    }
    for (int p = 0; p < np; p++) {
        mps[p].mnId = p;
        mps[p].pos = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) mps[p].pos.at<float>(k) = pts[3 * p + k];
    }
    for (int e = 0; e < ne; e++) {
        KeyFrame& kf = kfs[ecam[e]];
        const size_t idx = kf.mvKeysUn.size();
        cv::KeyPoint kp;
        kp.pt.x = eobs[3 * e]; kp.pt.y = eobs[3 * e + 1];
        kp.octave = (int)kf.mvInvLevelSigma2.size();      // one table entry per observation keeps the exact float
        kf.mvInvLevelSigma2.push_back(einv[e]);
        kf.mvKeysUn.push_back(kp);
        kf.mvuRight.push_back(eobs[3 * e + 2]);
        kf.mvpMapPoints.push_back(&mps[ept[e]]);
        mps[ept[e]].observations[&kf] = idx;
    }
    KeyFrame* cur = &kfs[firstFree];
    for (int c = 0; c < nc; c++) if (!fixed[c] && c != firstFree) cur->covisible.push_back(&kfs[c]);
    Map map;
    bool stop = false;
    eaofusion::LocalBundleAdjustment<MapPoint>(cur, &stop, &map);
    for (int c = 0; c < nc; c++) wr(out, kfs[c].Tcw.ptr<float>(0), 16);
    for (int p = 0; p < np; p++) wr(out, mps[p].pos.ptr<float>(0), 3);
    int32_t erased = 0, normals = 0;
    for (int c = 0; c < nc; c++) erased += kfs[c].erased;
    for (int p = 0; p < np; p++) normals += mps[p].normalUpdates;
    wr(out, &erased, 1); wr(out, &normals, 1);
    bool stopNow = true;                                  // abort flag set on entry: silent return, nothing written back
    std::vector<float> before(16);
    for (int k = 0; k < 16; k++) before[k] = cur->Tcw.ptr<float>(0)[k];
    eaofusion::LocalBundleAdjustment<MapPoint>(cur, &stopNow, &map);
    int32_t same = 1;
    for (int k = 0; k < 16; k++) same &= (before[k] == cur->Tcw.ptr<float>(0)[k]);
    wr(out, &same, 1);
    // ---------------------------------------------------------------- Frame::ComputeStereoMatches over two extractors
    {
        int32_t sw = 0, sh = 0;
        rd(in, &sw, 1); rd(in, &sh, 1);
        cv::Mat imL(sh, sw, CV_8U), imR(sh, sw, CV_8U);
        rd(in, imL.ptr(0), (size_t)sw * sh); rd(in, imR.ptr(0), (size_t)sw * sh);
        struct StereoFrame {
            int N = 0;
            std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
            cv::Mat mDescriptors, mDescriptorsRight;
            ORB_SLAM2::ORBextractor* mpORBextractorLeft = nullptr; ORB_SLAM2::ORBextractor* mpORBextractorRight = nullptr;
            float mb = 0, mbf = 0;
            std::vector<float> mvuRight, mvDepth;
        } SF;
        ORB_SLAM2::ORBextractor exL(1000, 1.2f, 8, 20, 7), exR(1000, 1.2f, 8, 20, 7);
        exL(imL, cv::Mat(), SF.mvKeys, SF.mDescriptors);
        exR(imR, cv::Mat(), SF.mvKeysRight, SF.mDescriptorsRight);
        SF.N = (int)SF.mvKeys.size();
        SF.mpORBextractorLeft = &exL; SF.mpORBextractorRight = &exR;
        SF.mbf = 40.0f; SF.mb = 40.0f / 535.4f;
        eaofusion::ComputeStereoMatches(SF);
        const int32_t ns = SF.N;
        wr(out, &ns, 1); wr(out, SF.mvuRight.data(), ns); wr(out, SF.mvDepth.data(), ns);
    }
    // ---------------------------------------------------------------- Optimizer::BundleAdjustment (keyframes + map points)
    {
        auto build = [&](std::vector<KeyFrame>& gk, std::vector<MapPoint>& gm) {
            gk.assign(nc, KeyFrame()); gm.assign(np, MapPoint());
            for (int c = 0; c < nc; c++) {
                gk[c].mnId = c;
                gk[c].fx = K[0]; gk[c].fy = K[1]; gk[c].cx = K[2]; gk[c].cy = K[3]; gk[c].mbf = K[4];
                gk[c].Tcw = cv::Mat(4, 4, CV_32F);
                for (int k = 0; k < 16; k++) gk[c].Tcw.ptr<float>(0)[k] = camT[c * 16 + k];
                gk[c].mvInvLevelSigma2.assign(1, 0.f);
            }
            for (int p = 0; p < np; p++) {
                gm[p].mnId = p;
                gm[p].pos = cv::Mat(3, 1, CV_32F);
                for (int k = 0; k < 3; k++) gm[p].pos.at<float>(k) = pts[3 * p + k];
            }
            for (int e = 0; e < ne; e++) {
                KeyFrame& kf = gk[ecam[e]];
                const size_t idx = kf.mvKeysUn.size();
                cv::KeyPoint kp;
                kp.pt.x = eobs[3 * e]; kp.pt.y = eobs[3 * e + 1];
                kp.octave = (int)kf.mvInvLevelSigma2.size();
                kf.mvInvLevelSigma2.push_back(einv[e]);
                kf.mvKeysUn.push_back(kp);
                kf.mvuRight.push_back(eobs[3 * e + 2]);
                kf.mvpMapPoints.push_back(&gm[ept[e]]);
                gm[ept[e]].observations[&kf] = idx;
            }
        };
        std::vector<KeyFrame> gk; std::vector<MapPoint> gm;
        build(gk, gm);
        std::vector<KeyFrame*> vk; std::vector<MapPoint*> vm; std::vector<MapPlane*> vpl;
        for (int c = nc - 1; c >= 0; c--) vk.push_back(&gk[c]);      // any order: the template sorts by mnId
        for (int p = 0; p < np; p++) vm.push_back(&gm[p]);
        MapPlane deadPlane; deadPlane.bad = true; vpl.push_back(&deadPlane);   // a bad plane is skipped (:205-206)
        eaofusion::BundleAdjustment(vk, vm, vpl, 10, nullptr, 0, false);       // nLoopKF == 0: SetPose / SetWorldPos
        for (int c = 0; c < nc; c++) wr(out, gk[c].Tcw.ptr<float>(0), 16);
        for (int p = 0; p < np; p++) wr(out, gm[p].pos.ptr<float>(0), 3);
        int32_t normalsG = 0;
        for (int p = 0; p < np; p++) normalsG += gm[p].normalUpdates;
        wr(out, &normalsG, 1);
        build(gk, gm);
        vk.clear(); vm.clear();
        for (int c = 0; c < nc; c++) vk.push_back(&gk[c]);
        for (int p = 0; p < np; p++) vm.push_back(&gm[p]);
        eaofusion::BundleAdjustment(vk, vm, vpl, 10, nullptr, 7, false);       // loop closing: results parked in mTcwGBA / mPosGBA
        int32_t parked = 1;
        for (int c = 0; c < nc; c++) {
            parked &= gk[c].mnBAGlobalForKF == 7 && !gk[c].mTcwGBA.empty();
            for (int k = 0; k < 16; k++) parked &= gk[c].Tcw.ptr<float>(0)[k] == camT[c * 16 + k];   // poses themselves untouched
            wr(out, gk[c].mTcwGBA.ptr<float>(0), 16);
        }
        for (int p = 0; p < np; p++) parked &= gm[p].normalUpdates == 0 && (gm[p].observations.empty() || (gm[p].mnBAGlobalForKF == 7 && !gm[p].mPosGBA.empty()));
        wr(out, &parked, 1);
        // a live map plane seen by every keyframe: measured coefficients = the plane in each camera frame (from the initial
        // poses) with a small deterministic disturbance; the test replays the same data through the C-ABI mirror
        build(gk, gm);
        vk.clear(); vm.clear();
        for (int c = 0; c < nc; c++) vk.push_back(&gk[c]);
        for (int p = 0; p < np; p++) vm.push_back(&gm[p]);
        MapPlane livePlane; livePlane.mnId = 3;
        livePlane.world = cv::Mat(4, 1, CV_32F);
        const float w4[4] = {0.12f, -0.2f, 0.97f, 3.4f};
        for (int k = 0; k < 4; k++) livePlane.world.at<float>(k) = w4[k];
        std::vector<float> plobs;
        for (int c = 0; c < nc; c++) {
            const float* T = &camT[c * 16];
            float nl[3], d;
            for (int r = 0; r < 3; r++) nl[r] = T[r * 4] * w4[0] + T[r * 4 + 1] * w4[1] + T[r * 4 + 2] * w4[2];
            d = w4[3] - (T[3] * nl[0] + T[7] * nl[1] + T[11] * nl[2]);
            cv::Mat m(4, 1, CV_32F);
            m.at<float>(0) = nl[0] + 0.004f * (float)((c % 3) - 1); m.at<float>(1) = nl[1] - 0.003f * (float)(c % 2); m.at<float>(2) = nl[2];
            m.at<float>(3) = d + 0.01f * (float)((c % 5) - 2);
            gk[c].mvPlaneCoefficients.assign(2, cv::Mat());
            gk[c].mvPlaneCoefficients[1] = m;
            livePlane.observations[&gk[c]] = 1;
            for (int k = 0; k < 4; k++) plobs.push_back(m.at<float>(k));
        }
        vpl.push_back(&livePlane);
        eaofusion::BundleAdjustment(vk, vm, vpl, 10, nullptr, 0, true);
        wr(out, plobs.data(), plobs.size());
        for (int c = 0; c < nc; c++) wr(out, gk[c].Tcw.ptr<float>(0), 16);
        wr(out, livePlane.world.ptr<float>(0), 4);
    }
    printf("adapter_test ok: %d keypoints, %d pose inliers, %d observations erased\n", nk, inliers, erased);
    return 0;
}
