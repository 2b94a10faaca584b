// mixed_load.cpp -- the hot path under the REFERENCE'S OWN CONCURRENCY.  Upstream runs three threads at once (src/System.cc:98-138):
//   Tracking      -- per frame, real time: ORBextractor::operator() -> TrackWithMotionModel / TrackReferenceKeyFrame -> TrackLocalMap (src/Tracking.cc:940-1207),
//   LocalMapping  -- Optimizer::LocalBundleAdjustment in a loop as keyframes arrive (src/LocalMapping.cc:42-116, :75),
//   LoopClosing   -- a global BundleAdjustment thread after a loop closure (src/LoopClosing.cc:590-594),
// and on one MI355X they share the chip.  This harness is NOT a tracker: thread T replays one frame's calls at a fixed cadence over stand-ins of the SLAM classes
// (accessors that lock and clone as upstream's do), thread L loops LocalBundleAdjustment through the class surface (BASELINE configs[3]) or eao_local_ba_batch over
// 25 windows, thread G loops the 1000-keyframe map-scale eao_bundle_adjustment.  Reported: p50 / p90 / p99 / max of T's per-frame latency idle, beside L, beside the
// batch, beside L + G; the stages' own p50 / p99; L's and G's call times alone and beside T; and whether every result equals the idle run's bit for bit.
//
//   mixed_load <problem.bin> <windows.bin> <map.bin> [frames] [period_us]       (files written by bench.py: mixed_load_inputs()); ONE JSON object on stdout.
// Built with hipcc (the tracked-frame chain takes the extractor's DEVICE outputs).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <eao_fusion.h>
// (the LBA adapter's C-ABI call is routed through a wrapper that stamps its entry and exit: where in L's cycle -- accessor walk, C-ABI, write-back -- T's slow frames fall)
namespace cabi {
struct Span { double walk0, abi0, abi1, end; };
static std::vector<Span> spans;              // thread L only
static double t_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double abi0 = 0, abi1 = 0;
template <class... A> eao_status lba(A... a) { abi0 = t_ms(); const eao_status st = ::eao_local_ba(a...); abi1 = t_ms(); return st; }
}  // namespace cabi
#define eao_local_ba cabi::lba
#include <eaofusion/OptimizerImpl.h>
#undef eao_local_ba
#include <eaofusion/ORBextractor.h>
#include <eaofusion/ORBmatcher.h>
#include <eaofusion/DeviceTracker.h>

// ---- stand-ins: the members the adapters touch, with upstream's locking / cloning behaviour (src/MapPoint.cc:68-91, 385-394; src/KeyFrame.cc:74-107, 268-300)
struct KeyFrame;
struct MapPoint {
    static std::mutex mGlobalMutex;
    std::mutex mMutexPos, mMutexFeatures;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnLastFrameSeen = ~0ul;
    bool mbTrackInView = true, mbBad = false;
    int mnTrackScaleLevel = 0, nObs = 1, nVisible = 0;
    float mTrackViewCos = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    cv::Mat mWorldPos, mDescriptor, mNormalVector;
    std::map<KeyFrame*, size_t> mObservations;
    cv::Mat GetWorldPos() { std::unique_lock<std::mutex> l(mMutexPos); return mWorldPos.clone(); }
    cv::Mat GetNormal() { std::unique_lock<std::mutex> l(mMutexPos); return mNormalVector.clone(); }
    void SetWorldPos(const cv::Mat& p) { std::unique_lock<std::mutex> l2(mGlobalMutex); std::unique_lock<std::mutex> l(mMutexPos); p.copyTo(mWorldPos); }
    cv::Mat GetDescriptor() { std::unique_lock<std::mutex> l(mMutexFeatures); return mDescriptor.clone(); }
    int Observations() { std::unique_lock<std::mutex> l(mMutexFeatures); return nObs; }
    void IncreaseVisible(int n = 1) { std::unique_lock<std::mutex> l(mMutexFeatures); nVisible += n; }
    bool isBad() { std::unique_lock<std::mutex> l(mMutexFeatures); std::unique_lock<std::mutex> l2(mMutexPos); return mbBad; }
    std::map<KeyFrame*, size_t> GetObservations() { std::unique_lock<std::mutex> l(mMutexFeatures); return mObservations; }
    void EraseObservation(KeyFrame* kf) { std::unique_lock<std::mutex> l(mMutexFeatures); mObservations.erase(kf); }
    void UpdateNormalAndDepth() {}
    float GetMinDistanceInvariance() { std::unique_lock<std::mutex> l(mMutexPos); return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { std::unique_lock<std::mutex> l(mMutexPos); return 1.2f * mfMaxDistance; }
    void SetDistances(float mn, float mx) { mfMinDistance = mn; mfMaxDistance = mx; }
protected:
    float mfMinDistance = 0, mfMaxDistance = 0;
};
std::mutex MapPoint::mGlobalMutex;

struct KeyFrame {
    std::mutex mMutexPose, mMutexFeatures, mMutexConnections;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul;
    float fx, fy, cx, cy, mbf;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    cv::Mat mDescriptors, Tcw;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
    bool isBad() { std::unique_lock<std::mutex> l(mMutexConnections); return false; }
    cv::Mat GetPose() { std::unique_lock<std::mutex> l(mMutexPose); return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { std::unique_lock<std::mutex> l(mMutexPose); T.copyTo(Tcw); }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { std::unique_lock<std::mutex> l(mMutexConnections); return mvpOrderedConnectedKeyFrames; }
    std::vector<MapPoint*> GetMapPointMatches() { std::unique_lock<std::mutex> l(mMutexFeatures); return mvpMapPoints; }
    void EraseMapPointMatch(MapPoint* mp) {
        std::unique_lock<std::mutex> l(mMutexFeatures);
        const auto it = mp->mObservations.find(this);
        if (it != mp->mObservations.end()) mvpMapPoints[it->second] = nullptr;
    }
};
struct Map { std::mutex mMutexMapUpdate; };
struct Frame {
    int N = 0;
    long unsigned int mnId = 9;
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    float mbf = 40.f, mb = 0, mfLogScaleFactor = std::log(1.2f);
    int mnScaleLevels = 8;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors, mTcw;
    std::vector<float> mvScaleFactors, mvuRight, mvDepth, mvInvLevelSigma2;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    void SetPose(cv::Mat T) { mTcw = T.clone(); }
};
float Frame::fx = 535.4f, Frame::fy = 539.2f, Frame::cx = 320.1f, Frame::cy = 247.6f;
float Frame::mnMinX = 0, Frame::mnMaxX = 640, Frame::mnMinY = 0, Frame::mnMaxY = 480;
float Frame::mfGridElementWidthInv = 64.f / 640.f, Frame::mfGridElementHeightInv = 48.f / 480.f;

template <typename T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), n * sizeof(T)); }
template <typename T> static std::vector<T> rdv(std::ifstream& f, size_t n) { std::vector<T> v(n); rd(f, v.data(), n); return v; }
static cv::Mat colN(const float* v, int n) { cv::Mat m(n, 1, CV_32F); for (int i = 0; i < n; i++) m.at<float>(i) = v[i]; return m; }
static cv::Mat mat44(const float* v) { cv::Mat m(4, 4, CV_32F); for (int i = 0; i < 16; i++) m.at<float>(i / 4, i % 4) = v[i]; return m; }
static cv::Mat desc32(const uint8_t* d) { cv::Mat m(1, 32, CV_8U); std::memcpy(m.data, d, 32); return m; }

using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
static uint64_t fnv(uint64_t h, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
static unsigned long long g_s = 88172645463325252ull;
static double rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) / 9007199254740992.0; }
#define HIPCHK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s\n", #x); std::exit(2); } } while (0)
#define EAOCHK(x) do { if ((x) != EAO_OK) { fprintf(stderr, "%s failed: %s\n", #x, eao_last_error()); std::exit(2); } } while (0)

struct Dist { std::vector<double> v; };
static double pct(std::vector<double> v, double q) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[std::min(v.size() - 1, (size_t)(q * (double)v.size()))]; }
static std::string stat_json(const std::vector<double>& v) {
    char b[256];
    double mean = 0; for (double x : v) mean += x; mean /= std::max<size_t>(1, v.size());
    std::snprintf(b, sizeof b, "{\"n\": %zu, \"p50\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"max\": %.4f, \"mean\": %.4f}", v.size(), pct(v, 0.50), pct(v, 0.90), pct(v, 0.99),
                  v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()), mean);
    return b;
}

// ---- flat BA problems (the C-ABI's arrays): windows.bin = int32 count, then per problem the arrays below; map.bin = one problem
struct FlatBA {
    int32_t nc, np, ne, its1, its2;
    std::vector<float> poses, pts, obs, inv;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ecam, ept;
    float K[5];
    std::vector<float> oT, oP; std::vector<uint8_t> oO;
    eao_ba_problem P; eao_ba_result R;
    void read(std::ifstream& in) {
        rd(in, &nc, 1); rd(in, &np, 1); rd(in, &ne, 1); rd(in, &its1, 1); rd(in, &its2, 1);
        poses = rdv<float>(in, 16 * (size_t)nc); fixed = rdv<uint8_t>(in, nc); pts = rdv<float>(in, 3 * (size_t)np);
        ecam = rdv<int32_t>(in, ne); ept = rdv<int32_t>(in, ne); obs = rdv<float>(in, 3 * (size_t)ne); inv = rdv<float>(in, ne); rd(in, K, 5);
        oT.resize(poses.size()); oP.resize(pts.size()); oO.resize(ne);
        P = eao_ba_problem{nc, np, ne, poses.data(), fixed.data(), pts.data(), ecam.data(), ept.data(), obs.data(), inv.data(), K[0], K[1], K[2], K[3], K[4], its1, its2};
        R = eao_ba_result{};
        R.cam_Tcw = oT.data(); R.points = oP.data(); R.edge_outlier = oO.data();
    }
    uint64_t hash() const { uint64_t h = fnv(1469598103934665603ull, oT.data(), oT.size() * 4); h = fnv(h, oP.data(), oP.size() * 4); return fnv(h, oO.data(), oO.size()); }
};

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int nFrames = argc > 4 ? std::atoi(argv[4]) : 1200;
    const int periodUs = argc > 5 ? std::atoi(argv[5]) : 2000;
    const int scenarioMask = argc > 6 ? std::atoi(argv[6]) : 15, variantMask = argc > 7 ? std::atoi(argv[7]) : 3;      // (diagnostic runs: a subset of the scenarios / variants)
    std::ifstream in(argv[1], std::ios::binary), inW(argv[2], std::ios::binary), inM(argv[3], std::ios::binary);
    if (!in || !inW || !inM) return 2;
    // ------------------------------------------------------------------ inputs (the layout of adapter_bench's problem.bin)
    int32_t H, W;
    rd(in, &H, 1); rd(in, &W, 1);
    cv::Mat img(H, W, CV_8UC1);
    rd(in, img.data, (size_t)H * W);
    // PoseOptimization(Frame*)
    int32_t pn;
    rd(in, &pn, 1);
    float pT[16], pK[5];
    rd(in, pT, 16);
    const std::vector<float> pXw = rdv<float>(in, 3 * (size_t)pn), pobs = rdv<float>(in, 3 * (size_t)pn), pinv = rdv<float>(in, pn);
    rd(in, pK, 5);
    // LocalBundleAdjustment window
    int32_t nc, np, ne;
    rd(in, &nc, 1); rd(in, &np, 1); rd(in, &ne, 1);
    const std::vector<float> poses = rdv<float>(in, 16 * (size_t)nc);
    const std::vector<uint8_t> fixed = rdv<uint8_t>(in, nc);
    const std::vector<float> pts = rdv<float>(in, 3 * (size_t)np);
    const std::vector<int32_t> ecam = rdv<int32_t>(in, ne), ept = rdv<int32_t>(in, ne);
    const std::vector<float> obs = rdv<float>(in, 3 * (size_t)ne), inv = rdv<float>(in, ne);
    float K[5];
    rd(in, K, 5);
    // the SearchByProjection pair
    int32_t N;
    rd(in, &N, 1);
    const std::vector<float> kx = rdv<float>(in, N), ky = rdv<float>(in, N), ang = rdv<float>(in, N), ur = rdv<float>(in, N);
    const std::vector<int32_t> oct = rdv<int32_t>(in, N);
    const std::vector<uint8_t> desc = rdv<uint8_t>(in, 32 * (size_t)N), occ = rdv<uint8_t>(in, N);
    float Tc[16], K6[6], sf[8];
    rd(in, Tc, 16); rd(in, K6, 6); rd(in, sf, 8);
    int32_t n;
    rd(in, &n, 1);
    float Tl[16];
    rd(in, Tl, 16);
    const std::vector<uint8_t> valid = rdv<uint8_t>(in, n);
    const std::vector<float> Xw = rdv<float>(in, 3 * (size_t)n);
    const std::vector<uint8_t> ldesc = rdv<uint8_t>(in, 32 * (size_t)n);
    const std::vector<int32_t> loct = rdv<int32_t>(in, n);
    const std::vector<float> lang = rdv<float>(in, n), px = rdv<float>(in, n), py = rdv<float>(in, n), pxr = rdv<float>(in, n), vc = rdv<float>(in, n);
    const std::vector<int32_t> lvl = rdv<int32_t>(in, n);
    const std::vector<uint8_t> skip = rdv<uint8_t>(in, n);
    int32_t nWin;
    rd(inW, &nWin, 1);
    std::vector<FlatBA> wins(nWin);
    for (auto& w : wins) w.read(inW);
    FlatBA gmap;
    gmap.read(inM);

    // ------------------------------------------------------------------ thread T, variant "device_chain": extractor (class surface) + the tracked-frame chain
    const int M = 1000, cap = 1024;
    Frame F0;
    for (int l = 0; l < 8; l++) { F0.mvScaleFactors.push_back(l ? F0.mvScaleFactors[l - 1] * 1.2f : 1.0f); F0.mvInvLevelSigma2.push_back(1.0f / (F0.mvScaleFactors[l] * F0.mvScaleFactors[l])); }
    F0.mTcw = cv::Mat::eye(4, 4, CV_32F);
    F0.mTcw.at<float>(0, 3) = 0.02f; F0.mTcw.at<float>(2, 3) = -0.03f;
    std::vector<MapPoint> tpts(M);
    std::vector<MapPoint*> local;
    std::vector<float> depth((size_t)W * H, 0.f);
    std::vector<unsigned char> tdesc;
    std::vector<int> seen;
    for (int m = 0; m < M; m++) {
        MapPoint& p = tpts[m];
        const float z = 2.f + 4.f * (float)rnd(), x = (float)(rnd() - 0.5) * z * 0.9f, y = (float)(rnd() - 0.5) * z * 0.7f;
        p.mWorldPos = cv::Mat(3, 1, CV_32F); p.mWorldPos.at<float>(0) = x; p.mWorldPos.at<float>(1) = y; p.mWorldPos.at<float>(2) = z;
        const float nrm = std::sqrt(x * x + y * y + z * z);
        p.mNormalVector = cv::Mat(3, 1, CV_32F); p.mNormalVector.at<float>(0) = x / nrm; p.mNormalVector.at<float>(1) = y / nrm; p.mNormalVector.at<float>(2) = z / nrm;
        const int o = (int)(rnd() * 7.99);
        p.SetDistances(0.7f * nrm, nrm * std::pow(1.2f, o - 0.5f));
        p.mDescriptor = cv::Mat(1, 32, CV_8U);
        for (int b = 0; b < 32; b++) p.mDescriptor.at<unsigned char>(0, b) = (unsigned char)(rnd() * 256);
        p.mbBad = rnd() < 0.02;
        local.push_back(&p);
        const float xc = x + 0.02f, zc = z - 0.03f;
        const float u = Frame::fx * xc / zc + Frame::cx + (float)(rnd() - 0.5) * 2.f, v = Frame::fy * y / zc + Frame::cy + (float)(rnd() - 0.5) * 2.f;
        if (u < 2 || u > W - 3 || v < 2 || v > H - 3) continue;
        cv::KeyPoint kp(u, v, 31.f, (float)(rnd() * 360.0), 50.f, o);
        F0.mvKeys.push_back(kp);
        seen.push_back(m);
        for (int b = 0; b < 32; b++) tdesc.push_back((unsigned char)(p.mDescriptor.at<unsigned char>(0, b) ^ (rnd() < 0.1 ? 1 << (int)(rnd() * 8) : 0)));
        depth[(size_t)(int)v * W + (int)u] = rnd() < 0.2 ? 0.f : zc;
    }
    F0.N = (int)F0.mvKeys.size();
    F0.mvKeysUn = F0.mvKeys;
    F0.mvpMapPoints.assign(F0.N, nullptr);
    F0.mvbOutlier.assign(F0.N, false);
    eao_keypoint* d_kps; uint8_t* d_desc; int32_t* d_n; float* d_depth;
    HIPCHK(hipMalloc((void**)&d_kps, cap * sizeof(eao_keypoint))); HIPCHK(hipMalloc((void**)&d_desc, cap * 32)); HIPCHK(hipMalloc((void**)&d_n, 4));
    HIPCHK(hipMalloc((void**)&d_depth, depth.size() * 4));
    HIPCHK(hipMemset(d_kps, 0, cap * sizeof(eao_keypoint))); HIPCHK(hipMemset(d_desc, 0, cap * 32));
    HIPCHK(hipMemcpy(d_kps, F0.mvKeys.data(), (size_t)F0.N * sizeof(eao_keypoint), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_desc, tdesc.data(), tdesc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_n, &F0.N, 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
    Frame Last = F0;
    Last.mnId = 8;
    for (int k = 0; k < Last.N; k++) { Last.mvpMapPoints[k] = &tpts[seen[k]]; Last.mvbOutlier[k] = (k % 17) == 0; }

    // ------------------------------------------------------------------ thread T, variant "class_surface": extractor + SearchByProjection x 2 + PoseOptimization x 2
    std::vector<MapPoint> pmps(pn);
    Frame PF;
    PF.N = pn; PF.mbf = pK[4];
    PF.mvKeysUn.resize(pn); PF.mvuRight.resize(pn); PF.mvpMapPoints.resize(pn); PF.mvbOutlier.assign(pn, false);
    PF.mvInvLevelSigma2.assign(pinv.begin(), pinv.end());
    for (int i = 0; i < pn; i++) {
        pmps[i].mWorldPos = colN(&pXw[3 * (size_t)i], 3);
        PF.mvKeysUn[i].pt.x = pobs[3 * (size_t)i]; PF.mvKeysUn[i].pt.y = pobs[3 * (size_t)i + 1]; PF.mvKeysUn[i].octave = i;
        PF.mvuRight[i] = pobs[3 * (size_t)i + 2];
        PF.mvpMapPoints[i] = &pmps[i];
    }
    Frame Cur, SLast;
    Cur.N = N; Cur.mbf = K6[4]; Cur.mb = K6[5];
    Cur.mvKeysUn.resize(N); Cur.mvuRight = ur; Cur.mvScaleFactors.assign(sf, sf + 8); Cur.mDescriptors = cv::Mat(N, 32, CV_8U);
    std::memcpy(Cur.mDescriptors.data, desc.data(), desc.size());
    for (int i = 0; i < N; i++) { Cur.mvKeysUn[i].pt.x = kx[i]; Cur.mvKeysUn[i].pt.y = ky[i]; Cur.mvKeysUn[i].angle = ang[i]; Cur.mvKeysUn[i].octave = oct[i]; }
    Cur.mvKeys = Cur.mvKeysUn; Cur.mTcw = mat44(Tc); Cur.mvbOutlier.assign(N, false);
    std::vector<MapPoint> holders(N), smps(n);
    std::vector<MapPoint*> vp(n);
    for (int i = 0; i < n; i++) {
        MapPoint& m = smps[i];
        m.mWorldPos = colN(&Xw[3 * (size_t)i], 3); m.mDescriptor = desc32(&ldesc[32 * (size_t)i]);
        m.mTrackProjX = px[i]; m.mTrackProjY = py[i]; m.mTrackProjXR = pxr[i]; m.mTrackViewCos = vc[i]; m.mnTrackScaleLevel = lvl[i]; m.mbTrackInView = !skip[i];
        vp[i] = &m;
    }
    SLast.N = n; SLast.mvKeys.resize(n); SLast.mvKeysUn.resize(n); SLast.mvpMapPoints.resize(n); SLast.mvbOutlier.assign(n, false); SLast.mTcw = mat44(Tl);
    for (int i = 0; i < n; i++) { SLast.mvKeys[i].octave = loct[i]; SLast.mvKeysUn[i].angle = lang[i]; SLast.mvpMapPoints[i] = valid[i] ? &smps[i] : nullptr; }

    // ------------------------------------------------------------------ thread L's window (class surface), rebuilt before every call as LocalMapping hands it over
    std::vector<KeyFrame> kfs(nc);
    std::vector<MapPoint> mps(np);
    Map map;
    KeyFrame* pKF = nullptr;
    auto buildWindow = [&] {
        pKF = nullptr;
        for (int c = 0; c < nc; c++) {
            KeyFrame& k = kfs[c];
            k.mnId = (unsigned long)c + (fixed[c] ? 0 : 100); k.mnBALocalForKF = k.mnBAFixedForKF = ~0ul;
            k.fx = K[0]; k.fy = K[1]; k.cx = K[2]; k.cy = K[3]; k.mbf = K[4];
            k.Tcw = mat44(&poses[16 * (size_t)c]);
            k.mvKeysUn.clear(); k.mvuRight.clear(); k.mvpMapPoints.clear(); k.mvInvLevelSigma2.clear(); k.mvpOrderedConnectedKeyFrames.clear();
            if (!fixed[c]) pKF = &k;
        }
        for (int c = 0; c < nc; c++) if (!fixed[c] && &kfs[c] != pKF) pKF->mvpOrderedConnectedKeyFrames.push_back(&kfs[c]);
        for (int p = 0; p < np; p++) { mps[p].mnId = p; mps[p].mnBALocalForKF = ~0ul; mps[p].mWorldPos = colN(&pts[3 * (size_t)p], 3); mps[p].mObservations.clear(); }
        for (int e = 0; e < ne; e++) {
            KeyFrame& k = kfs[ecam[e]];
            cv::KeyPoint kp;
            kp.pt.x = obs[3 * (size_t)e]; kp.pt.y = obs[3 * (size_t)e + 1]; kp.octave = (int)k.mvKeysUn.size();
            mps[ept[e]].mObservations[&k] = k.mvKeysUn.size();
            k.mvKeysUn.push_back(kp); k.mvuRight.push_back(obs[3 * (size_t)e + 2]); k.mvInvLevelSigma2.push_back(inv[e]); k.mvpMapPoints.push_back(&mps[ept[e]]);
        }
    };
    auto windowHash = [&] {
        uint64_t h = 1469598103934665603ull;
        for (int c = 0; c < nc; c++) h = fnv(h, kfs[c].Tcw.data, 64);
        for (int p = 0; p < np; p++) h = fnv(h, mps[p].mWorldPos.data, 12);
        for (int p = 0; p < np; p++) { const size_t s = mps[p].mObservations.size(); h = fnv(h, &s, sizeof s); }
        return h;
    };

    // ------------------------------------------------------------------ the three thread bodies
    struct TOut { std::vector<double> total, extract, stageA, stageB, start; std::set<uint64_t> hashes; int late = 0; };
    std::atomic<bool> stopBg{false};
    auto runT = [&](int variant, TOut& out) {
        ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
        ex.keepPyramid = false;
        std::vector<cv::KeyPoint> keys;
        cv::Mat descriptors;
        eaofusion::DeviceTracker trk(F0, cap, 1024);
        trk.SetLocalMap(local);
        eaofusion::ORBmatcher m1(0.8f, true), m2(0.9f, true);
        const int warm = 30;
        const auto t00 = Clock::now();
        for (int f = 0; f < nFrames + warm; f++) {
            // untimed: the state a new frame starts from
            Frame C = F0;
            C.mnId = 9;
            C.mTcw = F0.mTcw.clone(); C.mTcw.at<float>(0, 3) += 0.004f;
            if (variant == 1) { Cur.mvpMapPoints.assign(N, nullptr); for (int k = 0; k < N; k++) if (occ[k]) Cur.mvpMapPoints[k] = &holders[k]; PF.mTcw = mat44(pT); Cur.mTcw = mat44(Tc); }
            const auto due = t00 + std::chrono::microseconds((long long)f * periodUs);
            if (Clock::now() > due) { if (f >= warm) out.late++; } else std::this_thread::sleep_until(due);
            const auto t0 = Clock::now();
            ex(img, cv::Mat(), keys, descriptors);
            const double tE = ms_since(t0);
            uint64_t h = fnv(1469598103934665603ull, keys.data(), keys.size() * sizeof(cv::KeyPoint));
            double tA, tB;
            if (variant == 0) {
                int nMap = 0, nSearch = 0;
                const int nm = trk.TrackWithMotionModel(C, Last, d_kps, d_desc, d_n, d_depth, W, W, H, 15.f, false, nullptr, &nMap, &nSearch);
                tA = ms_since(t0) - tE;
                const int ni = trk.TrackLocalMap(C, d_kps, d_desc, d_n, d_depth, W, W, H, 3.0f, 0.8f, nullptr);
                tB = ms_since(t0) - tE - tA;
                h = fnv(h, &nm, 4); h = fnv(h, &ni, 4); h = fnv(h, C.mTcw.data, 64);
                for (int k = 0; k < C.N; k++) { const bool o = C.mvbOutlier[k]; const void* p = C.mvpMapPoints[k]; h = fnv(h, &o, 1); h = fnv(h, &p, sizeof p); }
            } else {
                const int a = m2.SearchByProjection(Cur, SLast, 7.0f, false);
                const int i1 = eaofusion::PoseOptimization<MapPoint>(&PF);
                tA = ms_since(t0) - tE;
                for (int k = 0; k < N; k++) Cur.mvpMapPoints[k] = occ[k] ? &holders[k] : nullptr;      // (the local-map search starts from the occupancy the file carries)
                const int b = m1.SearchByProjection(Cur, vp, 1.0f);
                const int i2 = eaofusion::PoseOptimization<MapPoint>(&PF);
                tB = ms_since(t0) - tE - tA;
                h = fnv(h, &a, 4); h = fnv(h, &b, 4); h = fnv(h, &i1, 4); h = fnv(h, &i2, 4); h = fnv(h, PF.mTcw.data, 64);
                for (int k = 0; k < N; k++) { const void* p = Cur.mvpMapPoints[k]; h = fnv(h, &p, sizeof p); }
            }
            const double tot = ms_since(t0);
            h = fnv(h, descriptors.data, (size_t)descriptors.rows * 32);
            if (f >= warm) { out.total.push_back(tot); out.extract.push_back(tE); out.stageA.push_back(tA); out.stageB.push_back(tB); out.start.push_back(cabi::t_ms() - tot); }
            out.hashes.insert(h);
        }
    };
    struct BgOut { std::vector<double> ms; std::set<uint64_t> hashes; std::atomic<int> calls{0}; };
    auto runL = [&](BgOut& out, int minCalls) {
        bool stop = false;
        for (int i = 0; !stopBg.load() || i < minCalls; i++) {
            buildWindow();
            const auto t0 = Clock::now();
            const double w0 = cabi::t_ms();
            eaofusion::LocalBundleAdjustment<MapPoint>(pKF, &stop, &map);
            out.ms.push_back(ms_since(t0));
            out.calls++;
            cabi::spans.push_back(cabi::Span{w0, cabi::abi0, cabi::abi1, cabi::t_ms()});
            out.hashes.insert(windowHash());
        }
    };
    auto runLB = [&](BgOut& out, int minCalls) {
        std::vector<eao_ba_problem> P; std::vector<eao_ba_result> R;
        for (auto& w : wins) { P.push_back(w.P); R.push_back(w.R); }
        for (int i = 0; !stopBg.load() || i < minCalls; i++) {
            const auto t0 = Clock::now();
            EAOCHK(eao_local_ba_batch(P.data(), nWin, nullptr, R.data()));
            out.ms.push_back(ms_since(t0));
            out.calls++;
            uint64_t h = 0; for (auto& w : wins) h ^= w.hash() * 1099511628211ull + (h << 7);
            out.hashes.insert(h);
        }
    };
    auto runG = [&](BgOut& out, int minCalls) {
        for (int i = 0; !stopBg.load() || i < minCalls; i++) {
            const auto t0 = Clock::now();
            EAOCHK(eao_bundle_adjustment(&gmap.P, 0, nullptr, &gmap.R));
            out.ms.push_back(ms_since(t0));
            out.calls++;
            out.hashes.insert(gmap.hash());
        }
    };

    // ------------------------------------------------------------------ scenarios
    std::printf("{\n  \"frames\": %d, \"period_us\": %d, \"stream_priority\": \"%s\",\n", nFrames, periodUs, getenv("EAO_STREAM_PRIORITY") ? getenv("EAO_STREAM_PRIORITY") : "default (on)");
    {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        std::printf("  \"device_priority_range\": [%d, %d],\n", least, greatest);
    }
    // the background jobs alone (warm-up + their own pace)
    BgOut Lalone, LBalone, Galone;
    stopBg = true;
    runL(Lalone, 14); runLB(LBalone, 8); runG(Galone, 5);
    auto tail = [](const std::vector<double>& v, size_t skip) { return std::vector<double>(v.begin() + std::min(skip, v.size()), v.end()); };
    std::printf("  \"alone\": {\"lba_class_surface_ms\": %s, \"lba_batch25_ms\": %s, \"map_ba_ms\": %s},\n", stat_json(tail(Lalone.ms, 3)).c_str(), stat_json(tail(LBalone.ms, 2)).c_str(),
                stat_json(tail(Galone.ms, 2)).c_str());
    bool identical = true;
    const char* vname[2] = {"device_chain", "class_surface"};
    for (int variant = 0; variant < 2; variant++) {
        if (!(variantMask >> variant & 1)) continue;
        std::printf("  \"%s\": {\n", vname[variant]);
        std::set<uint64_t> ref;
        const char* sname[4] = {"idle", "beside_lba", "beside_lba_batch25", "beside_lba_and_map_ba"};
        for (int sc = 0; sc < 4; sc++) {
            if (!(scenarioMask >> sc & 1)) { if (sc == 3) std::printf("    \"skipped\": true\n"); continue; }
            TOut T;
            BgOut L, LB, G;
            stopBg = false;
            std::vector<std::thread> bg;
            if (sc == 1 || sc == 3) bg.emplace_back([&] { runL(L, 1); });
            if (sc == 2) bg.emplace_back([&] { runLB(LB, 1); });
            if (sc == 3) bg.emplace_back([&] { runG(G, 1); });
            // the background threads are long-lived in the reference (LocalMapping, LoopClosing): their first calls -- stream and arena creation, pinned mirrors -- are
            // behind them when the first frame arrives
            while ((sc == 1 || sc == 3) && L.calls < 2) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            while (sc == 2 && LB.calls < 2) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            while (sc == 3 && G.calls < 1) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            runT(variant, T);
            stopBg = true;
            for (auto& t : bg) t.join();
            if (sc == 0) ref = T.hashes;
            const bool same = T.hashes == ref && T.hashes.size() == 1 && (L.hashes.empty() || L.hashes == Lalone.hashes) && (LB.hashes.empty() || LB.hashes == LBalone.hashes) &&
                              (G.hashes.empty() || G.hashes == Galone.hashes);
            identical = identical && same;
            char why[256];
            std::snprintf(why, sizeof why, "\"distinct_results\": {\"frames\": %zu, \"frames_equal_idle\": %s, \"lba\": %zu, \"lba_equal_alone\": %s, \"lba_batch\": %zu, \"lba_batch_equal_alone\": %s, \"map_ba\": %zu, \"map_ba_equal_alone\": %s}",
                          T.hashes.size(), T.hashes == ref ? "true" : "false", L.hashes.size(), L.hashes.empty() || L.hashes == Lalone.hashes ? "true" : "false", LB.hashes.size(),
                          LB.hashes.empty() || LB.hashes == LBalone.hashes ? "true" : "false", G.hashes.size(), G.hashes.empty() || G.hashes == Galone.hashes ? "true" : "false");
            std::printf("    \"%s\": {\"frame_ms\": %s, \"extract_ms\": %s, \"%s\": %s, \"%s\": %s, \"late_frames\": %d, \"results_identical\": %s", sname[sc], stat_json(T.total).c_str(),
                        stat_json(T.extract).c_str(), variant == 0 ? "motion_model_ms" : "search_last_frame_plus_pose_ms", stat_json(T.stageA).c_str(),
                        variant == 0 ? "local_map_ms" : "search_local_map_plus_pose_ms", stat_json(T.stageB).c_str(), T.late, same ? "true" : "false");
            std::printf(", %s", why);
            if (!L.ms.empty()) std::printf(", \"lba_class_surface_ms\": %s", stat_json(L.ms).c_str());
            if (sc == 1 && !cabi::spans.empty()) {
                // where T's slow frames (beyond 1.25 x the median) START in thread L's cycle: [idle: L rebuilds its window | walk: the adapter reads the object graph |
                // abi_first_third / rest: inside eao_local_ba (its first third holds the upload and the bulk enqueue) | write_back]; `all` = every frame, for the base rate
                const double med = pct(T.total, 0.5);
                int slow[5] = {0, 0, 0, 0, 0}, all[5] = {0, 0, 0, 0, 0};
                for (size_t i = 0; i < T.total.size(); i++) {
                    const double ts = T.start[i] + 0.5 * T.total[i];
                    int ph = 0;
                    for (const cabi::Span& sp : cabi::spans) {
                        if (ts < sp.walk0 || ts > sp.end) continue;
                        ph = ts < sp.abi0 ? 1 : ts < sp.abi0 + (sp.abi1 - sp.abi0) / 3 ? 2 : ts < sp.abi1 ? 3 : 4;
                        break;
                    }
                    all[ph]++;
                    if (T.total[i] > 1.25 * med) slow[ph]++;
                }
                std::printf(", \"slow_frames_by_lba_phase\": {\"idle\": [%d, %d], \"walk\": [%d, %d], \"abi_first_third\": [%d, %d], \"abi_rest\": [%d, %d], \"write_back\": [%d, %d], \"note\": \"[slow, all] frames whose midpoint falls in the phase\"}",
                            slow[0], all[0], slow[1], all[1], slow[2], all[2], slow[3], all[3], slow[4], all[4]);
            }
            cabi::spans.clear();
            if (!LB.ms.empty()) std::printf(", \"lba_batch25_ms\": %s", stat_json(LB.ms).c_str());
            if (!G.ms.empty()) std::printf(", \"map_ba_ms\": %s", stat_json(G.ms).c_str());
            std::printf("}%s\n", sc == 3 ? "" : ",");
            std::fflush(stdout);
        }
        std::printf("  },\n");
    }
    std::printf("  \"results_identical\": %s\n}\n", identical ? "true" : "false");
    return identical ? 0 : 1;
}
