// adapter_bench.cpp -- times the hot path AT THE CLASS SURFACE the repository declares as its drop-in boundary:
//   ORB_SLAM2::ORBextractor::operator()                                  (reference src/Frame.cc:616-622 behind the constructors :192-194)
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)       (src/Tracking.cc:2639)
//   ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, m) (src/Tracking.cc:1753)
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)       (src/Tracking.cc:1577)
//   Optimizer::PoseOptimization(Frame*)                                  (src/Tracking.cc:2186, 2247)
//   Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*)             (src/LocalMapping.cc:75)
// over stand-ins of the SLAM classes whose accessors cost what upstream's cost (a mutex + a cv::Mat clone per GetWorldPos / GetDescriptor, a std::map
// copy per GetObservations), at BASELINE.json's sizes: one 640 x 480 frame, 1000 correspondences, 20 + 4 keyframes x 3000 map points.
//
// Beside every class-surface time stands the time the SAME call spent inside the C-ABI: the entry points the adapters call are routed through timing
// wrappers (the macros below rename them inside the adapter headers only), so "adapter overhead" = call - C-ABI is measured on one and the same call,
// not on a hand-flattened twin of the problem.
//
//   adapter_bench <problem.bin>      (written by bench.py: class_surface_problem()); prints ONE JSON object on stdout.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <eao_fusion.h>

namespace cabi {
static double inside_ns = 0;
struct Scope {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~Scope() { inside_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); }
};
template <class... A> eao_status orb_extract(A... a) { Scope s; return ::eao_orb_extract_ref(a...); }
template <class... A> eao_status orb_pyramid(A... a) { Scope s; return ::eao_orb_pyramid(a...); }
template <class... A> eao_status sbp_points(A... a) { Scope s; return ::eao_search_by_projection_points(a...); }
template <class... A> eao_status sbp_frames(A... a) { Scope s; return ::eao_search_by_projection_frames(a...); }
template <class... A> eao_status sbow(A... a) { Scope s; return ::eao_search_by_bow(a...); }
template <class... A> eao_status pose(A... a) { Scope s; return ::eao_pose_optimization(a...); }
static bool stub_lba = false;      // `adapter_bench problem.bin lba-walk`: the library call replaced by an identity result -- the adapter's own walk, timed on a box without a GPU
inline void dump_problem(const eao_ba_problem* p) {
    if (const char* dump = std::getenv("EAO_WALK_DUMP")) {      // tests/test_adapter_walk_cpu.py: the problem the adapter hands to the library, once
        static bool done = false;
        if (!done) {
            done = true;
            std::ofstream f(dump, std::ios::binary);
            const int32_t hdr[3] = {p->n_cams, p->n_points, p->n_edges};
            f.write((const char*)hdr, sizeof hdr);
            f.write((const char*)p->cam_Tcw, sizeof(float) * 16 * (size_t)p->n_cams); f.write((const char*)p->cam_fixed, (size_t)p->n_cams);
            f.write((const char*)p->points, sizeof(float) * 3 * (size_t)p->n_points);
            f.write((const char*)p->edge_cam, sizeof(int32_t) * (size_t)p->n_edges); f.write((const char*)p->edge_point, sizeof(int32_t) * (size_t)p->n_edges);
            f.write((const char*)p->edge_obs, sizeof(float) * 3 * (size_t)p->n_edges); f.write((const char*)p->edge_inv_sigma2, sizeof(float) * (size_t)p->n_edges);
        }
    }
}
static int lba_points = 0, lba_edges = 0;      // what the adapter handed over: the points the LOCAL keyframes see (upstream's window), a little less than the generator's map
inline eao_status lba(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r) {
    lba_points = p->n_points; lba_edges = p->n_edges;
    Scope s;
    if (!stub_lba) return ::eao_local_ba(p, stop, r);
    dump_problem(p);
    std::memcpy(r->cam_Tcw, p->cam_Tcw, sizeof(float) * 16 * (size_t)p->n_cams);
    std::memcpy(r->points, p->points, sizeof(float) * 3 * (size_t)p->n_points);
    std::memset(r->edge_outlier, 0, (size_t)p->n_edges);
    r->aborted = 0; r->iters[0] = r->iters[1] = 0; r->chi2[0] = r->chi2[1] = 0;
    return EAO_OK;
}
static int gba_points = 0, gba_edges = 0;
inline eao_status gba(const eao_ba_problem* p, const eao_ba_planes* pl, int32_t robust, const volatile uint8_t* stop, eao_ba_result* r, float* planes_out) {
    gba_points = p->n_points; gba_edges = p->n_edges;
    Scope s;
    if (!stub_lba) return ::eao_bundle_adjustment_planes(p, pl, robust, stop, r, planes_out);
    dump_problem(p);
    std::memcpy(r->cam_Tcw, p->cam_Tcw, sizeof(float) * 16 * (size_t)p->n_cams);
    std::memcpy(r->points, p->points, sizeof(float) * 3 * (size_t)p->n_points);
    r->aborted = 0; r->iters[0] = r->iters[1] = 0; r->chi2[0] = r->chi2[1] = 0;
    return EAO_OK;
}
}  // namespace cabi
#define eao_bundle_adjustment_planes cabi::gba
#define eao_orb_extract_ref cabi::orb_extract
#define eao_orb_pyramid cabi::orb_pyramid
#define eao_search_by_projection_points cabi::sbp_points
#define eao_search_by_projection_frames cabi::sbp_frames
#define eao_search_by_bow cabi::sbow
#define eao_pose_optimization cabi::pose
#define eao_local_ba cabi::lba
#include <eaofusion/ORBextractor.h>
#include <eaofusion/ORBmatcher.h>
#include <eaofusion/OptimizerImpl.h>
#undef eao_orb_extract_ref
#undef eao_orb_pyramid
#undef eao_search_by_projection_points
#undef eao_search_by_projection_frames
#undef eao_search_by_bow
#undef eao_pose_optimization
#undef eao_local_ba
#undef eao_bundle_adjustment_planes

// ---- stand-ins: the members the adapters touch, with upstream's locking / cloning behaviour (src/MapPoint.cc:68-91, 385-394; src/KeyFrame.cc:74-107, 268-300)
struct KeyFrame;
struct MapPoint {
    static std::mutex mGlobalMutex;
    std::mutex mMutexPos, mMutexFeatures;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnLastFrameSeen = 0;
    bool mbTrackInView = true, mbBad = false;
    int mnTrackScaleLevel = 0, nObs = 1;
    float mTrackViewCos = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    cv::Mat mWorldPos, mDescriptor;
    std::map<KeyFrame*, size_t> mObservations;
    cv::Mat GetWorldPos() { std::unique_lock<std::mutex> l(mMutexPos); return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat& p) { std::unique_lock<std::mutex> l2(mGlobalMutex); std::unique_lock<std::mutex> l(mMutexPos); p.copyTo(mWorldPos); }
    cv::Mat GetDescriptor() { std::unique_lock<std::mutex> l(mMutexFeatures); return mDescriptor.clone(); }
    int Observations() { std::unique_lock<std::mutex> l(mMutexFeatures); return nObs; }
    bool isBad() { std::unique_lock<std::mutex> l(mMutexFeatures); std::unique_lock<std::mutex> l2(mMutexPos); return mbBad; }
    std::map<KeyFrame*, size_t> GetObservations() { std::unique_lock<std::mutex> l(mMutexFeatures); return mObservations; }
#ifdef EAO_BENCH_EDITED_MAPPOINT
    // the two allocation-free accessors a checkout may add to include/MapPoint.h; the LBA adapter detects and uses them
    template <class F> void ForEachObservation(F&& f) { std::unique_lock<std::mutex> l(mMutexFeatures); for (auto& o : mObservations) f(o.first, o.second); }
    void GetWorldPos(float* xyz) { std::unique_lock<std::mutex> l(mMutexPos); for (int i = 0; i < 3; i++) xyz[i] = mWorldPos.at<float>(i); }
#endif
    void EraseObservation(KeyFrame* kf) { std::unique_lock<std::mutex> l(mMutexFeatures); mObservations.erase(kf); }
    void UpdateNormalAndDepth() {}
    cv::Mat mPosGBA;                              // Optimizer::BundleAdjustment with nLoopKF != 0
    long unsigned int mnBAGlobalForKF = 0;
};
struct MapPlane {                                 // (the map of the bench holds none: the members Optimizer::BundleAdjustment's template names)
    long unsigned int mnId = 0, mnBAGlobalForKF = 0;
    cv::Mat mWorldPos, mPosGBA;
    std::map<KeyFrame*, int> mObservations;
    bool isBad() { return false; }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat& p) { p.copyTo(mWorldPos); }
    std::map<KeyFrame*, int> GetObservations() { return mObservations; }
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
    std::vector<cv::Mat> mvPlaneCoefficients;
    cv::Mat mTcwGBA;
    long unsigned int mnBAGlobalForKF = 0;
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
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors, mTcw;
    std::vector<float> mvScaleFactors, mvuRight, mvInvLevelSigma2;
    float fx, fy, cx, cy, mbf, mb = 0;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    void SetPose(cv::Mat T) { mTcw = T.clone(); }
};
float Frame::mnMinX = 0, Frame::mnMaxX = 640, Frame::mnMinY = 0, Frame::mnMaxY = 480;
float Frame::mfGridElementWidthInv = 64.f / 640.f, Frame::mfGridElementHeightInv = 48.f / 480.f;

template <typename T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), n * sizeof(T)); }
template <typename T> static std::vector<T> rdv(std::ifstream& f, size_t n) { std::vector<T> v(n); rd(f, v.data(), n); return v; }
static cv::Mat colN(const float* v, int n) { cv::Mat m(n, 1, CV_32F); for (int i = 0; i < n; i++) m.at<float>(i) = v[i]; return m; }
static cv::Mat mat44(const float* v) { cv::Mat m(4, 4, CV_32F); for (int i = 0; i < 16; i++) m.at<float>(i / 4, i % 4) = v[i]; return m; }
static cv::Mat desc32(const uint8_t* d) { cv::Mat m(1, 32, CV_8U); std::memcpy(m.data, d, 32); return m; }

struct Stat { double call_ms, cabi_ms; };
static bool g_lbaOnly = false, g_inLba = false;      // `adapter_bench problem.bin lba`: only the LocalBundleAdjustment section is measured and printed (the EDITED-MapPoint build, row 2c)
// median over `reps` of (whole call, time inside the C-ABI); prep() restores the inputs outside the timed region
template <class Prep, class Call>
static Stat measure(int warm, int reps, Prep&& prep, Call&& call) {
    std::vector<double> tc, ti;
    if (g_lbaOnly && !g_inLba) { warm = 0; reps = 1; }
    if (cabi::stub_lba && !g_inLba) { prep(); return Stat{0, 0}; }
    if (cabi::stub_lba) reps = 200;
    for (int r = 0; r < warm + reps; r++) {
        prep();
        cabi::inside_ns = 0;
        const auto t0 = std::chrono::steady_clock::now();
        call();
        const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
        if (r >= warm) { tc.push_back(ns); ti.push_back(cabi::inside_ns); }
    }
    std::sort(tc.begin(), tc.end()); std::sort(ti.begin(), ti.end());
    return Stat{tc[tc.size() / 2] * 1e-6, ti[ti.size() / 2] * 1e-6};
}
static void emit(const char* name, const Stat& s, const char* more, bool last = false) {
    if (g_lbaOnly) {
        if (!g_inLba) return;
        last = true;
    }
    std::printf("  \"%s\": {\"call_ms\": %.4f, \"c_abi_ms\": %.4f, \"adapter_overhead_ms\": %.4f, \"adapter_overhead_frac_of_c_abi\": %.4f%s%s}%s\n", name, s.call_ms, s.cabi_ms,
                s.call_ms - s.cabi_ms, s.cabi_ms > 0 ? (s.call_ms - s.cabi_ms) / s.cabi_ms : 0.0, more[0] ? ", " : "", more, last ? "" : ",");
}

// `adapter_bench <map.bin> gba` (or gba-walk: the library call replaced by an identity result): the whole-map call alone
static int map_section(const char* path) {
    // ------------------------------------------------------------------ Optimizer::BundleAdjustment(vpKFs, vpMP, vpMPl, 10, &stop, nLoopKF, false) over a whole map
    // (argv[3] = the flat map of bench.py's mixed_load_inputs: LoopClosing::RunGlobalBundleAdjustment's call, src/LoopClosing.cc:594 -> src/Optimizer.cc:47-323)
    {
        std::ifstream mf(path, std::ios::binary);
        if (!mf) return 2;
        int32_t hdr[5];
        rd(mf, hdr, 5);
        const int nc = hdr[0], np = hdr[1], ne = hdr[2];
        const std::vector<float> poses = rdv<float>(mf, 16 * (size_t)nc);
        const std::vector<uint8_t> fixed = rdv<uint8_t>(mf, nc);
        const std::vector<float> pts = rdv<float>(mf, 3 * (size_t)np);
        const std::vector<int32_t> ecam = rdv<int32_t>(mf, ne), ept = rdv<int32_t>(mf, ne);
        const std::vector<float> obs = rdv<float>(mf, 3 * (size_t)ne), inv = rdv<float>(mf, ne);
        float K[5];
        rd(mf, K, 5);
        std::vector<KeyFrame> kfs(nc);
        std::vector<MapPoint> mps(np);
        std::vector<KeyFrame*> vk; std::vector<MapPoint*> vm; std::vector<MapPlane*> vpl;
        auto build = [&] {
            for (int c = 0; c < nc; c++) {
                KeyFrame& k = kfs[c];
                k.mnId = fixed[c] ? 0 : (unsigned long)c + 1;      // (upstream fixes the keyframe with mnId 0, src/Optimizer.cc:91)
                k.fx = K[0]; k.fy = K[1]; k.cx = K[2]; k.cy = K[3]; k.mbf = K[4];
                k.Tcw = mat44(&poses[16 * (size_t)c]);
                k.mvKeysUn.clear(); k.mvuRight.clear(); k.mvInvLevelSigma2.clear();
            }
            for (int p = 0; p < np; p++) { mps[p].mnId = p; mps[p].mWorldPos = colN(&pts[3 * (size_t)p], 3); mps[p].mObservations.clear(); }
            for (int e = 0; e < ne; e++) {
                KeyFrame& k = kfs[ecam[e]];
                cv::KeyPoint kp;
                kp.pt.x = obs[3 * (size_t)e]; kp.pt.y = obs[3 * (size_t)e + 1]; kp.octave = (int)k.mvKeysUn.size();
                mps[ept[e]].mObservations[&k] = k.mvKeysUn.size();
                k.mvKeysUn.push_back(kp); k.mvuRight.push_back(obs[3 * (size_t)e + 2]); k.mvInvLevelSigma2.push_back(inv[e]);
            }
            vk.clear(); vm.clear();
            for (int c = 0; c < nc; c++) vk.push_back(&kfs[c]);
            for (int p = 0; p < np; p++) vm.push_back(&mps[(size_t)((long long)p * 7919 % np)]);      // (Map::GetAllMapPoints: the order of a std::set of pointers -- any order)
        };
        bool stop = false;
        std::vector<double> tc, ti;
        for (int r = 0; r < 2 + 5; r++) {
            build();
            cabi::inside_ns = 0;
            const auto t0 = std::chrono::steady_clock::now();
            eaofusion::BundleAdjustment(vk, vm, vpl, 10, &stop, 0, false);
            const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
            if (r >= 2) { tc.push_back(ns); ti.push_back(cabi::inside_ns); }
        }
        std::sort(tc.begin(), tc.end()); std::sort(ti.begin(), ti.end());
        const Stat s = {tc[tc.size() / 2] * 1e-6, ti[ti.size() / 2] * 1e-6};
        char more[384];
        std::snprintf(more, sizeof more, "\"keyframes\": %d, \"map_points\": %d, \"edges\": %d, \"problem_points\": %d, \"problem_edges\": %d", nc, np, ne, cabi::gba_points, cabi::gba_edges);
        std::printf("{\n  \"bundle_adjustment_map\": {\"call_ms\": %.4f, \"c_abi_ms\": %.4f, \"adapter_overhead_ms\": %.4f, %s}\n}\n", s.call_ms, s.cabi_ms, s.call_ms - s.cabi_ms, more);
    }
    return 0;

}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (argc > 2 && (std::string(argv[2]) == "gba" || std::string(argv[2]) == "gba-walk")) {
        cabi::stub_lba = std::string(argv[2]) == "gba-walk";
        return map_section(argv[1]);
    }
    cabi::stub_lba = argc > 2 && std::string(argv[2]) == "lba-walk";
    g_lbaOnly = cabi::stub_lba || (argc > 2 && std::string(argv[2]) == "lba");
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) return 2;
    char more[384];
    std::printf("{\n");
    // ------------------------------------------------------------------ ORBextractor::operator(), as Frame::ExtractORB calls it
    int32_t H, W;
    rd(in, &H, 1); rd(in, &W, 1);
    cv::Mat img(H, W, CV_8UC1);
    rd(in, img.data, (size_t)H * W);
    if (!cabi::stub_lba) {
        ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> keys;
        cv::Mat descriptors;
        ex.keepPyramid = false;
        const Stat a = measure(10, 60, [] {}, [&] { ex(img, cv::Mat(), keys, descriptors); });
        std::snprintf(more, sizeof more, "\"keypoints\": %d, \"keepPyramid\": false", (int)keys.size());
        emit("orb_extractor_call", a, more);
        ex.keepPyramid = true;
        const Stat b = measure(5, 40, [] {}, [&] { ex(img, cv::Mat(), keys, descriptors); });
        std::snprintf(more, sizeof more, "\"keepPyramid\": true, \"level7_cols\": %d", ex.mvImagePyramid[7].cols);
        emit("orb_extractor_call_with_pyramid", b, more);
    }
    // ------------------------------------------------------------------ Optimizer::PoseOptimization(Frame*)
    {
        int32_t n;
        rd(in, &n, 1);
        float T[16], K[5];
        rd(in, T, 16);
        const std::vector<float> Xw = rdv<float>(in, 3 * (size_t)n), obs = rdv<float>(in, 3 * (size_t)n), inv = rdv<float>(in, n);
        rd(in, K, 5);
        std::vector<MapPoint> mps(n);
        Frame F;
        F.N = n; F.fx = K[0]; F.fy = K[1]; F.cx = K[2]; F.cy = K[3]; F.mbf = K[4];
        F.mvKeysUn.resize(n); F.mvuRight.resize(n); F.mvpMapPoints.resize(n); F.mvbOutlier.assign(n, false);
        F.mvInvLevelSigma2.assign(inv.begin(), inv.end());      // the information values come per correspondence: keypoint i names "level" i of this table
        for (int i = 0; i < n; i++) {
            mps[i].mWorldPos = colN(&Xw[3 * (size_t)i], 3);
            F.mvKeysUn[i].pt.x = obs[3 * (size_t)i]; F.mvKeysUn[i].pt.y = obs[3 * (size_t)i + 1]; F.mvKeysUn[i].octave = i;
            F.mvuRight[i] = obs[3 * (size_t)i + 2];
            F.mvpMapPoints[i] = &mps[i];
        }
        int inl = 0;
        const Stat s = measure(5, 40, [&] { F.mTcw = mat44(T); }, [&] { inl = eaofusion::PoseOptimization<MapPoint>(&F); });
        std::snprintf(more, sizeof more, "\"correspondences\": %d, \"inliers\": %d", n, inl);
        emit("pose_optimization", s, more);
    }
    // ------------------------------------------------------------------ Optimizer::LocalBundleAdjustment(pKF, &stop, pMap)
    {
        int32_t nc, np, ne;
        rd(in, &nc, 1); rd(in, &np, 1); rd(in, &ne, 1);
        const std::vector<float> poses = rdv<float>(in, 16 * (size_t)nc);
        const std::vector<uint8_t> fixed = rdv<uint8_t>(in, nc);
        const std::vector<float> pts = rdv<float>(in, 3 * (size_t)np);
        const std::vector<int32_t> ecam = rdv<int32_t>(in, ne), ept = rdv<int32_t>(in, ne);
        const std::vector<float> obs = rdv<float>(in, 3 * (size_t)ne), inv = rdv<float>(in, ne);
        float K[5];
        rd(in, K, 5);
        std::vector<KeyFrame> kfs(nc);
        std::vector<MapPoint> mps(np);
        Map map;
        KeyFrame* pKF = nullptr;
        auto build = [&] {      // the window as LocalMapping hands it over: the newest free keyframe + its covisible ones; fixed cameras reached through the observations
            pKF = nullptr;
            for (int c = 0; c < nc; c++) {
                KeyFrame& k = kfs[c];
                k.mnId = (unsigned long)c + (fixed[c] ? 0 : 100); k.mnBALocalForKF = k.mnBAFixedForKF = ~0ul;      // (ids: fixed ones sort first; none is 0 unless fixed)
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
        bool stop = false;
        g_inLba = true;
        const Stat s = measure(3, 15, build, [&] { eaofusion::LocalBundleAdjustment<MapPoint>(pKF, &stop, &map); });
        // What ANY implementation behind this signature pays to the reference's object model: exactly the accessor calls upstream's own function makes on this
        // window (src/Optimizer.cc:680-738, 800-905, 1085-1137) -- GetVectorCovisibleKeyFrames, GetMapPointMatches, isBad, GetObservations twice per point,
        // GetPose, GetWorldPos, SetPose, SetWorldPos, UpdateNormalAndDepth -- with nothing between them (no graph, no optimisation).
        volatile double sink = 0;
        auto floor_walk = [&] {
            double acc = 0;
            std::vector<KeyFrame*> local(1, pKF), fixedK;
            pKF->mnBALocalForKF = pKF->mnId;
            for (KeyFrame* n : pKF->GetVectorCovisibleKeyFrames()) { n->mnBALocalForKF = pKF->mnId; if (!n->isBad()) local.push_back(n); }
            std::vector<MapPoint*> lm;
            for (KeyFrame* k : local) for (MapPoint* m : k->GetMapPointMatches()) if (m && !m->isBad() && m->mnBALocalForKF != pKF->mnId) { lm.push_back(m); m->mnBALocalForKF = pKF->mnId; }
            for (MapPoint* m : lm) { const std::map<KeyFrame*, size_t> o = m->GetObservations(); for (const auto& ob : o) if (ob.first->mnBALocalForKF != pKF->mnId && ob.first->mnBAFixedForKF != pKF->mnId) { ob.first->mnBAFixedForKF = pKF->mnId; if (!ob.first->isBad()) fixedK.push_back(ob.first); } }
            for (KeyFrame* k : local) acc += k->GetPose().at<float>(0, 0);
            for (KeyFrame* k : fixedK) acc += k->GetPose().at<float>(0, 0);
            for (MapPoint* m : lm) {
                acc += m->GetWorldPos().at<float>(0);
                const std::map<KeyFrame*, size_t> o = m->GetObservations();
                for (const auto& ob : o) if (!ob.first->isBad()) acc += ob.first->mvKeysUn[ob.second].pt.x + ob.first->mvuRight[ob.second];
            }
            std::unique_lock<std::mutex> lock(map.mMutexMapUpdate);
            for (KeyFrame* k : local) { cv::Mat T = k->GetPose(); k->SetPose(T); }
            for (MapPoint* m : lm) { cv::Mat X = m->GetWorldPos(); m->SetWorldPos(X); m->UpdateNormalAndDepth(); }
            sink = acc;
        };
        const Stat fl = measure(3, 15, build, floor_walk);
        std::snprintf(more, sizeof more, "\"keyframes\": %d, \"map_points\": %d, \"edges\": %d, \"window_points\": %d, \"window_edges\": %d, \"reference_accessor_walk_ms\": %.4f, \"adapter_overhead_beyond_accessor_walk_ms\": %.4f", nc, np, ne,
                      cabi::lba_points, cabi::lba_edges, fl.call_ms, s.call_ms - s.cabi_ms - fl.call_ms);
        emit("local_bundle_adjustment", s, more);
        g_inLba = false;
    }
    if (cabi::stub_lba) { std::printf("}\n"); return 0; }
    // ------------------------------------------------------------------ the two SearchByProjection variants of the tracking loop
    {
        int32_t N;
        rd(in, &N, 1);
        const std::vector<float> kx = rdv<float>(in, N), ky = rdv<float>(in, N), ang = rdv<float>(in, N), ur = rdv<float>(in, N);
        const std::vector<int32_t> oct = rdv<int32_t>(in, N);
        const std::vector<uint8_t> desc = rdv<uint8_t>(in, 32 * (size_t)N), occ = rdv<uint8_t>(in, N);
        float Tc[16], K[6], sf[8];
        rd(in, Tc, 16); rd(in, K, 6); rd(in, sf, 8);
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
        Frame Cur, Last;
        Cur.N = N; Cur.fx = K[0]; Cur.fy = K[1]; Cur.cx = K[2]; Cur.cy = K[3]; Cur.mbf = K[4]; Cur.mb = K[5];
        Cur.mvKeysUn.resize(N); Cur.mvuRight = ur; Cur.mvScaleFactors.assign(sf, sf + 8); Cur.mDescriptors = cv::Mat(N, 32, CV_8U);
        std::memcpy(Cur.mDescriptors.data, desc.data(), desc.size());
        for (int i = 0; i < N; i++) { Cur.mvKeysUn[i].pt.x = kx[i]; Cur.mvKeysUn[i].pt.y = ky[i]; Cur.mvKeysUn[i].angle = ang[i]; Cur.mvKeysUn[i].octave = oct[i]; }
        Cur.mvKeys = Cur.mvKeysUn; Cur.mTcw = mat44(Tc); Cur.mvbOutlier.assign(N, false);
        std::vector<MapPoint> holders(N), mps(n);
        std::vector<MapPoint*> vp(n);
        for (int i = 0; i < n; i++) {
            MapPoint& m = mps[i];
            m.mWorldPos = colN(&Xw[3 * (size_t)i], 3); m.mDescriptor = desc32(&ldesc[32 * (size_t)i]);
            m.mTrackProjX = px[i]; m.mTrackProjY = py[i]; m.mTrackProjXR = pxr[i]; m.mTrackViewCos = vc[i]; m.mnTrackScaleLevel = lvl[i]; m.mbTrackInView = !skip[i];
            vp[i] = &m;
        }
        auto reset = [&] { Cur.mvpMapPoints.assign(N, nullptr); for (int k = 0; k < N; k++) if (occ[k]) Cur.mvpMapPoints[k] = &holders[k]; };
        eaofusion::ORBmatcher m1(0.8f, true);
        int nm = 0;
        const Stat a = measure(5, 40, reset, [&] { nm = m1.SearchByProjection(Cur, vp, 1.0f); });
        std::snprintf(more, sizeof more, "\"keypoints\": %d, \"map_points\": %d, \"matches\": %d", N, n, nm);
        emit("search_by_projection_local_map", a, more);
        Last.N = n; Last.mvKeys.resize(n); Last.mvKeysUn.resize(n); Last.mvpMapPoints.resize(n); Last.mvbOutlier.assign(n, false); Last.mTcw = mat44(Tl);
        for (int i = 0; i < n; i++) { Last.mvKeys[i].octave = loct[i]; Last.mvKeysUn[i].angle = lang[i]; Last.mvpMapPoints[i] = valid[i] ? &mps[i] : nullptr; }
        eaofusion::ORBmatcher m2(0.9f, true);
        const Stat b = measure(5, 40, reset, [&] { nm = m2.SearchByProjection(Cur, Last, 7.0f, false); });
        std::snprintf(more, sizeof more, "\"keypoints\": %d, \"last_frame_points\": %d, \"matches\": %d", N, n, nm);
        emit("search_by_projection_last_frame", b, more);
    }
    // ------------------------------------------------------------------ SearchByBoW(KeyFrame*, Frame&, matches)
    {
        struct Raw { int n; std::vector<float> ang; std::vector<int32_t> mp; std::vector<uint8_t> desc; std::map<unsigned, std::vector<unsigned> > fv; } r[2];
        for (int q = 0; q < 2; q++) {
            rd(in, &r[q].n, 1);
            r[q].ang = rdv<float>(in, r[q].n); r[q].mp = rdv<int32_t>(in, r[q].n); r[q].desc = rdv<uint8_t>(in, 32 * (size_t)r[q].n);
            int32_t nn;
            rd(in, &nn, 1);
            const std::vector<uint32_t> id = rdv<uint32_t>(in, nn);
            const std::vector<int32_t> st = rdv<int32_t>(in, nn + 1);
            const std::vector<uint32_t> idx = rdv<uint32_t>(in, st[nn]);
            for (int k = 0; k < nn; k++) r[q].fv[id[k]] = std::vector<unsigned>(idx.begin() + st[k], idx.begin() + st[k + 1]);
        }
        KeyFrame kf;
        Frame F;
        std::vector<MapPoint> mps(r[0].n);
        kf.N = r[0].n; kf.mvKeysUn.resize(kf.N); kf.mDescriptors = cv::Mat(kf.N, 32, CV_8U); kf.mFeatVec = r[0].fv; kf.mvpMapPoints.resize(kf.N);
        std::memcpy(kf.mDescriptors.data, r[0].desc.data(), r[0].desc.size());
        for (int i = 0; i < kf.N; i++) { kf.mvKeysUn[i].angle = r[0].ang[i]; kf.mvpMapPoints[i] = r[0].mp[i] >= 0 ? &mps[i] : nullptr; }
        F.N = r[1].n; F.mvKeys.resize(F.N); F.mDescriptors = cv::Mat(F.N, 32, CV_8U); F.mFeatVec = r[1].fv;
        std::memcpy(F.mDescriptors.data, r[1].desc.data(), r[1].desc.size());
        for (int i = 0; i < F.N; i++) F.mvKeys[i].angle = r[1].ang[i];
        std::vector<MapPoint*> matches;
        eaofusion::ORBmatcher m(0.7f, true);
        int nm = 0;
        const Stat s = measure(5, 40, [] {}, [&] { nm = m.SearchByBoW(&kf, F, matches); });
        std::snprintf(more, sizeof more, "\"keyframe_keypoints\": %d, \"frame_keypoints\": %d, \"matches\": %d", kf.N, F.N, nm);
        emit("search_by_bow_kf_frame", s, more, true);
    }
    std::printf("}\n");
    return 0;
}
