// Drives the templates of include/eaofusion/Frame.h (IsInFrustum, AssignFeaturesToGrid, ComputeStereoFromRGBD) against
// stand-ins of the reference's Frame / MapPoint with the member names src/Frame.cc and src/Tracking.cc use, and compares
// with a plain restatement of the reference loops written here (float Mats: double accumulation, one rounding).
// Exit code 0 = every table agrees.
#include <cmath>
#include <cstdio>
#include <vector>

#include <eaofusion/Frame.h>

struct MapPoint {
    cv::Mat pos, normal;
    bool bad = false, mbTrackInView = false;
    float mTrackProjX = -7, mTrackProjY = -7, mTrackProjXR = -7, mTrackViewCos = -7;
    int mnTrackScaleLevel = -7;
    long mnLastFrameSeen = -1;
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    void SetDistances(float mn, float mx) { mfMinDistance = mn; mfMaxDistance = mx; }
protected:
    float mfMinDistance = 0, mfMaxDistance = 0;
};

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64
struct Frame {
    int N = 0;
    long mnId = 5;
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    float mbf = 40.f, mfLogScaleFactor = std::log(1.2f);
    cv::Mat mRcw, mtcw, mOw, mTcw;
    cv::Mat GetCameraCenter() { return mOw.clone(); }
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
};
float Frame::fx = 517.3f, Frame::fy = 516.5f, Frame::cx = 318.6f, Frame::cy = 255.3f;
float Frame::mnMinX = -2.5f, Frame::mnMaxX = 643.25f, Frame::mnMinY = -1.75f, Frame::mnMaxY = 481.5f;
float Frame::mfGridElementWidthInv = 64.f / (643.25f + 2.5f), Frame::mfGridElementHeightInv = 48.f / (481.5f + 1.75f);

static unsigned long long g_seed = 88172645463325252ull;
static double urand() { g_seed ^= g_seed << 13; g_seed ^= g_seed >> 7; g_seed ^= g_seed << 17; return (double)(g_seed >> 11) / 9007199254740992.0; }

int main() {
    int bad = 0;
    Frame F;
    // pose: a small rotation about y and a translation
    const float a = 0.2f;
    F.mRcw = cv::Mat(3, 3, CV_32F); F.mtcw = cv::Mat(3, 1, CV_32F); F.mOw = cv::Mat(3, 1, CV_32F);
    const float R[9] = {std::cos(a), 0, std::sin(a), 0, 1, 0, -std::sin(a), 0, std::cos(a)}, t[3] = {0.1f, -0.05f, 0.3f};
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) F.mRcw.at<float>(r, c) = R[3 * r + c]; F.mtcw.at<float>(r) = t[r]; }
    for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += -(double)R[3 * k + r] * t[k]; F.mOw.at<float>(r) = (float)s; }
    F.mTcw = cv::Mat::eye(4, 4, CV_32F);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) F.mTcw.at<float>(r, c) = R[3 * r + c]; F.mTcw.at<float>(r, 3) = t[r]; }
    // ---- 1. IsInFrustum over a local map
    const int np = 3000;
    std::vector<MapPoint> store(np);
    std::vector<MapPoint*> local(np);
    for (int i = 0; i < np; i++) {
        MapPoint& p = store[i];
        p.pos = cv::Mat(3, 1, CV_32F); p.normal = cv::Mat(3, 1, CV_32F);
        float X[3] = {(float)(urand() * 12 - 6), (float)(urand() * 8 - 4), (float)(urand() * 12 - 2)};
        double nn = 0; float nv[3];
        for (int k = 0; k < 3; k++) { nv[k] = X[k] - F.mOw.at<float>(k) + (float)(urand() - 0.5) * 2.0f; nn += (double)nv[k] * nv[k]; }
        for (int k = 0; k < 3; k++) { p.pos.at<float>(k) = X[k]; p.normal.at<float>(k) = (float)(nv[k] / std::sqrt(nn)); }
        const float d = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        p.SetDistances(d * (float)(0.3 + urand()), d * (float)(0.8 + 2 * urand()));
        p.bad = i % 17 == 0;
        p.mnLastFrameSeen = i % 11 == 0 ? F.mnId : -1;
        local[i] = &p;
    }
    const int nToMatch = eaofusion::IsInFrustum(F, local, 0.5f, [&](MapPoint* p) { return p->mnLastFrameSeen == F.mnId || p->isBad(); });
    int expect = 0;
    for (int i = 0; i < np; i++) {
        MapPoint& p = store[i];
        const bool skipped = p.mnLastFrameSeen == F.mnId || p.bad;
        // restatement of src/Frame.cc:638-695
        bool in = false; float u = 0, v = 0, ur = 0, vc = 0; int lvl = 0;
        do {
            float Pc[3];
            for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)R[3 * r + k] * (double)p.pos.at<float>(k); Pc[r] = (float)(s + (double)t[r]); }
            if (Pc[2] < 0.0f) break;
            const float invz = 1.0f / Pc[2];
            u = Frame::fx * Pc[0] * invz + Frame::cx; v = Frame::fy * Pc[1] * invz + Frame::cy;
            if (u < Frame::mnMinX || u > Frame::mnMaxX || v < Frame::mnMinY || v > Frame::mnMaxY) break;
            float PO[3]; double n2 = 0, dot = 0;
            for (int k = 0; k < 3; k++) { PO[k] = p.pos.at<float>(k) - F.mOw.at<float>(k); n2 += (double)PO[k] * PO[k]; dot += (double)PO[k] * p.normal.at<float>(k); }
            const float dist = (float)std::sqrt(n2);
            if (dist < p.GetMinDistanceInvariance() || dist > p.GetMaxDistanceInvariance()) break;
            vc = (float)(dot / (double)dist);
            if (vc < 0.5f) break;
            struct Raw : MapPoint { static float mx(MapPoint* q) { return q->*(&Raw::mfMaxDistance); } };
            const float ratio = Raw::mx(&p) / dist;
            lvl = (int)std::ceil(std::log(ratio) / F.mfLogScaleFactor);
            ur = u - F.mbf * invz;
            in = true;
        } while (false);
        if (skipped) {
            if (p.mbTrackInView || p.mTrackProjX != -7) { bad++; }
            continue;
        }
        expect += in;
        if (p.mbTrackInView != in) { bad++; continue; }
        if (in && (p.mTrackProjX != u || p.mTrackProjY != v || p.mTrackProjXR != ur || p.mTrackViewCos != vc || p.mnTrackScaleLevel != lvl)) bad++;
        if (!in && p.mTrackProjX != -7) bad++;
    }
    if (nToMatch != expect || expect < 100) { fprintf(stderr, "IsInFrustum: %d in view, expected %d\n", nToMatch, expect); bad++; }
    fprintf(stderr, "IsInFrustum: %d of %d in view, %d mismatches so far\n", nToMatch, np, bad);
    // ---- 2. AssignFeaturesToGrid
    F.N = 2500;
    F.mvKeys.resize(F.N); F.mvKeysUn.resize(F.N);
    for (int i = 0; i < F.N; i++) {
        F.mvKeys[i].pt.x = (float)(urand() * 639.9); F.mvKeys[i].pt.y = (float)(urand() * 479.9);
        F.mvKeysUn[i].pt.x = F.mvKeys[i].pt.x + (float)(urand() * 8 - 4); F.mvKeysUn[i].pt.y = F.mvKeys[i].pt.y + (float)(urand() * 8 - 4);
    }
    eaofusion::AssignFeaturesToGrid(F);
    std::vector<std::size_t> ref[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    for (int i = 0; i < F.N; i++) {   // src/Frame.cc:604-611, 751-761
        const int px = (int)std::round((F.mvKeysUn[i].pt.x - Frame::mnMinX) * Frame::mfGridElementWidthInv);
        const int py = (int)std::round((F.mvKeysUn[i].pt.y - Frame::mnMinY) * Frame::mfGridElementHeightInv);
        if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) continue;
        ref[px][py].push_back(i);
    }
    size_t total = 0;
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) { if (ref[i][j] != F.mGrid[i][j]) bad++; total += ref[i][j].size(); }
    fprintf(stderr, "AssignFeaturesToGrid: %zu of %d keypoints inside the grid, %d mismatches so far\n", total, F.N, bad);
    // ---- 3. ComputeStereoFromRGBD
    cv::Mat depth(480, 640, CV_32F);
    for (int y = 0; y < 480; y++)
        for (int x = 0; x < 640; x++) depth.at<float>(y, x) = urand() < 0.2 ? 0.f : (float)(0.3 + urand() * 8);
    eaofusion::ComputeStereoFromRGBD(F, depth);
    int nd = 0;
    for (int i = 0; i < F.N; i++) {
        const float d = depth.at<float>((int)F.mvKeys[i].pt.y, (int)F.mvKeys[i].pt.x);
        const float ed = d > 0 ? d : -1.f, eu = d > 0 ? F.mvKeysUn[i].pt.x - F.mbf / d : -1.f;
        if (F.mvDepth[i] != ed || F.mvuRight[i] != eu) bad++;
        nd += d > 0;
    }
    fprintf(stderr, "ComputeStereoFromRGBD: %d of %d keypoints with depth, %d mismatches in all\n", nd, F.N, bad);
    return bad ? 1 : 0;
}
