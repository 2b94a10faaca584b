// Drives the remaining ORBmatcher templates of include/eaofusion/ORBmatcher.h (SearchByBoW x2, SearchForTriangulation,
// SearchForInitialization, SearchByProjection loop / relocalisation variants, Fuse x2, SearchBySim3) against stand-ins of
// the reference's KeyFrame / Frame / MapPoint with the member names src/ORBmatcher.cc uses.  Reads a scene written by
// tests/test_gpu_adapters.py, writes every result table; the Python side compares them with the C-ABI called directly.
#include <cstdio>
#include <fstream>
#include <map>
#include <set>
#include <vector>

#include <eaofusion/ORBmatcher.h>

using eaofusion::ORBmatcher;

struct KeyFrame;
struct MapPoint {
    cv::Mat pos, normal, descriptor;
    bool bad = false;
    std::map<KeyFrame*, size_t> obs;
    MapPoint* replacedBy = nullptr;
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    cv::Mat GetDescriptor() { return descriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int Observations() { return (int)obs.size(); }
    bool IsInKeyFrame(KeyFrame* kf) { return obs.count(kf) != 0; }
    int GetIndexInKeyFrame(KeyFrame* kf) { return obs.count(kf) ? (int)obs[kf] : -1; }
    void AddObservation(KeyFrame* kf, size_t idx) { obs[kf] = idx; }
    // upstream's Replace ends in pMP->ComputeDistinctiveDescriptors() (src/MapPoint.cc:177-215): the SURVIVOR's descriptor changes.  Here it takes over the absorbed
    // point's descriptor with its bytes rotated by one -- a change later searches cannot miss (a stand-in that only marks the loser hides stale searches)
    void Replace(MapPoint* other) {
        bad = true; replacedBy = other;
        cv::Mat d = descriptor.clone();
        for (int k = 0; k < 32; k++) d.ptr(0)[k] = descriptor.ptr(0)[(k + 1) % 32];
        other->descriptor = d;
    }
    void SetDistances(float mn, float mx) { mfMinDistance = mn; mfMaxDistance = mx; }
protected:
    float mfMinDistance = 0, mfMaxDistance = 0;
};

struct KeyFrame {
    int N = 0;
    float fx, fy, cx, cy, mbf;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    cv::Mat mDescriptors;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 640, mnMaxY = 480, mnGridCols = 64, mnGridRows = 48;
    float mfGridElementWidthInv = 64.f / 640.f, mfGridElementHeightInv = 48.f / 480.f, mfLogScaleFactor = 0;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    std::vector<MapPoint*> mvpMapPoints;
    cv::Mat Rcw, tcw, Ow;
    std::vector<std::pair<int, MapPoint*> > added;
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (auto* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
    void AddMapPoint(MapPoint* p, size_t idx) { mvpMapPoints[idx] = p; added.push_back({(int)idx, p}); }
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
};

struct Frame {
    int N = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvScaleFactors;
    cv::Mat mDescriptors, mTcw;
    float mfLogScaleFactor = 0, fx, fy, cx, cy, mbf = 0, mb = 0;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
};
float Frame::mnMinX = 0, Frame::mnMaxX = 640, Frame::mnMinY = 0, Frame::mnMaxY = 480;
float Frame::mfGridElementWidthInv = 64.f / 640.f, Frame::mfGridElementHeightInv = 48.f / 480.f;

template <typename T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), n * sizeof(T)); }
template <typename T> static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), n * sizeof(T)); }
static cv::Mat col3(const float* v) { cv::Mat m(3, 1, CV_32F); for (int i = 0; i < 3; i++) m.at<float>(i) = v[i]; return m; }

struct RawFrame {
    int n = 0;
    std::vector<float> x, y, ang, ur;
    std::vector<int32_t> oct, mp;
    std::vector<uint8_t> desc;
    std::map<unsigned, std::vector<unsigned> > fv;
    void read(std::ifstream& f) {
        rd(f, &n, 1);
        x.resize(n); y.resize(n); ang.resize(n); ur.resize(n); oct.resize(n); mp.resize(n); desc.resize((size_t)n * 32);
        rd(f, x.data(), n); rd(f, y.data(), n); rd(f, ang.data(), n); rd(f, ur.data(), n); rd(f, oct.data(), n); rd(f, mp.data(), n);
        rd(f, desc.data(), desc.size());
        int nn = 0;
        rd(f, &nn, 1);
        std::vector<uint32_t> id(nn); std::vector<int32_t> st(nn + 1);
        rd(f, id.data(), nn); rd(f, st.data(), nn + 1);
        std::vector<uint32_t> idx(st[nn]);
        rd(f, idx.data(), idx.size());
        for (int k = 0; k < nn; k++) fv[id[k]] = std::vector<unsigned>(idx.begin() + st[k], idx.begin() + st[k + 1]);
    }
    template <class T>
    void fill(T& K) const {
        K.N = n;
        K.mvKeysUn.resize(n); K.mvuRight = ur;
        for (int i = 0; i < n; i++) { K.mvKeysUn[i].pt.x = x[i]; K.mvKeysUn[i].pt.y = y[i]; K.mvKeysUn[i].angle = ang[i]; K.mvKeysUn[i].octave = oct[i]; }
        K.mDescriptors = cv::Mat(n, 32, CV_8U);
        for (int i = 0; i < n; i++) std::memcpy(K.mDescriptors.ptr(i), &desc[(size_t)i * 32], 32);
        K.mFeatVec = fv;
    }
};

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    RawFrame r1, r2;
    r1.read(in); r2.read(in);
    int np = 0;
    rd(in, &np, 1);
    std::vector<uint8_t> pact(np), pdesc((size_t)np * 32);
    std::vector<float> pX((size_t)np * 3), pN((size_t)np * 3), pmin(np), pmax(np);
    rd(in, pact.data(), np); rd(in, pX.data(), pX.size()); rd(in, pN.data(), pN.size()); rd(in, pmin.data(), np); rd(in, pmax.data(), np);
    rd(in, pdesc.data(), pdesc.size());
    float T1[16], T2[16], Kc[4], bf, F12[9], S[16], R12[9], t12[3], sf[8], s2[8], is2[8], logsf;
    rd(in, T1, 16); rd(in, T2, 16); rd(in, Kc, 4); rd(in, &bf, 1); rd(in, F12, 9); rd(in, S, 16); rd(in, R12, 9); rd(in, t12, 3);
    rd(in, sf, 8); rd(in, s2, 8); rd(in, is2, 8); rd(in, &logsf, 1);
    if (!in) { fprintf(stderr, "short scene file\n"); return 3; }

    std::vector<MapPoint> store(np);
    for (int i = 0; i < np; i++) {
        store[i].pos = col3(&pX[(size_t)i * 3]); store[i].normal = col3(&pN[(size_t)i * 3]);
        store[i].descriptor = cv::Mat(1, 32, CV_8U);
        std::memcpy(store[i].descriptor.ptr(0), &pdesc[(size_t)i * 32], 32);
        store[i].SetDistances(pmin[i], pmax[i]);     // raw mfMinDistance / mfMaxDistance
        store[i].bad = !pact[i];
    }
    auto make_kf = [&](const RawFrame& r, const float* T) {
        KeyFrame K;
        r.fill(K);
        K.fx = Kc[0]; K.fy = Kc[1]; K.cx = Kc[2]; K.cy = Kc[3]; K.mbf = bf;
        K.mvScaleFactors.assign(sf, sf + 8); K.mvLevelSigma2.assign(s2, s2 + 8); K.mvInvLevelSigma2.assign(is2, is2 + 8); K.mfLogScaleFactor = logsf;
        K.mvpMapPoints.assign(r.n, nullptr);
        K.Rcw = cv::Mat(3, 3, CV_32F); K.tcw = cv::Mat(3, 1, CV_32F); K.Ow = cv::Mat(3, 1, CV_32F);
        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) K.Rcw.at<float>(a, b) = T[a * 4 + b]; K.tcw.at<float>(a) = T[a * 4 + 3]; }
        for (int a = 0; a < 3; a++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)T[k * 4 + a] * (double)T[k * 4 + 3];
            K.Ow.at<float>(a) = (float)(-s);
        }
        return K;
    };
    KeyFrame K1 = make_kf(r1, T1), K2 = make_kf(r2, T2);
    for (int k = 0; k < r1.n; k++) if (r1.mp[k] >= 0) K1.mvpMapPoints[k] = &store[r1.mp[k]];
    for (int k = 0; k < r2.n; k++) if (r2.mp[k] >= 0) K2.mvpMapPoints[k] = &store[r2.mp[k]];
    auto mat = [](const float* v, int rows, int cols) { cv::Mat m(rows, cols, CV_32F); for (int a = 0; a < rows; a++) for (int b = 0; b < cols; b++) m.at<float>(a, b) = v[a * cols + b]; return m; };
    ORBmatcher m75(0.75f, true);
    auto dump_ptrs = [&](const std::vector<MapPoint*>& v) {
        std::vector<int32_t> idx(v.size());
        for (size_t i = 0; i < v.size(); i++) idx[i] = v[i] ? (int32_t)(v[i] - store.data()) : -1;
        const int32_t n = (int32_t)idx.size();
        wr(out, &n, 1); wr(out, idx.data(), idx.size());
    };

    // 1. SearchByBoW(KF, KF)
    { std::vector<MapPoint*> m12; const int32_t n = m75.SearchByBoW(&K1, &K2, m12); wr(out, &n, 1); dump_ptrs(m12); }
    // 2. SearchByBoW(KF, Frame)
    Frame F2;
    r2.fill(F2);
    F2.mvKeys = F2.mvKeysUn; F2.mvScaleFactors.assign(sf, sf + 8); F2.mfLogScaleFactor = logsf;
    F2.fx = Kc[0]; F2.fy = Kc[1]; F2.cx = Kc[2]; F2.cy = Kc[3];
    F2.mvpMapPoints.assign(r2.n, nullptr); F2.mTcw = mat(T2, 4, 4);
    { std::vector<MapPoint*> mf; const int32_t n = m75.SearchByBoW(&K1, F2, mf); wr(out, &n, 1); dump_ptrs(mf); }
    // 3. SearchForTriangulation: keypoints with a map point at an even index count as occupied
    {
        KeyFrame A = K1, B = K2;
        for (int k = 0; k < A.N; k++) if (!(A.mvpMapPoints[k] && k % 2 == 0)) A.mvpMapPoints[k] = nullptr;
        for (int k = 0; k < B.N; k++) if (!(B.mvpMapPoints[k] && k % 3 == 0)) B.mvpMapPoints[k] = nullptr;
        std::vector<std::pair<size_t, size_t> > pairs;
        const int32_t n = m75.SearchForTriangulation(&A, &B, mat(F12, 3, 3), pairs, false);
        std::vector<int32_t> m12(A.N, -1);
        for (auto& p : pairs) m12[p.first] = (int32_t)p.second;
        wr(out, &n, 1); wr(out, m12.data(), m12.size());
        // ... and the batched template (round 4): three neighbours (B, B again, A itself) in one call = three single calls
        KeyFrame A2 = A;
        std::vector<KeyFrame*> nb = {&B, &B, &A2};
        std::vector<cv::Mat> vF = {mat(F12, 3, 3), mat(F12, 3, 3), mat(F12, 3, 3)};
        std::vector<std::vector<std::pair<size_t, size_t> > > vv;
        m75.SearchForTriangulationBatch(&A, nb, vF, vv, false);
        for (size_t q = 0; q < nb.size(); q++) {
            std::vector<std::pair<size_t, size_t> > one;
            m75.SearchForTriangulation(&A, nb[q], vF[q], one, false);
            if (one != vv[q]) { std::fprintf(stderr, "SearchForTriangulationBatch: neighbour %zu differs from its single call (%zu vs %zu pairs)\n", q, vv[q].size(), one.size()); return 3; }
        }
        if (vv[0] != pairs) { std::fprintf(stderr, "SearchForTriangulationBatch: neighbour 0 differs from the single call above\n"); return 3; }
        // ... and over keyframe handles (round 5): resident frames, selection on the device; then after the keyframe gained a map point (refresh)
        eaofusion::KeyFrameHandles H;
        std::vector<const eao_keyframe*> hs = {H.of(nb[0]), H.of(nb[1]), H.of(nb[2])};
        std::vector<std::vector<std::pair<size_t, size_t> > > vh;
        m75.SearchForTriangulationBatch(H.of(&A), &A, hs, nb, vF, vh, false);
        if (vh != vv) { std::fprintf(stderr, "SearchForTriangulationBatch over keyframe handles differs from the host-array batch\n"); return 3; }
        if (H.size() != 3) { std::fprintf(stderr, "KeyFrameHandles: %zu handles for 3 distinct keyframes\n", H.size()); return 3; }
        if (!pairs.empty()) {
            static MapPoint extra;
            A.mvpMapPoints[pairs[0].first] = &extra;      // the first matched keypoint of A now holds a map point: upstream skips it
            H.refresh(&A);
            m75.SearchForTriangulationBatch(H.of(&A), &A, hs, nb, vF, vh, false);
            std::vector<std::pair<size_t, size_t> > one;
            m75.SearchForTriangulation(&A, nb[0], vF[0], one, false);
            if (vh[0] != one || (one.size() && one[0].first == pairs[0].first)) { std::fprintf(stderr, "KeyFrameHandles::refresh: the occupancy did not reach the device\n"); return 3; }
            A.mvpMapPoints[pairs[0].first] = nullptr;
        }
        std::fprintf(stderr, "SearchForTriangulationBatch over keyframe handles: identical (%zu + %zu + %zu pairs)\n", vh[0].size(), vh[1].size(), vh[2].size());
    }
    // 4. SearchForInitialization
    {
        Frame F1;
        r1.fill(F1);
        F1.mvKeys = F1.mvKeysUn; F1.mvpMapPoints.assign(r1.n, nullptr); F1.mvScaleFactors.assign(sf, sf + 8);
        std::vector<cv::Point2f> prev(r1.n);
        for (int i = 0; i < r1.n; i++) { prev[i].x = r1.x[i]; prev[i].y = r1.y[i]; }
        std::vector<int> m12;
        ORBmatcher m9(0.9f, true);
        const int32_t n = m9.SearchForInitialization(F1, F2, prev, m12, 100);
        wr(out, &n, 1); wr(out, m12.data(), m12.size());
        std::vector<float> pm((size_t)r1.n * 2);
        for (int i = 0; i < r1.n; i++) { pm[2 * i] = prev[i].x; pm[2 * i + 1] = prev[i].y; }
        wr(out, pm.data(), pm.size());
    }
    // 5. SearchByProjection(KF, Scw, points, matched, th)
    std::vector<MapPoint*> all(np);
    for (int i = 0; i < np; i++) all[i] = &store[i];
    {
        std::vector<MapPoint*> matched(K2.N, nullptr);
        for (int k = 0; k < K2.N; k += 13) matched[k] = &store[0];         // pre-occupied slots; point 0 is "already found"
        const int32_t n = m75.SearchByProjection(&K2, mat(S, 4, 4), all, matched, 10);
        for (int k = 0; k < K2.N; k += 13) if (matched[k] == &store[0]) matched[k] = nullptr;
        wr(out, &n, 1); dump_ptrs(matched);
    }
    // 6. SearchByProjection(Frame, KF, already found, th, ORBdist): the points of K1 into frame F2
    {
        std::set<MapPoint*> found;
        for (int i = 0; i < np; i += 17) found.insert(&store[i]);
        const int32_t n = m75.SearchByProjection(F2, &K1, found, 15.f, 100);
        wr(out, &n, 1); dump_ptrs(F2.mvpMapPoints);
    }
    // 7. SearchBySim3
    { std::vector<MapPoint*> m12(K1.N, nullptr); const float s12 = 1.0f; const int32_t n = m75.SearchBySim3(&K1, &K2, m12, s12, mat(R12, 3, 3), mat(t12, 3, 1), 7.5f); wr(out, &n, 1); dump_ptrs(m12); }
    // 8. Fuse(KF, Scw, ...) into an empty copy of K2: every hit is an AddMapPoint
    {
        KeyFrame B = K2;
        B.mvpMapPoints.assign(B.N, nullptr);
        std::vector<MapPoint*> repl(np, nullptr);
        const int32_t n = m75.Fuse(&B, mat(S, 4, 4), all, 3.0f, repl);
        std::vector<int32_t> best(np, -1);
        for (auto& a : B.added) best[a.second - store.data()] = a.first;
        wr(out, &n, 1); wr(out, best.data(), best.size());
    }
    // 9. Fuse(KF, points, th) into K2 with its own points: replacements and additions
    std::vector<cv::Mat> descBefore(np);      // (Replace changes the survivor's descriptor: the shared `store` gets its own back for the blocks below)
    for (int i = 0; i < np; i++) descBefore[i] = store[i].descriptor.clone();
    {
        KeyFrame B = K2;
        for (int k = 0; k < B.N; k++) if (B.mvpMapPoints[k]) B.mvpMapPoints[k]->AddObservation(&B, k);
        std::vector<MapPoint> fresh(store.begin(), store.end());          // same geometry, not yet in the keyframe
        std::vector<MapPoint*> cand(np);
        for (int i = 0; i < np; i++) { fresh[i].obs.clear(); fresh[i].bad = !pact[i]; cand[i] = &fresh[i]; }
        const int32_t n = m75.Fuse(&B, cand, 3.0f);
        int32_t replaced = 0, addedN = (int32_t)B.added.size();
        for (auto& p : fresh) if (p.replacedBy) replaced++;
        for (auto& p : store) if (p.replacedBy) replaced++;
        wr(out, &n, 1); wr(out, &replaced, 1); wr(out, &addedN, 1);
    }
    for (int i = 0; i < np; i++) store[i].descriptor = descBefore[i];
    // 9b. FuseBatch (round 4): the same candidates against three target keyframes (K2, K1, K2 again -- the third target meets what the first one changed) in one
    //     library call, against three Fuse calls in a row on an identical second world: per-target counts, every keyframe slot, every replacement
    {
        struct World {
            std::vector<MapPoint> own, fresh;      // the keyframes' own points (copies of `store`) and the candidates (same geometry, in no keyframe)
            KeyFrame T[3];
            std::vector<MapPoint*> cand;
        };
        auto build = [&](World& w) {
            w.own.assign(store.begin(), store.end());
            w.fresh.assign(store.begin(), store.end());
            for (auto& p : w.own) { p.obs.clear(); p.replacedBy = nullptr; }
            w.T[0] = K2; w.T[1] = K1; w.T[2] = K2;
            for (int q = 0; q < 3; q++) {
                w.T[q].added.clear();
                for (int k = 0; k < w.T[q].N; k++) {
                    MapPoint*& p = w.T[q].mvpMapPoints[k];
                    if (p) { p = &w.own[p - store.data()]; if (q != 2 || (k % 3)) p->AddObservation(&w.T[q], k); else p = nullptr; }      // (the third target starts with a third of its slots empty)
                }
            }
            w.cand.resize(np);
            static KeyFrame elsewhere[4];      // every second candidate is seen by four other keyframes: it wins the Observations() comparison and survives its fusions
            for (int i = 0; i < np; i++) {
                w.fresh[i].obs.clear(); w.fresh[i].replacedBy = nullptr; w.fresh[i].bad = !pact[i]; w.cand[i] = &w.fresh[i];
                if (i % 2) for (int z = 0; z < 4; z++) w.fresh[i].AddObservation(&elsewhere[z], (size_t)i);
            }
        };
        World wa, wb;
        build(wa); build(wb);
        int nSeq[3], total = 0;
        for (int q = 0; q < 3; q++) { nSeq[q] = m75.Fuse(&wa.T[q], wa.cand, 3.0f); total += nSeq[q]; }
        std::vector<KeyFrame*> targets = {&wb.T[0], &wb.T[1], &wb.T[2]};
        std::vector<int> nBatch;
        const int totalB = m75.FuseBatch(targets, wb.cand, 3.0f, &nBatch);
        // ... and a third world through keyframe handles (frames resident on the device, eao_kf_fuse_search)
        World wc;
        build(wc);
        {
            eaofusion::KeyFrameHandles H;
            std::vector<KeyFrame*> tc = {&wc.T[0], &wc.T[1], &wc.T[2]};
            std::vector<const eao_keyframe*> hs = {H.of(tc[0]), H.of(tc[1]), H.of(tc[2])};
            std::vector<int> nH;
            const int totalH = m75.FuseBatch(tc, wc.cand, 3.0f, &nH, &hs);
            int dh = totalH != total;
            for (int q = 0; q < 3; q++) {
                if (nH[q] != nSeq[q]) dh++;
                for (int k = 0; k < wa.T[q].N; k++) {
                    MapPoint* pa = wa.T[q].mvpMapPoints[k]; MapPoint* pc = wc.T[q].mvpMapPoints[k];
                    const long ia = !pa ? -1 : (pa >= wa.own.data() && pa < wa.own.data() + wa.own.size()) ? pa - wa.own.data() : 100000 + (pa - wa.fresh.data());
                    const long ic = !pc ? -1 : (pc >= wc.own.data() && pc < wc.own.data() + wc.own.size()) ? pc - wc.own.data() : 100000 + (pc - wc.fresh.data());
                    if (ia != ic) dh++;
                }
            }
            if (dh) { std::fprintf(stderr, "FuseBatch over keyframe handles: %d differences from three Fuse calls (%d / %d)\n", dh, totalH, total); return 3; }
            std::fprintf(stderr, "FuseBatch over keyframe handles: identical to three Fuse calls (%zu handles)\n", H.size());
        }
        auto id = [&](World& w, MapPoint* p) -> long { if (!p) return -1; if (p >= w.own.data() && p < w.own.data() + w.own.size()) return p - w.own.data(); return 100000 + (p - w.fresh.data()); };
        int diff = totalB != total;
        for (int q = 0; q < 3; q++) {
            if (nBatch[q] != nSeq[q]) diff++;
            for (int k = 0; k < wa.T[q].N; k++) if (id(wa, wa.T[q].mvpMapPoints[k]) != id(wb, wb.T[q].mvpMapPoints[k])) diff++;
            if (wa.T[q].added.size() != wb.T[q].added.size()) diff++;
        }
        for (int i = 0; i < np; i++) {
            if (wa.fresh[i].bad != wb.fresh[i].bad || id(wa, wa.fresh[i].replacedBy) != id(wb, wb.fresh[i].replacedBy)) diff++;
            if (wa.own[i].bad != wb.own[i].bad || id(wa, wa.own[i].replacedBy) != id(wb, wb.own[i].replacedBy)) diff++;
            if (wa.fresh[i].obs.size() != wb.fresh[i].obs.size()) diff++;
        }
        // (the third target -- K2 again -- fuses nothing in EITHER world: the candidates that survived the first target carry their absorbed partners' rotated
        //  descriptors by then.  A batch that searched all targets up front on the initial descriptors would fuse dozens there: that is the difference this block exists to catch.)
        if (diff || total < 100 || nSeq[0] == 0 || nSeq[1] == 0) { std::fprintf(stderr, "FuseBatch: %d differences from three Fuse calls (fused %d / %d: %d %d %d)\n", diff, totalB, total, nSeq[0], nSeq[1], nSeq[2]); return 3; }
        std::fprintf(stderr, "FuseBatch: %d fused over three targets (%d %d %d), identical to three Fuse calls\n", total, nSeq[0], nSeq[1], nSeq[2]);
    }
    // 10. MapPoint::ComputeDistinctiveDescriptors over the observations of each map point (K1 and K2 keypoints)
    {
        std::vector<std::vector<cv::Mat> > sets(np);
        for (int k = 0; k < r1.n; k++) if (r1.mp[k] >= 0) sets[r1.mp[k]].push_back(K1.mDescriptors.row(k));
        for (int k = 0; k < r2.n; k++) if (r2.mp[k] >= 0) sets[r2.mp[k]].push_back(K2.mDescriptors.row(k));
        sets[0].clear();
        const std::vector<int> best = ORBmatcher::DistinctiveDescriptors(sets);
        std::vector<int32_t> b32(best.begin(), best.end());
        wr(out, b32.data(), b32.size());
    }
    printf("search_adapter_test ok\n");
    return 0;
}
