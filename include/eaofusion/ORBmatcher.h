// ORBmatcher.h -- the distance side of the reference matcher (reference include/ORBmatcher.h:37-102,
// src/ORBmatcher.cc:37-39, 1603-1665) on top of the C-ABI.
//
// What is here: constructor, TH_LOW / TH_HIGH / HISTO_LENGTH, static DescriptorDistance, ComputeThreeMaxima, the two
// per-frame guided searches SearchByProjection(Frame&, vector<MapPoint*>&, th) and SearchByProjection(Frame& Cur,
// const Frame& Last, th, bMono) (candidate lists + distances on the GPU, the greedy assignment replayed inside the
// library), and the batched primitive the remaining Search* routines are built from: best-two Hamming search of a set
// of query descriptors against a set of train descriptors under a candidate mask, with the reference's tie order.
// The remaining searches (BoW x2, triangulation, initialisation, Sim3, loop / relocalisation projection, Fuse x2) are
// templates further down: they flatten what upstream reads through the KeyFrame / MapPoint accessors, call the library
// (candidates + distances on the GPU, upstream's selection loops replayed in order) and apply the result the way upstream
// does -- including the map mutations of Fuse, which stay on the caller's side of the C-ABI.
//
// HOW IT IS BOUND: the reference header include/ORBmatcher.h stays as it is; src/ORBmatcher.cc leaves
// the build and src/ORBmatcher_hip.cc (whole file in INTEGRATION.md) defines ORB_SLAM2::ORBmatcher's members as one-line
// forwards to the class below.  Hence its own namespace and its own include guard: both headers are seen by that file.
#ifndef EAOFUSION_ORBMATCHER_H
#define EAOFUSION_ORBMATCHER_H

#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../eao_fusion.h"
#include "cv_compat.h"

namespace eaofusion {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (1 x 32 CV_8U each), reference :1649-1665.
    // One pair per call is an upload, a kernel launch and a download (tens of microseconds):
    // kept for completeness of the class surface only.  The reference's two outside callers are loops and are routed to
    // the batched entry points instead -- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:281) to
    // DistinctiveDescriptor() below, Frame::ComputeStereoMatches (src/Frame.cc:914) to eaofusion::ComputeStereoMatches.
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
        uint16_t d = 0;
        check(eao_hamming_matrix(a.ptr(0), 1, b.ptr(0), 1, &d), "eao_hamming_matrix");
        return d;
    }

    // D[i * nb + j] for descriptor matrices (rows of 32 bytes, continuous)
    static void DistanceMatrix(const cv::Mat& A, const cv::Mat& B, std::vector<uint16_t>& D) {
        D.resize((size_t)A.rows * B.rows);
        if (A.rows && B.rows) check(eao_hamming_matrix(A.ptr(0), A.rows, B.ptr(0), B.rows, D.data()), "eao_hamming_matrix");
    }

    // per query row: best / second-best distance and their train indices over the candidates allowed by `mask`
    // (A.rows x B.rows bytes, nullptr = all).  Equals the result of the reference's candidate loops
    // "if(dist<bestDist){bestDist2=bestDist;bestDist=dist;bestIdx=idx;} else if(dist<bestDist2) bestDist2=dist;" (:102-114).
    static void BestTwo(const cv::Mat& A, const cv::Mat& B, const unsigned char* mask, std::vector<eao_best2>& out) {
        out.resize(A.rows);
        if (A.rows && B.rows) check(eao_hamming_best2(A.ptr(0), A.rows, B.ptr(0), B.rows, mask, out.data()), "eao_hamming_best2");
    }

    // ---- guided searches of the per-frame tracking loop.  Templates over the reference's Frame / MapPoint classes (same
    //      member names as src/ORBmatcher.cc uses); src/ORBmatcher_hip.cc instantiates them with the reference's classes.

    // reference :45-129 (TrackLocalMap -> SearchLocalPoints)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3) {
        const int M = (int)vpMapPoints.size();
        static thread_local std::vector<float> px, py, pxr, vc;
        static thread_local std::vector<int32_t> lvl, match;
        static thread_local std::vector<uint8_t> skip, desc;
        static thread_local FrameArrays fa;
        px.resize(M); py.resize(M); pxr.resize(M); vc.resize(M); lvl.resize(M); match.assign(M, -1); skip.resize(M); desc.resize((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            MapPointT* pMP = vpMapPoints[i];
            skip[i] = (!pMP->mbTrackInView || pMP->isBad()) ? 1 : 0;
            px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR; vc[i] = pMP->mTrackViewCos;
            lvl[i] = pMP->mnTrackScaleLevel;
            if (!skip[i]) { const cv::Mat d = pMP->GetDescriptor(); std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32); }
        }
        const eao_frame_view v = view(F, fa);
        int nm = 0;
        check(eao_search_by_projection_points(&v, M, px.data(), py.data(), pxr.data(), vc.data(), lvl.data(), desc.data(), skip.data(),
                                              th, mfNNratio, match.data(), &nm), "eao_search_by_projection_points");
        for (int i = 0; i < M; i++) if (match[i] >= 0) F.mvpMapPoints[match[i]] = vpMapPoints[i];
        return nm;
    }

    // reference :1328-1472 (TrackWithMotionModel)
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
        const int NL = LastFrame.N;
        static thread_local std::vector<uint8_t> valid, desc;
        static thread_local std::vector<float> Xw, ang;
        static thread_local std::vector<int32_t> oct, cm;
        static thread_local FrameArrays fa;
        valid.resize(NL); desc.resize((size_t)NL * 32); Xw.resize((size_t)NL * 3); ang.resize(NL); oct.resize(NL); cm.assign(CurrentFrame.N, -1);
        for (int i = 0; i < NL; i++) {
            auto* pMP = LastFrame.mvpMapPoints[i];
            valid[i] = (pMP && !LastFrame.mvbOutlier[i]) ? 1 : 0;
            oct[i] = LastFrame.mvKeys[i].octave; ang[i] = LastFrame.mvKeysUn[i].angle;
            if (!valid[i]) continue;
            const cv::Mat X = pMP->GetWorldPos();
            for (int k = 0; k < 3; k++) Xw[(size_t)i * 3 + k] = X.template at<float>(k);
            const cv::Mat d = pMP->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        float Tc[16], Tl[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { Tc[r * 4 + c] = CurrentFrame.mTcw.template at<float>(r, c); Tl[r * 4 + c] = LastFrame.mTcw.template at<float>(r, c); }
        const eao_frame_view v = view(CurrentFrame, fa);
        int nm = 0;
        check(eao_search_by_projection_frames(&v, Tc, Tl, NL, valid.data(), Xw.data(), desc.data(), oct.data(), ang.data(), CurrentFrame.fx,
                                              CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf, CurrentFrame.mb, th, bMono ? 1 : 0,
                                              mbCheckOrientation ? 1 : 0, cm.data(), &nm), "eao_search_by_projection_frames");
        for (int k = 0; k < CurrentFrame.N; k++) if (cm[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[cm[k]];
        return nm;
    }


    // =====================================================================================================================
    // The remaining guided searches.  KeyFrameT / FrameT / MapPointT are the reference classes (same member names as
    // src/ORBmatcher.cc uses); feature vectors are DBoW2::FeatureVector, i.e. std::map<unsigned, std::vector<unsigned>>.

    // reference :290-403 (LoopClosing::ComputeSim3 / DetectLoop)
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th) {
        std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT*>(NULL));
        PointArrays pa;
        const eao_map_points mp = gather(vpPoints, pa, [&](MapPointT* p) { return !p->isBad() && !spAlreadyFound.count(p); });
        FrameArrays fa;
        eao_frame_view v = kfview(*pKF, fa);
        for (int k = 0; k < v.n; k++) fa.occ[k] = vpMatched[k] ? 1 : 0;
        float S[16];
        mat44(Scw, S);
        std::vector<int32_t> km(v.n, -1);
        int nm = 0;
        check(eao_search_by_projection_sim3(&v, S, pKF->fx, pKF->fy, pKF->cx, pKF->cy, &mp, th, km.data(), &nm), "eao_search_by_projection_sim3");
        for (int k = 0; k < v.n; k++) if (km[k] >= 0) vpMatched[k] = vpPoints[km[k]];
        return nm;
    }

    // reference :1474-1601 (Tracking::Relocalization)
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist) {
        const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
        PointArrays pa;
        const eao_map_points mp = gather(vpMPs, pa, [&](MapPointT* p) { return p && !p->isBad() && !sAlreadyFound.count(p); });
        std::vector<float> ang(vpMPs.size());
        for (size_t i = 0; i < vpMPs.size(); i++) ang[i] = pKF->mvKeysUn[i].angle;
        FrameArrays fa;
        eao_frame_view v = view(CurrentFrame, fa);
        for (int k = 0; k < v.n; k++) fa.occ[k] = CurrentFrame.mvpMapPoints[k] ? 1 : 0;
        v.log_scale_factor = CurrentFrame.mfLogScaleFactor;
        float T[16];
        mat44(CurrentFrame.mTcw, T);
        std::vector<int32_t> cm(v.n, -1);
        int nm = 0;
        check(eao_search_by_projection_kf(&v, T, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, &mp, ang.data(), th, ORBdist,
                                          mbCheckOrientation ? 1 : 0, cm.data(), &nm), "eao_search_by_projection_kf");
        for (int k = 0; k < v.n; k++) if (cm[k] >= 0) CurrentFrame.mvpMapPoints[k] = vpMPs[cm[k]];
        return nm;
    }

    // reference :159-288 (Tracking::TrackReferenceKeyFrame, Relocalization)
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
        const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(NULL));
        const int n1 = (int)vpMapPointsKF.size(), n2 = F.N;
        static thread_local std::vector<uint8_t> valid1, d1, d2;
        static thread_local std::vector<float> a1, a2;
        static thread_local FeatVecArrays f1, f2;
        valid1.resize(n1); a1.resize(n1); a2.resize(n2);
        for (int i = 0; i < n1; i++) {
            valid1[i] = (vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad()) ? 1 : 0;
            a1[i] = pKF->mvKeysUn[i].angle;
        }
        for (int j = 0; j < n2; j++) a2[j] = F.mvKeys[j].angle;
        const uint8_t* p1 = rows32(pKF->mDescriptors, n1, d1);
        const uint8_t* p2 = rows32(F.mDescriptors, n2, d2);
        const eao_feature_vector fv1 = flatten(pKF->mFeatVec, f1), fv2 = flatten(F.mFeatVec, f2);
        std::vector<int32_t> m12(n1, -1);
        int nm = 0;
        check(eao_search_by_bow(0, n1, p1, a1.data(), valid1.data(), &fv1, n2, p2, a2.data(), nullptr, &fv2, mfNNratio,
                                mbCheckOrientation ? 1 : 0, m12.data(), &nm), "eao_search_by_bow");
        for (int i = 0; i < n1; i++) if (m12[i] >= 0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
        return nm;
    }

    // reference :522-655 (LoopClosing::ComputeSim3)
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
        const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
        vpMatches12 = std::vector<MapPointT*>(n1, static_cast<MapPointT*>(NULL));
        std::vector<uint8_t> v1(n1), v2(n2), d1((size_t)n1 * 32), d2((size_t)n2 * 32);
        std::vector<float> a1(n1), a2(n2);
        for (int i = 0; i < n1; i++) {
            v1[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;
            a1[i] = pKF1->mvKeysUn[i].angle;
            std::memcpy(&d1[(size_t)i * 32], pKF1->mDescriptors.ptr(i), 32);
        }
        for (int j = 0; j < n2; j++) {
            v2[j] = (vpMapPoints2[j] && !vpMapPoints2[j]->isBad()) ? 1 : 0;
            a2[j] = pKF2->mvKeysUn[j].angle;
            std::memcpy(&d2[(size_t)j * 32], pKF2->mDescriptors.ptr(j), 32);
        }
        FeatVecArrays f1, f2;
        const eao_feature_vector fv1 = flatten(pKF1->mFeatVec, f1), fv2 = flatten(pKF2->mFeatVec, f2);
        std::vector<int32_t> m12(n1, -1);
        int nm = 0;
        check(eao_search_by_bow(1, n1, d1.data(), a1.data(), v1.data(), &fv1, n2, d2.data(), a2.data(), v2.data(), &fv2, mfNNratio,
                                mbCheckOrientation ? 1 : 0, m12.data(), &nm), "eao_search_by_bow");
        for (int i = 0; i < n1; i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
        return nm;
    }

    // reference :657-823 (LocalMapping::CreateNewMapPoints)
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo) {
        // epipole in the second image (:663-670): C2 = R2w * Cw + t2w with cv::gemm's double accumulation
        const cv::Mat Cw = pKF1->GetCameraCenter(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
        float C2[3];
        for (int r = 0; r < 3; r++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)R2w.template at<float>(r, k) * (double)Cw.template at<float>(k);
            C2[r] = (float)(acc + (double)t2w.template at<float>(r));
        }
        const float invz = 1.0f / C2[2];
        const float ex = pKF2->fx * C2[0] * invz + pKF2->cx, ey = pKF2->fy * C2[1] * invz + pKF2->cy;
        FrameArrays fa1, fa2;
        eao_frame_view v1 = kfview(*pKF1, fa1), v2 = kfview(*pKF2, fa2);
        for (int k = 0; k < v1.n; k++) fa1.occ[k] = pKF1->GetMapPoint(k) ? 1 : 0;
        for (int k = 0; k < v2.n; k++) fa2.occ[k] = pKF2->GetMapPoint(k) ? 1 : 0;
        FeatVecArrays f1, f2;
        const eao_feature_vector fv1 = flatten(pKF1->mFeatVec, f1), fv2 = flatten(pKF2->mFeatVec, f2);
        float Fm[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Fm[r * 3 + c] = F12.template at<float>(r, c);
        std::vector<int32_t> m12(v1.n, -1);
        int nm = 0;
        check(eao_search_for_triangulation(&v1, &fv1, &v2, &fv2, Fm, ex, ey, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &nm),
              "eao_search_for_triangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nm);
        for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return nm;
    }

    // reference :405-520 (Tracking::MonocularInitialization)
    template <class FrameT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
        const int n1 = (int)F1.mvKeysUn.size();
        std::vector<int32_t> oct(n1), m12(n1, -1);
        std::vector<float> ang(n1), pm((size_t)n1 * 2);
        std::vector<uint8_t> d1((size_t)n1 * 32);
        for (int i = 0; i < n1; i++) {
            oct[i] = F1.mvKeysUn[i].octave; ang[i] = F1.mvKeysUn[i].angle;
            pm[2 * i] = vbPrevMatched[i].x; pm[2 * i + 1] = vbPrevMatched[i].y;
            std::memcpy(&d1[(size_t)i * 32], F1.mDescriptors.ptr(i), 32);
        }
        FrameArrays fa;
        const eao_frame_view v2 = view(F2, fa);
        int nm = 0;
        check(eao_search_for_initialization(n1, oct.data(), ang.data(), d1.data(), &v2, pm.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                            m12.data(), &nm), "eao_search_for_initialization");
        vnMatches12.assign(m12.begin(), m12.end());
        for (int i = 0; i < n1; i++) { vbPrevMatched[i].x = pm[2 * i]; vbPrevMatched[i].y = pm[2 * i + 1]; }
        return nm;
    }

    // Round 4 -- ALL neighbour keyframes of LocalMapping::CreateNewMapPoints in one library call (src/LocalMapping.cc:211-290 calls SearchForTriangulation per
    // neighbour inside its loop): vF12[k] is ComputeF12(pKF1, vpNeighKFs[k]) (:266), vvMatchedPairs[k] what the k-th single call would return.  The loop's own
    // baseline test (:240-263) decides which neighbours the caller passes.
    template <class KeyFrameT>
    void SearchForTriangulationBatch(KeyFrameT* pKF1, const std::vector<KeyFrameT*>& vpNeighKFs, const std::vector<cv::Mat>& vF12,
                                     std::vector<std::vector<std::pair<size_t, size_t> > >& vvMatchedPairs, const bool bOnlyStereo) {
        const size_t nb = vpNeighKFs.size();
        vvMatchedPairs.assign(nb, std::vector<std::pair<size_t, size_t> >());
        if (!nb) return;
        FrameArrays fa1;
        eao_frame_view v1 = kfview(*pKF1, fa1);
        for (int k = 0; k < v1.n; k++) fa1.occ[k] = pKF1->GetMapPoint(k) ? 1 : 0;
        FeatVecArrays f1;
        const eao_feature_vector fv1 = flatten(pKF1->mFeatVec, f1);
        std::vector<FrameArrays> fa2(nb);
        std::vector<FeatVecArrays> f2(nb);
        std::vector<eao_frame_view> v2(nb);
        std::vector<eao_feature_vector> fv2(nb);
        std::vector<const eao_frame_view*> pv(nb);
        std::vector<const eao_feature_vector*> pf(nb);
        std::vector<float> Fm(9 * nb), ex(nb), ey(nb);
        const cv::Mat Cw = pKF1->GetCameraCenter();
        for (size_t q = 0; q < nb; q++) {
            KeyFrameT* pKF2 = vpNeighKFs[q];
            const cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();      // the epipole in the second image, as in the single call (:663-670)
            float C2[3];
            for (int r = 0; r < 3; r++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += (double)R2w.template at<float>(r, k) * (double)Cw.template at<float>(k);
                C2[r] = (float)(acc + (double)t2w.template at<float>(r));
            }
            const float invz = 1.0f / C2[2];
            ex[q] = pKF2->fx * C2[0] * invz + pKF2->cx; ey[q] = pKF2->fy * C2[1] * invz + pKF2->cy;
            v2[q] = kfview(*pKF2, fa2[q]);
            for (int k = 0; k < v2[q].n; k++) fa2[q].occ[k] = pKF2->GetMapPoint(k) ? 1 : 0;
            fv2[q] = flatten(pKF2->mFeatVec, f2[q]);
            pv[q] = &v2[q]; pf[q] = &fv2[q];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Fm[9 * q + r * 3 + c] = vF12[q].template at<float>(r, c);
        }
        std::vector<int32_t> m12(nb * (size_t)v1.n, -1), nm(nb, 0);
        check(eao_search_for_triangulation_batch(&v1, &fv1, (int)nb, pv.data(), pf.data(), Fm.data(), ex.data(), ey.data(), bOnlyStereo ? 1 : 0,
                                                 mbCheckOrientation ? 1 : 0, m12.data(), nm.data()), "eao_search_for_triangulation_batch");
        for (size_t q = 0; q < nb; q++) {
            vvMatchedPairs[q].reserve(nm[q]);
            for (int i = 0; i < v1.n; i++) if (m12[q * v1.n + i] >= 0) vvMatchedPairs[q].push_back(std::make_pair((size_t)i, (size_t)m12[q * v1.n + i]));
        }
    }

    // Round 5 -- the same call over KEYFRAME HANDLES (KeyFrameHandles further down: a keyframe is uploaded once and searched many times -- the neighbours of this
    // new keyframe are mostly the neighbours of the next one): the whole search, selection included, runs on the device (eao_kf_search_for_triangulation).
    template <class KeyFrameT>
    void SearchForTriangulationBatch(const eao_keyframe* h1, KeyFrameT* pKF1, const std::vector<const eao_keyframe*>& h2s, const std::vector<KeyFrameT*>& vpNeighKFs,
                                     const std::vector<cv::Mat>& vF12, std::vector<std::vector<std::pair<size_t, size_t> > >& vvMatchedPairs, const bool bOnlyStereo) {
        const size_t nb = vpNeighKFs.size();
        vvMatchedPairs.assign(nb, std::vector<std::pair<size_t, size_t> >());
        if (!nb) return;
        std::vector<float> Fm(9 * nb), ex(nb), ey(nb);
        const cv::Mat Cw = pKF1->GetCameraCenter();
        for (size_t q = 0; q < nb; q++) {
            KeyFrameT* pKF2 = vpNeighKFs[q];
            const cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();      // the epipole in the second image, as in the single call (:663-670)
            float C2[3];
            for (int r = 0; r < 3; r++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += (double)R2w.template at<float>(r, k) * (double)Cw.template at<float>(k);
                C2[r] = (float)(acc + (double)t2w.template at<float>(r));
            }
            const float invz = 1.0f / C2[2];
            ex[q] = pKF2->fx * C2[0] * invz + pKF2->cx; ey[q] = pKF2->fy * C2[1] * invz + pKF2->cy;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Fm[9 * q + r * 3 + c] = vF12[q].template at<float>(r, c);
        }
        const int n1 = eao_keyframe_size(h1);
        std::vector<int32_t> m12(nb * (size_t)n1, -1), nm(nb, 0);
        check(eao_kf_search_for_triangulation(h1, (int)nb, h2s.data(), Fm.data(), ex.data(), ey.data(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), nm.data()),
              "eao_kf_search_for_triangulation");
        for (size_t q = 0; q < nb; q++) {
            vvMatchedPairs[q].reserve(nm[q]);
            for (int i = 0; i < n1; i++) if (m12[q * n1 + i] >= 0) vvMatchedPairs[q].push_back(std::make_pair((size_t)i, (size_t)m12[q * n1 + i]));
        }
    }

    // reference :825-975 (LocalMapping::SearchInNeighbors).  The search runs for all points first; replacing / adding
    // observations then happens in index order exactly as upstream interleaves it, re-checking the entry conditions a
    // previous replacement may have changed.
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0) {
        PointArrays pa;
        const eao_map_points mp = gather(vpMapPoints, pa, [&](MapPointT* p) { return p && !p->isBad() && !p->IsInKeyFrame(pKF); });
        FrameArrays fa;
        const eao_frame_view v = kfview(*pKF, fa);
        const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
        float pose[15];
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pose[r * 3 + c] = Rcw.template at<float>(r, c); pose[9 + r] = tcw.template at<float>(r); pose[12 + r] = Ow.template at<float>(r); }
        std::vector<int32_t> best(vpMapPoints.size(), -1);
        int n = 0;
        check(eao_fuse_search(&v, 0, pose, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, &mp, th, best.data(), &n), "eao_fuse_search");
        int nFused = 0;
        for (size_t i = 0; i < vpMapPoints.size(); i++) {
            if (best[i] < 0) continue;
            MapPointT* pMP = vpMapPoints[i];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, best[i]);
                pKF->AddMapPoint(pMP, best[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // Round 4 -- the points of LocalMapping::SearchInNeighbors against ALL its target keyframes in one library call (src/LocalMapping.cc:484-494 / :509-519 call
    // Fuse once per target): the searches of every target run up front on the map as it is, then the targets are applied IN ORDER exactly as the single calls apply
    // them -- a point that an earlier target's fusion made bad, or put into the keyframe at hand, is skipped as upstream's loop head would skip it (:851-861).
    // Round 5: an earlier target's fusion can also change what a LATER target's search reads -- pMPinKF->Replace(pMP) ends in
    // pMP->ComputeDistinctiveDescriptors() (src/MapPoint.cc:177-215), so the surviving candidate meets the next targets with a NEW descriptor.  The points a target
    // leaves in that state are therefore searched again (they alone, against the remaining targets, one more library call per target that produced any -- with
    // keyframe handles a few tens of microseconds) before the next target is applied: the result equals the single calls in a row, descriptor changes included
    // (tests/cpp/search_adapter_test.cpp 9b, whose stand-in Replace now changes the survivor's descriptor).  Returns the sum of the targets' nFused; the
    // per-target counts in *nFusedPerKF when given.  One camera for all targets (the batch entry point takes fx .. mbf once), as in the reference.
    // `handles` (may be NULL): the targets' keyframe handles -- the searches then run on frames resident in HBM (eao_kf_fuse_search) instead of uploading them.
    template <class KeyFrameT, class MapPointT>
    int FuseBatch(const std::vector<KeyFrameT*>& vpTargetKFs, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0, std::vector<int>* nFusedPerKF = nullptr,
                  const std::vector<const eao_keyframe*>* handles = nullptr) {
        const size_t nk = vpTargetKFs.size(), np = vpMapPoints.size();
        if (nFusedPerKF) nFusedPerKF->assign(nk, 0);
        if (!nk || !np) return 0;
        std::vector<FrameArrays> fa(handles ? 0 : nk);
        std::vector<eao_frame_view> v(handles ? 0 : nk);
        std::vector<const eao_frame_view*> pv(handles ? 0 : nk);
        std::vector<float> poses(15 * nk);
        for (size_t q = 0; q < nk; q++) {
            KeyFrameT* pKF = vpTargetKFs[q];
            if (!handles) { v[q] = kfview(*pKF, fa[q]); pv[q] = &v[q]; }
            const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
            float* pose = &poses[15 * q];
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pose[r * 3 + c] = Rcw.template at<float>(r, c); pose[9 + r] = tcw.template at<float>(r); pose[12 + r] = Ow.template at<float>(r); }
        }
        KeyFrameT* k0 = vpTargetKFs[0];
        // search `pts` against targets q0 .. nk-1; row q - q0 of `out` = target q
        auto search = [&](const std::vector<MapPointT*>& pts, size_t q0, std::vector<int32_t>& out) {
            PointArrays pa;
            const eao_map_points mp = gather(pts, pa, [&](MapPointT* p) { return p && !p->isBad(); });
            const size_t nt = nk - q0;
            out.assign(nt * pts.size(), -1);
            std::vector<int32_t> nf(nt, 0);
            if (handles)
                check(eao_kf_fuse_search((int)nt, handles->data() + q0, 0, &poses[15 * q0], k0->fx, k0->fy, k0->cx, k0->cy, k0->mbf, &mp, th, out.data(), nf.data()), "eao_kf_fuse_search");
            else
                check(eao_fuse_search_batch((int)nt, pv.data() + q0, 0, &poses[15 * q0], k0->fx, k0->fy, k0->cx, k0->cy, k0->mbf, &mp, th, out.data(), nf.data()), "eao_fuse_search_batch");
        };
        std::vector<int32_t> best;
        search(vpMapPoints, 0, best);
        int total = 0;
        std::vector<size_t> changed;            // candidates whose descriptor this target's fusions changed
        std::vector<MapPointT*> sub;
        std::vector<int32_t> again;
        for (size_t q = 0; q < nk; q++) {
            KeyFrameT* pKF = vpTargetKFs[q];
            int nFused = 0;
            changed.clear();
            for (size_t i = 0; i < np; i++) {
                const int k = best[q * np + i];
                if (k < 0) continue;
                MapPointT* pMP = vpMapPoints[i];
                if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
                MapPointT* pMPinKF = pKF->GetMapPoint(k);
                if (pMPinKF) {
                    if (!pMPinKF->isBad()) {
                        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                        else { pMPinKF->Replace(pMP); changed.push_back(i); }      // pMP survives with a recomputed descriptor
                    }
                } else {
                    pMP->AddObservation(pKF, k);
                    pKF->AddMapPoint(pMP, k);
                }
                nFused++;
            }
            if (nFusedPerKF) (*nFusedPerKF)[q] = nFused;
            total += nFused;
            if (!changed.empty() && q + 1 < nk) {
                sub.resize(changed.size());
                for (size_t c = 0; c < changed.size(); c++) sub[c] = vpMapPoints[changed[c]];
                search(sub, q + 1, again);
                for (size_t t = q + 1; t < nk; t++)
                    for (size_t c = 0; c < changed.size(); c++) best[t * np + changed[c]] = again[(t - q - 1) * changed.size() + c];
            }
        }
        return total;
    }

    // reference :977-1100 (LoopClosing::SearchAndFuse)
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, cv::Mat Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint) {
        const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
        PointArrays pa;
        const eao_map_points mp = gather(vpPoints, pa, [&](MapPointT* p) { return !p->isBad() && !spAlreadyFound.count(p); });
        FrameArrays fa;
        const eao_frame_view v = kfview(*pKF, fa);
        float S[16];
        mat44(Scw, S);
        std::vector<int32_t> best(vpPoints.size(), -1);
        int n = 0;
        check(eao_fuse_search(&v, 1, S, pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0.f, &mp, th, best.data(), &n), "eao_fuse_search");
        int nFused = 0;
        for (size_t i = 0; i < vpPoints.size(); i++) {
            if (best[i] < 0) continue;
            MapPointT* pMP = vpPoints[i];
            MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, best[i]);
                pKF->AddMapPoint(pMP, best[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // reference :1102-1326 (LoopClosing::ComputeSim3)
    template <class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th) {
        const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> already1(N1, false), already2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPointT* pMP = vpMatches12[i];
            if (pMP) {
                already1[i] = true;
                const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) already2[idx2] = true;
            }
        }
        PointArrays pa1, pa2;
        int cursor = 0;
        const eao_map_points mp1 = gather(vpMapPoints1, pa1, [&](MapPointT* p) { const int i = cursor++; return p && !already1[i] && !p->isBad(); });
        cursor = 0;
        const eao_map_points mp2 = gather(vpMapPoints2, pa2, [&](MapPointT* p) { const int i = cursor++; return p && !already2[i] && !p->isBad(); });
        FrameArrays fa1, fa2;
        const eao_frame_view v1 = kfview(*pKF1, fa1), v2 = kfview(*pKF2, fa2);
        float T1[16], T2[16], R[9], t[3];
        pose44(pKF1->GetRotation(), pKF1->GetTranslation(), T1);
        pose44(pKF2->GetRotation(), pKF2->GetTranslation(), T2);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = R12.template at<float>(r, c); t[r] = t12.template at<float>(r); }
        std::vector<int32_t> m12(N1, -1);
        int nf = 0;
        check(eao_search_by_sim3(&v1, T1, &mp1, &v2, T2, &mp2, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, s12, R, t, th, m12.data(), &nf),
              "eao_search_by_sim3");
        for (int i = 0; i < N1; i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
        return nf;
    }

    // reference src/MapPoint.cc:242-307 (MapPoint::ComputeDistinctiveDescriptors), for one map point or a batch: returns,
    // per point, the row of `descriptors[s]` (the observing keyframes' descriptors in std::map order, bad keyframes
    // skipped) with the least median distance to the rest, or -1 for an empty set.  The body of
    // MapPoint::ComputeDistinctiveDescriptors becomes: gather vDescriptors as upstream, then
    //   mDescriptor = vDescriptors[ORBmatcher::DistinctiveDescriptor(vDescriptors)].clone();
    static std::vector<int> DistinctiveDescriptors(const std::vector<std::vector<cv::Mat> >& descriptors) {
        std::vector<int32_t> start(descriptors.size() + 1, 0), best(descriptors.size(), -1);
        std::vector<uint8_t> flat;
        for (size_t s = 0; s < descriptors.size(); s++) {
            for (size_t k = 0; k < descriptors[s].size(); k++) {
                const unsigned char* d = descriptors[s][k].ptr(0);
                flat.insert(flat.end(), d, d + 32);
            }
            start[s + 1] = (int32_t)(flat.size() / 32);
        }
        if (!descriptors.empty())
            check(eao_distinctive_descriptors((int32_t)descriptors.size(), start.data(), flat.data(), best.data()), "eao_distinctive_descriptors");
        return std::vector<int>(best.begin(), best.end());
    }
    static int DistinctiveDescriptor(const std::vector<cv::Mat>& vDescriptors) {
        return DistinctiveDescriptors(std::vector<std::vector<cv::Mat> >(1, vDescriptors))[0];
    }

    // reference src/ORBmatcher.cc:37-39
    static constexpr int TH_LOW = 50;
    static constexpr int TH_HIGH = 100;
    static constexpr int HISTO_LENGTH = 30;

    // reference :1603-1644 (rotation-consistency histogram): indices of the three fullest bins, -1 when a bin is
    // below 10% of the fullest
    static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
        // running top-3 by bin population; a later bin only displaces an earlier one when strictly fuller
        int top[3] = {0, 0, 0};
        int* idx[3] = {&ind1, &ind2, &ind3};
        for (int bin = 0; bin < L; bin++) {
            const int pop = (int)histo[bin].size();
            for (int rank = 0; rank < 3; rank++) {
                if (pop > top[rank]) {
                    for (int q = 2; q > rank; q--) { top[q] = top[q - 1]; *idx[q] = *idx[q - 1]; }
                    top[rank] = pop;
                    *idx[rank] = bin;
                    break;
                }
            }
        }
        const float floor10 = 0.1f * (float)top[0];
        if (top[1] < floor10) { ind2 = -1; ind3 = -1; }
        else if (top[2] < floor10) { ind3 = -1; }
    }

    friend class KeyFrameHandles;
protected:
    struct FrameArrays { std::vector<float> x, y, ang, ur, sf, s2, is2; std::vector<int32_t> oct; std::vector<uint8_t> occ, desc; };
    struct PointArrays { std::vector<uint8_t> active, desc; std::vector<float> Xw, normal, dmin, dmax, draw; };
    struct FeatVecArrays { std::vector<uint32_t> id, index; std::vector<int32_t> start; };
    // mfMaxDistance is what PredictScale divides (src/MapPoint.cc:385-394); upstream exposes it only through
    // GetMaxDistanceInvariance() = 1.2f * mfMaxDistance, which does not round-trip in float.  A derived class may name
    // the protected member of its base, and the resulting pointer-to-member applies to any MapPoint: an exact read
    // without touching MapPoint.h.  (A one-line public accessor in MapPoint.h is the alternative.)
    template <class MapPointT>
    struct MaxDistance : MapPointT {
        static float of(MapPointT* p) { return p->*(&MaxDistance::mfMaxDistance); }
    };
    // flatten the map points the searches read; active(p) evaluates upstream's entry conditions IN INDEX ORDER
    template <class MapPointT, class Pred>
    static eao_map_points gather(const std::vector<MapPointT*>& pts, PointArrays& a, Pred&& active) {
        const size_t n = pts.size();
        a.active.assign(n, 0); a.desc.assign(n * 32, 0); a.Xw.assign(n * 3, 0.f); a.normal.assign(n * 3, 0.f);
        a.dmin.assign(n, 0.f); a.dmax.assign(n, 0.f); a.draw.assign(n, 0.f);
        for (size_t i = 0; i < n; i++) {
            MapPointT* p = pts[i];
            if (!active(p)) continue;
            a.active[i] = 1;
            const cv::Mat X = p->GetWorldPos(), Nn = p->GetNormal(), d = p->GetDescriptor();
            for (int k = 0; k < 3; k++) { a.Xw[i * 3 + k] = X.template at<float>(k); a.normal[i * 3 + k] = Nn.template at<float>(k); }
            a.dmin[i] = p->GetMinDistanceInvariance(); a.dmax[i] = p->GetMaxDistanceInvariance(); a.draw[i] = MaxDistance<MapPointT>::of(p);
            std::memcpy(&a.desc[i * 32], d.ptr(0), 32);
        }
        eao_map_points m;
        m.n = (int32_t)n; m.active = a.active.data(); m.Xw = a.Xw.data(); m.normal = a.normal.data(); m.min_dist_inv = a.dmin.data();
        m.max_dist_inv = a.dmax.data(); m.max_dist = a.draw.data(); m.desc = a.desc.data();
        return m;
    }
    template <class FeatVecT>
    static eao_feature_vector flatten(const FeatVecT& fv, FeatVecArrays& a) {
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
    // n rows of 32 descriptor bytes as ONE block: the matrix itself when its rows are contiguous (Frame / KeyFrame::mDescriptors come out of one
    // create(n, 32, CV_8U)), a packed copy in `pack` otherwise
    static const uint8_t* rows32(const cv::Mat& D, int n, std::vector<uint8_t>& pack) {
        if (n == 0) return nullptr;
        if (D.isContinuous()) return D.ptr(0);
        pack.resize((size_t)n * 32);
        for (int i = 0; i < n; i++) std::memcpy(&pack[(size_t)i * 32], D.ptr(i), 32);
        return pack.data();
    }
    static void mat44(const cv::Mat& M, float* out) {
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[r * 4 + c] = M.at<float>(r, c);
    }
    static void pose44(const cv::Mat& R, const cv::Mat& t, float* out) {
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) out[r * 4 + c] = R.at<float>(r, c); out[r * 4 + 3] = t.at<float>(r); }
        out[12] = out[13] = out[14] = 0.f; out[15] = 1.f;
    }
    // KeyFrame flavour of view(): per-keyframe grid bounds (const members, include/KeyFrame.h:193-194,255-262), no
    // occupancy (each search defines its own), level sigmas and the log scale factor
    template <class KeyFrameT>
    static eao_frame_view kfview(KeyFrameT& K, FrameArrays& a) {
        const int N = K.N;
        a.x.resize(N); a.y.resize(N); a.ang.resize(N); a.ur.resize(N); a.oct.resize(N); a.occ.assign(N, 0);
        for (int i = 0; i < N; i++) {
            a.x[i] = K.mvKeysUn[i].pt.x; a.y[i] = K.mvKeysUn[i].pt.y; a.ang[i] = K.mvKeysUn[i].angle; a.oct[i] = K.mvKeysUn[i].octave;
            a.ur[i] = K.mvuRight[i];
        }
        const uint8_t* descPtr = rows32(K.mDescriptors, N, a.desc);
        a.sf.assign(K.mvScaleFactors.begin(), K.mvScaleFactors.end());
        a.s2.assign(K.mvLevelSigma2.begin(), K.mvLevelSigma2.end());
        a.is2.assign(K.mvInvLevelSigma2.begin(), K.mvInvLevelSigma2.end());
        eao_frame_view v;
        v.n = N; v.kp_x = a.x.data(); v.kp_y = a.y.data(); v.kp_octave = a.oct.data(); v.kp_angle = a.ang.data(); v.u_right = a.ur.data();
        v.descriptors = descPtr; v.occupied = a.occ.data();
        v.min_x = K.mnMinX; v.min_y = K.mnMinY; v.max_x = K.mnMaxX; v.max_y = K.mnMaxY;
        v.grid_inv_w = K.mfGridElementWidthInv; v.grid_inv_h = K.mfGridElementHeightInv;
        v.grid_cols = K.mnGridCols; v.grid_rows = K.mnGridRows;
        v.scale_factors = a.sf.data(); v.nlevels = (int)a.sf.size();
        v.log_scale_factor = K.mfLogScaleFactor; v.level_sigma2 = a.s2.data(); v.inv_level_sigma2 = a.is2.data();
        return v;
    }
    // flatten the members of Frame the searches read (src/ORBmatcher.cc, src/Frame.cc:696-761) into an eao_frame_view
    template <class FrameT>
    static eao_frame_view view(FrameT& F, FrameArrays& a) {
        const int N = F.N;
        a.x.resize(N); a.y.resize(N); a.ang.resize(N); a.ur.resize(N); a.oct.resize(N); a.occ.resize(N);
        for (int i = 0; i < N; i++) {
            a.x[i] = F.mvKeysUn[i].pt.x; a.y[i] = F.mvKeysUn[i].pt.y; a.ang[i] = F.mvKeysUn[i].angle; a.oct[i] = F.mvKeysUn[i].octave;
            a.ur[i] = F.mvuRight[i];
            a.occ[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
        }
        const uint8_t* descPtr = rows32(F.mDescriptors, N, a.desc);
        a.sf.assign(F.mvScaleFactors.begin(), F.mvScaleFactors.end());
        eao_frame_view v;
        v.n = N; v.kp_x = a.x.data(); v.kp_y = a.y.data(); v.kp_octave = a.oct.data(); v.kp_angle = a.ang.data(); v.u_right = a.ur.data();
        v.descriptors = descPtr; v.occupied = a.occ.data();
        v.min_x = FrameT::mnMinX; v.min_y = FrameT::mnMinY; v.max_x = FrameT::mnMaxX; v.max_y = FrameT::mnMaxY;
        v.grid_inv_w = FrameT::mfGridElementWidthInv; v.grid_inv_h = FrameT::mfGridElementHeightInv;
        v.grid_cols = 64; v.grid_rows = 48;   // FRAME_GRID_COLS / FRAME_GRID_ROWS (include/Frame.h:89-90)
        v.scale_factors = a.sf.data(); v.nlevels = (int)a.sf.size();
        v.log_scale_factor = 0.f; v.level_sigma2 = nullptr; v.inv_level_sigma2 = nullptr;
        return v;
    }
    static void check(eao_status st, const char* what) {
        if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
    }
    float mfNNratio;
    bool mbCheckOrientation;
};

// Keyframe handles for the LocalMapping / LoopClosing side (round 5): of(pKF) uploads a keyframe the first time it is asked for -- keypoints, descriptors, grid
// order, feature vector, scale tables -- and hands out the resident handle afterwards; refresh(pKF) re-sends its occupancy (GetMapPoint(k) != NULL: what
// SearchForTriangulation skips) after map points were added or erased; erase(pKF) when the keyframe is culled (KeyFrame::SetBadFlag).  One instance per thread
// that searches (LocalMapping owns one); the handles themselves may be searched from several threads.
class KeyFrameHandles {
public:
    KeyFrameHandles() {}
    ~KeyFrameHandles() { for (auto& e : map_) eao_keyframe_destroy(e.second); }
    KeyFrameHandles(const KeyFrameHandles&) = delete;
    KeyFrameHandles& operator=(const KeyFrameHandles&) = delete;
    template <class KeyFrameT>
    const eao_keyframe* of(KeyFrameT* pKF) {
        auto it = map_.find((const void*)pKF);
        if (it != map_.end()) return it->second;
        ORBmatcher::FrameArrays fa;
        ORBmatcher::FeatVecArrays fv;
        eao_frame_view v = ORBmatcher::kfview(*pKF, fa);
        for (int k = 0; k < v.n; k++) fa.occ[k] = pKF->GetMapPoint(k) ? 1 : 0;
        const eao_feature_vector f = ORBmatcher::flatten(pKF->mFeatVec, fv);
        eao_keyframe* h = nullptr;
        ORBmatcher::check(eao_keyframe_create(&v, &f, &h), "eao_keyframe_create");
        map_[(const void*)pKF] = h;
        return h;
    }
    template <class KeyFrameT>
    void refresh(KeyFrameT* pKF) {
        auto it = map_.find((const void*)pKF);
        if (it == map_.end()) { (void)of(pKF); return; }
        const int n = eao_keyframe_size(it->second);
        std::vector<uint8_t> occ(n > 0 ? n : 1, 0);
        for (int k = 0; k < n; k++) occ[k] = pKF->GetMapPoint(k) ? 1 : 0;
        ORBmatcher::check(eao_keyframe_update_points(it->second, occ.data()), "eao_keyframe_update_points");
    }
    template <class KeyFrameT>
    void erase(KeyFrameT* pKF) {
        auto it = map_.find((const void*)pKF);
        if (it == map_.end()) return;
        eao_keyframe_destroy(it->second);
        map_.erase(it);
    }
    size_t size() const { return map_.size(); }
private:
    std::map<const void*, eao_keyframe*> map_;
};

}  // namespace eaofusion
#endif  // EAOFUSION_ORBMATCHER_H
