// ORBmatcher.h -- the distance side of the reference matcher (reference include/ORBmatcher.h:37-102,
// src/ORBmatcher.cc:37-39, 1603-1665) on top of the C-ABI.
//
// What is here: constructor, TH_LOW / TH_HIGH / HISTO_LENGTH, static DescriptorDistance, ComputeThreeMaxima, the two
// per-frame guided searches SearchByProjection(Frame&, vector<MapPoint*>&, th) and SearchByProjection(Frame& Cur,
// const Frame& Last, th, bMono) (candidate lists + distances on the GPU, the greedy assignment replayed inside the
// library), and the batched primitive the remaining Search* routines are built from: best-two Hamming search of a set
// of query descriptors against a set of train descriptors under a candidate mask, with the reference's tie order.
// The other eight Search*/Fuse routines (BoW, triangulation, Sim3, fuse) are the next row of the scope table.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../eao_fusion.h"
#include "cv_compat.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (1 x 32 CV_8U each), reference :1649-1665.
    // One pair per call costs a kernel launch; hot loops should use BestTwo / DistanceMatrix below.
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
    //      member names as src/ORBmatcher.cc uses), so `ORBmatcher::SearchByProjection(F, vpMapPoints, th)` in Tracking.cc
    //      resolves to these when src/ORBmatcher.cc's two bodies are removed.

    // reference :45-129 (TrackLocalMap -> SearchLocalPoints)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3) {
        const int M = (int)vpMapPoints.size();
        std::vector<float> px(M), py(M), pxr(M), vc(M);
        std::vector<int32_t> lvl(M), match(M, -1);
        std::vector<uint8_t> skip(M), desc((size_t)M * 32);
        for (int i = 0; i < M; i++) {
            MapPointT* pMP = vpMapPoints[i];
            skip[i] = (!pMP->mbTrackInView || pMP->isBad()) ? 1 : 0;
            px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR; vc[i] = pMP->mTrackViewCos;
            lvl[i] = pMP->mnTrackScaleLevel;
            if (!skip[i]) { const cv::Mat d = pMP->GetDescriptor(); std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32); }
        }
        FrameArrays fa;
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
        std::vector<uint8_t> valid(NL), desc((size_t)NL * 32);
        std::vector<float> Xw((size_t)NL * 3), ang(NL);
        std::vector<int32_t> oct(NL), cm(CurrentFrame.N, -1);
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
        FrameArrays fa;
        const eao_frame_view v = view(CurrentFrame, fa);
        int nm = 0;
        check(eao_search_by_projection_frames(&v, Tc, Tl, NL, valid.data(), Xw.data(), desc.data(), oct.data(), ang.data(), CurrentFrame.fx,
                                              CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf, CurrentFrame.mb, th, bMono ? 1 : 0,
                                              mbCheckOrientation ? 1 : 0, cm.data(), &nm), "eao_search_by_projection_frames");
        for (int k = 0; k < CurrentFrame.N; k++) if (cm[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[cm[k]];
        return nm;
    }

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

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

protected:
    struct FrameArrays { std::vector<float> x, y, ang, ur, sf; std::vector<int32_t> oct; std::vector<uint8_t> occ, desc; };
    // flatten the members of Frame the searches read (src/ORBmatcher.cc, src/Frame.cc:696-761) into an eao_frame_view
    template <class FrameT>
    static eao_frame_view view(FrameT& F, FrameArrays& a) {
        const int N = F.N;
        a.x.resize(N); a.y.resize(N); a.ang.resize(N); a.ur.resize(N); a.oct.resize(N); a.occ.resize(N); a.desc.resize((size_t)N * 32);
        for (int i = 0; i < N; i++) {
            a.x[i] = F.mvKeysUn[i].pt.x; a.y[i] = F.mvKeysUn[i].pt.y; a.ang[i] = F.mvKeysUn[i].angle; a.oct[i] = F.mvKeysUn[i].octave;
            a.ur[i] = F.mvuRight[i];
            a.occ[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
            std::memcpy(&a.desc[(size_t)i * 32], F.mDescriptors.ptr(i), 32);
        }
        a.sf.assign(F.mvScaleFactors.begin(), F.mvScaleFactors.end());
        eao_frame_view v;
        v.n = N; v.kp_x = a.x.data(); v.kp_y = a.y.data(); v.kp_octave = a.oct.data(); v.kp_angle = a.ang.data(); v.u_right = a.ur.data();
        v.descriptors = a.desc.data(); v.occupied = a.occ.data();
        v.min_x = FrameT::mnMinX; v.min_y = FrameT::mnMinY; v.max_x = FrameT::mnMaxX; v.max_y = FrameT::mnMaxY;
        v.grid_inv_w = FrameT::mfGridElementWidthInv; v.grid_inv_h = FrameT::mfGridElementHeightInv;
        v.grid_cols = 64; v.grid_rows = 48;   // FRAME_GRID_COLS / FRAME_GRID_ROWS (include/Frame.h:89-90)
        v.scale_factors = a.sf.data(); v.nlevels = (int)a.sf.size();
        return v;
    }
    static void check(eao_status st, const char* what) {
        if (st != EAO_OK) throw std::runtime_error(std::string(what) + ": " + eao_last_error());
    }
    float mfNNratio;
    bool mbCheckOrientation;
};

// reference src/ORBmatcher.cc:37-39
inline const int ORBmatcher::TH_HIGH = 100;
inline const int ORBmatcher::TH_LOW = 50;
inline const int ORBmatcher::HISTO_LENGTH = 30;

}  // namespace ORB_SLAM2
#endif  // ORBMATCHER_H
