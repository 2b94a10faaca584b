// ORBmatcher.h -- the distance side of the reference matcher (reference include/ORBmatcher.h:37-102,
// src/ORBmatcher.cc:37-39, 1603-1665) on top of the C-ABI.
//
// What is here: constructor, TH_LOW / TH_HIGH / HISTO_LENGTH, static DescriptorDistance, ComputeThreeMaxima, and
// the batched primitive the Search* routines are built from: best-two Hamming search of a set of query descriptors
// against a set of train descriptors under an arbitrary candidate mask, with the reference's tie order (first
// candidate wins, strict '<').  The ten Search*/Fuse routines walk Frame / KeyFrame / MapPoint objects and are
// greedy over queries; their candidate gating and assignment replay stay on the host and feed this
// primitive with the mask -- they are the next row of the scope table (DESIGN.md "what comes next").
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

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
