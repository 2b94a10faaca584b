// match_internal.h -- shared between match.hip and search.hip (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

namespace eao {
namespace match {

struct Query {
    float x, y, r;            // window centre and half-size
    int minLevel, maxLevel;   // GetFeaturesInArea level gate (both -1: no level test, KeyFrame::GetFeaturesInArea)
    float urRef, urTol;       // stereo gate: |urRef - uRight| > urTol rejects (only for keypoints with uRight > 0)
    int active;
};

// per query: items[start .. start + count) = (distance << 16 | keypoint index), in upstream's candidate order
struct Lists {
    std::vector<int> start, count;
    std::vector<unsigned> items;
};

// A frame whose arrays already live on the device in the layout k_match_candidates walks (an eao_keyframe handle, csrc/keyframe.hip: uploaded and put into
// grid order ONCE): device addresses; n keypoints, no of them inside the grid.
struct Resident {
    const float* kx; const float* ky; const float* ur; const int* oct; const uint8_t* desc;
    const int* order; const unsigned short* cellx; const unsigned short* celly; const int* colStart;
    int n, no;
};

// uploads the frame + queries, runs the candidate kernel, downloads the compact lists (res: the frame is resident -- only the queries travel)
eao_status build_lists(const eao_frame_view* F, const std::vector<Query>& q, const uint8_t* qdesc, Lists& L, const Resident* res = nullptr);
// the same for nf frames in one upload / one synchronisation (frames may share their query descriptors: same pointer, same count);
// res (may be NULL): per frame NULL or its resident form
eao_status build_lists_multi(int nf, const eao_frame_view* const* Fs, const std::vector<Query>* qs, const uint8_t* const* qdescs, Lists* Ls,
                             const Resident* const* res = nullptr);

// Hamming distance of explicit pairs (ia[k] of set A, ib[k] of set B): the bag-of-words bucket searches
eao_status pair_distances(const uint8_t* descA, int nA, const uint8_t* descB, int nB, const std::vector<int>& ia,
                          const std::vector<int>& ib, std::vector<unsigned short>& dist);

}  // namespace match
}  // namespace eao
