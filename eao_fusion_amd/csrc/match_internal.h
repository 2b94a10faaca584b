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

// uploads the frame + queries, runs the candidate kernel, downloads the compact lists
eao_status build_lists(const eao_frame_view* F, const std::vector<Query>& q, const uint8_t* qdesc, Lists& L);
// the same for nf frames in one upload / one synchronisation (frames may share their query descriptors: same pointer, same count)
eao_status build_lists_multi(int nf, const eao_frame_view* const* Fs, const std::vector<Query>* qs, const uint8_t* const* qdescs, Lists* Ls);

// Hamming distance of explicit pairs (ia[k] of set A, ib[k] of set B): the bag-of-words bucket searches
eao_status pair_distances(const uint8_t* descA, int nA, const uint8_t* descB, int nB, const std::vector<int>& ia,
                          const std::vector<int>& ib, std::vector<unsigned short>& dist);

}  // namespace match
}  // namespace eao
