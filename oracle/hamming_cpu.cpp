// oracle/hamming_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
//
// ORBmatcher::DescriptorDistance            reference src/ORBmatcher.cc:1649-1665 (8 x 32-bit SWAR popcount)
// brute-force all-pairs / best-two search    BASELINE.json configs[2]; tie order = first candidate wins
//                                            (strict '<' everywhere in src/ORBmatcher.cc, e.g. :102-114)
// Pinned by: exhaustive agreement with __builtin_popcount in tests/test_oracle_hamming.py, plus the
// all-zero / all-one known answers (0 and 256).  PARITY otherwise UNPINNED (no upstream fixtures exist).
#include <cstdint>
#include <vector>
#include <climits>
#include <algorithm>
#include <cstring>

namespace {
inline int swar_distance(const uint32_t* pa, const uint32_t* pb) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}
}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    uint32_t wa[8], wb[8];
    std::memcpy(wa, a, 32);
    std::memcpy(wb, b, 32);
    return swar_distance(wa, wb);
}

// D[i*nb + j] = distance(A_i, B_j)
void orc_hamming_matrix(const uint8_t* A, int na, const uint8_t* B, int nb, uint16_t* D) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) D[(size_t)i * nb + j] = (uint16_t)orc_descriptor_distance(A + 32 * i, B + 32 * j);
}

// per row i of A: best (smallest) and second-best distance over the B_j with mask[i*nb+j] != 0 (mask may be
// null = all allowed).  The scan replays the reference's candidate loop (src/ORBmatcher.cc:83-115 shape):
//   if d < best { second = best; best = d; idx = j } else if d < second { second = d }
// out[i] = {best, second, idx, pad}; no candidate => {256, 256, -1}.
struct Best2 { int32_t best, second, idx, idx2; };
void orc_hamming_best2(const uint8_t* A, int na, const uint8_t* B, int nb, const uint8_t* mask, Best2* out) {
    for (int i = 0; i < na; i++) {
        int best = 256, second = 256, idx = -1, idx2 = -1;
        for (int j = 0; j < nb; j++) {
            if (mask && !mask[(size_t)i * nb + j]) continue;
            int d = orc_descriptor_distance(A + 32 * i, B + 32 * j);
            if (d < best) { second = best; idx2 = idx; best = d; idx = j; }
            else if (d < second) { second = d; idx2 = j; }
        }
        out[i] = {best, second, idx, idx2};
    }
}


// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307) for a batch of map points: per set the N x N distance
// table, each row sorted, median = sorted[(int)(0.5 * (N - 1))], least median wins (strict <, so the first index).
void orc_distinctive_descriptors(int n_sets, const int32_t* set_start, const uint8_t* desc, int32_t* best) {
    for (int s = 0; s < n_sets; s++) {
        const int beg = set_start[s], N = set_start[s + 1] - beg;
        best[s] = -1;
        if (N <= 0) continue;
        std::vector<int> D((size_t)N * N);
        for (int i = 0; i < N; i++) {
            D[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; j++) {
                uint32_t a[8], b[8];
                std::memcpy(a, desc + 32 * (size_t)(beg + i), 32); std::memcpy(b, desc + 32 * (size_t)(beg + j), 32);
                int d = 0;
                for (int w = 0; w < 8; w++) d += __builtin_popcount(a[w] ^ b[w]);
                D[(size_t)i * N + j] = d; D[(size_t)j * N + i] = d;
            }
        }
        int bestMedian = INT_MAX, bestIdx = 0;
        for (int i = 0; i < N; i++) {
            std::vector<int> v(D.begin() + (size_t)i * N, D.begin() + (size_t)(i + 1) * N);
            std::sort(v.begin(), v.end());
            const int median = v[(size_t)(0.5 * (N - 1))];
            if (median < bestMedian) { bestMedian = median; bestIdx = i; }
        }
        best[s] = bestIdx;
    }
}
}  // extern "C"
