// tests/cpp/host_replay_provider.cpp -- TEST INFRASTRUCTURE (CPU suite, sanitizer pass): a list provider for the product's HOST replay code.
//
// eao_fusion_amd/csrc/search.hip is the host half of all eleven guided searches: per-query geometry, then order-dependent selection loops over candidate
// lists that the product gets from its GPU kernels (csrc/match.hip: eao::match::build_lists / build_lists_multi / pair_distances).  Here the same three
// functions are answered by the ORACLE's candidate lister (oracle/match_cpu.cpp: orc_candidate_lists, orc_pair_distance), so that search.hip -- compiled
// unchanged as plain C++, with AddressSanitizer + UndefinedBehaviorSanitizer when tools/run_sanitizers.sh asks -- can be driven without a GPU by
// tests/test_host_replay_cpu.py: caller-supplied indices, octaves, NaN poses and ragged feature vectors then meet the sanitizers.  Nothing of this file is
// linked into libeaofusion_hip.so; the shared object built from it is called libeaosearch_hosttest.so and lives in a scratch directory.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../eao_fusion_amd/csrc/common.h"
#include "../../eao_fusion_amd/csrc/match_internal.h"

extern "C" int orc_candidate_lists(const eao_frame_view* F, int nq, const void* queries, const uint8_t* qdesc, int32_t* start, int32_t* count, uint32_t* items, int items_cap);
extern "C" int orc_pair_distance(const uint8_t* a, const uint8_t* b);

namespace {
thread_local char g_err[512] = "";
}
namespace eao {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
eao_status require_device() { return EAO_OK; }      // (the host replay has no device work of its own)
void range_push(const char*) {}
void range_pop() {}

namespace match {
eao_status build_lists_multi(int nf, const eao_frame_view* const* Fs, const std::vector<Query>* qs, const uint8_t* const* qdescs, Lists* Ls, const Resident* const*) {
    for (int f = 0; f < nf; f++) {
        const int nq = (int)qs[f].size(), n = Fs[f]->n;
        Ls[f].start.assign(nq, 0); Ls[f].count.assign(nq, 0); Ls[f].items.clear();
        if (n == 0 || nq == 0) continue;
        EAO_REQUIRE(n < 65536, "at most 65535 keypoints per frame (indices are packed in 16 bits)");
        std::vector<uint32_t> items((size_t)nq * n + 1);
        const int tot = orc_candidate_lists(Fs[f], nq, qs[f].data(), qdescs[f], Ls[f].start.data(), Ls[f].count.data(), items.data(), (int)items.size());
        if (tot < 0) { set_error("candidate lists overflow"); return EAO_ERR_INTERNAL; }
        Ls[f].items.assign(items.begin(), items.begin() + tot);
    }
    return EAO_OK;
}
eao_status build_lists(const eao_frame_view* F, const std::vector<Query>& q, const uint8_t* qdesc, Lists& L, const Resident*) { return build_lists_multi(1, &F, &q, &qdesc, &L, nullptr); }
eao_status pair_distances(const uint8_t* descA, int nA, const uint8_t* descB, int nB, const std::vector<int>& ia, const std::vector<int>& ib,
                          std::vector<unsigned short>& dist) {
    dist.assign(ia.size(), 0);
    for (size_t k = 0; k < ia.size(); k++) {
        if (ia[k] < 0 || ia[k] >= nA || ib[k] < 0 || ib[k] >= nB) { set_error("pair %zu out of range", k); return EAO_ERR_INTERNAL; }
        dist[k] = (unsigned short)orc_pair_distance(descA + 32 * (size_t)ia[k], descB + 32 * (size_t)ib[k]);
    }
    return EAO_OK;
}
}  // namespace match
}  // namespace eao

extern "C" const char* eao_last_error(void) { return g_err; }
