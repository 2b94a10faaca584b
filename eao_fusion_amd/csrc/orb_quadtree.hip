// orb_quadtree.hip -- k_quadtree: ORBextractor::DistributeOctTree (src/ORBextractor.cc:537-763) for one (frame, level) per workgroup, and its launch.
// (Split from orb.hip in round 6, a pure move; the file header of orb.hip describes the kernel.)
#include "orb_internal.h"

using namespace eao::orb;

namespace {

// ---------------------------------------------------------------------------------------------- quad-tree
// In-place exclusive scan of a[0..n) by the whole kQT-thread block; returns the total.  Caller guarantees a[]
// is fully written and visible (barrier) before the call; the function ends with a barrier.
template <int kQT>
__device__ int block_excl_scan(int* a, int n, int* wtmp) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n + kQT - 1) / kQT;
    const int b = min(t * per, n), e = min(b + per, n);
    int ssum = 0;
    for (int i = b; i < e; i++) ssum += a[i];
    // inclusive scan over the wave on the VALU (DPP row shifts inside the rows of 16, then the row broadcasts 15 / 31) instead
    // of six ds_bpermute round trips: the quad-tree calls this scan some thirty times per level, each on its critical path
    int v = ssum;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2 and 3
    __syncthreads();  // protect wtmp from the previous call's readers
    if (lane == 63) wtmp[wv] = v;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kQT / 64; w++) {
        const int x = wtmp[w];
        if (w < wv) woff += x;
        total += x;
    }
    int run = woff + v - ssum;
    for (int i = b; i < e; i++) {
        const int x = a[i];
        a[i] = run;
        run += x;
    }
    __syncthreads();
    return total;
}

// The same exclusive scan by ONE wave, in place, without a workgroup barrier (the caller fences at wavefront scope).
__device__ __forceinline__ int wave_excl_scan(int* a, int n, int lane) {
    const int per = (n + 63) >> 6;
    const int b = min(lane * per, n), e = min(b + per, n);
    int ssum = 0;
    for (int i = b; i < e; i++) ssum += a[i];
    int v = ssum;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2 and 3
    const int total = __builtin_amdgcn_readlane(v, 63);
    int run = v - ssum;
    for (int i = b; i < e; i++) {
        const int x = a[i];
        a[i] = run;
        run += x;
    }
    eao::wave_sync();
    return total;
}

constexpr unsigned kQtLeaf = 0xFFFFFFFFu;          // mid of an entry that holds one key (every key then maps to quadrant 0)
__device__ __forceinline__ unsigned box_mid(short4 bx) {
    return (unsigned)(bx.x + ((bx.z - bx.x + 1) >> 1)) | ((unsigned)(bx.y + ((bx.w - bx.y + 1) >> 1)) << 16);
}
__device__ __forceinline__ int quadrant_mid(unsigned key, unsigned mid) {
    const unsigned x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    return (x < (mid & 0xFFFFu) ? 0 : 1) + (y < (mid >> 16) ? 0 : 2);
}
__device__ __forceinline__ int quadrant(unsigned key, short4 bx) {
    const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    const int mx = bx.x + ((bx.z - bx.x + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)
    const int my = bx.y + ((bx.w - bx.y + 1) >> 1);
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);
}

// One workgroup per (level, frame).  List entries are (box, count, creation rank); `nodeof[k]` is the list
// position of candidate k's node.  Every pass (a) histograms children of all multi-key nodes, (b) picks the set
// of nodes that upstream would split in this pass and their processing order, (c) lays out the new list exactly
// as upstream's push_front/erase sequence would leave it.
struct QtArgs {
    const Geom* g; const unsigned* cellcand; const int* cellcnt; unsigned* levelkps; int* levelcnt; int f, l, M;
    long long* dbg;
    unsigned* candOut;   // global copy of the gathered candidates (read back by eao_orb_level_candidates)
};

// The candidate keys and their node index live in LDS when the level's M candidates fit the launch's LDS budget
// (g->qtLdsCand; always at the default 1000-feature settings), else in the global scratch arrays: the body is inlined
// once per placement.
template <int kQT, class KeyPtr, class NofPtr>
__device__ __forceinline__ void quadtree_body(const QtArgs& A, unsigned char* smem, int* wtmp, int* shv, KeyPtr keys, NofPtr nof) {
    const Geom* __restrict__ g = A.g;
    int& sh_S = shv[0]; int& sh_phase = shv[1]; int& sh_done = shv[2]; int& sh_rstar = shv[3]; int& sh_nexp = shv[4];
    long long* dbg = A.dbg;
    // (phase stamps of diagnostic runs, thread 0 only, kept in LDS: as per-thread registers they cost 22 VGPRs of a kernel that
    //  spills at 1024 threads)
    __shared__ long long dacc[11];
    if (dbg && threadIdx.x == 0) { for (int i = 0; i < 10; i++) dacc[i] = 0; dacc[10] = clock64(); }
#define QSTAMP(i) do { if (dbg && threadIdx.x == 0) { const long long now_ = clock64(); dacc[i] += now_ - dacc[10]; dacc[10] = now_; } } while (0)
    const int t = threadIdx.x, lane = t & 63;
    const int l = A.l, f = A.f, M = A.M;
    const LevelGeom L = g->L[l];
    const int LC = L.listCap, N = L.quota;
    // ---- LDS carve-up
    short4* box0 = reinterpret_cast<short4*>(smem);
    short4* box1 = box0 + LC;
    int* cnt0 = reinterpret_cast<int*>(box1 + LC);
    int* cnt1 = cnt0 + LC;
    int* crk0 = cnt1 + LC;
    int* crk1 = crk0 + LC;
    unsigned* mid0 = reinterpret_cast<unsigned*>(crk1 + LC);   // split point of a multi-key entry (mx | my << 16), kQtLeaf otherwise
    unsigned* mid1 = mid0 + LC;
    int* childcnt = reinterpret_cast<int*>(mid1 + LC);          // 4 per entry
    int* childpos = childcnt + 4 * LC;  // 4 per entry
    unsigned long long* rkey = reinterpret_cast<unsigned long long*>(childpos);   // sort keys of a careful pass (before childpos is filled)
    int* newpos = childpos + 4 * LC;
    int* order = newpos + LC;
    int* vlist = order + LC;
    int* procRank = vlist + LC;
    int* scanB = procRank + LC;
    int* scanA = scanB + LC;            // g->scanCap entries (>= LC and >= nCells); holds the exclusive cell prefix on entry

    // ---- gather this level's candidates in upstream order: cells row-major, corners row-major inside a cell.  One
    // thread per cell, sixteen independent loads in flight (a cell holds a handful of corners).
    const long long cslot = (long long)f * g->totalCells + L.cellBase;
    for (int c = t; c < L.nCells; c += kQT) {
        const int o = scanA[c], n = (c + 1 < L.nCells ? scanA[c + 1] : M) - o;
        const unsigned* srcc = A.cellcand + (cslot + c) * g->cellCap;
        // (sixteen loads in flight: a cell of the benchmark frames holds ~11 corners, so one memory round trip instead of three)
        for (int j0 = 0; j0 < n; j0 += 16) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = srcc[min(j0 + u, n - 1)];
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (j0 + u < n) { keys[o + j0 + u] = v[u]; A.candOut[o + j0 + u] = v[u]; }
        }
    }
    __syncthreads();
    // ---- initial nodes
    const int nIni = L.nIni;
    if (t < nIni) {
        box0[t] = make_short4((short)(int)(L.hX * (float)t), 0, (short)(int)(L.hX * (float)(t + 1)), (short)L.boxH);
        cnt0[t] = 0;
        crk0[t] = t;
    }
    __syncthreads();
    for (int k0 = 0; k0 < M; k0 += kQT) {   // (every candidate lands in one of <= 16 nodes: count by ballot, not by 3000 atomics on one word)
        const int k = k0 + t;
        int ini = -1;
        if (k < M) {
            ini = min((int)((float)(keys[k] & 0xFFF) / L.hX), nIni - 1);
            nof[k] = (unsigned short)ini;
        }
        for (int i = 0; i < nIni; i++) {
            const unsigned long long m = __ballot(ini == i);
            if (m && (t & 63) == 0) atomicAdd(&cnt0[i], __popcll(m));
        }
    }
    __syncthreads();
    if (t == 0) {  // drop empty initial nodes (nIni <= 16)
        int S = 0;
        for (int i = 0; i < nIni; i++) {
            newpos[i] = S;
            if (cnt0[i] > 0) { box0[S] = box0[i]; cnt0[S] = cnt0[i]; crk0[S] = S; mid0[S] = cnt0[i] > 1 ? box_mid(box0[i]) : kQtLeaf; S++; }
        }
        sh_S = S; sh_phase = 0; sh_done = 0;
    }
    __syncthreads();
    if (nIni > 1) {
        for (int k = t; k < M; k += kQT) nof[k] = (unsigned short)newpos[nof[k]];
        __syncthreads();
    }

    QSTAMP(0);
    int diters = 0;
    short4* box = box0; short4* nbox = box1;
    unsigned* mid = mid0; unsigned* nmid = mid1;
    int* cnt = cnt0; int* ncnt = cnt1;
    int* crk = crk0; int* ncrk = crk1;
    // Every pass: two sweeps over the M candidates by the whole workgroup (child histograms, re-homing) and, between them, the
    // list logic over the S <= N nodes.  The list logic runs in WAVE 0 ALONE, wave-synchronously (DPP scans, wavefront-scope
    // fences, no workgroup barrier): as a sequence of block-wide steps it was ~20 barriers per pass with a handful of
    // instructions between them -- 43 k of level 0's 132 k cycles.  Four barriers per pass remain (six in a careful pass,
    // whose O(n^2) ranking stays block-wide).
#define QT_WAVE_FENCE() eao::wave_sync()
    const int wv = t >> 6;
    for (int iter = 0; iter < 64 && !sh_done; iter++) {
        const int S = sh_S, phase = sh_phase;
        // (1) reset of the per-node scratch (everyone) ...
        for (int i = t; i < S; i += kQT) {
            procRank[i] = -1;
            childcnt[4 * i] = 0; childcnt[4 * i + 1] = 0; childcnt[4 * i + 2] = 0; childcnt[4 * i + 3] = 0;
        }
        if (t == 0) { sh_rstar = 0x7FFFFFFF; sh_nexp = 0; }
        __syncthreads();
        // ... and the multi-key entries in list order (wave 0, beside the other waves' share of the histogram sweep)
        int nCand = 0;
        if (wv == 0) {
            for (int i = lane; i < S; i += 64) scanA[i] = cnt[i] > 1 ? 1 : 0;
            QT_WAVE_FENCE();
            nCand = wave_excl_scan(scanA, S, lane);
            for (int i = lane; i < S; i += 64)
                if (cnt[i] > 1) vlist[scanA[i]] = i;
            QT_WAVE_FENCE();
        }
        QSTAMP(1);
        // (2) child histograms of every candidate
        // The candidates are in spatial order (cells row-major, corners row-major inside a cell), so neighbouring lanes mostly
        // hit the same (node, quadrant) bin: one LDS atomic per RUN of equal bins in the wave instead of one per candidate
        // (while the tree is shallow, thousands of atomics would otherwise queue on a handful of words).
        // (FOUR rows of kQT candidates per trip: node, key, split point of all four are fetched before any of them is used -- as one row
        //  per trip the sweep was a chain of three dependent LDS reads and an atomic, twelve times over for level 0: 7 k cycles per pass)
        for (int k0 = 0; k0 < M; k0 += 4 * kQT) {
            int nd[4], bin[4];
            unsigned ky[4], md[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int k = k0 + u * kQT + t; nd[u] = k < M ? (int)nof[k] : -1; ky[u] = k < M ? keys[k] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) md[u] = nd[u] >= 0 ? mid[nd[u]] : kQtLeaf;
#pragma unroll
            for (int u = 0; u < 4; u++) bin[u] = md[u] != kQtLeaf ? 4 * nd[u] + quadrant_mid(ky[u], md[u]) : -1;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (k0 + u * kQT >= M) break;          // (uniform)
                const int prev = __builtin_amdgcn_update_dpp(-2, bin[u], 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps -2): VALU, not the LDS crossbar
                const bool head = bin[u] != prev;
                const unsigned long long hm = __ballot(head);
                if (head && bin[u] >= 0) {
                    const unsigned long long rest = lane == 63 ? 0ull : (hm >> (lane + 1));
                    const int run = rest ? __ffsll((long long)rest) : 64 - lane;
                    atomicAdd(&childcnt[bin[u]], run);
                }
            }
        }
        __syncthreads();
        QSTAMP(2);
        // (3) processing order: list order (full pass) or (size, creation rank) descending (careful pass)
        if (phase == 1) {
            // rank by (size, creation rank) descending, block-wide: the two sort fields are packed into one 64-bit key per node
            // (size << 32 | creation rank; the pairs are unique) and laid out densely (childpos is free here, 8-byte aligned),
            // so that the counting loop is one broadcast LDS read and one compare per node.  When there are fewer nodes than
            // threads, 2 / 4 / ... adjacent lanes share a node's loop and add their counts with DPP shuffles.
            if (wv == 0) {
                for (int j = lane; j < nCand; j += 64) { const int me = vlist[j]; rkey[j] = ((unsigned long long)(unsigned)cnt[me] << 32) | (unsigned)crk[me]; }
                if (lane == 0) sh_nexp = nCand;         // (nCand lives in wave 0's registers: hand it to the others; reset below)
            }
            __syncthreads();
            const int nC2 = sh_nexp;
            int sl = 1;
            while (2 * sl * nC2 <= kQT && sl < 16) sl *= 2;
            for (int j0 = 0; j0 < nC2; j0 += kQT / sl) {
                const int j = j0 + t / sl, part = t & (sl - 1);
                int r = 0;
                if (j < nC2) {
                    const unsigned long long mk = rkey[j];
#pragma unroll 8
                    for (int u = part; u < nC2; u += sl) r += rkey[u] > mk;
                }
                for (int d = sl >> 1; d >= 1; d >>= 1) r += __shfl_xor(r, d);
                if (j < nC2 && part == 0) order[r] = vlist[j];
            }
            __syncthreads();
            if (t == 0) sh_nexp = 0;
        }
        QSTAMP(3);
        if (wv == 0) {
            if (phase == 0) {
                for (int j = lane; j < nCand; j += 64) order[j] = vlist[j];
                QT_WAVE_FENCE();
            }
            // (4) growth prefix in processing order; the careful pass stops at the first prefix reaching N
            for (int r = lane; r < nCand; r += 64) {
                const int i = order[r];
                scanA[r] = (childcnt[4 * i] > 0) + (childcnt[4 * i + 1] > 0) + (childcnt[4 * i + 2] > 0) + (childcnt[4 * i + 3] > 0);
            }
            QT_WAVE_FENCE();
            wave_excl_scan(scanA, nCand, lane);
            int rstar = 0x7FFFFFFF;
            if (phase == 1) {
                for (int r = lane; r < nCand; r += 64) {
                    const int i = order[r];
                    const int ne = (childcnt[4 * i] > 0) + (childcnt[4 * i + 1] > 0) + (childcnt[4 * i + 2] > 0) + (childcnt[4 * i + 3] > 0);
                    if (S + scanA[r] + ne - (r + 1) >= N) rstar = min(rstar, r);
                }
                for (int d = 32; d >= 1; d >>= 1) rstar = min(rstar, __shfl_xor(rstar, d));
            }
            QSTAMP(4);
            const int nProc = (phase == 1 && rstar != 0x7FFFFFFF) ? rstar + 1 : nCand;
            int totalChildren = 0;
            if (nProc > 0) {
                const int i = order[nProc - 1];
                totalChildren = scanA[nProc - 1] + (childcnt[4 * i] > 0) + (childcnt[4 * i + 1] > 0) + (childcnt[4 * i + 2] > 0) + (childcnt[4 * i + 3] > 0);
            }
            for (int r = lane; r < nProc; r += 64) procRank[order[r]] = r;
            QT_WAVE_FENCE();
            for (int i = lane; i < S; i += 64) scanB[i] = procRank[i] < 0 ? 1 : 0;
            QT_WAVE_FENCE();
            wave_excl_scan(scanB, S, lane);
            QSTAMP(5);
            // (5) new list: children of the LAST processed node first (each as n4,n3,n2,n1), untouched entries after
            int myexp = 0;
            for (int i = lane; i < S; i += 64) {
                const int r = procRank[i];
                if (r < 0) {
                    const int p = totalChildren + scanB[i];
                    nbox[p] = box[i]; ncnt[p] = cnt[i]; ncrk[p] = crk[i]; nmid[p] = mid[i];
                    childpos[4 * i] = p; childpos[4 * i + 1] = p; childpos[4 * i + 2] = p; childpos[4 * i + 3] = p;   // (every key of an untouched entry moves with it)
                } else {
                    const short4 b = box[i];
                    const short mx = (short)(b.x + ((b.z - b.x + 1) >> 1)), my = (short)(b.y + ((b.w - b.y + 1) >> 1));
                    const int ne = (childcnt[4 * i] > 0) + (childcnt[4 * i + 1] > 0) + (childcnt[4 * i + 2] > 0) + (childcnt[4 * i + 3] > 0);
                    int p = totalChildren - (scanA[r] + ne);
                    for (int q = 3; q >= 0; q--) {
                        const int c = childcnt[4 * i + q];
                        if (c > 0) {
                            short4 nb;
                            nb.x = (q & 1) ? mx : b.x; nb.z = (q & 1) ? b.z : mx;
                            nb.y = (q & 2) ? my : b.y; nb.w = (q & 2) ? b.w : my;
                            nbox[p] = nb; ncnt[p] = c; ncrk[p] = 4 * r + q; nmid[p] = c > 1 ? box_mid(nb) : kQtLeaf;
                            childpos[4 * i + q] = p;
                            p++;
                            myexp += c > 1;
                        }
                    }
                }
            }
            for (int d = 32; d >= 1; d >>= 1) myexp += __shfl_xor(myexp, d);
            // (7) upstream's termination tests (src/ORBextractor.cc:660-737)
            if (lane == 0) {
                const int S2 = totalChildren + S - nProc;
                sh_S = S2;
                if (S2 >= N || S2 == S) sh_done = 1;
                else if (phase == 0 && S2 + 3 * myexp > N) sh_phase = 1;
            }
        }
        __syncthreads();
        QSTAMP(6);
        // (6) re-home the candidates
        for (int k = t; k < M; k += kQT) {
            const int nd = nof[k];
            nof[k] = (unsigned short)childpos[4 * nd + quadrant_mid(keys[k], mid[nd])];
        }
        __syncthreads();
        QSTAMP(7);
        diters++;
        short4* tb = box; box = nbox; nbox = tb;
        unsigned* tm = mid; mid = nmid; nmid = tm;
        int* ti = cnt; cnt = ncnt; ncnt = ti;
        ti = crk; crk = ncrk; ncrk = ti;
    }
#undef QT_WAVE_FENCE
    // ---- best response per node, first candidate wins ties (strict '>' at src/ORBextractor.cc:752)
    const int S = sh_S;
    unsigned* best = reinterpret_cast<unsigned*>(scanB);
    for (int i = t; i < S; i += kQT) best[i] = 0;
    __syncthreads();
    for (int k = t; k < M; k += kQT) atomicMax(&best[nof[k]], ((keys[k] >> 24) << 20) | (0xFFFFFu - (unsigned)k));
    __syncthreads();
    unsigned* out = A.levelkps + (long long)f * g->totalKpCap + L.kpBase;
    for (int i = t; i < S; i += kQT) {
        const unsigned k = 0xFFFFFu - (best[i] & 0xFFFFFu);
        out[i] = keys[k];
    }
    if (t == 0) A.levelcnt[f * g->nlevels + l] = S;
    QSTAMP(8);
    if (dbg && t == 0 && A.dbg) {
        long long* o = dbg + 16 * l;
        for (int i = 0; i < 9; i++) o[i] = dacc[i];
        o[9] = diters; o[10] = M; o[11] = S;
    }
#undef QSTAMP
}

template <int kQT>
__device__ __noinline__ void quadtree_global(const Geom* __restrict__ g, const unsigned* cellcand, const int* cellcnt, unsigned* cand, unsigned short* nodeof,
                                             unsigned* levelkps, int* levelcnt, int* candcnt, int f, int l, unsigned char* base, long long* dbg, int* wtmp, int* shv) {
    const int t = threadIdx.x;
    const LevelGeom L = g->L[l];
    int* scanA = reinterpret_cast<int*>(base + (size_t)L.listCap * (2 * sizeof(short4) + sizeof(int) * kQtNodeInts));
    const long long cslot = (long long)f * g->totalCells + L.cellBase;
    for (int i = t; i < L.nCells; i += kQT) scanA[i] = cellcnt[cslot + i];
    __syncthreads();
    const int M = block_excl_scan<kQT>(scanA, L.nCells, wtmp);
    if (t == 0) candcnt[f * g->nlevels + l] = M;
    if (M == 0) {
        if (t == 0) levelcnt[f * g->nlevels + l] = 0;
        return;
    }
    QtArgs A = {g, cellcand, cellcnt, levelkps, levelcnt, f, l, M, dbg, cand + (long long)f * g->totalCandCap + L.candBase};
    quadtree_body<kQT>(A, base, wtmp, shv, cand + (long long)f * g->totalCandCap + L.candBase, nodeof + (long long)f * g->totalCandCap + L.candBase);
}

template <int kQT>
__global__ __launch_bounds__(kQT, kQT == 256 ? 4 : 1) void k_quadtree(const Geom* __restrict__ g, const unsigned* __restrict__ cellcand,
                                                  const int* __restrict__ cellcnt, unsigned* __restrict__ cand,
                                                  unsigned short* __restrict__ nodeof, unsigned* __restrict__ levelkps,
                                                  int* __restrict__ levelcnt, int* __restrict__ candcnt, int f0, long long* dbg, int l0,
                                                  unsigned char* __restrict__ qtnodes) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int wtmp[kQT / 64];
    __shared__ int shv[8];
    const int t = threadIdx.x;
    // level-major dispatch (frames fastest): the long level-0 workgroups of EVERY frame start first and the short top levels
    // fill the gaps behind them -- with the levels of a frame dispatched together the last frames' level 0 was the tail
    const int l = blockIdx.y + l0, f = blockIdx.x + f0;   // the launch covers levels l0 .. l0 + gridDim.y - 1
    const LevelGeom L = g->L[l];
    if (g->qtNodesGlobal) {   // node lists too large for LDS: the same algorithm over a global workspace (slower, never refused)
        quadtree_global<kQT>(g, cellcand, cellcnt, cand, nodeof, levelkps, levelcnt, candcnt, f, l, qtnodes + (long long)f * g->qtNodeFrameBytes + L.nodeOff,
                        (dbg && f == f0) ? dbg : nullptr, wtmp, shv);
        return;
    }
    int* scanA = reinterpret_cast<int*>(smem + (size_t)L.listCap * (2 * sizeof(short4) + sizeof(int) * kQtNodeInts));
    const long long cslot = (long long)f * g->totalCells + L.cellBase;
    for (int i = t; i < L.nCells; i += kQT) scanA[i] = cellcnt[cslot + i];
    __syncthreads();
    const int M = block_excl_scan<kQT>(scanA, L.nCells, wtmp);
    if (t == 0) candcnt[f * g->nlevels + l] = M;
    if (M == 0) {
        if (t == 0) levelcnt[f * g->nlevels + l] = 0;
        return;
    }
    QtArgs A = {g, cellcand, cellcnt, levelkps, levelcnt, f, l, M, (dbg && f == f0) ? dbg : nullptr, cand + (long long)f * g->totalCandCap + L.candBase};
    if (M <= g->qtLdsCand) {
        unsigned* keysL = reinterpret_cast<unsigned*>(smem + g->qtKeysOff);
        quadtree_body<kQT>(A, smem, wtmp, shv, keysL, reinterpret_cast<unsigned short*>(keysL + g->qtLdsCand));
    } else {
        quadtree_body<kQT>(A, smem, wtmp, shv, cand + (long long)f * g->totalCandCap + L.candBase, nodeof + (long long)f * g->totalCandCap + L.candBase);
    }
}

}  // namespace

void eao::orb::launch_quadtree(const eao_orb* h, hipStream_t str, int nb, int f0, int lFirst, int nLev) {
    if (nb < 36) hipLaunchKernelGGL(k_quadtree<kQTSmall>, dim3(nb, nLev), dim3(kQTSmall), h->quadLds, str, h->d_geom.p, h->d_cellcand.p, h->d_cellcnt.p, h->d_cand.p,
                       h->d_nodeof.p, h->d_levelkps.p, h->d_levelcnt.p, h->d_candcnt.p, f0, h->d_dbg, lFirst, h->d_qtnodes.p);
    else hipLaunchKernelGGL(k_quadtree<kQTLarge>, dim3(nb, nLev), dim3(kQTLarge), h->quadLds, str, h->d_geom.p, h->d_cellcand.p, h->d_cellcnt.p, h->d_cand.p,
                       h->d_nodeof.p, h->d_levelkps.p, h->d_levelcnt.p, h->d_candcnt.p, f0, h->d_dbg, lFirst, h->d_qtnodes.p);
}

eao_status eao::orb::quadtree_reserve_lds(size_t bytes) {      // per-function, process-wide state: only ever raised (another handle with a larger nfeatures may be in use)
    static std::atomic<int> cur{0};
    int have = cur.load();
    while ((int)bytes > have) {
        EAO_HIP(hipFuncSetAttribute((const void*)k_quadtree<kQTSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        EAO_HIP(hipFuncSetAttribute((const void*)k_quadtree<kQTLarge>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (cur.compare_exchange_weak(have, (int)bytes)) break;
    }
    return EAO_OK;
}
