// match.hip -- guided ORB matching for MI355X (gfx950): the two per-frame SearchByProjection variants.
//
// Stands behind ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (reference src/ORBmatcher.cc:45-129)
// and ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (:1328-1472), with
// Frame::GetFeaturesInArea / PosInGrid (src/Frame.cc:696-761) as the candidate generator.
//
// Upstream is greedy over queries (a keypoint claimed by an earlier query is skipped by later ones), so only the
// candidate generation + Hamming distances are data parallel:
//   k_match_candidates  one wavefront per query walks the frame's keypoints in GRID ORDER (cell column-major, then
//                       insertion order -- the order upstream's nested cell loops visit them), applies the cell-range,
//                       level, window and stereo gates with the reference's float expressions, computes the 256-bit
//                       Hamming distance of the survivors and appends (index, distance) through ballot compaction, so
//                       every query's list is already in upstream's candidate order.  Two sweeps (count, reserve a
//                       segment with one atomic, fill) keep the output compact.
// The assignment loops over those lists are replayed on the host exactly as upstream runs them: csrc/search.hip (a host-only translation unit since round 4).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "match_internal.h"
#include "chain_internal.h"

using eao::match::Query;
using eao::match::Lists;

namespace {

struct FrameDev {
    int n, nOrdered;
    const int* nOrderedDev;   // when set: the number of ordered entries lives on the device (chained tracking)
    const float* kx; const float* ky; const int* oct; const float* ur;
    const uint4* desc;        // n x 2
    const int* order;         // keypoint indices in grid order
    const unsigned short* cellx; const unsigned short* celly;   // per ordered entry
    const int* colStart;      // when set (cols + 1 entries): first ordered entry of every grid column -- the walk list is in (column, row,
                              // index) order, so a query's candidates all lie in [colStart[x0], colStart[x1 + 1])
    float minX, minY, invW, invH;
    int cols, rows;
};

__device__ __forceinline__ int dist256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// out: packed (distance << 16 | keypoint index); segStart/segCount per query; cursor = global fill position
__global__ __launch_bounds__(256) void k_match_candidates(FrameDev F, const Query* __restrict__ q, const uint4* __restrict__ qdesc, int nq,
                                                          unsigned* __restrict__ out, int outCap, int* __restrict__ segStart,
                                                          int* __restrict__ segCount, int* __restrict__ cursor, eao::match::QueryBuild B) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    Query Q;
    if (B.inView) {      // the tracker's chain: the window of local map point qi, built here (QueryBuild, chain_internal.h)
        const int m = qi;
        Q.active = (B.active[m] && B.inView[m] && !B.skip[m]) ? 1 : 0;
        int lvl = Q.active ? B.level[m] : 0;
        if (Q.active && (lvl < 0 || lvl >= B.nlevels)) { if (lane == 0) atomicOr(B.errFlags, 1); Q.active = 0; lvl = 0; }   // (upstream would index out of range)
        float r = (double)B.viewCos[m] > refc::VIEWCOS_NARROW ? refc::RADIUS_NARROW : refc::RADIUS_WIDE;          // RadiusByViewingCos, :131-137
        if (B.th != 1.0f) r *= B.th;
        const float rs = Q.active ? r * B.scale[lvl] : 0.f;
        Q.x = B.projX[m]; Q.y = B.projY[m]; Q.r = rs;
        Q.minLevel = lvl - 1; Q.maxLevel = lvl;
        Q.urRef = B.projXR[m]; Q.urTol = rs;
        if (lane == 0) B.qOut[m] = Q;
    } else {
        Q = q[qi];
    }
    if (F.nOrderedDev) F.nOrdered = *F.nOrderedDev;
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    bool any = Q.active != 0;
    if (any) {   // Frame::GetFeaturesInArea, src/Frame.cc:701-717 (float expressions kept as written upstream)
        x0 = max(0, (int)floorf((Q.x - F.minX - Q.r) * F.invW));
        x1 = min(F.cols - 1, (int)ceilf((Q.x - F.minX + Q.r) * F.invW));
        y0 = max(0, (int)floorf((Q.y - F.minY - Q.r) * F.invH));
        y1 = min(F.rows - 1, (int)ceilf((Q.y - F.minY + Q.r) * F.invH));
        if (x0 >= F.cols || x1 < 0 || y0 >= F.rows || y1 < 0) any = false;
    }
    const bool checkLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
    const uint4 d0 = qdesc[2 * qi], d1 = qdesc[2 * qi + 1];
    int base = 0, total = 0;
    // (the tracker's chain hands the column starts over: a window of one or two grid columns then costs one round of the wave per
    //  sweep instead of a walk over every keypoint of the frame -- 18 rounds for 1100 keypoints)
    int oBeg = 0, oEnd = F.nOrdered;
    if (any && F.colStart) { oBeg = F.colStart[x0]; oEnd = min(F.colStart[x1 + 1], F.nOrdered); }
    // A window of one or two grid columns is at most four rounds of the wave: the candidates (index and distance) of such a window stay in the
    // lanes' registers while the segment is reserved, and the second sweep -- three dependent loads per round all over again -- is left out.
    if (any && oEnd - oBeg <= 4 * 64) {
        int ci[4], cd[4], run = 0;
        unsigned long long cm[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            ci[r] = -1; cd[r] = 0; cm[r] = 0;
            const int o = oBeg + 64 * r + lane;
            if (oBeg + 64 * r >= oEnd) continue;          // (uniform)
            bool pass = false;
            int i = 0;
            if (o < oEnd) {
                const int cx = F.cellx[o], cy = F.celly[o];
                if (cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1) {
                    i = F.order[o];
                    const int oc = F.oct[i];
                    bool ok = true;
                    if (checkLevels) {
                        if (oc < Q.minLevel) ok = false;
                        if (Q.maxLevel >= 0 && oc > Q.maxLevel) ok = false;
                    }
                    if (ok) {
                        const float dx = F.kx[i] - Q.x, dy = F.ky[i] - Q.y;
                        ok = fabsf(dx) < Q.r && fabsf(dy) < Q.r;
                    }
                    if (ok) {
                        const float u = F.ur[i];
                        if (u > 0 && fabsf(Q.urRef - u) > Q.urTol) ok = false;
                    }
                    pass = ok;
                }
            }
            cm[r] = __ballot(pass);
            ci[r] = pass ? i : -1;
            if (pass) cd[r] = dist256(d0, d1, F.desc[2 * i], F.desc[2 * i + 1]);
            run += __popcll(cm[r]);
        }
        total = run;
        if (lane == 0) {
            base = total ? atomicAdd(cursor, total) : 0;
            segStart[qi] = base;
            segCount[qi] = total;
        }
        base = __shfl(base, 0);
        int off = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (ci[r] >= 0) {
                const int pos = base + off + __popcll(cm[r] & ((1ull << lane) - 1));
                if (pos < outCap) out[pos] = ((unsigned)cd[r] << 16) | (unsigned)ci[r];
            }
            off += __popcll(cm[r]);
        }
        return;
    }
    for (int sweep = 0; sweep < 2; sweep++) {
        int run = 0;
        if (any) {
            for (int o0 = oBeg; o0 < oEnd; o0 += 64) {
                const int o = o0 + lane;
                bool pass = false;
                int i = 0;
                if (o < oEnd) {
                    const int cx = F.cellx[o], cy = F.celly[o];
                    if (cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1) {
                        i = F.order[o];
                        const int oc = F.oct[i];
                        bool ok = true;
                        if (checkLevels) {
                            if (oc < Q.minLevel) ok = false;
                            if (Q.maxLevel >= 0 && oc > Q.maxLevel) ok = false;
                        }
                        if (ok) {
                            const float dx = F.kx[i] - Q.x, dy = F.ky[i] - Q.y;
                            ok = fabsf(dx) < Q.r && fabsf(dy) < Q.r;
                        }
                        if (ok) {
                            const float u = F.ur[i];
                            if (u > 0 && fabsf(Q.urRef - u) > Q.urTol) ok = false;
                        }
                        pass = ok;
                    }
                }
                const unsigned long long m = __ballot(pass);
                if (sweep == 1 && pass) {
                    const int pos = base + run + __popcll(m & ((1ull << lane) - 1));
                    if (pos < outCap) out[pos] = ((unsigned)dist256(d0, d1, F.desc[2 * i], F.desc[2 * i + 1]) << 16) | (unsigned)i;
                }
                run += __popcll(m);
            }
        }
        if (sweep == 0) {
            total = run;
            if (lane == 0) {
                base = total ? atomicAdd(cursor, total) : 0;
                segStart[qi] = base;
                segCount[qi] = total;
            }
            base = __shfl(base, 0);
            if (total == 0) break;
        }
    }
}

struct PinBuf {   // grow-only pinned host buffer (device-visible: kernels write results straight into it)
    unsigned char* p = nullptr;
    size_t n = 0;
    eao_status reserve(size_t need) {
        if (need <= n) return EAO_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        const size_t cap = need + (need >> 2) + 4096;
        EAO_HIP(hipHostMalloc((void**)&p, cap, hipHostMallocMapped));
        n = cap;
        return EAO_OK;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};
struct HostView {   // what the old std::vector staging offered
    PinBuf buf;
    eao_status resize(size_t n) { return buf.reserve(n); }
    unsigned char* data() { return buf.p; }
};
struct Ctx {   // per-thread workspace, grow-only
    hipStream_t stream = nullptr;
    eao::DevBuf<unsigned char> dev;
    HostView host;           // pinned staging of everything uploaded: ONE asynchronous H2D copy per call
    PinBuf out;              // candidate items, written by the kernel over PCIe (zero-copy), read after the stream sync
    eao::DevBuf<int> metaDev; // segStart[nq], segCount[nq], cursor (device: the cursor is an atomic)
    PinBuf meta;             // their pinned copy (asynchronous D2H in the same stream)
    std::vector<int> cellOf, cellStart;      // scratch of the grid order's counting sort
    ~Ctx() { if (stream) (void)hipStreamDestroy(stream); }
};
thread_local Ctx g_ctx;

}  // namespace

// uploads the frames + their queries, runs the candidate kernel once per frame, downloads the compact lists: ONE staging block, one upload, one
// download and one synchronisation for all of them (round 4: the batched LocalMapping-side searches -- a single frame is a batch of one).
// qdesc[f]: the query descriptors of frame f (nq[f] x 32 bytes); frames may share them (Fuse: the same map points into every target keyframe)
eao_status eao::match::build_lists_multi(int nf, const eao_frame_view* const* Fs, const std::vector<Query>* qs, const uint8_t* const* qdescs, Lists* Ls,
                                         const Resident* const* res) {
    Ctx& c = g_ctx;
    eao_status st = eao::require_device();
    if (st) return st;
    if (!c.stream) EAO_HIP(eao::create_stream(&c.stream, eao::StreamClass::Latency));
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    struct Ord { int cx, cy, i; };
    struct Plan { int n, nq, no; size_t oKx, oKy, oUr, oOc, oOr, oCx, oCy, oDe, oQ, oQd, oOut, oMeta; size_t outCap; std::vector<Ord> ord; };
    std::vector<Plan> plan(nf);
    size_t off = 0, outOff = 0, metaOff = 0;
    // shared query descriptors are staged once
    std::vector<size_t> qdOff(nf, 0);
    for (int f = 0; f < nf; f++) {
        const eao_frame_view* F = Fs[f];
        Plan& P = plan[f];
        P.n = F->n; P.nq = (int)qs[f].size(); P.no = 0;
        Ls[f].start.assign(P.nq, 0); Ls[f].count.assign(P.nq, 0); Ls[f].items.clear();
        if (P.n == 0 || P.nq == 0) { P.n = 0; P.nq = 0; continue; }
        EAO_REQUIRE(P.n < 65536, "at most 65535 keypoints per frame (indices are packed in 16 bits)");
        const Resident* R = res ? res[f] : nullptr;
        if (R) {      // the frame is on the device already, in grid order: only its queries (and their descriptors) travel
            EAO_REQUIRE(R->n == P.n, "resident frame %d holds %d keypoints, the view %d", f, R->n, P.n);
            P.no = R->no;
            const size_t no = std::max(P.no, 1), nq = P.nq;
            P.oQ = off; off = al(off + sizeof(Query) * nq);
            int shared = -1;
            for (int g = 0; g < f && shared < 0; g++) if (plan[g].nq == P.nq && qdescs[g] == qdescs[f]) shared = g;
            if (shared >= 0) P.oQd = plan[shared].oQd;
            else { P.oQd = off; off = al(off + 32 * nq); }
            P.outCap = nq * no;
            P.oOut = outOff; outOff += P.outCap;
            P.oMeta = metaOff; metaOff += 2 * nq + 1;
            continue;
        }
        // grid order: PosInGrid (src/Frame.cc:751-761) then cell column-major, insertion (= index) order inside a cell
        // (a counting sort by cell -- the keypoints of a cell keep their index order: a comparison sort of a thousand records took a third of a whole search call)
        {
            const size_t nCells = (size_t)F->grid_cols * F->grid_rows;
            std::vector<int>& cellOf = c.cellOf; std::vector<int>& cellStart = c.cellStart;
            cellOf.resize(P.n); cellStart.assign(nCells + 1, 0);
            for (int i = 0; i < P.n; i++) {
                const int px = (int)std::round((F->kp_x[i] - F->min_x) * F->grid_inv_w);
                const int py = (int)std::round((F->kp_y[i] - F->min_y) * F->grid_inv_h);
                const bool in = px >= 0 && px < F->grid_cols && py >= 0 && py < F->grid_rows;
                cellOf[i] = in ? px * F->grid_rows + py : -1;
                if (in) cellStart[cellOf[i] + 1]++;
            }
            for (size_t q = 0; q < nCells; q++) cellStart[q + 1] += cellStart[q];
            P.ord.resize(cellStart[nCells]);
            for (int i = 0; i < P.n; i++)
                if (cellOf[i] >= 0) P.ord[cellStart[cellOf[i]]++] = {cellOf[i] / F->grid_rows, cellOf[i] % F->grid_rows, i};
        }
        P.no = (int)P.ord.size();
        const size_t n = P.n, no = std::max(P.no, 1), nq = P.nq;
        // per frame: kx ky ur (float n) | oct (int n) | order (int no) | cellx celly (u16 no) | desc (32 n) | queries | (qdesc unless shared with an earlier frame)
        P.oKx = off; off = al(off + 4 * n);
        P.oKy = off; off = al(off + 4 * n);
        P.oUr = off; off = al(off + 4 * n);
        P.oOc = off; off = al(off + 4 * n);
        P.oOr = off; off = al(off + 4 * no);
        P.oCx = off; off = al(off + 2 * no);
        P.oCy = off; off = al(off + 2 * no);
        P.oDe = off; off = al(off + 32 * n);
        P.oQ = off; off = al(off + sizeof(Query) * nq);
        int shared = -1;
        for (int g = 0; g < f && shared < 0; g++) if (plan[g].nq == P.nq && qdescs[g] == qdescs[f]) shared = g;
        if (shared >= 0) P.oQd = plan[shared].oQd;
        else { P.oQd = off; off = al(off + 32 * nq); }
        P.outCap = nq * no;
        P.oOut = outOff; outOff += P.outCap;
        P.oMeta = metaOff; metaOff += 2 * nq + 1;
    }
    if (off == 0) return EAO_OK;
    EAO_REQUIRE(outOff < ((size_t)1 << 31), "candidate lists too large (%zu entries)", outOff);
    if ((st = c.host.resize(off))) return st;
    unsigned char* hb = c.host.data();
    for (int f = 0; f < nf; f++) {
        const eao_frame_view* F = Fs[f];
        const Plan& P = plan[f];
        if (!P.nq) continue;
        const size_t n = P.n;
        if (res && res[f]) {
            std::memcpy(hb + P.oQ, qs[f].data(), sizeof(Query) * (size_t)P.nq);
            std::memcpy(hb + P.oQd, qdescs[f], 32 * (size_t)P.nq);
            continue;
        }
        std::memcpy(hb + P.oKx, F->kp_x, 4 * n); std::memcpy(hb + P.oKy, F->kp_y, 4 * n);
        std::memcpy(hb + P.oUr, F->u_right, 4 * n); std::memcpy(hb + P.oOc, F->kp_octave, 4 * n);
        for (int k = 0; k < P.no; k++) {
            ((int*)(hb + P.oOr))[k] = P.ord[k].i;
            ((unsigned short*)(hb + P.oCx))[k] = (unsigned short)P.ord[k].cx;
            ((unsigned short*)(hb + P.oCy))[k] = (unsigned short)P.ord[k].cy;
        }
        std::memcpy(hb + P.oDe, F->descriptors, 32 * n);
        std::memcpy(hb + P.oQ, qs[f].data(), sizeof(Query) * (size_t)P.nq);
        std::memcpy(hb + P.oQd, qdescs[f], 32 * (size_t)P.nq);
    }
    if ((st = c.dev.reserve(off))) return st;
    if ((st = c.out.reserve(std::max(outOff, (size_t)1) * sizeof(unsigned)))) return st;
    if ((st = c.meta.reserve(metaOff * sizeof(int)))) return st;
    if ((st = c.metaDev.reserve(metaOff))) return st;
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, hb, off, hipMemcpyHostToDevice, s));
    EAO_HIP(hipMemsetAsync(c.metaDev.p, 0, metaOff * sizeof(int), s));   // (the cursors)
    for (int f = 0; f < nf; f++) {
        const eao_frame_view* F = Fs[f];
        const Plan& P = plan[f];
        if (!P.nq) continue;
        FrameDev D;
        D.nOrderedDev = nullptr;
        D.n = P.n; D.nOrdered = P.no;
        if (const Resident* R = res ? res[f] : nullptr) {
            D.kx = R->kx; D.ky = R->ky; D.ur = R->ur; D.oct = R->oct; D.order = R->order; D.cellx = R->cellx; D.celly = R->celly; D.colStart = R->colStart;
            D.desc = (const uint4*)R->desc;
        } else {
        D.kx = (const float*)(c.dev.p + P.oKx); D.ky = (const float*)(c.dev.p + P.oKy); D.ur = (const float*)(c.dev.p + P.oUr);
        D.oct = (const int*)(c.dev.p + P.oOc); D.order = (const int*)(c.dev.p + P.oOr);
        D.cellx = (const unsigned short*)(c.dev.p + P.oCx); D.celly = (const unsigned short*)(c.dev.p + P.oCy); D.colStart = nullptr;
        D.desc = (const uint4*)(c.dev.p + P.oDe);
        }
        D.minX = F->min_x; D.minY = F->min_y; D.invW = F->grid_inv_w; D.invH = F->grid_inv_h; D.cols = F->grid_cols; D.rows = F->grid_rows;
        int* md = c.metaDev.p + P.oMeta;
        hipLaunchKernelGGL(k_match_candidates, dim3(eao::cdiv(P.nq, 4)), dim3(256), 0, s, D, (const Query*)(c.dev.p + P.oQ),
                           (const uint4*)(c.dev.p + P.oQd), P.nq, (unsigned*)c.out.p + P.oOut, (int)std::min(P.outCap, (size_t)0x7FFFFFFF), md,
                           md + P.nq, md + 2 * (size_t)P.nq, eao::match::QueryBuild{});
    }
    int* meta = (int*)c.meta.p;
    EAO_HIP(hipMemcpyAsync(meta, c.metaDev.p, metaOff * sizeof(int), hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    for (int f = 0; f < nf; f++) {
        const Plan& P = plan[f];
        if (!P.nq) continue;
        const int* m = meta + P.oMeta;
        const int totalItems = m[2 * (size_t)P.nq];
        const unsigned* o = (const unsigned*)c.out.p + P.oOut;
        Ls[f].items.assign(o, o + std::max(totalItems, 0));
        for (int k = 0; k < P.nq; k++) { Ls[f].start[k] = m[k]; Ls[f].count[k] = m[P.nq + k]; }
    }
    return EAO_OK;
}
eao_status eao::match::build_lists(const eao_frame_view* F, const std::vector<Query>& q, const uint8_t* qdesc, Lists& L, const Resident* res) {
    return build_lists_multi(1, &F, &q, &qdesc, &L, res ? &res : nullptr);
}

namespace {

__global__ __launch_bounds__(256) void k_pair_distances(const uint4* __restrict__ A, const uint4* __restrict__ B, const int2* __restrict__ pairs,
                                                        int n, unsigned short* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int2 p = pairs[k];
    out[k] = (unsigned short)dist256(A[2 * p.x], A[2 * p.x + 1], B[2 * p.y], B[2 * p.y + 1]);
}

}  // namespace

eao_status eao::match::pair_distances(const uint8_t* descA, int nA, const uint8_t* descB, int nB, const std::vector<int>& ia,
                                      const std::vector<int>& ib, std::vector<unsigned short>& dist) {
    Ctx& c = g_ctx;
    eao_status st = eao::require_device();
    if (st) return st;
    if (!c.stream) EAO_HIP(eao::create_stream(&c.stream, eao::StreamClass::Latency));
    const size_t np = ia.size();
    dist.assign(np, 0);
    if (np == 0) return EAO_OK;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t oA = off; off = al(off + 32 * (size_t)nA);
    const size_t oB = off; off = al(off + 32 * (size_t)nB);
    const size_t oP = off; off = al(off + 8 * np);
    const size_t oD = off; off = al(off + 2 * np);
    if ((st = c.host.resize(off))) return st;
    unsigned char* hb = c.host.data();
    std::memcpy(hb + oA, descA, 32 * (size_t)nA);
    std::memcpy(hb + oB, descB, 32 * (size_t)nB);
    for (size_t k = 0; k < np; k++) { ((int*)(hb + oP))[2 * k] = ia[k]; ((int*)(hb + oP))[2 * k + 1] = ib[k]; }
    if ((st = c.dev.reserve(off))) return st;
    hipStream_t s = c.stream;
    EAO_HIP(hipMemcpyAsync(c.dev.p, hb, oD, hipMemcpyHostToDevice, s));
    if ((st = c.out.reserve(2 * np))) return st;
    hipLaunchKernelGGL(k_pair_distances, dim3(eao::cdiv((int)np, 256)), dim3(256), 0, s, (const uint4*)(c.dev.p + oA), (const uint4*)(c.dev.p + oB),
                       (const int2*)(c.dev.p + oP), (int)np, (unsigned short*)c.out.p);
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    std::memcpy(dist.data(), c.out.p, 2 * np);
    return EAO_OK;
}

#include "chain_internal.h"
eao_status eao::match::enqueue_candidates_device(const FrameDevArgs& F, const Query* q, const uint8_t* qdesc, int nq, unsigned* out, int outCap,
                                                 int* segStart, int* segCount, int* cursor, hipStream_t s, bool cursorIsZero, const QueryBuild* build) {
    if (nq <= 0) return EAO_OK;
    FrameDev D;
    D.n = F.cap; D.nOrdered = 0; D.nOrderedDev = F.nOrdered;
    D.kx = F.kx; D.ky = F.ky; D.oct = F.oct; D.ur = F.ur; D.desc = (const uint4*)F.desc;
    D.order = F.order; D.cellx = F.cellx; D.celly = F.celly; D.colStart = F.colStart;
    D.minX = F.minX; D.minY = F.minY; D.invW = F.invW; D.invH = F.invH; D.cols = F.cols; D.rows = F.rows;
    if (!cursorIsZero) EAO_HIP(hipMemsetAsync(cursor, 0, sizeof(int), s));
    QueryBuild B;
    std::memset(&B, 0, sizeof(B));
    if (build) B = *build;
    hipLaunchKernelGGL(k_match_candidates, dim3(eao::cdiv(nq, 4)), dim3(256), 0, s, D, q, (const uint4*)qdesc, nq, out, outCap, segStart, segCount, cursor, B);
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

