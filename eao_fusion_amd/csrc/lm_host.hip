// lm_host.hip -- host side of the Levenberg-Marquardt engine: per-thread contexts, BAJob (validation, arena, pinned mirror, active structure, the map-scale path's
// covisibility / tile structure, host-stepped trials), eao_local_ba / eao_local_ba_batch / eao_bundle_adjustment(_planes) and the traces.  Kernels: lba.hip, gba.hip
// (launched through BALaunch).  Shared pieces: lm_internal.h.  (Round 6: split out of csrc/lm.hip.)
#include "lm_internal.h"
#include "host_crew.h"

namespace eao {
namespace lm {
thread_local LMTraceHost g_trace;
thread_local LMContext g_ctx;

eao_status ctx_init(LMContext& c, bool ownStream, eao::StreamClass cls) {
    eao_status st = eao::require_device();
    if (st) return st;
    if (ownStream) {
        hipStream_t& q = c.byClass[(int)cls];
        if (!q) EAO_HIP(eao::create_stream(&q, cls));
        c.stream = q;
        if (!c.ev0) {
            EAO_HIP(hipEventCreate(&c.ev0));
            EAO_HIP(hipEventCreate(&c.ev1));
        }
    }
    if (!c.status) {
        EAO_HIP(hipHostMalloc((void**)&c.status, sizeof(BAStatus), hipHostMallocMapped));
        std::memset(c.status, 0, sizeof(BAStatus));
    }
    return EAO_OK;
}
}  // namespace lm
}  // namespace eao

// mode 0: Optimizer::LocalBundleAdjustment (two passes with the outlier pass between them, Huber kernels in the first).
// mode 1: Optimizer::BundleAdjustment over keyframes and map points (src/Optimizer.cc:55-323): ONE optimize(its_first) call,
//         Huber kernels only when `robust`, delta_mono = sqrt(5.99) (:94), no outlier pass, no observation is erased.
//         With `pl`: the MapPlane vertices / EdgePlane edges of :203-252 ride along as landmarks nPo.. / edges Ept.. .
namespace {


// One window in flight: LocalBundleAdjustment / BundleAdjustment of one problem on one context (device arena + pinned mirrors).
// The uploads of a batch group: window y of the launch is copied from its pinned host mirror (read over PCIe by the kernel itself) into
// its device arena, 16 bytes per lane.  Both ends are 16-byte aligned (arena offsets are multiples of 256).
struct BAUploadArgs { unsigned char* dst[8]; const unsigned char* src[8]; unsigned long long n16[8]; };
__global__ __launch_bounds__(256) void k_ba_upload(BAUploadArgs A) {
    const int w = blockIdx.y;
    const uint4* __restrict__ s = reinterpret_cast<const uint4*>(A.src[w]);
    uint4* __restrict__ d = reinterpret_cast<uint4*>(A.dst[w]);
    const unsigned long long n = A.n16[w];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) d[i] = s[i];
}

struct BAJob {
    const eao_ba_problem* p = nullptr; const volatile uint8_t* stop = nullptr; eao_ba_result* r = nullptr;
    int mode = 0, robust = 1; const eao_ba_planes* pl = nullptr; float* planes_out = nullptr;
    LMContext* c = nullptr; LMTraceHost* tr = nullptr;
    int nPo = 0, nPl = 0, Ept = 0, Epl = 0, nC = 0, nP = 0, E = 0;
    bool hasPl = false, trivial = false, chained = false, pollStop = false, lazy = false;
    int nPairsLong = 0, nPairsSlots = 0;      // map-scale path: launch slots (lpOrder) of the four-wave assembly kernel / of both kernels
    BADev D; BADev* dW = nullptr;
    BALaunch L;
    int curHost = 0;
    SE3* outCams = nullptr; double* outPts = nullptr; double* outPlanes = nullptr; unsigned char* outCls = nullptr;

    void write_records(BADev* dst) const {      // the two records of this window (see BA_WIN)
        dst[0] = D; dst[0].ctl = D.ctl0; dst[0].lm = D.lm0;
        dst[1] = D; dst[1].ctl = D.ctl0 + 8; dst[1].lm = D.lm0 + 8;
    }
    bool batchable() const { return !trivial && chained && L.d.usePairs && L.d.solveTiles && !hasPl && !L.d.bigPath && D.nFree > 0 && D.nL > 0 && mode == 0; }

    // validation, arena, pinned mirror, upload (two copies on `s`), active structure.  No kernel is launched here.
    // deferUpload (batches): NO call into the HIP runtime at all -- the pinned mirror is filled and [upSrc, upSrc + upBytes) is left
    // for the group's leader, which moves every window of its group with ONE launch of k_ba_upload (the copies' enqueue calls
    // serialise inside the runtime: 50 of them were most of a batch's 0.55 ms of set-up, and more host threads made it worse).
    const unsigned char* upSrc = nullptr; unsigned char* upDst = nullptr; size_t upBytes = 0;
    eao_status prepare(hipStream_t s, bool deferUpload = false) {
        eao::Range rg("lm: window set-up + upload");
        EAO_REQUIRE(p && r && r->cam_Tcw && r->points && (p->n_edges == 0 || r->edge_outlier || mode == 1), "null argument");
        EAO_REQUIRE(p->n_cams > 0 && p->n_points >= 0 && p->n_edges >= 0, "bad sizes");
        if (pl && pl->n_planes <= 0) pl = nullptr;
        EAO_REQUIRE(!pl || (mode == 1 && pl->plane_world && planes_out && pl->n_pedges >= 0 && (pl->n_pedges == 0 || (pl->pedge_plane && pl->pedge_cam && pl->pedge_obs))),
                    "bad plane arguments");
        LMContext& c = *this->c;
        eao_status st;
        tr->clear();
        static const bool hostStamps = getenv("EAO_DEBUG_STAMPS") != nullptr;      // host phases of the set-up, in ms on stderr
        const auto hs0 = std::chrono::steady_clock::now();
        double hsT[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto hs_lap = [&](int k) { if (hostStamps) hsT[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count(); };
        nPo = p->n_points; nPl = pl ? pl->n_planes : 0; Ept = p->n_edges; Epl = pl ? pl->n_pedges : 0;
        nC = p->n_cams; nP = nPo + nPl; E = Ept + Epl;          // landmarks = points then planes, edges = point edges then plane edges
        hasPl = nPl > 0;
        const eao_ba_problem* p = this->p; const eao_ba_planes* pl = this->pl;
        const int nPo = this->nPo, Ept = this->Ept;
        auto edge_cam = [=](int e) { return e < Ept ? p->edge_cam[e] : pl->pedge_cam[e - Ept]; };
        auto edge_lm = [=](int e) { return e < Ept ? p->edge_point[e] : nPo + pl->pedge_plane[e - Ept]; };
        r->iters[0] = r->iters[1] = 0; r->aborted = 0; r->chi2[0] = r->chi2[1] = 0;
        if (stop && *stop) {  // src/Optimizer.cc:961-963: nothing is optimised; poses go through the same SE3 round trip
            r->aborted = 1;
            for (int i = 0; i < nC; i++) se3_to_Tcw_f32(se3_from_Tcw_f32(p->cam_Tcw + 16 * i), r->cam_Tcw + 16 * i);
            for (size_t i = 0; i < (size_t)nPo * 3; i++) r->points[i] = p->points[i];
            for (int i = 0; i < nPl; i++) { double c4[4]; plane_from_f32(pl->plane_world + 4 * i, c4); for (int k = 0; k < 4; k++) planes_out[4 * i + k] = (float)c4[k]; }
            if (Ept && r->edge_outlier) std::memset(r->edge_outlier, 0, Ept);
            trivial = true;
            return EAO_OK;
        }
        bool edgesByLandmark = true;      // the edge list is grouped landmark by landmark, ascending (what the adapters and every generator produce): ptEdges is then the identity
        int nFreeIn = 0;
        for (int i = 0; i < nC; i++) nFreeIn += p->cam_fixed[i] ? 0 : 1;
        EAO_REQUIRE(nFreeIn <= kBigMaxFree, "at most %d free keyframes in this build (got %d)", kBigMaxFree, nFreeIn);
        // more free keyframes than the single-workgroup solvers take (or EAO_BA_SOLVER=big, the harness's A/B switch): the
        // map-scale path, dense system in HBM factorised by the whole chip (k_bal_*)
        static const char* solverEnv0 = getenv("EAO_BA_SOLVER");
        // (measured, LocalBundleAdjustment wall time, tools/dbg_ba_sizes.py: the LDS / global-scratch single-workgroup solver with
        //  the slab assembly takes 5.8 ms at 31 free keyframes and 29 ms at 64, the map-scale path 3.6 and 6.9 ms -- so everything
        //  beyond the register-tile solver goes there; that older path was removed in round 5)
        const bool bigPath = nFreeIn > kTileMaxFree || (nFreeIn > 0 && solverEnv0 && !strcmp(solverEnv0, "big"));
        // ---- round 6: the set-up of a map-scale call runs as a SESSION of the host crew (HostCrew: one wake-up, then passes handed over through one polled word):
        //      validation + counts, the observer / camera lists, the pair counts, the problem pack and the pair lists are each a pass over landmark or camera ranges.
        //      EAO_BA_SETUP_THREADS: 1 = the serial walks, n > 1 = a session of n threads on any map (the tests), unset = a session from 20 000 edges on.
        const int envSetupT = getenv("EAO_BA_SETUP_THREADS") ? atoi(getenv("EAO_BA_SETUP_THREADS")) : 0;      // (read per call)
        struct SessionGuard {
            bool open = false;
            ~SessionGuard() { if (open) host_crew().session_end(); }
        } session;
        if (bigPath && !t_inCrew && envSetupT != 1 && (Ept >= 20000 || envSetupT > 1)) {
            const int hw = (int)std::thread::hardware_concurrency();
            const int nT = envSetupT > 1 ? envSetupT : std::max(2, std::min(12, hw / 2));
            session.open = host_crew().session_begin(nT - 1);
        }
        // ... and, in the same pass over the edges, the edge counts per camera and per landmark the active structure starts from
        std::vector<int>& cnt = c.scratch;
        cnt.assign((size_t)nC + nP, 0);
        int* const camCnt = cnt.data(); int* const ptCnt = camCnt + nC;
        // ---- the parallel form of that pass (sessions; no plane edges): chunks of the edge list cut at landmark boundaries.  Every chunk validates its edges, checks that
        //      the landmarks ascend, counts each landmark's edges (a landmark's run belongs to one chunk) and its own edges per camera, looks for a camera that appears
        //      twice in a landmark, and counts what the covisibility structure needs (free observers, pair entries).  Anything unexpected -- an index out of range, a
        //      landmark out of order -- and the serial pass below runs instead (and words the error).
        constexpr int kQ = 48;
        static thread_local std::vector<int> cb, cl, chunkFree, camCntQ, camLastQ, chunkBad, cmE;
        static thread_local std::vector<long long> chunkEnt;
        bool countedInChunks = false;
        if (session.open && Epl == 0 && Ept > 0) {
            const int* const ecam = p->edge_cam; const int* const ept = p->edge_point; const uint8_t* const fixedp = p->cam_fixed;
            cb.assign(kQ + 1, Ept);
            for (int q = 0; q < kQ; q++) {      // chunk q = edges [cb[q], cb[q + 1]); boundaries moved forward to the end of a run of equal landmarks
                int e = (int)((long long)Ept * q / kQ);
                while (e > 0 && e < Ept && ept[e] == ept[e - 1]) e++;
                cb[q] = std::min(e, Ept);
            }
            cb[0] = 0;
            for (int q = 1; q <= kQ; q++) cb[q] = std::max(cb[q], cb[q - 1]);      // (monotone; an empty chunk is harmless)
            chunkFree.assign(kQ, 0); chunkEnt.assign(kQ, 0); chunkBad.assign(2 * kQ, -1);
            camCntQ.assign((size_t)kQ * nC, 0); camLastQ.assign((size_t)kQ * nC, -1);
            {
                int* const cfp = chunkFree.data(); long long* const cep = chunkEnt.data(); int* const ccq = camCntQ.data(); int* const clq = camLastQ.data();
                const int* const cbp = cb.data(); int* const bad = chunkBad.data();
                const int nC_ = nC, nPo_ = nPo;
                host_crew().session_pass(kQ, [=](int q) {
                    int* const cc = ccq + (size_t)q * nC_; int* const last = clq + (size_t)q * nC_;
                    int freeN = 0; long long ent = 0;
                    const int e1 = cbp[q + 1];
                    int prev = cbp[q] > 0 ? ept[cbp[q] - 1] : -1;
                    for (int e = cbp[q]; e < e1;) {
                        const int lmk = ept[e];
                        if (lmk <= prev || lmk >= nPo_) { bad[2 * q] = -2; return; }
                        int m = 0, run = 0;
                        for (; e < e1 && ept[e] == lmk; e++, run++) {
                            const int ec = ecam[e];
                            if ((unsigned)ec >= (unsigned)nC_) { bad[2 * q] = -2; return; }
                            if (last[ec] == lmk && bad[2 * q] == -1) { bad[2 * q] = ec; bad[2 * q + 1] = lmk; }
                            last[ec] = lmk;
                            cc[ec]++;
                            m += fixedp[ec] ? 0 : 1;
                        }
                        __atomic_store_n(&ptCnt[lmk], run, __ATOMIC_RELAXED);      // (a landmark out of order could be written by two chunks: the serial pass then starts over)
                        freeN += m; ent += (long long)m * (m + 1) / 2;
                        prev = lmk;
                    }
                    cfp[q] = freeN; cep[q] = ent;
                });
            }
            countedInChunks = true;
            for (int q = 0; q < kQ; q++) countedInChunks = countedInChunks && chunkBad[2 * q] != -2;
            if (countedInChunks) {
                for (int q = 0; q < kQ; q++)
                    if (chunkBad[2 * q] >= 0) { eao::set_error("two edges join camera %d and point %d", chunkBad[2 * q], chunkBad[2 * q + 1]); return EAO_ERR_INVALID; }
                for (int i = 0; i < nC; i++) {      // a camera's count; per chunk: where the chunk's edges go inside the camera's list
                    int run = 0;
                    for (int q = 0; q < kQ; q++) { const int c0 = camCntQ[(size_t)q * nC + i]; camCntQ[(size_t)q * nC + i] = run; run += c0; }
                    camCnt[i] = run;
                }
            } else std::fill(cnt.begin(), cnt.end(), 0);
        }
        if (!countedInChunks) {
            for (int e = 0, prev = 0; e < Ept; e++) {
                const int ec = p->edge_cam[e], ep = p->edge_point[e];
                EAO_REQUIRE(ec >= 0 && ec < nC && ep >= 0 && ep < nPo, "edge %d out of range", e);
                edgesByLandmark = edgesByLandmark && ep >= prev; prev = ep;
                camCnt[ec]++; ptCnt[ep]++;
            }
            for (int e = 0, prev = 0; e < Epl; e++) {
                const int ec = pl->pedge_cam[e], ep = pl->pedge_plane[e];
                EAO_REQUIRE(ec >= 0 && ec < nC && ep >= 0 && ep < nPl, "plane edge %d out of range", e);
                edgesByLandmark = edgesByLandmark && ep >= prev; prev = ep;
                camCnt[ec]++; ptCnt[nPo + ep]++;
            }
        }
        size_t lpEntries = 0, lpPairsMax = 0;
        // ---- round 5: the covisibility structure of the map-scale path, CAMERA-MAJOR.  For every free camera i1 (ascending) the landmarks it observes in ascending
        //      order, and for each of them its observers i2 >= i1: the pairs (i1, i2) of camera i1 are counted in a counter array of nF entries that stays in the
        //      cache, come out sorted, and their entries are later written into ONE contiguous range per camera -- in ascending landmark order, which is the order the
        //      assembly's fixed-order sums need.  (Rounds 3-5 walked the landmarks and scattered every (pair, landmark) entry through a counter per pair of the
        //      whole nF (nF + 1) / 2 triangle, three times -- once into a byte matrix for the tile structure, once to count, once to fill: 9.8 + 2.4 ms of host time in
        //      front of 26.7 ms of device time on the banded 1000-keyframe map.)  Same arrays as before, bit for bit.
        //      Every landmark's observer list is sorted by camera, so a camera's partners i2 >= i1 in a landmark are the SUFFIX behind its own entry: no test per
        //      observer (it failed half the time and mispredicted).  Both walks -- counting and filling -- are split over the host crew by camera ranges of equal
        //      size taken from a shared counter (a camera's pairs and entries are its own: no two workers write the same word).
        static thread_local std::vector<int> fidx, lmOff, lmCam, lmEdge, cmOff, cmLm, cmU, prA, prB, prStart, cmPairStart, pcur;
        // workers for the two walks: the crew unless this thread is one of its own (a map-scale window inside a batch call), or the map is small
        auto crew_for = [&](size_t work, int nChunks, const std::function<void(int)>& chunk) {
            if (session.open) { host_crew().session_pass(nChunks, chunk); return; }
            const int hw = (int)std::thread::hardware_concurrency();
            const int nT = t_inCrew || envSetupT == 1 || (work < 200000 && envSetupT <= 0) ? 1 : std::max(1, std::min(envSetupT > 0 ? envSetupT : std::min(12, hw / 2), nChunks));
            if (nT == 1) { for (int q = 0; q < nChunks; q++) chunk(q); return; }
            std::atomic<int> next(0);
            auto body = [&]() { for (int q; (q = next.fetch_add(1)) < nChunks;) chunk(q); };
            host_crew().run(nT - 1, [&](int) { body(); }, body);
        };
        // ---- the ORDER, TILE structure and launch SCHEDULE of the map-scale system (GbaPlan, gba.hip; round 5 built the tile structure here in natural keyframe order):
        //      which 64 x 64 tiles of the lower triangle can ever be non-zero -- the tiles a covisible camera pair's 6 x 6 block touches in the elimination order, the
        //      diagonal, the tile row of the right-hand side, and the fill-in of the elimination worked out at tile level (the block form of the symbolic factorisation a
        //      sparse LDL^T starts with, solvers/linear_solver_eigen.h:95-112).  Memory and the launches' grids follow this structure.  The plan is a pure function of
        //      the pair list; the context keeps it while the list's hash stays the same.
        GbaPlan& plan = c.plan;             // (read by the launches of this window, long after this function has returned: the context's)
        int bigT = 0, bigTiles = 0;
        if (bigPath) {
            // free cameras with at least one edge, in ascending order (the numbering the active structure below gives them: camIdx)
            fidx.assign((size_t)nC, -1);
            lmOff.assign((size_t)nP + 1, 0);
            int nFa = 0;
            for (int i = 0; i < nC; i++) if (camCnt[i] && !p->cam_fixed[i]) fidx[i] = nFa++;      // (camCnt: the validation pass)
            // per landmark: its free observers and their edges, in edge order (the order of the active structure's ptEdges)
            // ---- round 6: the observer lists and the camera lists in PARALLEL PASSES over landmark ranges (the serial walks below took 2.3 of the 4.8 ms of host
            //      set-up in front of the 1000-keyframe map's 7.5 ms of device time).  With the edges listed landmark by landmark (what the adapters and every
            //      generator produce; no plane edges) a landmark's edges are contiguous, so a chunk of the edge list cut at landmark boundaries owns its landmarks:
            //      the validation pass above has counted every chunk's free observers, pair entries and edges per camera; the pass here writes the observer lists
            //      (sorted by camera) and files every entry under its camera at the position the chunks before it left -- a camera's list comes out in ascending
            //      landmark order, the same arrays as the serial walk, element for element (tests/test_gpu_lm.py::test_map_scale_set_up_on_the_host_crew), and the
            //      camera's edge list of the active structure (camEdges) with it.
            const bool parallelLists = countedInChunks;
            if (parallelLists) {
                const int Q = kQ;
                const int* const ecam = p->edge_cam; const int* const ept = p->edge_point;
                cl.assign(Q + 1, nP);
                cl[0] = 0;
                for (int q = 1; q < Q; q++) cl[q] = cb[q] < Ept ? ept[cb[q]] : nP;
                // chunk bases; per camera the start of its list and, per chunk, where the chunk's entries go
                static thread_local std::vector<int> chunkBase;
                chunkBase.assign(Q + 1, 0);
                for (int q = 0; q < Q; q++) { chunkBase[q + 1] = chunkBase[q] + chunkFree[q]; lpEntries += (size_t)chunkEnt[q]; }
                EAO_REQUIRE(lpEntries < ((size_t)1 << 31), "covisibility structure too large (%zu pair entries)", lpEntries);
                hs_lap(8);
                const int total = chunkBase[Q];
                lmCam.resize((size_t)total + 1); lmEdge.resize((size_t)total + 1);
                cmOff.assign((size_t)nFa + 1, 0);
                for (int i = 0; i < nC; i++) if (fidx[i] >= 0) cmOff[fidx[i] + 1] = camCnt[i];
                for (int f = 0; f < nFa; f++) cmOff[f + 1] += cmOff[f];
                cmLm.resize(cmOff[nFa]); cmU.resize(cmOff[nFa]); cmE.resize(cmOff[nFa]);
                hs_lap(9);
                {
                    int* const lmOffp = lmOff.data(); int* const lc = lmCam.data(); int* const le = lmEdge.data(); const int* const fi = fidx.data();
                    int* const ccq = camCntQ.data(); const int* const cbp = cb.data(); const int* const clp = cl.data(); const int* const basep = chunkBase.data();
                    const int* const cmOffp = cmOff.data(); int* const cmLmp = cmLm.data(); int* const cmUp = cmU.data(); int* const cmEp = cmE.data();
                    const int nFa_ = nFa;
                    static thread_local std::vector<int> ccF;      // per chunk and free camera: where the chunk's entries go inside the camera's list (camCntQ, renumbered)
                    ccF.resize((size_t)Q * nFa);
                    for (int q = 0; q < Q; q++)
                        for (int i = 0; i < nC; i++) if (fidx[i] >= 0) ccF[(size_t)q * nFa + fidx[i]] = ccq[(size_t)q * nC + i];
                    int* const ccFp = ccF.data();
                    crew_for((size_t)Ept * 8, Q, [=](int q) {
                        int* const cc = ccFp + (size_t)q * nFa_;
                        int at = basep[q], e = cbp[q];
                        const int lEnd = clp[q + 1];
                        for (int lmk = clp[q]; lmk < lEnd; lmk++) {
                            lmOffp[lmk] = at;      // (a landmark without edges: an empty list)
                            const int first = at;
                            for (; e < cbp[q + 1] && ept[e] == lmk; e++) { const int f = fi[ecam[e]]; if (f >= 0) { lc[at] = f; le[at] = e; at++; } }
                            for (int u = first + 1; u < at; u++) {      // observers by camera (insertion sort: a handful per landmark, mostly in order already)
                                const int cf = lc[u], ce = le[u];
                                int v = u;
                                for (; v > first && lc[v - 1] > cf; v--) { lc[v] = lc[v - 1]; le[v] = le[v - 1]; }
                                lc[v] = cf; le[v] = ce;
                            }
                            for (int u = first; u < at; u++) { const int f = lc[u], pos = cmOffp[f] + cc[f]++; cmLmp[pos] = lmk; cmUp[pos] = u; cmEp[pos] = le[u]; }
                        }
                    });
                    lmOff[nP] = total;
                }
                hs_lap(10);
            } else {
                bool byLandmark = true;        // the edges come landmark by landmark (the adapters and every generator list them so): the observer lists are then a filtered copy
                for (int e = 0, prev = 0; e < E; e++) { const int lmk = edge_lm(e); byLandmark = byLandmark && lmk >= prev; prev = lmk; if (fidx[edge_cam(e)] >= 0) lmOff[lmk + 1]++; }
                for (int i = 0; i < nP; i++) {
                    const int m = lmOff[i + 1];
                    lpEntries += (size_t)m * (m + 1) / 2;
                    lmOff[i + 1] += lmOff[i];
                }
                EAO_REQUIRE(lpEntries < ((size_t)1 << 31), "covisibility structure too large (%zu pair entries)", lpEntries);
                hs_lap(8);
                lmCam.resize((size_t)lmOff[nP] + 1); lmEdge.resize((size_t)lmOff[nP] + 1);      // (+ 1: the branch-free append writes one slot ahead)
                cmOff.assign((size_t)nFa + 1, 0);
                if (byLandmark) {
                    // (plain pointers and a branch-free append: the loop is a stream of 2 E loads and at most 2 E stores)
                    int* const lc = lmCam.data(); int* const le = lmEdge.data(); int* const co = cmOff.data() + 1; const int* const fi = fidx.data();
                    const int* const ecam = p->edge_cam; const int* const pcam = pl ? pl->pedge_cam : nullptr;
                    int at = 0;
                    for (int e = 0; e < Ept; e++) { const int f = fi[ecam[e]]; lc[at] = f; le[at] = e; const int ok = f >= 0; at += ok; if (ok) co[f]++; }
                    for (int e = Ept; e < E; e++) { const int f = fi[pcam[e - Ept]]; lc[at] = f; le[at] = e; const int ok = f >= 0; at += ok; if (ok) co[f]++; }
                } else {
                    pcur.assign(lmOff.begin(), lmOff.end() - 1);
                    for (int e = 0; e < E; e++) {
                        const int f = fidx[edge_cam(e)];
                        if (f < 0) continue;
                        const int at = pcur[edge_lm(e)]++;
                        lmCam[at] = f; lmEdge[at] = e; cmOff[f + 1]++;
                    }
                }
                hs_lap(9);
                for (int i = 0; i < nP; i++)           // observers by camera (insertion sort: a handful per landmark, mostly in order already)
                    for (int u = lmOff[i] + 1; u < lmOff[i + 1]; u++) {
                        const int cf = lmCam[u], ce = lmEdge[u];
                        int v = u;
                        for (; v > lmOff[i] && lmCam[v - 1] > cf; v--) { lmCam[v] = lmCam[v - 1]; lmEdge[v] = lmEdge[v - 1]; }
                        lmCam[v] = cf; lmEdge[v] = ce;
                    }
                hs_lap(10);
                // per free camera: its landmarks in ascending order (a counting sort over the landmarks, walked in ascending order), each with the position of the
                // camera's own entry in that landmark's list
                for (int f = 0; f < nFa; f++) cmOff[f + 1] += cmOff[f];
                cmLm.resize(cmOff[nFa]); cmU.resize(cmOff[nFa]);
                {   // (camera ranges on the crew: every worker walks all observer lists and files the entries of ITS cameras -- a camera's list is written by one worker, in landmark order)
                    const int nRanges = std::max(1, std::min(16, nFa / 32));
                    const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const cmOffp = cmOff.data();
                    int* const cmLmp = cmLm.data(); int* const cmUp = cmU.data();
                    const int nPl_ = nP;
                    crew_for(lpEntries, nRanges, [=](int q) {
                        const int f0 = (int)((long long)nFa * q / nRanges), f1 = (int)((long long)nFa * (q + 1) / nRanges);
                        static thread_local std::vector<int> curv;
                        curv.assign(cmOffp + f0, cmOffp + f1);
                        int* const cur = curv.data();
                        for (int i = 0; i < nPl_; i++)
                            for (int u = lmOffp[i]; u < lmOffp[i + 1]; u++) {
                                const int f = lmCamp[u];
                                if (f < f0 || f >= f1) continue;
                                const int at = cur[f - f0]++;
                                cmLmp[at] = i; cmUp[at] = u;
                            }
                    });
                }
            }
            hs_lap(5);
            // the pairs of every camera and their entry counts: chunks of cameras, each into lists of its own, joined in camera order
            constexpr int kChunkCams = 4;
            const int nChunks = (nFa + kChunkCams - 1) / kChunkCams;
            static thread_local std::vector<std::vector<int>> chB, chCnt;
            chB.resize(nChunks); chCnt.resize(nChunks);
            cmPairStart.assign((size_t)nFa + 1, 0);
            {
                int* const pairsOfCam = cmPairStart.data() + 1;
                const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const cmOffp = cmOff.data();
                const int* const cmLmp = cmLm.data(); const int* const cmUp = cmU.data();
                std::vector<int>* const chBp = chB.data(); std::vector<int>* const chCntp = chCnt.data();
                crew_for(lpEntries, nChunks, [=](int q) {
                    // (the per-thread scratch through plain pointers: in a shared library every use of a thread_local object is a call into the TLS runtime,
                    //  and the two loops below made one per observer -- the pass took 0.57 ms where the walk itself needs 0.15)
                    static thread_local std::vector<int> cnt2v, touchedv;
                    cnt2v.assign((size_t)nFa, 0); touchedv.resize((size_t)nFa);
                    int* const cnt2 = cnt2v.data(); int* const touched = touchedv.data();
                    std::vector<int>& oB = chBp[q]; std::vector<int>& oC = chCntp[q];
                    oB.clear(); oC.clear();
                    for (int i1 = q * kChunkCams; i1 < std::min(nFa, (q + 1) * kChunkCams); i1++) {
                        int nt = 0;
                        for (int k = cmOffp[i1]; k < cmOffp[i1 + 1]; k++)
                            for (int u = cmUp[k], ue = lmOffp[cmLmp[k] + 1]; u < ue; u++) { const int i2 = lmCamp[u]; if (cnt2[i2]++ == 0) touched[nt++] = i2; }
                        std::sort(touched, touched + nt);
                        for (int k = 0; k < nt; k++) { const int i2 = touched[k]; oB.push_back(i2); oC.push_back(cnt2[i2]); cnt2[i2] = 0; }
                        pairsOfCam[i1] = nt;
                    }
                });
            }
            hs_lap(6);
            prA.clear(); prB.clear(); prStart.clear();
            int run = 0;
            for (int q = 0; q < nChunks; q++) {
                size_t at = 0;
                for (int i1 = q * kChunkCams; i1 < std::min(nFa, (q + 1) * kChunkCams); i1++) {
                    const int np = cmPairStart[i1 + 1];
                    for (int k = 0; k < np; k++, at++) { prA.push_back(i1); prB.push_back(chB[q][at]); prStart.push_back(run); run += chCnt[q][at]; }
                    cmPairStart[i1 + 1] = (int)prA.size();
                }
            }
            prStart.push_back(run);
            EAO_REQUIRE((size_t)run == lpEntries, "internal: covisibility count mismatch (%d entries counted, %zu expected)", run, lpEntries);
            lpPairsMax = prA.size();
            hs_lap(7);
            {
                const int forceP = getenv("EAO_BA_ND") ? atoi(getenv("EAO_BA_ND")) : 0;      // (read per call: A/B runs and the tests -- 1 = natural order, p > 1 = p segments)
                const uint64_t key = gba_pattern_hash(nFa, prA, prB) ^ ((uint64_t)(unsigned)forceP << 48);
                if (!plan.valid || plan.key != key || plan.nFa != nFa) {
                    gba_build_plan(nFa, prA, prB, forceP, plan);
                    plan.key = key; plan.valid = true;
                    if (hostStamps) fprintf(stderr, "[eao map-scale plan] %d free keyframes (bandwidth %d%s): %d segment(s), %d separator keyframes, %d rows in %d tiles; %zu factorisation launches "
                                            "(natural order: %d), %zu back-substitution launches, %zu work records\n", nFa, plan.bandwidth, plan.rcm ? ", reverse Cuthill-McKee line" : "", plan.P,
                                            plan.nSep, plan.N, plan.bigTiles, plan.launches.size(), plan.chainNatural, plan.sbLaunches.size(), plan.work.size() / 2);
                }
            }
            bigT = plan.T; bigTiles = plan.bigTiles;
            EAO_REQUIRE(plan.work.size() < ((size_t)1 << 28), "tile structure too large (%zu work records)", plan.work.size() / 2);
        }
        hs_lap(0);
        size_t need = 0;
        need += (size_t)E * (3 * 4 + 4 + 4 + 4 + 1 + 4 + 4 + 4 + 1 + 24 + 18 * 8);
        need += (size_t)nP * (3 + 3 + 9 + 3 + 3 + 1 + 1) * 8 + (size_t)nP * 16 + (bigPath ? 64 : (size_t)nP * nC * 4);
        need += (size_t)nC * (2 * sizeof(SE3) + 36 * 8 + 6 * 8 + 6 * 8 + 16);
        need += (size_t)nP * 8 * sizeof(int4) + 256;
        need += (size_t)nP * 9 * 8 + 2048 + 256;      // Tl, ul, the zero block
        need += 128 * 256 + (size_t)nPl * 4 * 8 * 2 + (size_t)Epl * 4 * 8 + 2 * sizeof(BADev) + (size_t)nP + 1024;     // (+ k_ba_backsub's workgroup sums)
        if (bigPath) {
            need += (2 * ((size_t)bigTiles << 12) + 2 * (size_t)plan.N * kBigNB) * 8;
            need += (3 * lpEntries + 5 * lpPairsMax + 72 + 4 * (lmCam.size() + 2) + (size_t)nP + nC + 16) * 4 + (plan.tileMap.size() + plan.rowOf.size() + plan.rowCam.size() + plan.diagList.size() + 8) * 4 + (plan.work.size() + plan.sb.size()) * sizeof(int4) + 4096;
        } else {
            need += 2 * ((size_t)(nC * 6 + 6) * (nC * 6 + 34) + 8) * 8;
            need += (size_t)nC * (nC + 1) / 2 * ((size_t)nP + 64) * (4 + 16);   // landmark lists / item records of the camera pairs
            need += (size_t)nP * 9 * 8 + 1024;
            need += (size_t)nC * (nC + 1) / 2 * 4;
        }
        if ((st = c.bytes.reserve(need))) return st;
        if (bigPath && getenv("EAO_DEBUG_STAMPS"))
            fprintf(stderr, "[eao map-scale arena] %.1f MB for this problem (%d x %d tile grid, %d live tiles = %.1f MB in the two pools, %zu work records), context arena %.1f MB\n",
                    need / 1e6, bigT, bigT, bigTiles, 2.0 * bigTiles * 32768 / 1e6, plan.work.size() / 2, c.bytes.n / 1e6);
        Arena a{c.bytes.p, c.bytes.n};
        std::memset(&D, 0, sizeof(D));
        D.nCams = nC; D.nPts = nP; D.nEdges = E;
        D.cam.fx = p->fx; D.cam.fy = p->fy; D.cam.cx = p->cx; D.cam.cy = p->cy; D.cam.bf = p->bf; D.cam.bf_f = p->bf;
        D.cam.deltaMono = (float)std::sqrt(mode == 1 ? refc::GBA_HUBER2_MONO : refc::LBA_HUBER2_MONO);
        D.cam.deltaStereo = (float)std::sqrt(mode == 1 ? refc::GBA_HUBER2_STEREO : refc::LBA_HUBER2_STEREO);
        // ---- the uploaded part of the arena (problem, initial state, adjacency, zeroed control block, the window record
        //      itself) is mirrored in pinned host memory: filled in place, sent with two copies
        const size_t off0 = a.off;
        float* dobs = a.take<float>((size_t)E * 3); float* dinfo = a.take<float>(E);
        int* decam = a.take<int>(E); int* dept = a.take<int>(E);
        SE3* dcams = a.take<SE3>(nC);
        double* dpts = a.take<double>((size_t)nP * 3);
        unsigned char* dflag = a.take<unsigned char>(E);
        int* dcamIdx = a.take<int>(nC); int* dptIdx = a.take<int>(nP); int* dactCam = a.take<int>(nC); int* dactPt = a.take<int>(nP);
        int* dptStart = a.take<int>(nP + 1); int* dptEdges = a.take<int>(E); int* dcamStart = a.take<int>(nC + 1); int* dcamEdges = a.take<int>(E);
        int* dctl = a.take<int>(16);   // two control blocks: see BADecision
        int* dlpStart = a.take<int>(bigPath ? lpPairsMax + 1 : 1);
        int* dlpPair = a.take<int>(bigPath ? 2 * lpPairsMax : 1);
        int* dlpOrder = a.take<int>(bigPath ? 2 * lpPairsMax + 64 : 1);
        const size_t nObs = bigPath ? lmCam.size() : 1;      // (observer list entries + 1)
        int* dlmOff = a.take<int>(bigPath ? (size_t)nP + 1 : 1); int* dlmCam = a.take<int>(nObs); int* dlmEdge = a.take<int>(nObs);
        int* dcmOff = a.take<int>(bigPath ? cmOff.size() : 1); int* dcmLm = a.take<int>(nObs); int* dcmU = a.take<int>(nObs);
        int* dbigTile = a.take<int>(bigPath ? plan.tileMap.size() : 1);
        int4* dbigWork = a.take<int4>(bigPath ? std::max<size_t>(plan.work.size(), 1) : 1);
        int* dbigRow = a.take<int>(bigPath ? std::max<size_t>(plan.rowOf.size(), 1) : 1);
        int* dbigRowCam = a.take<int>(bigPath ? std::max<size_t>(plan.rowCam.size(), 1) : 1);
        int4* dbigSB = a.take<int4>(bigPath ? std::max<size_t>(plan.sb.size(), 1) : 1);
        int* dbigDiagList = a.take<int>(bigPath ? std::max<size_t>(plan.diagList.size(), 1) : 1);
        double* dpl0 = a.take<double>((size_t)nPl * 4 + 1);
        double* dpmeas = a.take<double>((size_t)Epl * 4 + 1);
        dW = a.take<BADev>(2);
        const size_t off1 = (a.off + 255) & ~(size_t)255;
        // ---- device-only part
        int* dtable = bigPath ? nullptr : a.take<int>((size_t)nP * nC);      // (the map-scale path finds a landmark's edges in its pair lists)
        int* dlpPts = a.take<int>(bigPath ? lpEntries : 1);                  // (filled by k_bal_pair_fill)
        int* dlpE1 = a.take<int>(bigPath ? lpEntries : 1);
        int* dlpE2 = a.take<int>(bigPath ? lpEntries : 1);
        D.slot = a.take<int4>((size_t)std::max(nP, 1) * 8);
        D.camEdgeL = a.take<int>(E);
        const bool pairPath = !bigPath && nFreeIn > 0 && nFreeIn <= kTileMaxFree;
        const int nPairsMax = nFreeIn * (nFreeIn + 1) / 2;
        D.pairCnt = a.take<int>(bigPath ? 1 : std::max(nPairsMax, 1));
        D.pairPts = a.take<int>(pairPath ? (size_t)nPairsMax * std::max(nP, 1) : 1);
        static const bool envNoW = getenv("EAO_BA_WMODE") && !atoi(getenv("EAO_BA_WMODE"));      // (A/B switch: the VALU pair kernels)
        const bool wmode = pairPath && !hasPl && !envNoW;
        D.wmode = wmode ? 1 : 0;
        D.pairItems = a.take<int4>(wmode ? (size_t)nPairsMax * std::max(nP, 1) : 1);
        D.Tl = a.take<double>((size_t)std::max(nP, 1) * 6); D.ul = a.take<double>(((size_t)std::max(nP, 1) + 1) * 3);
        D.cls = a.take<unsigned char>(E);
        SE3* dcamsT = a.take<SE3>(nC);
        double* dptsT = a.take<double>((size_t)nP * 3);
        D.plBuf[0] = dpl0; D.plBuf[1] = a.take<double>((size_t)nPl * 4 + 1); D.pmeas = dpmeas;
        D.nPtsOnly = nPo; D.nEdgesPt = Ept;
        D.deltaPlane = (float)std::sqrt(refc::PLANE_CHI2); D.infoAngle = refc::PLANE_ANGLE_INFO / (1.0 * 1.0); D.infoDist = refc::PLANE_DIST_INFO_ROOT * refc::PLANE_DIST_INFO_ROOT;   // src/Optimizer.cc:203-208
        D.err = a.take<double>((size_t)E * 3);
        D.Hpp = a.take<double>((size_t)nC * 36); D.bp = a.take<double>((size_t)nC * 6);
        D.Hll = a.take<double>((size_t)nP * 9); D.bl = a.take<double>((size_t)nP * 3);
        D.Hpl = a.take<double>(((size_t)E + 1) * 18);      // (+ the zero block of k_ba_schur_pairs_mfma)
        D.sys = a.take<double>(bigPath ? 8 : std::max((size_t)(nFreeIn * 6) * (nFreeIn * 6 + 1), (size_t)tile_geom(std::max(nFreeIn, 1)).nTiles * 256) + 8);
        D.big = a.take<double>(bigPath ? ((size_t)bigTiles << 12) : 8);
        D.bigL = a.take<double>(bigPath ? ((size_t)bigTiles << 12) : 8);
        D.bigTile = dbigTile; D.bigT = bigT; D.bigTiles = bigTiles; D.bigWork = dbigWork; D.bigDense = bigPath && bigTiles == bigT * (bigT + 1) / 2 ? 1 : 0;
        D.bigDiag = a.take<double>(bigPath ? (size_t)plan.N * kBigNB : 8);
        D.bigLinv = a.take<double>(bigPath ? (size_t)plan.N * kBigNB : 8);
        D.bigN = bigPath ? plan.N : 0; D.bigRow = dbigRow; D.bigRowCam = dbigRowCam; D.bigSB = dbigSB; D.bigDiagList = dbigDiagList;
        D.bigFail = a.take<int>(4);
        D.lpStart = dlpStart; D.lpPair = dlpPair; D.lpOrder = dlpOrder; D.lpPts = dlpPts; D.lpE1 = dlpE1; D.lpE2 = dlpE2;
        D.lmOff = dlmOff; D.lmCam = dlmCam; D.lmEdge = dlmEdge; D.cmOff = dcmOff; D.cmLm = dcmLm; D.cmU = dcmU;
        D.xp = a.take<double>((size_t)nC * 6); D.xl = a.take<double>((size_t)nP * 3);
        D.partChi = a.take<double>(nP); D.partScale = a.take<double>(nP);
        D.lm0 = a.take<double>(16);
        D.solveOk = a.take<int>(4);
        D.doneCnt = a.take<int>(4);
        D.wgPart = a.take<double>(2 * (size_t)eao::cdiv(std::max(nP, 1) * 8, 256) + 2);
        long long* ddbg = a.take<long long>(32);
        D.dbg = getenv("EAO_DEBUG_STAMPS") ? ddbg : nullptr;
        EAO_REQUIRE(a.off <= a.cap, "internal: arena overflow");
        D.obs = dobs; D.info = dinfo; D.ecam = decam; D.ept = dept; D.eflag = dflag;
        D.camIdx = dcamIdx; D.ptIdx = dptIdx; D.actCam = dactCam; D.actPt = dactPt;
        D.ptStart = dptStart; D.ptEdges = dptEdges; D.camStart = dcamStart; D.camEdges = dcamEdges; D.table = dtable;
        D.camsBuf[0] = dcams; D.camsBuf[1] = dcamsT; D.ptsBuf[0] = dpts; D.ptsBuf[1] = dptsT;
        D.ctl0 = dctl; D.ctl = dctl; D.lm = D.lm0;
        D.status = c.status;
        if (c.pinCap < off1) {
            if (c.pin) (void)hipHostFree(c.pin);
            c.pin = nullptr; c.pinCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pin, off1 + (off1 >> 2), hipHostMallocDefault));
            c.pinCap = off1 + (off1 >> 2);
        }
        const size_t outBytes = (size_t)nC * sizeof(SE3) + (size_t)nP * 24 + (size_t)nPl * 32 + (((size_t)E + 15) & ~(size_t)15) + 64;
        if (c.pinOutCap < outBytes) {
            if (c.pinOut) (void)hipHostFree(c.pinOut);
            c.pinOut = nullptr; c.pinOutCap = 0;
            EAO_HIP(hipHostMalloc((void**)&c.pinOut, outBytes + (outBytes >> 2), hipHostMallocMapped));
            c.pinOutCap = outBytes + (outBytes >> 2);
        }
        outCams = (SE3*)c.pinOut;
        outPts = (double*)(c.pinOut + (((size_t)nC * sizeof(SE3) + 15) & ~(size_t)15));
        outPlanes = outPts + (size_t)nP * 3;
        outCls = (unsigned char*)(outPlanes + (size_t)nPl * 4);
        D.outCams = outCams; D.outPts = outPts; D.outPlanes = outPlanes; D.outCls = outCls;
        auto hostp = [&](const void* dev) { return c.pin + ((const unsigned char*)dev - a.base); };
        size_t offSplit = off0;
        {
            unsigned char* const hf = (unsigned char*)hostp(dflag);      // edge flags: bit0 stereo, bit2 robust kernel present (bit1 = level 1 is only ever set on the device)
            double* const hp = (double*)hostp(dpts);
            {
                unsigned char* const hobs = hostp(dobs); unsigned char* const hinfo = hostp(dinfo); unsigned char* const hecam = hostp(decam); unsigned char* const hept = hostp(dept);
                const eao_ba_problem* const pp = p;
                const int Ept_ = Ept, nPo_ = nPo; const unsigned char rb = robust ? 4 : 0;
                const int nPack = session.open ? 16 : 1;
                crew_for(0, nPack, [=](int q) {
                    const size_t e0 = (size_t)Ept_ * q / nPack, e1 = (size_t)Ept_ * (q + 1) / nPack;
                    std::memcpy(hobs + e0 * 12, pp->edge_obs + e0 * 3, (e1 - e0) * 12);
                    std::memcpy(hinfo + e0 * 4, pp->edge_inv_sigma2 + e0, (e1 - e0) * 4);
                    std::memcpy(hecam + e0 * 4, pp->edge_cam + e0, (e1 - e0) * 4);
                    std::memcpy(hept + e0 * 4, pp->edge_point + e0, (e1 - e0) * 4);
                    for (size_t e = e0; e < e1; e++) hf[e] = (unsigned char)((!(pp->edge_obs[3 * e + 2] < 0) ? 1 : 0) | rb);
                    for (size_t i = (size_t)nPo_ * 3 * q / nPack, i1 = (size_t)nPo_ * 3 * (q + 1) / nPack; i < i1; i++) hp[i] = pp->points[i];
                });
            }
            if (hasPl) {
                std::memset(hostp(dobs) + (size_t)Ept * 12, 0, (size_t)Epl * 12);
                std::memset(hostp(dinfo) + (size_t)Ept * 4, 0, (size_t)Epl * 4);
                int* hc2 = (int*)hostp(decam); int* hp2 = (int*)hostp(dept);
                for (int e = Ept; e < E; e++) { hc2[e] = edge_cam(e); hp2[e] = edge_lm(e); }
                double* hpl = (double*)hostp(dpl0); double* hpm = (double*)hostp(dpmeas);
                for (int i = 0; i < nPl; i++) plane_from_f32(pl->plane_world + 4 * i, hpl + 4 * i);          // Converter::toPlane3D (:217)
                for (int e = 0; e < Epl; e++) plane_from_f32(pl->pedge_obs + 4 * e, hpm + 4 * e);           // (:239)
            }
            SE3* hc = (SE3*)hostp(dcams);
            for (int i = 0; i < nC; i++) hc[i] = se3_from_Tcw_f32(p->cam_Tcw + 16 * i);
            for (size_t i = (size_t)nPo * 3; i < (size_t)nP * 3; i++) hp[i] = 0;
            for (int e = Ept; e < E; e++) hf[e] = 8 | 4;      // EdgePlane: always a Huber kernel (:246-248)
            std::memset(hostp(dctl), 0, 16 * sizeof(int));
            // The problem itself (observations, indices, initial state, flags) is on its way to the device while the host builds
            // the active structure below; the structure follows in a second copy.
            offSplit = (size_t)((unsigned char*)dcamIdx - a.base) & ~(size_t)255;
            hs_lap(1);
            if (!deferUpload) EAO_HIP(hipMemcpyAsync(a.base + off0, c.pin + off0, offSplit - off0, hipMemcpyHostToDevice, s));
            // ---- active structure: SparseOptimizer::initializeOptimization(level 0) + buildIndexMapping
            int* camIdx = (int*)hostp(dcamIdx); int* ptIdx = (int*)hostp(dptIdx);
            int* actCam = (int*)hostp(dactCam); int* actPt = (int*)hostp(dactPt);
            int* ptStart = (int*)hostp(dptStart); int* ptEdges = (int*)hostp(dptEdges);
            int* camStart = (int*)hostp(dcamStart); int* camEdges = (int*)hostp(dcamEdges);
            // (camCnt / ptCnt: counted with the validation pass above)
            int nF = 0, nL = 0;
            for (int i = 0; i < nC; i++) { camIdx[i] = -1; if (camCnt[i] && !p->cam_fixed[i]) { actCam[nF] = i; camIdx[i] = nF++; } }
            ptStart[0] = 0;
            for (int i = 0; i < nP; i++) { ptIdx[i] = -1; if (ptCnt[i]) { actPt[nL] = i; ptIdx[i] = nL; ptStart[nL + 1] = ptStart[nL] + ptCnt[i]; nL++; } }
            camStart[0] = 0;
            for (int i = 0; i < nF; i++) camStart[i + 1] = camStart[i] + camCnt[actCam[i]];
            for (int i = 0; i < nL; i++) ptCnt[actPt[i]] = ptStart[i];         // counters become fill cursors
            for (int i = 0; i < nF; i++) camCnt[actCam[i]] = camStart[i];
            bool dupChecked = false;
            if (countedInChunks && bigPath) {      // (sessions: the duplicate test rode along with the validation pass, a camera's edges came with its landmark list -- cmE)
                EAO_REQUIRE((int)cmE.size() == camStart[nF], "internal: camera edge lists built for another set of free keyframes");
                const int* const cmEp = cmE.data();
                const int E_ = E, tot = camStart[nF];
                crew_for((size_t)E, 16, [=](int q) {
                    for (int e = (int)((long long)E_ * q / 16), e1 = (int)((long long)E_ * (q + 1) / 16); e < e1; e++) ptEdges[e] = e;
                    const int k0 = (int)((long long)tot * q / 16), k1 = (int)((long long)tot * (q + 1) / 16);
                    std::memcpy(camEdges + k0, cmEp + k0, (size_t)(k1 - k0) * sizeof(int));
                });
                dupChecked = true;
            } else if (edgesByLandmark) {      // (the landmarks' edge lists, concatenated in landmark order, ARE the edge list; the one-edge-per-pair test rides along)
                static thread_local std::vector<int> camLast;
                camLast.assign((size_t)nC, -1);
                for (int e = 0; e < E; e++) {
                    const int cam = edge_cam(e), lmk = edge_lm(e);
                    ptEdges[e] = e;
                    if (camIdx[cam] >= 0) camEdges[camCnt[cam]++] = e;
                    if (camLast[cam] == lmk) { eao::set_error("two edges join camera %d and point %d", cam, lmk); return EAO_ERR_INVALID; }
                    camLast[cam] = lmk;
                }
                dupChecked = true;
            } else {
                for (int e = 0; e < E; e++) {
                    const int cam = edge_cam(e);
                    ptEdges[ptCnt[edge_lm(e)]++] = e;
                    if (camIdx[cam] >= 0) camEdges[camCnt[cam]++] = e;
                }
            }
            // one edge per (camera, point) pair: the device's edge table has one slot per pair
            if (!dupChecked) {
                for (int i = 0; i < nC; i++) camCnt[i] = -1;                        // now: last point seen with this camera
                for (int l = 0; l < nL; l++)
                    for (int k = ptStart[l]; k < ptStart[l + 1]; k++) {
                        const int cam = edge_cam(ptEdges[k]);
                        if (camCnt[cam] == l) { eao::set_error("two edges join camera %d and point %d", cam, actPt[l]); return EAO_ERR_INVALID; }
                        camCnt[cam] = l;
                    }
            }
            D.nFree = nF; D.nL = nL;
            hs_lap(2);
            if (bigPath && nF > 0) {
                // covisibility CSR: for every camera pair (i1 <= i2) sharing a landmark, the landmark blocks in ascending order
                // (counting sort over the landmarks' observer lists; the diagonal pairs carry each camera's own landmarks)
                int* lpStart = (int*)hostp(dlpStart); int* lpPair = (int*)hostp(dlpPair);
                EAO_REQUIRE(plan.valid && plan.nFa == nF && plan.T == bigT, "internal: tile structure built for another system size");
                std::memcpy(hostp(dbigTile), plan.tileMap.data(), plan.tileMap.size() * sizeof(int));
                if (!plan.work.empty()) std::memcpy(hostp(dbigWork), plan.work.data(), plan.work.size() * sizeof(int4));
                std::memcpy(hostp(dbigRow), plan.rowOf.data(), plan.rowOf.size() * sizeof(int));
                std::memcpy(hostp(dbigRowCam), plan.rowCam.data(), plan.rowCam.size() * sizeof(int));
                if (!plan.sb.empty()) std::memcpy(hostp(dbigSB), plan.sb.data(), plan.sb.size() * sizeof(int4));
                if (!plan.diagList.empty()) std::memcpy(hostp(dbigDiagList), plan.diagList.data(), plan.diagList.size() * sizeof(int));
                // (the pairs, their entry counts and every camera's landmark list were worked out above, before the arena was sized)
                EAO_REQUIRE((int)cmPairStart.size() == nF + 1, "internal: covisibility structure built for another set of free keyframes");
                const int nz = (int)prA.size();
                for (int k = 0; k < nz; k++) { lpPair[2 * k] = prA[k]; lpPair[2 * k + 1] = prB[k]; }
                std::memcpy(lpStart, prStart.data(), ((size_t)nz + 1) * sizeof(int));
                // the observer lists travel instead of the pairs' entries (k_bal_pair_fill writes those on the device: see there)
                EAO_REQUIRE((int)lmOff.size() == nP + 1 && (int)cmOff.size() == nF + 1 && lmCam.size() == nObs && cmLm.size() + 1 == nObs, "internal: observer lists built for another problem");
                {
                    int* const hLmOff = (int*)hostp(dlmOff); int* const hLmCam = (int*)hostp(dlmCam); int* const hLmEdge = (int*)hostp(dlmEdge);
                    int* const hCmOff = (int*)hostp(dcmOff); int* const hCmLm = (int*)hostp(dcmLm); int* const hCmU = (int*)hostp(dcmU);
                    const int* const lmOffp = lmOff.data(); const int* const lmCamp = lmCam.data(); const int* const lmEdgep = lmEdge.data();
                    const int* const cmOffp = cmOff.data(); const int* const cmLmp = cmLm.data(); const int* const cmUp = cmU.data();
                    const size_t nO = nObs - 1, nPp = (size_t)nP + 1, nFp = (size_t)nF + 1;
                    const int nCopy = session.open ? 12 : 1;
                    crew_for(0, nCopy, [=](int q) {
                        auto part = [&](int* dst, const int* src, size_t n) { const size_t i0 = n * q / nCopy, i1 = n * (q + 1) / nCopy; std::memcpy(dst + i0, src + i0, (i1 - i0) * sizeof(int)); };
                        part(hLmOff, lmOffp, nPp); part(hLmCam, lmCamp, nO); part(hLmEdge, lmEdgep, nO);
                        part(hCmOff, cmOffp, nFp); part(hCmLm, cmLmp, nO); part(hCmU, cmUp, nO);
                    });
                }
                D.nPairsNZ = nz;
                nPairsLong = 0;
                {   // launch order: long pairs first; and inside each class the pairs are dealt to the eight XCDs by camera range -- workgroup b runs on XCD b % 8, a pair
                    // list is sorted by its first camera, and the pairs of neighbouring cameras share their landmarks: dealt round-robin, every landmark's blocks were
                    // pulled into all eight L2s (the assembly re-reads each block once per pair of its landmark: 570 MB per launch on the banded 1000-keyframe map);
                    // with one contiguous camera range per XCD (equal shares of the entries) they stay in one or two.  A slot of -1 is an idle workgroup.
                    int* lpOrder = (int*)hostp(dlpOrder);
                    const int kLong = getenv("EAO_BA_PAIR_LONG") ? atoi(getenv("EAO_BA_PAIR_LONG")) : kBigPairLong;      // (tests: the four-wave kernel on small maps)
                    static thread_local std::vector<int> cls, grp[8];
                    size_t at = 0;
                    const size_t cap = 2 * (size_t)lpPairsMax + 64;
                    bool fits = true;
                    auto deal = [&](bool longOnes) -> int {
                        cls.clear();
                        long long tot = 0;
                        for (int k = 0; k < nz; k++) if ((lpStart[k + 1] - lpStart[k] > kLong) == longOnes) { cls.push_back(k); tot += lpStart[k + 1] - lpStart[k]; }
                        if (cls.empty()) return 0;
                        for (auto& g8 : grp) g8.clear();
                        long long run = 0;
                        int x = 0, lastCam = -1;
                        for (int k : cls) {      // a new XCD only at a camera boundary, once the running share of the entries is reached
                            const int cam = lpPair[2 * k];
                            if (cam != lastCam && x < 7 && run * 8 >= tot * (x + 1)) x++;
                            lastCam = cam;
                            grp[x].push_back(k);
                            run += lpStart[k + 1] - lpStart[k];
                        }
                        size_t len = 0;
                        for (auto& g8 : grp) len = std::max(len, g8.size());
                        if (at + 8 * len > cap) { fits = false; return 0; }
                        for (size_t sl = 0; sl < len; sl++)
                            for (int q = 0; q < 8; q++) lpOrder[at++] = sl < grp[q].size() ? grp[q][sl] : -1;
                        return (int)(8 * len);
                    };
                    nPairsLong = deal(true);
                    nPairsSlots = nPairsLong + deal(false);
                    if (!fits) {      // (one camera holds most of the pairs: plain order)
                        at = 0;
                        for (int k = 0; k < nz; k++) if (lpStart[k + 1] - lpStart[k] > kLong) lpOrder[at++] = k;
                        nPairsLong = (int)at;
                        for (int k = 0; k < nz; k++) if (lpStart[k + 1] - lpStart[k] <= kLong) lpOrder[at++] = k;
                        nPairsSlots = (int)at;
                    }
                }
            }
        }
        hs_lap(3);
        // ---- launch geometry and solver choice of this window
        const int nF = D.nFree, nL = D.nL;
        BADims& d = L.d;
        d = BADims();
        d.nF = nF; d.nL = nL; d.nP = nP; d.nC = nC; d.E = E; d.nPl = nPl; d.hasPl = hasPl; d.bigPath = bigPath;
        d.usePairs = pairPath && nF > 0 && nL > 0;
        d.wmode = D.wmode != 0 && d.usePairs;
        if (!d.wmode) D.wmode = 0;
        // solver choice: register tiles + MFMA up to kTileMaxFree free keyframes, the map-scale path beyond
        d.solveTiles = !bigPath && nF > 0 && nF <= kTileMaxFree;
        d.tileLds = tile_solver_lds(std::max(nF, 1));
        d.tiles3 = tile_geom(std::max(nF, 1)).nTiles <= 3 * (kTileThreads / 64);
        d.gB = big_geom(std::max(nF, 1));
        if (bigPath) { d.gB.N = plan.N; d.gB.RP = plan.RP; }
        d.nPairsNZ = D.nPairsNZ; d.nPairsLong = nPairsLong; d.nPairsSlots = nPairsSlots; d.big = D.big; d.bigTiles = bigTiles;
        d.plan = bigPath ? &plan : nullptr;
        d.bigCtl0 = D.ctl0;
        d.bigArgs = BigStepArgs{D.big, D.bigL, D.bigDiag, D.bigFail, D.bigWork, D.ctl0, D.dbg, d.gB.N, 0, {}};
        // the window record itself travels with the structure
        write_records((BADev*)hostp(dW));
        if (!deferUpload) EAO_HIP(hipMemcpyAsync(a.base + offSplit, c.pin + offSplit, off1 - offSplit, hipMemcpyHostToDevice, s));
        else { upSrc = c.pin + off0; upDst = a.base + off0; upBytes = (off1 - off0 + 15) & ~(size_t)15; }
        L.W = dW; L.nz = 1; L.s = s; L.seq = c.status->seq;
        c.status->ph[0].touched = c.status->ph[1].touched = 0;
        // map-scale runs (tens of milliseconds) are NOT enqueued speculatively when the caller can abort them: optimize() then
        // submits one LM iteration at a time and reads *stop in between, like g2o's forceStopFlag
        pollStop = bigPath && stop != nullptr;
        // ... and never more than two LM iterations ahead of the device otherwise (`lazy`, see optimize()): a map-scale iteration is ~50 launches, and
        // everything enqueued behind a rejected trial drains as no-ops at the launch rate -- 1.5 of 14 ms on the 200-keyframe benchmark map when all ten
        // iterations were enqueued up front
        lazy = bigPath && !pollStop;
        chained = E > 0 && (nF + nL) > 0 && !pollStop && !lazy;
        hs_lap(4);
        if (hostStamps && bigPath)
            fprintf(stderr, "[eao map-scale host set-up] observer / camera lists %.3f (counts %.3f, observer scatter %.3f, sort %.3f, camera scatter %.3f), pair counts %.3f, pair list %.3f, order + tiles + symbolic elimination + schedule (GbaPlan; cached per pattern) %.3f ms\n",
                    hsT[5], hsT[8], hsT[9] - hsT[8], hsT[10] - hsT[9], hsT[5] - hsT[10], hsT[6] - hsT[5], hsT[7] - hsT[6], hsT[0] - hsT[7]);
        if (hostStamps && bigPath)
            fprintf(stderr, "[eao map-scale host set-up] covisibility lists + pairs + plan %.3f, arena + problem pack %.3f, active structure %.3f, pair CSR + launch order %.3f, records + upload enqueue %.3f ms (cumulative %.3f)\n",
                    hsT[0], hsT[1] - hsT[0], hsT[2] - hsT[1], hsT[3] - hsT[2], hsT[4] - hsT[3], hsT[4]);
        return EAO_OK;
    }

    eao_status wait_status(int want) {
        EAO_HIP(hipStreamSynchronize(L.s));
        if (c->status->seq != want) { eao::set_error("LM status hand-off out of sequence"); return EAO_ERR_INTERNAL; }
        return EAO_OK;
    }
    eao_status set_ctl(int halt, int iters, int nBad) {
        const int v[5] = {halt, curHost, iters, kStRunning, nBad};
        EAO_HIP(hipMemcpyAsync(D.ctl0, v, sizeof(v), hipMemcpyHostToDevice, L.s));
        return EAO_OK;
    }
    // ---- SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg::solve per iteration.
    // Iterations are enqueued in bulk (one trial each, no host round trip); the device finishes clean iterations itself
    // and halts the window on anything else, which the host then replays trial by trial like g2o's do/while.
    // resume: the first bulk segment of this call was already enqueued (and has finished) -- start from its outcome
    eao_status optimize(int phase, int iterations, int* itersDone, double* chiOut, const BAPhase* resume) {
        *itersDone = 0;
        LMContext& c = *this->c;
        eao_status st;
        if (D.nFree + D.nL == 0) return EAO_OK;   // "_ivMap.size() == 0": nothing to optimise
        bool needErrors = true, ok = true;
        double currentChi = 0;
        int nBad = 0, done = 0;
        while (done < iterations && !(stop && *stop && !resume) && ok) {
            if (!resume) {
                // ---- bulk segment: every remaining iteration, one trial each.  The control block is clean at the start of
                //      an optimize() call (zeros from the upload / reset by the outlier pass); after a takeover it is rewritten.
                if (done > 0 && (st = set_ctl(0, done, nBad))) return st;
                if (lazy) {
                    // one iteration per enqueue, the next one as soon as the decision of the one before the last has landed in the pinned status block.  The
                    // poll is a pacing hint only: a stale read enqueues later (or one no-op iteration more), never something else -- what the host acts on is
                    // read after the stream synchronisation below, as in the bulk path.
                    volatile const int* pseq = &c.status->seq;
                    volatile const int* pstat = &c.status->status;
                    static thread_local std::vector<int> seqAt;
                    seqAt.assign((size_t)iterations + 1, 0);
                    for (int enq = done; enq < iterations; enq++) {
                        if (enq - done >= 2) {
                            while (*pseq - seqAt[enq - 2] < 0) { if (hipStreamQuery(L.s) != hipErrorNotReady) break; }
                            if (*pstat != kStRunning) break;
                        }
                        L.bulk(enq, enq + 1, needErrors && enq == done);
                        seqAt[enq] = L.seq;
                    }
                } else L.bulk(done, pollStop ? std::min(iterations, done + 1) : iterations, needErrors);
                needErrors = false;
                EAO_HIP(hipStreamSynchronize(L.s));
            }
            const BAPhase S = resume ? *resume : c.status->ph[phase];
            resume = nullptr;
            for (int k = done; k < S.iters && k < 32; k++) {
                tr->lambda.push_back(c.status->trLambda[32 * phase + k]); tr->chi2.push_back(c.status->trChi[32 * phase + k]);
                tr->trials.push_back(c.status->trTrials[32 * phase + k]);
            }
            tr->linearizations += S.iters - done;
            done = S.iters; nBad = S.nBad; curHost = S.cur; currentChi = S.chi;
            if (S.status == kStEmpty) { done = -1; break; }
            if (S.status == kStTerminate) { ok = false; break; }
            if (S.status == kStRunning && done < iterations && pollStop) continue;   // next iteration (after a look at *stop)
            if (S.status != kStTakeover) break;           // all requested iterations done
            // ---- host takeover of iteration `done`: its first trial was rejected (or rho == 0 / NaN)
            tr->linearizations++;
            const double iniChi = S.chi;
            double rho = S.rho;
            int qmax = 1;
            bool accepted = S.accepted != 0;
            while (rho < 0 && qmax < refc::LM_MAX_TRIALS && !(stop && *stop)) {
                if ((st = set_ctl(0, done, nBad))) return st;
                L.relinearize();
                L.trial(0, 0, false, true);
                if ((st = wait_status(L.seq))) return st;
                rho = c.status->rho; accepted = c.status->accepted != 0; curHost = c.status->cur;
                if (accepted) currentChi = c.status->chi;
                qmax++;
            }
            needErrors = !accepted;             // pop(): residuals belong to the rejected state
            tr->lambda.push_back(c.status->lambda); tr->chi2.push_back(currentChi); tr->trials.push_back(qmax);
            done++;
            if (qmax == refc::LM_MAX_TRIALS || rho == 0) { ok = false; break; }
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) ok = false;
        }
        *itersDone = done;
        *chiOut = currentChi;
        return EAO_OK;
    }

    // After the chained enqueue (own or as part of a batch) has finished: takeovers, results.  The abort flag is read before
    // and after: a clean window takes less time than one g2o iteration on the CPU.
    eao_status complete() {
        eao::Range rg("lm: takeovers + results");
        LMContext& c = *this->c;
        eao_status st;
        hipStream_t s = L.s;
        const BAPhase A = c.status->ph[0], B = c.status->ph[1];
        if ((st = optimize(0, p->its_first, &r->iters[0], &r->chi2[0], chained ? &A : nullptr))) return st;
        const bool firstClean = chained && A.status != kStTakeover;
        const bool doMore = mode == 0 && (firstClean || !(stop && *stop));
        bool redo = !chained || (mode == 1 && !firstClean);
        if (doMore && E) {
            // outlier pass (src/Optimizer.cc:978-1008): chi2 of the residual each edge last computed + depth test.  g2o's
            // initializeOptimization(0) would now drop the level-1 edges (and vertices left without edges) from the active
            // set; here they stay in the lists with zero weight, which leaves every sum -- and a vertex without edges --
            // unchanged, and saves the host round trip of rebuilding and re-uploading the structure.
            if (firstClean) {
                if (B.status == kStTakeover) redo = true;
                if ((st = optimize(1, p->its_second, &r->iters[1], &r->chi2[1], &B))) return st;
            } else {
                redo = true;
                if ((st = set_ctl(0, 0, 0))) return st;            // the frozen window left "takeover" in the control block
                L.classify();
                if ((st = optimize(1, p->its_second, &r->iters[1], &r->chi2[1], nullptr))) return st;
            }
        }
        if (redo) {
            L.finish();
            EAO_HIP(hipStreamSynchronize(s));
        }
        EAO_HIP(hipGetLastError());
        if (D.dbg) {
            long long stt[32];
            EAO_HIP(hipMemcpy(stt, D.dbg, sizeof(stt), hipMemcpyDeviceToHost));
            fprintf(stderr, "[eao pair stamps] workgroup 0 (a diagonal pair): loads + accumulation %lld, block sum of 42 values %lld shader-cycles; linearisation: landmark workgroup 0 %lld, camera workgroup 0 %lld\n",
                    stt[13] >> 20, stt[13] & 0xFFFFF, stt[14], stt[15]);
            if (L.d.bigPath) fprintf(stderr, "[eao bal_backsolve stamps] super-block 1, workgroup 0: loads issued %lld, the 256-column triangle (8 blocks) %lld, removal from the columns to the left %lld shader-cycles\n",
                                     stt[25] - stt[24], stt[26] - stt[25], stt[27] - stt[26]);
            if (L.d.bigPath) fprintf(stderr, "[eao bal_step stamps] look-ahead workgroup of panel 2: prologue + loads + barrier %lld, row solves %lld, update %lld, tile to LDS and rows back %lld, 32 x 32 LDL^T %lld, store %lld shader-cycles\n",
                                     stt[17] - stt[16], stt[18] - stt[17], stt[19] - stt[18], stt[20] - stt[19], stt[21] - stt[20], stt[22] - stt[21]);
            fprintf(stderr, "[eao solve stamps] assemble %lld factor %lld (panel %lld trailing %lld / %lld) backsub %lld tail %lld shader-cycles; wall(100MHz) %lld %lld %lld %lld\n",
                    stt[2] - stt[0], stt[4] - stt[2], stt[10], stt[11], stt[12], stt[6] - stt[4], stt[8] - stt[6], stt[3] - stt[1], stt[5] - stt[3], stt[7] - stt[5], stt[9] - stt[7]);
        }
        for (int i = 0; i < nC; i++) se3_to_Tcw_f32(outCams[i], r->cam_Tcw + 16 * i);
        for (size_t i = 0; i < (size_t)nPo * 3; i++) r->points[i] = (float)outPts[i];
        for (size_t i = 0; i < (size_t)nPl * 4; i++) planes_out[i] = (float)outPlanes[i];           // Converter::toCvMat(Plane3D)
        if (Ept && r->edge_outlier) {
            if (mode == 0) std::memcpy(r->edge_outlier, outCls, Ept);
            else std::memset(r->edge_outlier, 0, Ept);
        }
        return EAO_OK;
    }
};

}  // namespace

static eao_status ba_run(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r, int mode, int robust,
                         const eao_ba_planes* pl = nullptr, float* planes_out = nullptr) {
    LMContext& c = g_ctx;
    eao_status st = ctx_init(c, true, mode == 0 ? eao::StreamClass::Background : eao::StreamClass::Bulk);
    if (st) return st;
    BAJob j;
    j.p = p; j.stop = stop; j.r = r; j.mode = mode; j.robust = robust; j.pl = pl; j.planes_out = planes_out; j.c = &c; j.tr = &g_trace;
    static const bool hostStamps = getenv("EAO_DEBUG_STAMPS") != nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
    EAO_HIP(hipEventRecord(c.ev0, c.stream));
    if ((st = j.prepare(c.stream))) return st;
    if (j.trivial) return EAO_OK;
    const double tPrep = ms();
    if ((st = j.L.attributes())) return st;
    j.L.setup();
    static const bool wallStamps = getenv("EAO_BA_WALL_STAMPS") != nullptr;
    static const bool spin = getenv("EAO_BA_SPIN") && atoi(getenv("EAO_BA_SPIN"));
    double tEnq = 0;
    if (j.chained) {
        j.L.chain(mode, p->its_first, p->its_second);
        tEnq = ms();
        if (spin) while (hipStreamQuery(c.stream) == hipErrorNotReady) {}
        EAO_HIP(hipStreamSynchronize(c.stream));
    }
    const double tSetup = ms();
    if ((st = j.complete())) return st;
    EAO_HIP(hipEventRecord(c.ev1, c.stream));
    EAO_HIP(hipStreamSynchronize(c.stream));
    EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, c.ev0, c.ev1));
    if (wallStamps && !j.L.d.bigPath) fprintf(stderr, "[eao window wall] prepare %.3f, enqueue %.3f, wait %.3f, takeovers + results %.3f ms; whole call %.3f ms\n", tPrep, tEnq - tPrep, tSetup - tEnq, ms() - tSetup, ms());
    if (hostStamps && j.L.d.bigPath) fprintf(stderr, "[eao map-scale wall] prepare %.3f, set-up launches (+ chain) %.3f, iterations + results %.3f ms; whole call %.3f ms\n", tPrep, tSetup - tPrep, ms() - tSetup, ms());
    return EAO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// LocalBundleAdjustment of MANY independent windows (the batched-sequence configuration: 25 windows of 20 keyframes): the
// window is the z dimension of every launch.  Each window owns a context (arena, pinned mirrors, status block); the
// host-side set-up of the windows runs on a few host threads; ONE chained enqueue serves them all; a window whose LM
// rejected a trial freezes by itself (its halt flag) and is finished by the host afterwards exactly like a single call.
namespace {
struct BABatchPool {
    std::vector<std::unique_ptr<LMContext>> ctx;
    std::vector<LMTraceHost> trace;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipStream_t> side;              // streams of the window groups beyond the first
    std::vector<hipEvent_t> sideDone;
    BADev* hW = nullptr; size_t hWCap = 0;      // pinned mirror of the window array
    eao::DevBuf<BADev> dW;
    ~BABatchPool() {
        if (hW) (void)hipHostFree(hW);
        for (hipEvent_t e : sideDone) (void)hipEventDestroy(e);
        for (hipStream_t q : side) (void)hipStreamDestroy(q);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
thread_local BABatchPool g_batch;
constexpr int kBatchGroups = 4, kBatchGroupMin = 4;     // default number of window groups / fewest windows worth a group

}  // namespace

extern "C" {

eao_status eao_local_ba(const eao_ba_problem* p, const volatile uint8_t* stop, eao_ba_result* r) { return ba_run(p, stop, r, 0, 1); }

eao_status eao_local_ba_batch(const eao_ba_problem* problems, int32_t n, const volatile uint8_t* stop, eao_ba_result* results) {
    EAO_REQUIRE(n >= 0 && (n == 0 || (problems && results)), "null argument");
    if (n == 0) return EAO_OK;
    eao_status st = eao::require_device();
    if (st) return st;
    BABatchPool& B = g_batch;
    if (!B.stream) {
        EAO_HIP(eao::create_stream(&B.stream, eao::StreamClass::Background));
        EAO_HIP(hipEventCreate(&B.ev0));
        EAO_HIP(hipEventCreate(&B.ev1));
    }
    while ((int)B.ctx.size() < n) B.ctx.emplace_back(new LMContext());
    if ((int)B.trace.size() < n) B.trace.resize(n);
    for (int w = 0; w < n; w++)
        if ((st = ctx_init(*B.ctx[w], false, eao::StreamClass::Background))) return st;
    if (B.hWCap < (size_t)n * 2) {
        if (B.hW) (void)hipHostFree(B.hW);
        B.hW = nullptr; B.hWCap = 0;
        EAO_HIP(hipHostMalloc((void**)&B.hW, (size_t)n * 2 * sizeof(BADev), hipHostMallocDefault));
        B.hWCap = (size_t)n * 2;
    }
    if ((st = B.dW.reserve((size_t)n * 2))) return st;
    g_trace.clear();
    std::vector<BAJob> jobs(n);
    for (int w = 0; w < n; w++) {
        BAJob& j = jobs[w];
        j.p = &problems[w]; j.stop = stop; j.r = &results[w]; j.mode = 0; j.robust = 1; j.c = B.ctx[w].get(); j.tr = &B.trace[w];
        j.c->status->seq = 0;        // every window of the batch sees the same hand-off sequence numbers
    }
    static const bool envTiming = getenv("EAO_BA_BATCH_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    EAO_HIP(hipEventRecord(B.ev0, B.stream));
    // ---- The windows are dealt to G groups (contiguous ranges).  A group is a chain of its own on its own stream: its
    //      host-side set-up (validation, pinned mirror, active structure, two uploads per window: ~0.1 ms of memcpy and
    //      counting each, on a few host threads), then ONE batched enqueue for its windows.  The first group's kernels start
    //      while the others are still being set up, and a group's one-workgroup-per-window solver (a tenth of the chip) and
    //      the tails of its other launches overlap the other groups' wide kernels.  More than four streams share hardware
    //      queues on this runtime and serialise (measured: 5+ groups are 40 % slower than one).
    static const int envThreads = getenv("EAO_BA_BATCH_THREADS") ? atoi(getenv("EAO_BA_BATCH_THREADS")) : 0;
    static const int envGroups = getenv("EAO_BA_BATCH_GROUPS") ? atoi(getenv("EAO_BA_BATCH_GROUPS")) : 0;
    const int hw = (int)std::thread::hardware_concurrency();
    // (set-up threads make no HIP call any more -- packing and counting only -- so they scale with the host's cores)
    const int nThreads = std::max(1, std::min(n, envThreads > 0 ? envThreads : std::min(16, std::max(1, hw / 2))));
    // Groups hold WHOLE ROWS of eight windows where the batch has them (BA_WIN pins a window to an XCD row by row, and deals an incomplete row over all eight):
    // 25 windows as 8 + 8 + 9 load every XCD with 3.125 windows, as 6 + 6 + 6 + 7 the fullest one with 3.5 (2.6 against 2.8 ms per call).
    const int fullRows = n / 8, rest = n - 8 * fullRows;
    const bool rowGroups = fullRows >= 1 && !getenv("EAO_BA_BATCH_EVEN");      // (A/B switch: the even split of rounds 2-3)
    const bool restGroup = rowGroups && rest >= kBatchGroupMin;                // an incomplete row large enough to be a group of its own (else it joins the last group)
    // (two groups while a frame-rate caller is alive in the process: common.h, note_latency_call)
    const int gWant = std::min(envGroups > 0 ? envGroups : (eao::latency_caller_alive() ? 2 : kBatchGroups), nThreads);
    const int G = std::max(1, rowGroups ? std::min(gWant, fullRows + (restGroup ? 1 : 0)) : std::min(gWant, n / kBatchGroupMin));
    std::vector<int> gStart(G + 1, n);
    for (int g = 0; g < G; g++) {
        if (!rowGroups) gStart[g] = (int)((long long)n * g / G);
        else if (restGroup && G > 1) gStart[g] = g == G - 1 ? 8 * fullRows : 8 * (int)((long long)fullRows * g / (G - 1));
        else gStart[g] = 8 * (int)((long long)fullRows * g / G);
    }
    while ((int)B.side.size() < G - 1) {
        hipStream_t q; hipEvent_t e;
        EAO_HIP(eao::create_stream(&q, eao::StreamClass::Background));
        EAO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        B.side.push_back(q); B.sideDone.push_back(e);
    }
    std::vector<eao_status> stw(n, EAO_OK), stg(G, EAO_OK);
    std::vector<std::string> errw(n), errg(G);
    std::vector<BALaunch> LG(G);
    std::vector<int> groupOf(n, -1), nInGroup(G, 0);
    std::vector<char> batched(n, 0);
    std::vector<double> msPrep(G, 0.0), msEnq(G, 0.0), msDone(G, 0.0);
    int dev = 0;
    EAO_HIP(hipGetDevice(&dev));
    // set-up: nThreads workers take the windows in order (window w's uploads go to its group's stream), so the first
    // group is complete after one round; each group's leader then enqueues its chain while the workers carry on
    std::vector<std::atomic<int>> prepared(G);
    for (auto& a : prepared) a.store(0);
    auto groupOfWindow = [&](int w) { int g = 0; while (gStart[g + 1] <= w) g++; return g; };
    auto streamOf = [&](int g) { return g == 0 ? B.stream : B.side[g - 1]; };
    // results: when a group's stream has drained, the set-up workers (idle since the first tenth of the call) take its windows' results out in parallel -- the last
    // group's nine windows were ~0.1 ms of conversions on one thread at the very end of the call
    const bool crewDone = !(nThreads == 1 && G == 1);
    std::mutex doneMu;
    std::condition_variable doneCv;
    std::vector<int> gState(G, 0);      // 0: running, 1: drained, 2: failed (nothing to take out)
    auto finish_window = [&](int w, int g) {
        BAJob& j = jobs[w];
        if (j.trivial || stw[w]) return;
        if (batched[w]) j.L.seq = LG[g].seq;
        stw[w] = j.complete();
        if (stw[w]) errw[w] = eao_last_error();
    };
    auto worker = [&](int t) {
        (void)hipSetDevice(dev);
        for (int w = t; w < n; w += nThreads) {
            const int g = groupOfWindow(w);
            stw[w] = jobs[w].prepare(streamOf(g), true);
            if (stw[w]) errw[w] = eao_last_error();
            prepared[g].fetch_add(1, std::memory_order_release);
        }
        if (!crewDone) return;
        for (int w = t; w < n; w += nThreads) {
            const int g = groupOfWindow(w);
            int state;
            { std::unique_lock<std::mutex> lk(doneMu); doneCv.wait(lk, [&] { return gState[g] != 0; }); state = gState[g]; }
            if (state == 1) finish_window(w, g);
        }
    };
    auto groupWork = [&](int g) {
        (void)hipSetDevice(dev);
        struct Announce {      // whatever way this group ends, the workers waiting for it are told
            std::mutex& mu; std::condition_variable& cv; int& state; int value = 2;
            ~Announce() { { std::lock_guard<std::mutex> lk(mu); state = value; } cv.notify_all(); }
        } announce{doneMu, doneCv, gState[g]};
        const int w0 = gStart[g], w1 = gStart[g + 1];
        hipStream_t sg = streamOf(g);
        while (prepared[g].load(std::memory_order_acquire) < w1 - w0) std::this_thread::yield();    // (the workers above)
        msPrep[g] = since(tp0);
        for (int w = w0; w < w1; w++)
            if (stw[w]) return;
        // the windows that share the batched enqueue (tile-solver path, something to optimise); the others -- windows beyond
        // 30 free keyframes, empty ones -- follow one by one on the same stream
        {   // this group's uploads: one launch per eight windows (see BAJob::prepare)
            BAUploadArgs U;
            int k = 0;
            auto flush = [&]() {
                if (k) hipLaunchKernelGGL(k_ba_upload, dim3(48, k), dim3(256), 0, sg, U);
                k = 0;
            };
            for (int w = w0; w < w1; w++) {
                if (jobs[w].trivial || !jobs[w].upBytes) continue;
                U.dst[k] = jobs[w].upDst; U.src[k] = jobs[w].upSrc; U.n16[k] = jobs[w].upBytes / 16;
                if (++k == 8) flush();
            }
            flush();
        }
        BALaunch& L = LG[g];
        L.s = sg; L.W = B.dW.p + 2 * w0; L.seq = 0; L.rot = w0 & 7;
        int first = -1, cnt = 0;
        for (int w = w0; w < w1; w++) {
            BAJob& j = jobs[w];
            if (!j.batchable()) continue;
            if (first >= 0 && (j.p->its_first != jobs[first].p->its_first || j.p->its_second != jobs[first].p->its_second)) continue;
            if (first < 0) { first = w; L.d = j.L.d; } else L.d.merge(j.L.d);
            j.write_records(B.hW + 2 * (w0 + cnt));
            cnt++;
            groupOf[w] = g; batched[w] = 1;
        }
        nInGroup[g] = cnt;
        auto fail = [&](eao_status e) { stg[g] = e; errg[g] = eao_last_error(); };
        if (cnt) {
            L.nz = cnt;
            if (hipMemcpyAsync(B.dW.p + 2 * w0, B.hW + 2 * w0, (size_t)cnt * 2 * sizeof(BADev), hipMemcpyHostToDevice, sg) != hipSuccess) {
                eao::set_error("hipMemcpyAsync of the window records failed");
                return fail(EAO_ERR_NO_DEVICE);
            }
            eao_status e = L.attributes();
            if (e) return fail(e);
            L.setup();
            L.chain(0, jobs[first].p->its_first, jobs[first].p->its_second);
        }
        for (int w = w0; w < w1; w++) {
            BAJob& j = jobs[w];
            if (batched[w] || j.trivial) continue;
            eao_status e = j.L.attributes();
            if (e) return fail(e);
            j.L.setup();
            if (j.chained) j.L.chain(0, j.p->its_first, j.p->its_second);
        }
        msEnq[g] = since(tp0);
        // results (and, for a window whose device-side run handed over to the host, the rest of its run) as soon as THIS
        // group's stream has drained: only the last group's ~10 us per window are not hidden behind the other groups' kernels
        if (g > 0 && hipEventRecord(B.sideDone[g - 1], sg) != hipSuccess) { eao::set_error("hipEventRecord failed"); return fail(EAO_ERR_NO_DEVICE); }
        if (hipStreamSynchronize(sg) != hipSuccess) { eao::set_error("hipStreamSynchronize: %s", hipGetErrorString(hipGetLastError())); return fail(EAO_ERR_NO_DEVICE); }
        msDone[g] = since(tp0);
        announce.value = 1;
        if (!crewDone)
            for (int w = w0; w < w1; w++) finish_window(w, g);
    };
    if (nThreads == 1 && G == 1) { worker(0); groupWork(0); }          // (a batch of one: no thread is involved)
    else host_crew().run(nThreads + G - 1, [&](int i) { if (i < nThreads) worker(i); else groupWork(i - nThreads + 1); }, [&] { groupWork(0); });
    bool failed = false;
    for (int w = 0; w < n; w++) failed = failed || stw[w];
    for (int g = 0; g < G; g++) failed = failed || stg[g];
    if (failed)
        for (int g = 1; g < G; g++) (void)hipStreamSynchronize(B.side[g - 1]);
    else
        for (int g = 1; g < G; g++) EAO_HIP(hipStreamWaitEvent(B.stream, B.sideDone[g - 1], 0));   // (device time of the call: ev0 .. ev1)
    for (int w = 0; w < n; w++)
        if (stw[w]) { eao::set_error("window %d: %s", w, errw[w].c_str()); (void)hipStreamSynchronize(B.stream); return stw[w]; }
    for (int g = 0; g < G; g++)
        if (stg[g]) { eao::set_error("%s", errg[g].c_str()); (void)hipStreamSynchronize(B.stream); return stg[g]; }
    int nBatched = 0;
    for (int g = 0; g < G; g++) nBatched += nInGroup[g];
    const double msPrepare = *std::max_element(msPrep.begin(), msPrep.end());
    const double msEnqueue = *std::max_element(msEnq.begin(), msEnq.end()), msSync = *std::max_element(msDone.begin(), msDone.end());
    for (int w = 0; w < n; w++)
        if (!jobs[w].trivial) g_trace.linearizations += jobs[w].tr->linearizations;
    EAO_HIP(hipEventRecord(B.ev1, B.stream));
    EAO_HIP(hipStreamSynchronize(B.stream));
    EAO_HIP(hipEventElapsedTime(&g_trace.deviceMs, B.ev0, B.ev1));
    if (envTiming)
        fprintf(stderr, "[eao_local_ba_batch] %d windows (%d batched, %d host threads): set-up + uploads enqueued %.3f ms, launches enqueued %.3f, device done %.3f, results out %.3f\n",
                n, nBatched, nThreads, msPrepare, msEnqueue, msSync, since(tp0));
    return EAO_OK;
}

eao_status eao_bundle_adjustment(const eao_ba_problem* p, int32_t robust, const volatile uint8_t* stop, eao_ba_result* r) {
    return ba_run(p, stop, r, 1, robust != 0);
}

eao_status eao_bundle_adjustment_planes(const eao_ba_problem* p, const eao_ba_planes* planes, int32_t robust, const volatile uint8_t* stop,
                                        eao_ba_result* r, float* planes_out) {
    return ba_run(p, stop, r, 1, robust != 0, planes, planes_out);
}

eao_status eao_last_lm_trace(double* lambda, double* chi2, int32_t* trials, int32_t cap, int32_t* n) {
    EAO_REQUIRE(n, "null argument");
    const int m = std::min((int)g_trace.lambda.size(), cap);
    for (int i = 0; i < m; i++) {
        if (lambda) lambda[i] = g_trace.lambda[i];
        if (chi2) chi2[i] = g_trace.chi2[i];
        if (trials) trials[i] = g_trace.trials[i];
    }
    *n = m;
    return EAO_OK;
}

eao_status eao_last_lm_timing(float* device_ms, int32_t* linearizations) {
    if (device_ms) *device_ms = g_trace.deviceMs;
    if (linearizations) *linearizations = g_trace.linearizations;
    return EAO_OK;
}

}  // extern "C"
