// lm_internal.h -- what the translation units of the Levenberg-Marquardt engine share (round 6: csrc/lm.hip, 5 236 lines, became pose.hip / lba.hip / gba.hip /
// lm_host.hip over this header -- a pure move, tools/ab_pose_bits.py and tools/ab_gba_bits.py report bit-identical results):
//   pose.hip     Optimizer::PoseOptimization (reference src/Optimizer.cc:325-673): the single-workgroup kernels, eao_pose_optimization(_batch), the tracker chain's hook
//   lba.hip      Optimizer::LocalBundleAdjustment (src/Optimizer.cc:675-1138): linearisation, pair assembly, register-tile solver, back substitution, the LM decision,
//                and BALaunch -- every launch of the engine, the window as grid.z
//   gba.hip      the map-scale path of Optimizer::BundleAdjustment (src/Optimizer.cc:47-323; more than 30 free keyframes): block-sparse tiles, panel LDL^T, back substitution
//   lm_host.hip  host side: per-thread contexts, BAJob (validation, arena, active structure, host-stepped trials), the batch call and its crew, the C-ABI entry points
// The arithmetic follows the reference's vendored g2o (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189, core/block_solver.hpp:354-604,
// types/types_six_dof_expmap.cpp, types/se3quat.h); g2o's object graph is not reproduced.  TYPES live in eao::lm (one definition for all units); the device helper
// FUNCTIONS are internal to each unit (anonymous namespace, all inline) -- the library is built without relocatable device code.
#pragma once
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "common.h"

namespace eao {
namespace lm {

struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };
struct Cam { double fx, fy, cx, cy, bf; float bf_f; double deltaMono, deltaStereo; };
constexpr int kPoseMaxPlanes = 32;

// A pointer member of a record that kernels read FROM MEMORY (the window records of the batched LM kernels).  A plain `T*`
// loaded from memory is a generic pointer to the compiler, and every access through it becomes a FLAT instruction: it
// takes the LDS path's counters as well, so the waits behind it are `vmcnt(0) lgkmcnt(0)` instead of counted ones, and it
// cannot use the scalar-base addressing mode.  On the device the member is declared in the global address space (same
// size and layout); converting it to `T*` is then a global -> generic cast the optimiser sees through, and the kernels'
// accesses become GLOBAL instructions with no change at their use sites.  Only device / pinned buffers are ever stored here.
template <typename T> struct GP {
#ifdef __HIP_DEVICE_COMPILE__
    __attribute__((address_space(1))) T* p;
#else
    T* p;
#endif
    __host__ __device__ __forceinline__ operator T*() const { return (T*)p; }
    __host__ __device__ __forceinline__ GP& operator=(T* q) { p = (decltype(p))q; return *this; }
};

struct BADev {
    int nCams, nPts, nEdges, nFree, nL;   // nFree / nL: active free cameras / active points of the current pass
    Cam cam;
    // problem (device)
    GP<const float> obs;       // E*3 as handed over (promoted to double where used, exactly like Converter / Eigen)
    GP<const float> info;      // E
    GP<const int> ecam;        // E
    GP<const int> ept;         // E
    GP<unsigned char> eflag;   // bit0 stereo, bit1 inactive (level 1: set on the device by the outlier pass), bit2 robust
    GP<const int> camIdx;      // nCams -> free block index or -1
    GP<const int> ptIdx;       // nPts  -> landmark block index or -1
    GP<const int> actCam;      // nFree -> camera
    GP<const int> actPt;       // nL    -> point
    // adjacency of the edges that were active when the window was set up; kernels skip edges whose bit1 was set since
    GP<const int> ptStart;     // nL+1   CSR by landmark block: all active edges of the point, insertion order
    GP<const int> ptEdges;
    GP<const int> camStart;    // nFree+1 CSR by free camera block
    GP<const int> camEdges;
    GP<int> camEdgeL;          // landmark block of each camEdges entry (resolved by k_ba_prepare)
    GP<int> pairCnt;           // per camera pair (i1 <= i2): number of landmarks both observe ...
    GP<int> pairPts;           // ... and their landmark blocks, ascending, nL slots per pair (k_ba_pairs)
    GP<int> table;             // nL * nFree: edge id of (point, free camera) or -1 (built and maintained on the device)
    GP<int4> slot;             // nL * 8: {edge, camera, free-camera index, point | more-than-8-edges << 31} of the landmark's k-th edge, edge = -1
                            // beyond its last one (k_ba_prepare): what the eight lanes of a landmark would otherwise chase through
                            // ptStart -> ptEdges -> ecam -> camIdx, four dependent round trips at the head of every launch
    // state: two buffers; ctl[1] says which one holds the current estimate, the other receives the trial
    GP<SE3> camsBuf[2];
    GP<double> ptsBuf[2];
    GP<int> ctl;               // [0] halt  [1] current buffer  [2] iterations done  [3] status  [4] nBad
    GP<double> err;            // E*3, last computed
    // system
    GP<double> Hpp;            // nFree*36
    GP<double> bp;             // nFree*6
    GP<double> Hll;            // nL*9
    GP<double> bl;             // nL*3
    GP<double> Hpl;            // E*18 (pose row block 6x3) for active edges with a free camera.  wmode = 1: the block is stored PRE-SCALED,
                               // W = Hpl C^-T with C C^T = Hll + lambda I of its landmark (see ba_chol3), so that Hpl (Hll + lambda I)^-1 Hpl'^T = W W'^T
    int wmode;                 // 1: windows on the matrix-core pair assembly (k_ba_schur_pairs_mfma): W / Tl / ul instead of Hpl / dinv3
    GP<double> Tl;             // nL*6: T = C^-T (upper triangle 00 01 02 11 12 22) per landmark block   (wmode 1)
    GP<double> ul;             // nL*3: u = C^-1 bl                                                     (wmode 1)
    GP<int4> pairItems;        // per camera pair (i1 <= i2), nL slots: {144 * edge (l, i1), 144 * edge (l, i2), 24 * l, l} of the landmarks both observe, ascending (k_ba_pairs)
    GP<double> sys;            // n*(n+1): assembled Schur system (upper triangle + rhs column)
    GP<int> doneCnt;           // workgroups of the running k_ba_backsub that have published their partial sums (zero between launches)
    GP<double> wgPart;         // their partial sums: 2 per workgroup
    GP<long long> dbg;         // optional phase stamps of k_ba_solve (diagnostic builds of the harness only)
    GP<double> xp;             // nFree*6
    GP<double> xl;             // nL*3
    GP<double> partChi;        // nL (robust chi2 of the point's edges at the last evaluated state)
    GP<double> partScale;      // nL
    GP<double> lm;             // [0] lambda [1] ni [2] currentChi [3] maxdiag
    // MapPlane vertices / EdgePlane edges of Optimizer::BundleAdjustment (src/Optimizer.cc:203-252): landmarks nPtsOnly.. are
    // planes (4 coefficients each, two state buffers like the points), edges nEdgesPt.. are plane edges (eflag bit3)
    int nPtsOnly, nEdgesPt;
    GP<double> plBuf[2];
    GP<const double> pmeas;    // 4 per plane edge: the measured plane, normalised
    double deltaPlane, infoAngle, infoDist;
    // map-scale path (k_bal_*): dense lower-triangular system in HBM, panel workspace, factored diagonal blocks, pair CSR
    // Round 5: BLOCK-SPARSE.  The reference factors this system with a sparse LDL^T (SimplicialLDLT + AMD, solvers/linear_solver_eigen.h:95-112) because a
    // map's covisibility is sparse; here the lower triangle lives as 64 x 64 TILES and only the tiles that the covisibility structure -- and the fill-in of its
    // elimination, worked out on the host at tile level -- can ever make non-zero exist: bigTile[ti * bigT + tj] = the tile's slot in both pools, or -1.
    // Memory and work follow the non-zero structure; a map in which every keyframe sees every other one keeps every tile and runs as before.
    GP<double> big;            // bigTiles x 64 x 64: the working matrix (row-major inside a tile)
    GP<double> bigL;           // bigTiles x 64 x 64: the factor L (rows below each panel's diagonal block); row N = z
    GP<double> bigDiag;        // (N / 32) * 32 * 32: unit-lower diagonal blocks, column-major (1 / d on the diagonal)
    GP<double> bigLinv;        // (N / 32) * 32 * 32: their inverses as unit-lower blocks (k_bal_linv, for the back substitution)
    GP<int> bigFail;
    GP<const int> bigTile;     // bigT * bigT
    int bigT, bigTiles, bigDense;
    // round 6 (GbaPlan): the system's rows follow the elimination order -- segments, then the separator block, each on a 64-row tile boundary, identity padding between
    int bigN;                  // rows / columns of the padded system (a multiple of 64); row bigN = the right-hand side, in a tile row of its own
    GP<const int> bigRow;      // nFree: first row of free camera i's six unknowns
    GP<const int> bigRowCam;   // bigN: the natural index (6 i + q) of the unknown a row holds, -1 for a padding row
    GP<const int4> bigSB;      // super-blocks of the back substitution: {first column, width, first / end column chunk (of 64) to its left that hears from it}
    GP<const int> bigDiagList; // panels whose diagonal block a k_bal_diag launch factors (no earlier panel reaches their tile row)
    GP<const int4> bigWork;    // per LAUNCH of the factorisation (GbaPlan::launches), ONE record pair per tile it updates -- {ti, tj, slot of tile (ti, tj), slot of the panel's tile in row ti},
                               // {slot of the panel's tile in row tj, panel, flags (1: factor panel `next`'s diagonal block on the spot, 2: archive tile row ti's l entries, 4: nothing but that), next};  (rounds 3-5: per panel)
                               // so a workgroup finds its tiles with one load; a launch's look-ahead records come first
    GP<const int> lpStart;     // nPairsNZ + 1
    GP<const int> lpPair;      // 2 * nPairsNZ: (i1, i2), i1 <= i2
    GP<const int> lpOrder;     // launch slots: the pairs with more than kBigPairLong entries first (four waves each), then the others (one wave each); -1 = idle slot
    GP<int> lpPts;             // landmark blocks of each pair, ascending
    GP<int> lpE1;              // ... and the landmark's edges in camera i1 / i2 (the dense point x camera table of the window path would be
    GP<int> lpE2;              //     nP x nC ints: 400 MB for a 1000-keyframe map).  Written by k_bal_pair_fill from the observer lists below
    GP<const int> lmOff;       // nPts + 1: per landmark its FREE observers, sorted by free-camera block ...
    GP<const int> lmCam;       // ... the block
    GP<const int> lmEdge;      // ... the edge
    GP<const int> cmOff;       // nFree + 1: per free camera its landmarks, ascending ...
    GP<const int> cmLm;        // ... the landmark (point index)
    GP<const int> cmU;         // ... the position of the camera's own entry in that landmark's observer list
    int nPairsNZ;
    // per-window addresses every kernel finds HERE (the kernels take an array of windows and blockIdx.z, see BA_WIN)
    GP<int> ctl0;              // the two control blocks (8 ints each); a launch runs on ctl0 + 8 * par
    GP<double> lm0;            // the two LM blocks (8 doubles each)
    GP<int> solveOk;
    GP<struct BAStatus> status;   // pinned host memory
    GP<unsigned char> cls;     // E: outlier table of the pass between the two optimize() calls
    GP<SE3> outCams; GP<double> outPts; GP<unsigned char> outCls; GP<double> outPlanes;   // pinned results (k_ba_finish)
};


// Every BA kernel takes the device array of window records: workgroup (x, y, z) works on window z, on the control / LM block
// pair `wpar` of that window (BADecision).  The array holds TWO records per window that differ only in ctl / lm (pair 0 and
// pair 1), so a kernel reads its record in place -- uniform, read-only loads on the scalar unit, exactly like kernel
// arguments (a local copy with the two pointers patched went through scratch memory: 1.17 -> 2.2 ms per window).  A single
// window is a batch of one.
// Batches pin every window to one XCD (speed only, any placement gives the same results): workgroups are dealt round-robin
// over the 8 XCDs in dispatch order, so with the plain (x = block, z = window) numbering the ~5 MB a window keeps re-reading
// (Hpl blocks, residuals, the edge table) would be pulled into all eight L2s -- 25 windows are then fabric-bound (the pair
// assembly alone moved 350 MB per launch).  Window w is served by XCD w % 8 only; the launch pads grid.z to a multiple of 8.
// The windows of an INCOMPLETE last row of eight share all eight XCDs (slot x serves window x mod rem with the other slots of that residue, the
// window's workgroups dealt round-robin among them): with 25 windows on 3 + 3 + ... + 4 the XCD that held four set the pace of every launch -- the
// kernels took as long for 25 windows as for 32 (k_ba_linearize 53.7 / 55.3 us, 43.0 for 24; profiles/r04_ba_xcd_balance.txt).
// wpar = block pair | rot << 4 | number of windows << 8 (rot: XCD of the group's first window -- the groups of a batch run
// concurrently and together should load the XCDs evenly);  bx = this workgroup's block index inside its window.
#define BA_WIN(P)                                                                                                   \
    unsigned bx = blockIdx.x, wz_ = blockIdx.z; (void)bx;                                                                  \
    {                                                                                                               \
        const unsigned nz_ = (unsigned)wpar >> 8;                                                                   \
        if (nz_ > 1) {                                                                                              \
            const unsigned b_ = blockIdx.x + gridDim.x * blockIdx.z, s_ = b_ >> 3;                                  \
            const unsigned xs_ = (b_ - ((unsigned)wpar >> 4)) & 7, row_ = s_ / gridDim.x;      /* XCD x: slot (x - rot) mod 8 */  \
            const unsigned rem_ = nz_ & 7;                                                                          \
            bx = s_ % gridDim.x;                                                                                    \
            wz_ = xs_ + 8 * row_;                                                                                   \
            if (rem_ && row_ == (nz_ >> 3)) {      /* the incomplete last row: its rem_ windows over all eight slots */   \
                const unsigned wq_ = xs_ % rem_, nsh_ = (8 - wq_ + rem_ - 1) / rem_;                                \
                if (bx % nsh_ != xs_ / rem_) return;                                                                \
                wz_ = 8 * row_ + wq_;                                                                               \
            }                                                                                                       \
            if (wz_ >= nz_) return;                                                                                 \
        }                                                                                                           \
    }                                                                                                               \
    const BADev& P = W[2 * wz_ + (wpar & 1)]

enum { kCtlHalt = 0, kCtlCur = 1, kCtlIters = 2, kCtlStatus = 3, kCtlNBad = 4, kCtlPhase = 5, kCtlAnyActive = 6 };   // phase: 0 / 1 = first / second optimize()
enum { kStRunning = 0, kStTakeover = 1, kStTerminate = 2, kStEmpty = 3 };   // empty: no level-0 edge left (g2o's optimize() returns -1)

struct BAPhase { double lambda, rho, chi; int accepted, cur, iters, status, nBad, touched; };
struct BAStatus {            // pinned host memory, written by k_ba_decide / k_ba_chi_init
    double lambda, rho, chi, tempChi;
    int accepted, solveOk, seq, cur;
    int iters, status, nBad, ntrace;
    double trLambda[64], trChi[64];   // [32 * phase + iteration]
    int trTrials[64];
    BAPhase ph[2];           // where each optimize() stood after its last decision (both calls may run in ONE enqueue)
};

constexpr int kTileMaxFree = 30;
constexpr int kTileThreads = 1024;
struct TileGeom { int n, n4, R, Tr, Tc, nTiles; };
constexpr int kBigMaxFree = 8192;
constexpr int kBigNB = 32;
struct BigGeom { int n, N, RP; };
constexpr int kBigPairLong = 2048;
constexpr int kBigSB = 256;      // columns per super-block of the back substitution

// The factorisation launches take what they need BY VALUE: the pool pointers, the halt flag's address, the work list -- and the first kBigByValue record pairs of the
// launch (its look-ahead workgroups come first in the list).  Read through the window record like the other LM kernels, a launch started with three dependent round
// trips to memory (record -> work list -> tiles) before its first useful load.
constexpr int kBigByValue = 8;      // the first records of a launch -- its look-ahead workgroups, the chain's critical path -- travel in the kernel arguments
struct BigStepArgs {
    double* big; double* bigL; double* bigDiag; int* bigFail; const int4* bigWork; const int* ctl; long long* dbg;
    int N, nByValue;
    int4 rec[2 * kBigByValue];
};

// Elimination order, tile structure and launch schedule of the map-scale path (gba.hip: gba_build_plan): a pure function of the covisibility pattern, kept in the
// thread's context and reused while the pattern's hash stays the same.
struct GbaPlan {
    uint64_t key = 0;
    bool valid = false;
    int nFa = 0, N = 0, T = 0, RP = 0, nbk = 0, bigTiles = 0;
    int P = 1, nSep = 0, sepStart = 0, rcm = 0, bandwidth = 0, chainNatural = 0, chainEstimate = 0;
    std::vector<int> rowOf, rowCam, segStart, tileMap, diagList;
    std::vector<int4> work, sb;
    struct Launch { int off, cnt, diagOff, diagCnt; };
    struct SbLaunch { int off, cnt, maxChunks; };
    std::vector<Launch> launches;
    std::vector<SbLaunch> sbLaunches;
};
uint64_t gba_pattern_hash(int nFa, const std::vector<int>& prA, const std::vector<int>& prB);
void gba_build_plan(int nFa, const std::vector<int>& prA, const std::vector<int>& prB, int forceP, GbaPlan& pl);

struct LMTraceHost {
    std::vector<double> lambda, chi2;
    std::vector<int> trials;
    float deviceMs = 0;
    int linearizations = 0;
    void clear() { lambda.clear(); chi2.clear(); trials.clear(); deviceMs = 0; linearizations = 0; }
};
extern thread_local LMTraceHost g_trace;      // (defined in lm_host.hip)

struct LMContext {  // per-thread device workspace, grow-only
    hipStream_t stream = nullptr;                   // the stream of the call in progress: one of byClass[] (every call ends synchronised, so the next may take another)
    hipStream_t byClass[3] = {nullptr, nullptr, nullptr};      // eao::StreamClass: PoseOptimization / LocalBundleAdjustment / map BundleAdjustment of this host thread
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BAStatus* status = nullptr;   // pinned + mapped
    unsigned char* pin = nullptr; // pinned host mirror of the input part of the arena: ONE H2D copy per upload
    size_t pinCap = 0;
    unsigned char* pinOut = nullptr;   // pinned results, written by k_ba_finish
    size_t pinOutCap = 0;
    eao::DevBuf<unsigned char> bytes;
    std::vector<int> scratch;     // host counters of the structure build (kept to avoid per-call allocation)
    // map-scale path: where every 32-column panel's work records start / the records themselves, as the HOST reads them when it enqueues the panel launches --
    // which, in a batch call, happens after every window has been prepared.  They belong to the window's context (round 5: as thread-local tables of the set-up
    // worker they were overwritten by the next map-scale window the same worker prepared, and the first window ran with the second one's panels --
    // tools/dbg_batch_two_maps.py, tests/test_gpu_lm.py::test_two_map_scale_windows_in_one_batch).
    GbaPlan plan;
    size_t used = 0;
    ~LMContext() {
        if (status) (void)hipHostFree(status);
        if (pin) (void)hipHostFree(pin);
        if (pinOut) (void)hipHostFree(pinOut);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (hipStream_t q : byClass) if (q) (void)hipStreamDestroy(q);
    }
};
extern thread_local LMContext g_ctx;          // (defined in lm_host.hip)
eao_status ctx_init(LMContext& c, bool ownStream, eao::StreamClass cls);

// bump allocator over one device buffer (256-byte aligned slices)
struct Arena {
    unsigned char* base;
    size_t cap, off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// Launch geometry of one window -- or, field by field, the largest of a batch (every kernel guards its own window's sizes).
struct BADims {
    int nF = 0, nL = 0, nP = 0, nC = 0, E = 0, nPl = 0;
    bool hasPl = false, bigPath = false, usePairs = false, solveTiles = false, tiles3 = true, wmode = false;
    size_t tileLds = 0;
    int nPairsNZ = 0;          // map-scale path (never batched)
    int nPairsLong = 0, nPairsSlots = 0;        // " : launch slots of the four-wave kernel (first in lpOrder) / of both
    BigStepArgs bigArgs{};     // " : what k_bal_step takes by value (ctl / wa0 / wb0x filled per launch)
    const int* bigCtl0 = nullptr;              // " : the control blocks on the device
    double* big = nullptr;     // "
    int bigTiles = 0;          // "
    const GbaPlan* plan = nullptr;           // " : the launch schedule (the context's: it outlives the call's launches; a POINTER: this struct is copied around)
    BigGeom gB{};
    void merge(const BADims& o) {
        nF = std::max(nF, o.nF); nL = std::max(nL, o.nL); nP = std::max(nP, o.nP); nC = std::max(nC, o.nC); E = std::max(E, o.E);
        tiles3 = tiles3 && o.tiles3; tileLds = std::max(tileLds, o.tileLds);
    }
};

// Where launches go: `nz` windows (device array W) on one stream.  Every kernel of the LM engine is launched from here, with
// the window as grid.z -- a single window is a batch of one.
struct BALaunch {      // (member functions: lba.hip -- they launch its kernels; the map-scale branch of a trial: gba.hip)
    const BADev* W = nullptr;
    int nz = 1;
    BADims d;
    hipStream_t s = nullptr;
    int seq = 0;
    int rot = 0;               // XCD of window 0 (BA_WIN)
    eao_status attributes() const;
    int wp(int par) const { return par | (nz > 1 ? (rot & 7) << 4 | nz << 8 : 0); }      // kernel argument (BA_WIN)
    unsigned gz() const { return nz > 1 ? (unsigned)((nz + 7) & ~7) : 1u; }   // windows are dealt to the XCDs: grid.z padded to 8
    int ptBlocks() const { return eao::cdiv(std::max(d.nL, 1) * 8, 256); }       // eight lanes per landmark
    int linBlocks() const;
    template <int NT> void lin_launch(int par, int first, int diagOnly) const;
    void lin_w(int par, int first, int diagOnly) const;
    void setup() const;
    void trial(int par, int bulk, bool firstTrial, bool withDecide);
    void relinearize();
    void bulk(int from, int to, bool withErrors);
    void classify() const;
    void finish() const;
    void chain(int mode, int itsFirst, int itsSecond);
};
// gba.hip: the dynamic-LDS limit of its back substitution; assembly + factorisation + back substitution of one trial of a map-scale window (BALaunch::trial)
eao_status gba_attributes();
void gba_enqueue_trial(const BALaunch& L, int par, bool firstTrial);
void gba_enqueue_pair_fill(const BALaunch& L);

}  // namespace lm
}  // namespace eao
using namespace eao::lm;      // NOLINT: an internal header of four translation units

namespace {

// 1/x: v_rcp_f64 + two Newton steps on the device (~1 ulp, a third of the instructions of an IEEE division), a plain
// division in host code
__host__ __device__ inline double recip(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// Several IEEE divisions by ONE denominator.  The compiler's sequence per fp64 division is v_div_scale x 2, v_rcp_f64 (quarter
// rate), four FMAs that refine the reciprocal, v_mul, v_fma, v_div_fmas, v_div_fixup: eleven instructions, seven of which
// depend on the denominator alone.  DivBy<true> runs those once (rcp + 4 FMA) and spends mul + 2 FMA per quotient -- the same
// operations on the same values, so the quotient is the SAME correctly rounded double, as long as v_div_scale would not have
// rescaled and v_div_fixup would not have patched anything: plain_den() admits denominators of magnitude 2^-400 .. 2^400 (not
// zero, NaN or infinite; depths in metres and their squares), the numerators are coordinates and products of coordinates
// (a quotient in the denormal range or a -0 numerator's sign could differ -- neither reaches a result).  The Jacobians of
// one BA edge divide 22 times by z or z^2 (types_six_dof_expmap.cpp:103-139,188-234 written out as upstream writes them):
// 242 -> 76 instructions per edge and role.  DivBy<false> is the plain division, taken lane by lane for any other denominator.
__device__ __forceinline__ bool plain_den(double d) { const double a = fabs(d); return a > 0x1p-400 && a < 0x1p400; }
template <bool SHARED> struct DivBy {
    double d, r;
    __device__ __forceinline__ explicit DivBy(double den) : d(den), r(0) {
        if (SHARED) {
            r = __builtin_amdgcn_rcp(den);
            r = fma(r, fma(-den, r, 1.0), r);
            r = fma(r, fma(-den, r, 1.0), r);
        }
    }
    __device__ __forceinline__ double operator()(double a) const {
        if (!SHARED) return a / d;
        const double q = a * r;
        return fma(fma(-d, q, a), r, q);
    }
};

// ============================================================================================ SE3 helpers

__host__ __device__ inline Quat quat_from_matrix(const double m[9]) {
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 * recip(t);
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {   // i = 0 (Eigen picks the largest diagonal; ties go to the lower index)
        t = sqrt(m[0] - m[4] - m[8] + 1.0);
        q.x = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[7] - m[5]) * t; q.y = (m[3] + m[1]) * t; q.z = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {    // i = 1
        t = sqrt(m[4] - m[8] - m[0] + 1.0);
        q.y = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[2] - m[6]) * t; q.z = (m[7] + m[5]) * t; q.x = (m[1] + m[3]) * t;
    } else {                                        // i = 2
        t = sqrt(m[8] - m[0] - m[4] + 1.0);
        q.z = 0.5 * t;
        t = 0.5 * recip(t);
        q.w = (m[3] - m[1]) * t; q.x = (m[2] + m[6]) * t; q.y = (m[5] + m[7]) * t;
    }
    return q;
}
__host__ __device__ inline void quat_normalize_pos(Quat& q) {
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double in = recip(sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w));
    q.x *= in; q.y *= in; q.z *= in; q.w *= in;
}
__host__ __device__ inline Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
__host__ __device__ inline void quat_rotate(const Quat& q, const double v[3], double out[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
__host__ __device__ inline void quat_to_matrix(const Quat& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void se3_map(const SE3& s, const double p[3], double out[3]) {
    quat_rotate(s.r, p, out);
    out[0] += s.t[0]; out[1] += s.t[1]; out[2] += s.t[2];
}
__host__ __device__ inline SE3 se3_exp(const double u[6]) {  // (omega, upsilon), types/se3quat.h:223-259
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double Om2[9], R[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { const double id = (i % 4 == 0) ? 1.0 : 0.0; R[i] = id + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        double st, ct;
#ifdef __HIP_DEVICE_COMPILE__
        // LM steps are small rotations: below half a radian the Taylor polynomials to x^15 / x^16 are exact to the last bit or two
        // (remainder < 2^-70) and cost 17 fused multiply-adds; the library's sincos (argument reduction, two kernels, ~150 fp64
        // instructions on ONE wave while the workgroup waits) only runs for larger angles.
        if (theta < 0.5) {
            const double z = theta * theta;
            double ps = -1.0 / 1307674368000.0;                     // -1/15!
            ps = fma(ps, z, 1.0 / 6227020800.0);                    // +1/13!
            ps = fma(ps, z, -1.0 / 39916800.0);                     // -1/11!
            ps = fma(ps, z, 1.0 / 362880.0);                        // +1/9!
            ps = fma(ps, z, -1.0 / 5040.0);                         // -1/7!
            ps = fma(ps, z, 1.0 / 120.0);                           // +1/5!
            ps = fma(ps, z, -1.0 / 6.0);                            // -1/3!
            st = fma(theta * z, ps, theta);
            double pc = 1.0 / 20922789888000.0;                     // +1/16!
            pc = fma(pc, z, -1.0 / 87178291200.0);                  // -1/14!
            pc = fma(pc, z, 1.0 / 479001600.0);                     // +1/12!
            pc = fma(pc, z, -1.0 / 3628800.0);                      // -1/10!
            pc = fma(pc, z, 1.0 / 40320.0);                         // +1/8!
            pc = fma(pc, z, -1.0 / 720.0);                          // -1/6!
            pc = fma(pc, z, 1.0 / 24.0);                            // +1/4!
            pc = fma(pc, z, -0.5);                                  // -1/2!
            ct = fma(pc, z, 1.0);
        } else
#endif
        sincos(theta, &st, &ct);                // one range reduction for both
        const double it = recip(theta), it2 = it * it;
        const double a = st * it, b = (1 - ct) * it2;
        const double c = (theta - st) * (it2 * it);
        for (int i = 0; i < 9; i++) {
            const double id = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = id + a * Om[i] + b * Om2[i];
            V[i] = id + b * Om[i] + c * Om2[i];
        }
    }
    SE3 s;
    s.r = quat_from_matrix(R);
    for (int i = 0; i < 3; i++) s.t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    quat_normalize_pos(s.r);
    return s;
}
__host__ __device__ inline SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 r;
    double rt[3];
    quat_rotate(a.r, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
    r.r = quat_mul(a.r, b.r);
    quat_normalize_pos(r.r);
    return r;
}
inline SE3 se3_from_Tcw_f32(const float* T) {  // Converter::toSE3Quat, reference src/Converter.cc:28-38
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    SE3 s;
    s.r = quat_from_matrix(R);
    s.t[0] = T[3]; s.t[1] = T[7]; s.t[2] = T[11];
    quat_normalize_pos(s.r);
    return s;
}
inline void se3_to_Tcw_f32(const SE3& s, float* T) {  // Converter::toCvMat(SE3Quat), reference src/Converter.cc:40-59
    double R[9];
    quat_to_matrix(s.r, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)s.t[i];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// 1/x by v_rcp_f64 + two Newton steps (~1 ulp): the solver's pivots
// one Newton step on v_rcp_f64: ~2^-46 relative error (the hardware seed carries single-precision accuracy)
__device__ inline double frcp1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ inline double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ inline void huber(double e, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho0 = e; rho1 = 1.; }
    else { const double s = sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}


// v + (v of the lane a DPP control selects): the building block of the cross-lane sums below -- VALU only, no LDS round trip
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
// sum over the 8 lanes of an aligned lane group; every lane gets the result.  The same tree as xor-shuffles by 1, 2, 4
// (pairs, quads, then the mirrored quad of the other half), so the sums are bit-identical to those -- but each step is a
// DPP add instead of a ds_bpermute round trip (66 of them per k_ba_linearize before).
__device__ __forceinline__ double group8_sum(double v) {
    v = dpp_add_f64<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp_add_f64<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]: lane ^ 2
    v = dpp_add_f64<0x141, 0xF>(v);    // row_half_mirror: lane i <-> 7 - i of its 8-lane half, i.e. the other quad's sum
    return v;
}

// ---- block-wide fixed-order sum of NV doubles per thread; result valid in thread 0 (and in `out` LDS after a barrier)
__device__ __forceinline__ double quad_sum(double v) {   // sum over the four lanes of a quad, same value in all four
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));          // quad_perm [2,3,0,1]
    return v + o;
}
// Sum over the eight lanes of a 16-lane DPP row that share this lane's parity (valid in lanes 0 and 1 of the row).
__device__ __forceinline__ double row_half_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x124, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x124, 0xF, 0xF, true));        // row_ror:4
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x128, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x128, 0xF, 0xF, true));        // row_ror:8
    return v + o;
}
// Fixed-order sum of NV accumulators over a block of NT threads through LDS: lane quads first (DPP), then NV x 8 column
// threads over NT/32 quad leaders each, then the last 8.  red: (NT/4)*NV doubles, part: 8*NV doubles; the totals land in
// part[0 .. NV).  (A 64-lane shuffle tree per value costs ~230 cycles per value; this is ~10x cheaper for NV ~ 28.)
template <int NV, int NT, int SEGS = 8>
__device__ inline void block_sum_lds(double (&acc)[NV], double* red, double* part) {
    static_assert(NV * SEGS <= NT && SEGS <= 8 && (NT / 4) % SEGS == 0, "column threads");
#pragma unroll
    for (int q = 0; q < NV; q++) acc[q] = quad_sum(acc[q]);
    if ((threadIdx.x & 3) == 0) {
        double* dst = red + (threadIdx.x >> 2) * NV;
#pragma unroll
        for (int q = 0; q < NV; q++) dst[q] = acc[q];
    }
    __syncthreads();
    constexpr int kSeg = NT / 4 / SEGS;
    if (threadIdx.x < NV * SEGS) {
        const int q = threadIdx.x % NV, seg = threadIdx.x / NV;
        double sacc = 0;
        for (int j = 0; j < kSeg; j++) sacc += red[(seg * kSeg + j) * NV + q];
        part[seg * NV + q] = sacc;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double sacc = 0;
        for (int seg = 0; seg < SEGS; seg++) sacc += part[seg * NV + threadIdx.x];
        part[threadIdx.x] = sacc;                  // only this thread reads or writes these eight slots
    }
    __syncthreads();
}

// Sum over the 64 lanes of a wave on the VALU: DPP quad permutes, half-row and row mirrors, then the row broadcasts 15 / 31;
// the total ends up in lane 63.  (A shuffle tree is six dependent ds_bpermute round trips per value: the single-value sums
// on the LM kernels' critical paths cost ~4.5 k cycles each that way.)
__device__ __forceinline__ double wave_sum_f64_lane63(double v) {
    v = dpp_add_f64<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141, 0xF>(v);    // row_half_mirror
    v = dpp_add_f64<0x140, 0xF>(v);    // row_mirror: every lane of a row holds the row sum
    v = dpp_add_f64<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3
    return v;
}
template <int NV, int NT>
__device__ inline void block_sum(double (&v)[NV], double* lds /* (NT/64)*NV */, double* out /* NV */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const double x = wave_sum_f64_lane63(v[k]);
        if (lane == 63) lds[wv * NV + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int w = 0; w < NT / 64; w++) s += lds[w * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// 6x6 LDLT solve of the damped pose system.  g2o's dense solver (solvers/linear_solver_dense.h:104-112) uses Eigen's
// diagonally pivoted LDLT and reports failure when the matrix is not positive; an unpivoted factorisation has the same
// inertia (Sylvester), hence the same success/failure decision, and the same solution up to rounding -- and it keeps
// every index static (registers, no scratch).
__device__ inline bool ldlt6_solve(const double* A, const double* b, double* x) {
    double a[6][6], inv[6], y[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) a[r][c] = A[r * 6 + c];
    bool positive = true;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const double d = a[r][r];
        if (!(d > 0)) positive = false;
        inv[r] = recip(d);
#pragma unroll
        for (int i = r + 1; i < 6; i++) {
            const double l = a[r][i] * inv[r];
#pragma unroll
            for (int c = i; c < 6; c++) a[i][c] = fma(-l, a[r][c], a[i][c]);
            a[i][r] = l;          // keep the multiplier below the diagonal
        }
    }
    if (!positive) return false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v = fma(-a[i][k], y[k], v);
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i] * inv[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v = fma(-a[k][i], x[k], v);
        x[i] = v;
    }
    return true;
}

// ------------------------------------------------------------------ planes (src/g2oAddition/Plane3D.h, EdgePlane.h)
// Plane3D keeps (n, -d) normalised with a non-negative fourth coefficient; the edge's error is (azimuth, elevation,
// distance) of the measured plane in the frame that rotates the predicted plane's normal onto +x.
__host__ __device__ inline void plane_normalize(double c[4]) {                 // Plane3D::normalize
    const double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const double s = 1. / n;
    for (int k = 0; k < 4; k++) c[k] = c[k] * s;
    if (c[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
}
inline void plane_from_f32(const float* v, double c[4]) {                      // Converter::toPlane3D, src/Converter.cc:215-225
    for (int k = 0; k < 4; k++) c[k] = v[k];
    if (v[3] < 0.0) for (int k = 0; k < 4; k++) c[k] = -c[k];
    plane_normalize(c);
}
// Plane3D::rotation: AngleAxis(azimuth, Z) * AngleAxis(-elevation, Y) as a quaternion product
__device__ inline void plane_rotation(const double* v, double Rn[9]) {
    const double az = atan2(v[1], v[0]);
    const double el = atan2(v[2], sqrt(v[0] * v[0] + v[1] * v[1]));
    const double ha = 0.5 * az, hb = 0.5 * (-el);
    const Quat qa{0, 0, sin(ha) * 1.0, cos(ha)}, qb{0, sin(hb) * 1.0, 0, cos(hb)};
    quat_to_matrix(quat_mul(qa, qb), Rn);
}
// Plane3D::oplus (VertexPlane::oplusImpl, src/g2oAddition/Plane3D.h:73-89, VertexPlane.h:35-38)
__device__ inline void plane_oplus(const double* c, const double* v, double out[4]) {
    const double sn = sin(v[1]), cs = cos(v[1]);
    const double n[3] = {cs * cos(v[0]), cs * sin(v[0]), sn};
    double R[9];
    plane_rotation(c, R);
    const double d = (-c[3]) + v[2];
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = R[r * 3] * n[0] + R[r * 3 + 1] * n[1] + R[r * 3 + 2] * n[2];
    out[3] = -d;
    plane_normalize(out);
}
__device__ inline void plane_error(const SE3& T, const double* world, const double* meas, double err[3]) {   // EdgePlane::computeError
    double R[9];
    quat_to_matrix(T.r, R);
    double v2[4];
#pragma unroll
    for (int r = 0; r < 3; r++) v2[r] = R[r * 3] * world[0] + R[r * 3 + 1] * world[1] + R[r * 3 + 2] * world[2];
    v2[3] = world[3] - (T.t[0] * v2[0] + T.t[1] * v2[1] + T.t[2] * v2[2]);
    if (v2[3] < 0.0) { v2[0] = -v2[0]; v2[1] = -v2[1]; v2[2] = -v2[2]; v2[3] = -v2[3]; }
    plane_normalize(v2);
    double Rn[9];
    plane_rotation(v2, Rn);
    double n[3];
#pragma unroll
    for (int r = 0; r < 3; r++) n[r] = Rn[r] * meas[0] + Rn[3 + r] * meas[1] + Rn[6 + r] * meas[2];   // rotation^T * normal
    err[0] = atan2(n[1], n[0]);
    err[1] = atan2(n[2], sqrt(n[0] * n[0] + n[1] * n[1]));
    err[2] = (-v2[3]) - (-meas[3]);
}
__device__ __forceinline__ const SE3* cur_cams(const BADev& P) { return P.camsBuf[P.ctl[kCtlCur]]; }
__device__ __forceinline__ const double* cur_pts(const BADev& P) { return P.ptsBuf[P.ctl[kCtlCur]]; }
__device__ __forceinline__ SE3* trial_cams(const BADev& P) { return P.camsBuf[P.ctl[kCtlCur] ^ 1]; }
__device__ __forceinline__ double* trial_pts(const BADev& P) { return P.ptsBuf[P.ctl[kCtlCur] ^ 1]; }

// (Hll + lambda I)^-1 by cofactors / determinant, as Eigen's fixed-size 3x3 inverse() (block_solver.hpp:392)
__device__ inline void dinv3(const double* Hll, double lambda, double Di[9]) {
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; i++) A[i] = Hll[i];
    A[0] += lambda; A[4] += lambda; A[8] += lambda;
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    Di[0] = c00 * id; Di[1] = (A[2] * A[7] - A[1] * A[8]) * id; Di[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Di[3] = c01 * id; Di[4] = (A[0] * A[8] - A[2] * A[6]) * id; Di[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Di[6] = c02 * id; Di[7] = (A[1] * A[6] - A[0] * A[7]) * id; Di[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

typedef double v4d __attribute__((ext_vector_type(4)));
__host__ __device__ inline TileGeom tile_geom(int nF) {
    TileGeom g;
    g.n = nF * 6; g.n4 = (g.n + 3) & ~3; g.R = g.n4 + 1;
    g.Tr = (g.R + 15) >> 4; g.Tc = (g.n4 + 15) >> 4;
    g.nTiles = g.Tc * g.Tr - g.Tc * (g.Tc - 1) / 2;      // column tj holds tile rows tj .. Tr-1
    return g;
}
__host__ __device__ inline void tile_of(const TileGeom& g, int idx, int& ti, int& tj) {
    int j = 0, off = 0;
    while (idx >= off + (g.Tr - j)) { off += g.Tr - j; j++; }
    tj = j; ti = j + (idx - off);
}
// L archive: row r (r = n4 is the right-hand side, i.e. z) starts at r (r - 1) / 2 + 4 r: r entries + 4 slack so that
// the panel threads store their four values unconditionally
__host__ __device__ inline int tile_lrow(int r) { return r * (r - 1) / 2 + 4 * r; }
__host__ inline size_t tile_solver_lds(int nF) {
    const TileGeom g = tile_geom(nF);
    return ((size_t)((tile_lrow(g.n4 + 1) + 1) & ~1) + 3 * (size_t)g.Tr * 16 * 4 + (size_t)g.n4 + 8) * sizeof(double);
}
__host__ __device__ inline int pair_index(int i1, int i2, int nF) { return i1 * nF - i1 * (i1 - 1) / 2 + (i2 - i1); }   // i1 <= i2
__host__ __device__ inline BigGeom big_geom(int nF) {
    BigGeom g;
    g.n = nF * 6; g.N = (g.n + kBigNB - 1) / kBigNB * kBigNB; g.RP = (g.N + 1 + 63) & ~63;
    return g;
}

}  // namespace
