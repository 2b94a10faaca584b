// orb_internal.h -- what the three translation units of the ORB extractor share (round 6: orb.hip was one 2 900-line file):
//   orb.hip           the pyramid / FAST / blur / orientation + description / stereo kernels and the enqueue of a call (every launch but the quad-tree's)
//   orb_quadtree.hip  k_quadtree (DistributeOctTree) and its launch
//   orb_host.hip      the handle: geometry of a frame size, streams and events, the C API entry points that launch nothing themselves
// Constants, the geometry records the host fills and the kernels read, the handle, and the functions that cross the files.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"


namespace eao {
namespace orb {

constexpr int kMaxLevels = 16;
constexpr int kEdge = refc::EDGE_THRESHOLD;
constexpr int kMinBorder = kEdge - 3;  // 16
constexpr int kTile = 72;              // max FAST sub-image edge (cell <= 60 px + 6)
// (tested region of a cell = sub-image minus the 3 px FAST margin on every side: at most 66 x 66)
constexpr int kTileStrideWide = 80;    // LDS row stride of the widest tile: 72 + up to 3 bytes of alignment phase, multiple of 4
constexpr int kMaxIni = 16;
constexpr size_t kPinnedOutMax = 1 << 20;   // host-API calls whose results fit go through mapped pinned memory (eao_orb_extract_batch)
constexpr int kFastXcdRun = 0;         // 0 = plain workgroup -> cell order (see k_fast_cells)

struct LevelGeom {
    int w, h, pitch, off;      // level image; off = byte offset inside one frame's pyramid block
    int cellBase, nCells;      // FAST cells of this level inside the per-frame cell table
    int quota;                 // mnFeaturesPerLevel
    int nIni, boxH;            // initial quad-tree nodes; maxBorderY - minBorderY
    float hX;
    int listCap;               // node-list / keypoint capacity of this level
    int kpBase;                // first keypoint slot of this level inside a frame
    int candBase, candCap;     // candidate scratch of this level inside a frame
    int tileBase, tilesX;      // blur: first workgroup of this level, 128-px strips per row of strips
    long long nodeOff;         // k_quadtree, global node lists: byte offset of this level inside a frame's workspace
    int scaledPatch;
    float scale;
};

struct Geom {
    int nlevels, W, H;
    int totalCells, cellCap;
    int totalKpCap, totalCandCap, totalTiles;
    int pyrFrameBytes;
    int iniTh, minTh;
    int scanCap;               // LDS scan workspace entries for k_quadtree
    int qtLdsCand, qtKeysOff;  // k_quadtree: candidates (key + node index) that fit in LDS, byte offset of the keys
    int qtNodesGlobal;         // the node lists do not fit in LDS (thousands of features on one level): global workspace instead
    long long qtNodeFrameBytes;
    int fastMaxTested, fastTileBytes, fastLdsBytes, fastStride;   // k_fast_cells dynamic LDS carve-up
    int umax[16];
    LevelGeom L[kMaxLevels];
};

struct CellDesc {
    short level, x0, y0, sw, sh, offX, offY, pad;
};

struct ImgSrc {  // level-0 source (caller's frames) + internal pyramid
    const uint8_t* img0;
    int pitch0;
    long long fs0;
    uint8_t* pyr;
};

__device__ __forceinline__ const uint8_t* level_ptr(const Geom* g, const ImgSrc& s, int l, int f, int* pitch) {
    if (l == 0) {
        *pitch = s.pitch0;
        return s.img0 + (long long)f * s.fs0;
    }
    *pitch = g->L[l].pitch;
    return s.pyr + (long long)f * g->pyrFrameBytes + g->L[l].off;
}

constexpr int kPyrTW = 128, kPyrTH = 32;      // the level-1 tile of a workgroup
constexpr int kPyrThreads = 256;
struct PyrLevel { int w, h, pitch, off; double invX, invY; int ldsOff, ldsW, ldsH; int coefXOff, coefYOff; };   // ldsW x ldsH: capacity of the level's tile
struct PyrArgs {
    int nlevels, tilesX, tilesY, pyrFrameBytes;
    int coefOff;                      // LDS offset of the coefficient tables (xofs, xw, yofs, yw: 4 x coefCap ints)
    int coefCap;
    PyrLevel L[kMaxLevels];           // L[0]: the input image (ldsOff / ldsW / ldsH: its staged source region)
    const int4* ranges;               // [level][tilesX + tilesY]: (own0, own1, req0, req1) of a tile column / tile row (device)
    const int2* coefX;                // per level: (source offset, weight pair) of every column / row (device)
    const int2* coefY;
};

// Threads per quad-tree workgroup: a template parameter.  The candidate sweeps scale with it, the block scans and barriers get
// dearer: 1024 threads are faster while the launch is latency-bound (one frame: device step 0.124 -> 0.105 ms, eight frames 0.143 ->
// 0.129), 256 when many workgroups compete for the CUs (64 frames: 0.323 vs 0.339 ms).
constexpr int kQTSmall = 1024, kQTLarge = 256;
constexpr int kQtNodeInts = 2 + 2 + 2 + 4 + 4 + 5;   // per list entry next to its two boxes: cnt, crk, mid (x2 each), childcnt, childpos (x4), five work arrays

constexpr size_t kProfEvents = 10;
struct GraphKey {
    const void* img; int pitch0; long long fs0; int batch; void* kps; void* desc; int cap; void* n; int lanes;
    bool operator==(const GraphKey& o) const {
        return img == o.img && pitch0 == o.pitch0 && fs0 == o.fs0 && batch == o.batch && kps == o.kps && desc == o.desc && cap == o.cap && n == o.n && lanes == o.lanes;
    }
};


// the same on the host (k_pyramid_fused reads tables built with it): lrintf rounds to nearest even like v_cvt_i32_f32
inline void resize_coef_host(int d, double inv, int slimit, bool clampHi, int* ofs, unsigned* wpair) {
    float fr = (float)(((double)d + 0.5) * inv - 0.5);
    int o = (int)std::floor(fr);
    fr -= (float)o;
    if (clampHi) {
        if (o < 0) { fr = 0; o = 0; }
        if (o >= slimit - 1) { fr = 0; o = slimit - 1; }
    }
    const int w0 = std::min(std::max((int)std::lrintf((1.f - fr) * 2048.f), -32768), 32767);
    const int w1 = std::min(std::max((int)std::lrintf(fr * 2048.f), -32768), 32767);
    *ofs = o;
    *wpair = ((unsigned)w0 & 0xFFFFu) | ((unsigned)w1 << 16);
}

// first destination index whose clamped source offset is >= bound (destinations 0 .. dn): the ownership boundary
__host__ __device__ inline int pyr_src_ofs(int d, double inv, int sn) {
    float fr = (float)(((double)d + 0.5) * inv - 0.5);
    int o = (int)floorf(fr);
    return o < 0 ? 0 : (o > sn - 1 ? sn - 1 : o);
}
__host__ __device__ inline int pyr_first_at_least(int bound, double inv, int sn, int dn) {
    if (bound <= 0) return 0;
    int d = (int)((double)bound / inv);          // estimate, then walk (the offset function is monotone)
    d = d < 0 ? 0 : (d > dn ? dn : d);
    while (d > 0 && pyr_src_ofs(d - 1, inv, sn) >= bound) d--;
    while (d < dn && pyr_src_ofs(d, inv, sn) < bound) d++;
    return d;
}

constexpr int kBlurRows = 16;    // output rows per strip
constexpr int kBlurSegW = 128;   // strip width
constexpr int kBlurStripsPerWg = 8;
inline int cv_round(double v) { return (int)std::lrint(v); }

}  // namespace orb
}  // namespace eao

struct eao_orb {
    using Geom = eao::orb::Geom; using CellDesc = eao::orb::CellDesc; using PyrArgs = eao::orb::PyrArgs; using ImgSrc = eao::orb::ImgSrc; using GraphKey = eao::orb::GraphKey;
    eao_orb_cfg cfg;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];
    // geometry of the current (W, H)
    Geom geom;
    bool geomValid = false;
    int batchCap = 0;
    std::vector<CellDesc> cells;
    size_t quadLds = 0;
    eao::DevBuf<int4> d_pyrRanges;
    eao::DevBuf<int2> d_pyrCoef;
    PyrArgs pyr;               // the fused pyramid launch of this geometry
    size_t pyrLds = 0;
    bool pyrFused = false;
    // device state
    hipStream_t stream = nullptr;
    // up to kLanes sub-batches can run as independent pipelines, each on its own (main, side) stream pair
    // (EAO_ORB_LANES, default 1: see the measurement note at enqueue())
    static constexpr int kLanes = 4;
    hipStream_t laneMain[kLanes] = {}, laneSide[kLanes] = {};
    // Round 6: the handle's private streams (its own main stream, the side stream of every call) take the priority of the stream the handle's FIRST call arrives on:
    // a host-API call (ORBextractor::operator(), what the Tracking thread makes) arrives on no stream -> Latency class; a device-API call arrives on the caller's stream
    // (PyTorch's default stream in bench.py) -> that stream's priority.  Measured (gpurun_out/r06g .. r06i, 64-frame step on PyTorch's default-priority stream): side
    // stream at the same (default) priority 0.2553 ms; a Latency-class side stream beside it 0.2665 ms (FAST level 0 and the blur overtake the main chain's seven dependent
    // resize launches); and a default-priority side stream created NEXT TO idle Latency-class streams of the same handle 0.43-0.64 ms, every stage twice as long -- one
    // handle's streams are all of one priority.
    bool evLastValid = false, capturing = false;   // evLastValid: a call has been enqueued on lastStream
    hipStream_t lastStream = nullptr;      // the stream of the previous call: compared, never dereferenced (its owner may have destroyed it)
    // Round 6: ordering between calls that come in on DIFFERENT streams without draining the device (under the reference's concurrency a drain makes the
    // Tracking thread wait for LocalMapping's whole bundle adjustment).  evLast is recorded behind a call's last kernel, a call on another stream waits for
    // it ON THE DEVICE (hipStreamWaitEvent).  An event record between two calls costs the stream a few microseconds, so a handle that only ever sees one
    // stream (the common case: the bench loop, one tracker) records nothing; the FIRST change of stream in a handle's life finds no event and drains once,
    // from then on every call leaves its event.  EAO_ORB_LAST_EVENT=always records from the first call on (no drain ever), =never is the rounds 1-5 drain.
    hipEvent_t evLast = nullptr;
    bool everyCallEvent = false, evLastRecorded = false;
    hipEvent_t evStart = nullptr, evFork[kLanes] = {}, evJoin[kLanes] = {}, evDone[kLanes] = {}, evFast0[kLanes] = {}, evMid[kLanes] = {};
    eao::DevBuf<Geom> d_geom;
    eao::DevBuf<CellDesc> d_cells;
    eao::DevBuf<uint8_t> d_pyr, d_blur, d_in;
    eao::DevBuf<unsigned> d_cellcand, d_cand, d_levelkps;
    eao::DevBuf<unsigned short> d_nodeof;
    eao::DevBuf<unsigned char> d_qtnodes;   // global node lists (only when they do not fit in LDS)
    eao::DevBuf<int> d_cellcnt, d_levelcnt, d_candcnt, d_nout;
    eao::DevBuf<eao_keypoint> d_kps;
    unsigned char* pinOut = nullptr;       // mapped pinned host memory: results of small host-API calls land here directly
    size_t pinOutCap = 0;
    unsigned char* pinUp = nullptr;        // pinned staging of small host-API uploads (upload_frames)
    size_t pinUpCap = 0;
    eao::DevBuf<uint8_t> d_desc;
    eao::DevBuf<float> d_xyr;
    eao::DevBuf<unsigned char> d_stereo;   // staging of eao_compute_stereo_matches
    unsigned char* pinPyr = nullptr;       // eao_orb_pyramid: the bordered levels of one frame in mapped pinned host memory
    size_t pinPyrCap = 0;
    eao_orb_level_view pyrViews[eao::orb::kMaxLevels] = {};
    int pyrFrame = -1, pyrBorder = -1;     // what pinPyr holds (of the last extraction; -1: nothing)
    int autoPyrBorder = -1;                // >= 0: single-frame host-API extractions export the bordered pyramid in the same stream pass (eao_orb_set_keep_pyramid)
    bool lastComplete = false;             // the last extraction was a synchronous host-API call: its products are final, nothing to wait for
    // last call (for stage taps)
    ImgSrc lastSrc{};
    int lastBatch = 0;
    bool profiling = false;
    long long* d_dbg = nullptr;   // EAO_DEBUG_STAMPS: per-level phase cycles of k_quadtree (diagnostic runs only)
    hipGraphExec_t graphExec = nullptr;
    GraphKey graphKey = {};
    std::vector<hipEvent_t> evs;   // kProfEvents events per profiled call, averaged by eao_orb_last_timing
    size_t evUsed = 0;
    // streaming host API (eao_orb_stream_*): a ring of pinned input / output slots, three streams (upload, extraction, download)
    struct StreamSlot {
        unsigned char* pinIn = nullptr; unsigned char* pinOut = nullptr;      // pinned host memory: frames in, [counts | keypoints | descriptors] out
        unsigned char* pinOutDev = nullptr;                                    // ... the output block as the device sees it (mapped)
        unsigned char* dIn = nullptr; unsigned char* dOut = nullptr;           // their device twins
        hipEvent_t evIn = nullptr, evDone = nullptr, evOut = nullptr;
        int batch = 0;
        bool submitted = false;
    };
    std::vector<StreamSlot> slots;
    hipStream_t sUp = nullptr, sRun = nullptr, sDown = nullptr;
    int sW = 0, sH = 0, sB = 0, sCap = 0, sPitch = 0;
    size_t sInBytes = 0, sOutBytes = 0, sOffK = 0, sOffD = 0;
};

namespace eao {
namespace orb {
// orb_host.hip
constexpr uintptr_t kNoCallerStream = ~(uintptr_t)0;      // a host-API call: the handle's streams are of the Latency class
int orb_last_event_mode();
eao_status ensure(eao_orb* h, int W, int H, int batch, hipStream_t caller = (hipStream_t)kNoCallerStream);
eao_status order_behind_last_call(eao_orb* h, hipStream_t st);
void stream_release(eao_orb* h);
eao_status wait_last_extraction(eao_orb* h);
eao_status upload_frames(eao_orb* h, const uint8_t* img, int width, int height, int stride, long long frame_stride, int batch);
// orb.hip
eao_status enqueue(eao_orb* h, const uint8_t* d_img, int pitch0, long long fs0, int batch, eao_keypoint* d_kps, uint8_t* d_desc, int cap, int* d_n, hipStream_t st);
eao_status enqueue_pyramid_export(eao_orb* h, int frame, int border);
// orb_quadtree.hip: one workgroup per (frame, level) of levels [lFirst, lFirst + nLev) of nb frames from frame f0 on
void launch_quadtree(const eao_orb* h, hipStream_t str, int nb, int f0, int lFirst, int nLev);
eao_status quadtree_reserve_lds(size_t bytes);      // raises the kernels' dynamic LDS limit (process-wide, only ever raised)
}  // namespace orb
}  // namespace eao
