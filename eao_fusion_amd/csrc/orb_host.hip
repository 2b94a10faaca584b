// orb_host.hip -- the eao_orb handle: geometry of a frame size, streams and events, ordering between calls, and the C API entry points of the extractor that launch
// no kernel themselves (those that do -- the streaming submit, the candidate tap, the stereo matcher -- sit beside the kernels in orb.hip).
// (Split from orb.hip in round 6, a pure move.)
#include "orb_internal.h"

using namespace eao::orb;

namespace eao {
namespace orb {

eao_status build_geometry(eao_orb* h, int W, int H) {
    const eao_orb_cfg& c = h->cfg;
    Geom& g = h->geom;
    std::memset(&g, 0, sizeof(g));
    g.nlevels = c.nlevels; g.W = W; g.H = H;
    g.iniTh = c.ini_th_fast; g.minTh = c.min_th_fast;
    for (int i = 0; i < 16; i++) g.umax[i] = h->umax[i];
    h->cells.clear();
    int off = 0, kpBase = 0, candBase = 0, tileBase = 0, maxCell = 0, scanCap = 0, maxList = 0, maxSw = 0, maxSh = 0;
    // first pass: level sizes and the largest FAST cell (fixes cellCap)
    for (int l = 0; l < c.nlevels; l++) {
        LevelGeom& L = g.L[l];
        const float s = h->invScale[l];
        L.w = cv_round((float)W * s);   // reference src/ORBextractor.cc:1111-1112
        L.h = cv_round((float)H * s);
        EAO_REQUIRE(L.w < 4096 && L.h < 4096, "level %d is %dx%d: coordinates are packed in 12 bits", l, L.w, L.h);
        const int maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
        const float width = (float)(maxBX - kMinBorder), height = (float)(maxBY - kMinBorder);
        const int nCols = (int)(width / (float)refc::FAST_CELL), nRows = (int)(height / (float)refc::FAST_CELL);
        EAO_REQUIRE(nCols >= 1 && nRows >= 1, "level %d (%dx%d) is smaller than one 30 px FAST cell plus borders", l, L.w, L.h);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        EAO_REQUIRE(wCell + 6 <= kTile && hCell + 6 <= kTile, "FAST cell %dx%d exceeds the LDS tile", wCell, hCell);
        maxCell = std::max(maxCell, ((wCell + 1) / 2) * ((hCell + 1) / 2));
    }
    g.cellCap = maxCell;  // 3x3 NMS with strict '>' keeps at most one corner per 2x2 block
    for (int l = 0; l < c.nlevels; l++) {
        LevelGeom& L = g.L[l];
        L.pitch = (L.w + 63) & ~63;
        L.off = off;
        off += ((L.pitch * ((L.h + kBlurRows - 1) / kBlurRows * kBlurRows)) + 255) & ~255;   // (padding rows: see k_blur7)
        const int maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
        const float width = (float)(maxBX - kMinBorder), height = (float)(maxBY - kMinBorder);
        const int nCols = (int)(width / (float)refc::FAST_CELL), nRows = (int)(height / (float)refc::FAST_CELL);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        L.cellBase = (int)h->cells.size();
        for (int i = 0; i < nRows; i++) {   // reference src/ORBextractor.cc:789-806
            const float iniY = (float)(kMinBorder + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(kMinBorder + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                CellDesc cd;
                cd.level = (short)l; cd.x0 = (short)iniX; cd.y0 = (short)iniY;
                cd.sw = (short)((int)maxX - (int)iniX); cd.sh = (short)((int)maxY - (int)iniY);
                cd.offX = (short)(j * wCell); cd.offY = (short)(i * hCell); cd.pad = 0;
                maxSw = std::max(maxSw, (int)cd.sw); maxSh = std::max(maxSh, (int)cd.sh);
                h->cells.push_back(cd);
            }
        }
        L.nCells = (int)h->cells.size() - L.cellBase;
        L.quota = h->quota[l];
        L.boxH = maxBY - kMinBorder;
        L.nIni = (int)std::round((float)(maxBX - kMinBorder) / (float)(maxBY - kMinBorder));  // :543
        EAO_REQUIRE(L.nIni >= 1 && L.nIni <= kMaxIni, "level %d aspect ratio gives %d initial quad-tree nodes (supported 1..%d)", l, L.nIni, kMaxIni);
        L.hX = (float)(maxBX - kMinBorder) / (float)L.nIni;
        L.listCap = std::max(L.quota + 3, 4 * L.nIni) + 1;
        EAO_REQUIRE(L.listCap < 65535, "quota too large");
        L.kpBase = kpBase; kpBase += L.listCap;
        L.candBase = candBase; L.candCap = L.nCells * g.cellCap; candBase += L.candCap;
        EAO_REQUIRE(L.candCap < (1 << 20), "level %d can hold %d FAST candidates; the quad-tree packs indices in 20 bits", l, L.candCap);
        L.tilesX = eao::cdiv(L.w, kBlurSegW);
        L.tileBase = tileBase; tileBase += eao::cdiv(L.tilesX * eao::cdiv(L.h, kBlurRows), kBlurStripsPerWg);
        L.scale = h->scale[l];
        L.scaledPatch = (int)(refc::PATCH_SIZE * h->scale[l]);
        scanCap = std::max(scanCap, std::max(L.listCap, L.nCells));
        maxList = std::max(maxList, L.listCap);
    }
    g.totalCells = (int)h->cells.size();
    g.totalKpCap = kpBase;
    g.totalCandCap = candBase;
    g.totalTiles = tileBase;
    g.pyrFrameBytes = off;
    // ---- fused pyramid: tile capacities by running the kernel's own range rules over every tile (host and device evaluate
    //      the same float / double expressions; -ffp-contract=off on both sides)
    {
        PyrArgs& P = h->pyr;
        std::memset(&P, 0, sizeof(P));
        h->pyrFused = false;
        static const bool envChain = getenv("EAO_ORB_PYRAMID") && !strcmp(getenv("EAO_ORB_PYRAMID"), "chain");
        if (c.nlevels > 1 && !envChain) {
            P.nlevels = c.nlevels; P.pyrFrameBytes = g.pyrFrameBytes;
            for (int l = 0; l < c.nlevels; l++) {
                P.L[l].w = g.L[l].w; P.L[l].h = g.L[l].h; P.L[l].pitch = g.L[l].pitch; P.L[l].off = g.L[l].off;
                if (l) { P.L[l].invX = 1. / ((double)g.L[l].w / g.L[l - 1].w); P.L[l].invY = 1. / ((double)g.L[l].h / g.L[l - 1].h); }
            }
            P.tilesX = eao::cdiv(g.L[1].w, kPyrTW); P.tilesY = eao::cdiv(g.L[1].h, kPyrTH);
            std::vector<int> capW(c.nlevels, 1), capH(c.nlevels, 1);
            const int nT = P.tilesX + P.tilesY;
            std::vector<int4> table((size_t)c.nlevels * nT, make_int4(0, 0, 0, 0));
            auto axis = [&](int tiles, int T, bool isX, std::vector<int>& capv) {
                for (int b = 0; b < tiles; b++) {
                    const int col = isX ? b : P.tilesX + b;
                    std::vector<int> o0(c.nlevels), o1(c.nlevels);
                    int b0 = std::min(b * T, isX ? g.L[1].w : g.L[1].h), b1 = std::min((b + 1) * T, isX ? g.L[1].w : g.L[1].h);
                    o0[1] = b0; o1[1] = b1;
                    for (int l = 2; l < c.nlevels; l++) {
                        const double inv = isX ? P.L[l].invX : P.L[l].invY;
                        const int sn = isX ? g.L[l - 1].w : g.L[l - 1].h, dn = isX ? g.L[l].w : g.L[l].h;
                        b0 = pyr_first_at_least(b0, inv, sn, dn); b1 = pyr_first_at_least(b1, inv, sn, dn);
                        o0[l] = b0; o1[l] = b1;
                    }
                    int r0 = o0[c.nlevels - 1], r1 = o1[c.nlevels - 1];
                    capv[c.nlevels - 1] = std::max(capv[c.nlevels - 1], r1 - r0);
                    for (int l = 1; l < c.nlevels; l++) { table[(size_t)l * nT + col].x = o0[l]; table[(size_t)l * nT + col].y = o1[l]; }
                    table[(size_t)(c.nlevels - 1) * nT + col].z = r0; table[(size_t)(c.nlevels - 1) * nT + col].w = r1;
                    for (int l = c.nlevels - 1; l >= 1; l--) {
                        int q0 = l > 1 ? o0[l - 1] : 0x7FFFFFFF, q1 = l > 1 ? o1[l - 1] : 0;
                        if (r1 > r0) {
                            const double inv = isX ? P.L[l].invX : P.L[l].invY;
                            const int sn = isX ? g.L[l - 1].w : g.L[l - 1].h;
                            q0 = std::min(q0, pyr_src_ofs(r0, inv, sn));
                            q1 = std::max(q1, std::min(pyr_src_ofs(r1 - 1, inv, sn) + 1, sn - 1) + 1);
                        }
                        if (q0 > q1) q0 = q1 = 0;
                        capv[l - 1] = std::max(capv[l - 1], q1 - q0);
                        table[(size_t)(l - 1) * nT + col].z = q0; table[(size_t)(l - 1) * nT + col].w = q1;
                        r0 = q0; r1 = q1;
                    }
                }
            };
            axis(P.tilesX, kPyrTW, true, capW);
            axis(P.tilesY, kPyrTH, false, capH);
            size_t lds = 0;
            int coefCap = 1;
            for (int l = 0; l < c.nlevels; l++) {
                P.L[l].ldsW = ((capW[l] + 3) & ~3) + 16; P.L[l].ldsH = capH[l] + 1;     // (+ slack: staging origin, three-word reads past the last pixel)
                P.L[l].ldsOff = (int)lds;
                lds += ((size_t)P.L[l].ldsW * P.L[l].ldsH + 15) & ~(size_t)15;
                coefCap = std::max(coefCap, std::max(capW[l], capH[l]));
            }
            coefCap = (coefCap + 7) & ~3;
            P.coefOff = (int)lds; P.coefCap = coefCap;
            lds += (size_t)coefCap * 16;
            h->pyrLds = lds;
            // (extreme level counts, and scale factors beyond 2 -- four pixels then read more than three source words -- keep
            //  the chain of k_resize launches)
            h->pyrFused = lds <= 64 * 1024 && h->cfg.scale_factor <= 2.0f;
            for (int l = 1; l < c.nlevels; l++)
                if (P.L[l].invX > 2.0 || P.L[l].invY > 2.0) h->pyrFused = false;
            if (h->pyrFused) {
                std::vector<int2> cxv, cyv;
                for (int l = 1; l < c.nlevels; l++) {
                    P.L[l].coefXOff = (int)cxv.size(); P.L[l].coefYOff = (int)cyv.size();
                    for (int d = 0; d < g.L[l].w; d++) { int o; unsigned w2; resize_coef_host(d, P.L[l].invX, g.L[l - 1].w, true, &o, &w2); cxv.push_back(make_int2(o, (int)w2)); }
                    for (int d = 0; d < g.L[l].h; d++) { int o; unsigned w2; resize_coef_host(d, P.L[l].invY, g.L[l - 1].h, false, &o, &w2); cyv.push_back(make_int2(o, (int)w2)); }
                }
                eao_status st2 = h->d_pyrRanges.reserve(table.size());
                if (st2) return st2;
                if ((st2 = h->d_pyrCoef.reserve(cxv.size() + cyv.size()))) return st2;
                EAO_HIP(hipMemcpy(h->d_pyrRanges.p, table.data(), table.size() * sizeof(int4), hipMemcpyHostToDevice));
                EAO_HIP(hipMemcpy(h->d_pyrCoef.p, cxv.data(), cxv.size() * sizeof(int2), hipMemcpyHostToDevice));
                EAO_HIP(hipMemcpy(h->d_pyrCoef.p + cxv.size(), cyv.data(), cyv.size() * sizeof(int2), hipMemcpyHostToDevice));
                P.ranges = h->d_pyrRanges.p;
                P.coefX = h->d_pyrCoef.p; P.coefY = h->d_pyrCoef.p + cxv.size();
            }
        }
    }
    g.scanCap = scanCap;
    g.fastMaxTested = std::max(1, (maxSw - 6) * (maxSh - 6));
    // tile | work list u16[maxT] (+ the second list, stacked from its end) | score map | 16-word survivor-position table
    g.fastStride = maxSw + 3 <= 48 ? 48 : kTileStrideWide;
    g.fastTileBytes = ((maxSh * g.fastStride) + 15) & ~15;
    g.fastLdsBytes = g.fastTileBytes + ((2 * g.fastMaxTested + 15) & ~15) + ((((maxSw - 4) * (maxSh - 4)) + 15) & ~15) + 64;
    // k_quadtree dynamic LDS: 2 short4 + 2 cnt + 2 crk + 4 childcnt + 4 childpos + newpos/order/vlist/procRank/scanB per entry + scanA
    h->quadLds = (size_t)maxList * (2 * sizeof(short4) + sizeof(int) * kQtNodeInts) + (size_t)scanCap * sizeof(int);
    g.qtNodesGlobal = 0; g.qtNodeFrameBytes = 0;
    if (h->quadLds > 100 * 1024) {   // (thousands of features on one level: upstream takes any N, src/ORBextractor.cc:539)
        g.qtNodesGlobal = 1;
        long long noff = 0;
        for (int l = 0; l < c.nlevels; l++) {
            g.L[l].nodeOff = noff;
            noff += (((long long)g.L[l].listCap * (2 * sizeof(short4) + sizeof(int) * kQtNodeInts) + (long long)scanCap * sizeof(int)) + 255) & ~255LL;
        }
        g.qtNodeFrameBytes = noff;
        h->quadLds = 0;
    }
    // candidate keys (4 B) + node indices (2 B) in LDS while two workgroups still fit a CU
    h->quadLds = (h->quadLds + 15) & ~(size_t)15;
    g.qtKeysOff = (int)h->quadLds;
    // Capacity: about four candidates per requested feature (the 640 x 480 benchmark frames leave ~3100 candidates on level 0
    // for 1000 features; a level that holds more falls back to the global arrays).  With the level-major dispatch the
    // smaller footprint pays: three workgroups per CU instead of two -- k_quadtree alone 210 -> 169 us at batch 256 (8192 ->
    // 4096 candidates; 3200: 143 us, but then the blur beside it is the longer of the two).
    static const size_t kQtLdsCandEnv = getenv("EAO_QT_LDS_CAND") ? (size_t)atoi(getenv("EAO_QT_LDS_CAND")) : 0;
    const size_t kQtLdsCand = kQtLdsCandEnv ? kQtLdsCandEnv : std::min<size_t>(8192, std::max<size_t>(2048, ((size_t)h->cfg.nfeatures * 4 + 63) & ~(size_t)63));
    g.qtLdsCand = (int)std::min<size_t>(kQtLdsCand, h->quadLds < 76 * 1024 ? (76 * 1024 - h->quadLds) / 6 : 0) & ~7;
    h->quadLds += (size_t)g.qtLdsCand * 6;
    if (g.qtNodesGlobal) { g.qtLdsCand = 0; h->quadLds = 0; }     // (everything in the global workspace)
    // the blur kernel hard-codes the taps; make sure the published construction gives them
    {
        float cf[7]; double sum = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3; cf[i] = (float)std::exp(-0.5 / 4.0 * x * x); sum += cf[i]; }
        const int expect[7] = {18, 34, 49, 55, 49, 34, 18};
        for (int i = 0; i < 7; i++)
            if (cv_round((double)(float)(cf[i] * (1. / sum)) * 256.0) != expect[i]) { eao::set_error("gaussian taps mismatch"); return EAO_ERR_INTERNAL; }
    }
    { eao_status st = h->d_geom.reserve(1); if (st) return st; }
    EAO_HIP(hipMemcpyAsync(h->d_geom.p, &g, sizeof(Geom), hipMemcpyHostToDevice, h->stream));
    { eao_status st = h->d_cells.reserve(h->cells.size()); if (st) return st; }
    EAO_HIP(hipMemcpyAsync(h->d_cells.p, h->cells.data(), h->cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    { eao_status sq = quadtree_reserve_lds(h->quadLds); if (sq) return sq; }
    h->geomValid = true;
    h->batchCap = 0;
    if (h->graphExec) { (void)hipGraphExecDestroy(h->graphExec); h->graphExec = nullptr; }
    return EAO_OK;
}

// EAO_ORB_LAST_EVENT: 0 = never (drain on a change of stream), 1 = adaptive (default), 2 = always
int orb_last_event_mode() {
    static const int mode = [] {
        const char* e = getenv("EAO_ORB_LAST_EVENT");
        if (!e) return 1;
        if (!strcmp(e, "always")) return 2;
        if (!strcmp(e, "never")) return 0;
        return 1;
    }();
    return mode;
}

eao_status ensure(eao_orb* h, int W, int H, int batch, hipStream_t caller) {
    eao_status st = eao::require_device();
    if (st) return st;
    if (!h->stream && batch > 0) {      // (batch 0: a geometry query -- eao_orb_max_keypoints -- needs no stream, and must not decide the handle's priority)
        // the priority of the handle's streams: see eao_orb (round 6)
        int least = 0, greatest = 0, p = 0;
        const bool follow = (uintptr_t)caller != kNoCallerStream && !(getenv("EAO_STREAM_PRIORITY") && !atoi(getenv("EAO_STREAM_PRIORITY"))) &&
                            hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest && hipStreamGetPriority(caller, &p) == hipSuccess;
        (void)hipGetLastError();
        auto make = [&](hipStream_t* q) -> hipError_t {
            if ((uintptr_t)caller == kNoCallerStream) return eao::create_stream(q, eao::StreamClass::Latency);
            if (!follow || p == 0) return hipStreamCreateWithFlags(q, hipStreamNonBlocking);
            return hipStreamCreateWithPriority(q, hipStreamNonBlocking, p);
        };
        EAO_HIP(make(&h->stream));
        EAO_HIP(hipEventCreateWithFlags(&h->evStart, hipEventDisableTiming));
        EAO_HIP(hipEventCreateWithFlags(&h->evLast, hipEventDisableTiming));
        h->everyCallEvent = orb_last_event_mode() == 2;
        // (HIP maps its streams onto a handful of hardware queues, and streams that share one execute in submission order: only the
        //  lanes the schedule can use get streams -- lane 0 unless EAO_ORB_LANES asks for more -- so that the streaming API's upload
        //  stream does not end up behind the extraction's side stream)
        const int lanesWanted = std::max(1, std::min(eao_orb::kLanes, getenv("EAO_ORB_LANES") ? atoi(getenv("EAO_ORB_LANES")) : 1));
        for (int i = 0; i < eao_orb::kLanes; i++) {
            if (i < lanesWanted) {
                EAO_HIP(make(&h->laneMain[i]));
                EAO_HIP(make(&h->laneSide[i]));
            }
            EAO_HIP(hipEventCreateWithFlags(&h->evFork[i], hipEventDisableTiming));
            EAO_HIP(hipEventCreateWithFlags(&h->evFast0[i], hipEventDisableTiming));
            EAO_HIP(hipEventCreateWithFlags(&h->evMid[i], hipEventDisableTiming));
            EAO_HIP(hipEventCreateWithFlags(&h->evJoin[i], hipEventDisableTiming));
            EAO_HIP(hipEventCreateWithFlags(&h->evDone[i], hipEventDisableTiming));
        }
    }
    if (!h->d_dbg && getenv("EAO_DEBUG_STAMPS")) {
        EAO_HIP(hipMalloc(&h->d_dbg, 16 * kMaxLevels * sizeof(long long)));
        EAO_HIP(hipMemset(h->d_dbg, 0, 16 * kMaxLevels * sizeof(long long)));
    }
    if (!h->geomValid || h->geom.W != W || h->geom.H != H) {
        st = build_geometry(h, W, H);
        if (st) return st;
    }
    if (batch > h->batchCap) {
        const Geom& g = h->geom;
        const size_t B = batch;
#define RES(buf, cnt) do { st = h->buf.reserve(cnt); if (st) return st; } while (0)
        RES(d_pyr, B * g.pyrFrameBytes);
        RES(d_blur, B * g.pyrFrameBytes);
        RES(d_cellcand, B * (size_t)g.totalCells * g.cellCap);
        RES(d_cellcnt, B * g.totalCells);
        RES(d_cand, B * (size_t)g.totalCandCap);
        RES(d_nodeof, B * (size_t)g.totalCandCap);
        if (g.qtNodesGlobal) RES(d_qtnodes, B * (size_t)g.qtNodeFrameBytes);
        RES(d_levelkps, B * (size_t)g.totalKpCap);
        RES(d_levelcnt, B * g.nlevels);
        RES(d_candcnt, B * g.nlevels);
#undef RES
        h->batchCap = batch;
        if (h->graphExec) { (void)hipGraphExecDestroy(h->graphExec); h->graphExec = nullptr; }   // buffers moved
    }
    return EAO_OK;
}

// A call that comes in on another stream than the handle's previous call shares its pyramid / candidate scratch with it: it runs behind the previous
// call's event on the device, or -- when that call left none (see eao_orb::evLast) -- behind a drain, after which every call of this handle leaves one.
// The previous stream itself is never touched (its owner may have destroyed it).
eao_status order_behind_last_call(eao_orb* h, hipStream_t st) {
    if (h->evLastRecorded) EAO_HIP(hipStreamWaitEvent(st, h->evLast, 0));
    else EAO_HIP(hipDeviceSynchronize());
    if (orb_last_event_mode() != 0) h->everyCallEvent = true;
    return EAO_OK;
}

void stream_release(eao_orb* h) {
    for (eao_orb::StreamSlot& sl : h->slots) {
        if (sl.evOut && sl.submitted) (void)hipEventSynchronize(sl.evOut);
        if (sl.pinIn) (void)hipHostFree(sl.pinIn);
        if (sl.pinOut) (void)hipHostFree(sl.pinOut);
        if (sl.dIn) (void)hipFree(sl.dIn);
        if (sl.dOut) (void)hipFree(sl.dOut);
        if (sl.evIn) (void)hipEventDestroy(sl.evIn);
        if (sl.evOut) (void)hipEventDestroy(sl.evOut);
        if (sl.evDone) (void)hipEventDestroy(sl.evDone);
    }
    h->slots.clear();
    for (hipStream_t* q : {&h->sUp, &h->sRun, &h->sDown})
        if (*q) { (void)hipStreamSynchronize(*q); (void)hipStreamDestroy(*q); *q = nullptr; }
}

// Readers of the last extraction's products (pyramid levels, candidate taps, the stereo matcher).  After a host-API call (eao_orb_extract / _batch: what the
// class-surface adapter makes) the products are final when the call returns -- nothing to wait for, and in particular no device-wide synchronise that would
// make the Tracking thread wait for the LocalMapping thread's bundle adjustment.  Only after a DEVICE-API call on a caller's stream, which
// the library must not touch again (its owner may have destroyed it, see enqueue_direct) and on which recording an event per call was measured at ~5 us, does
// the reader drain the device.
eao_status wait_last_extraction(eao_orb* h) {
    if (!h->lastComplete) {
        if (h->evLastRecorded) EAO_HIP(hipEventSynchronize(h->evLast));      // (round 6: the call left its event)
        else {
            EAO_HIP(hipDeviceSynchronize());
            if (orb_last_event_mode() != 0) h->everyCallEvent = true;           // this handle has readers behind device-API calls: later calls leave an event
        }
        h->lastComplete = true;
    }
    return EAO_OK;
}

// The caller's frames (pageable memory) to h->d_in in the device pitch.  Small uploads -- the per-frame calls -- go through a pinned staging buffer of the handle: the
// runtime's own path for pageable memory registers the caller's pages with the driver for the duration of the copy, and a registered range that the kernel touches
// meanwhile (another thread's munmap, page migration) has the driver take every queue of the process off the GPU for a millisecond or more -- one tracked frame in a
// thousand took 6 - 12 ms beside a looping LocalBundleAdjustment.  The staging copy costs ~15 us per 640 x 480 frame.  EAO_ORB_PINNED_IN=0: the runtime's path (A/B runs).
constexpr size_t kPinnedInMax = 8u << 20;
eao_status upload_frames(eao_orb* h, const uint8_t* img, int width, int height, int stride, long long frame_stride, int batch) {
    const Geom& g = h->geom;
    const long long fs0 = (long long)g.L[0].pitch * height;
    const size_t bytes = (size_t)batch * (size_t)fs0;
    static const bool envNoPinned = getenv("EAO_ORB_PINNED_IN") && !atoi(getenv("EAO_ORB_PINNED_IN"));
    if (bytes <= kPinnedInMax && !envNoPinned) {
        if (h->pinUpCap < bytes) {
            if (h->pinUp) (void)hipHostFree(h->pinUp);
            h->pinUp = nullptr; h->pinUpCap = 0;
            EAO_HIP(hipHostMalloc((void**)&h->pinUp, bytes, hipHostMallocDefault));
            h->pinUpCap = bytes;
        }
        if (stride == g.L[0].pitch && (batch == 1 || frame_stride == fs0)) std::memcpy(h->pinUp, img, bytes);
        else
            for (int f = 0; f < batch; f++)
                for (int y = 0; y < height; y++) std::memcpy(h->pinUp + f * fs0 + (size_t)y * g.L[0].pitch, img + (long long)f * frame_stride + (size_t)y * stride, (size_t)width);
        EAO_HIP(hipMemcpyAsync(h->d_in.p, h->pinUp, bytes, hipMemcpyHostToDevice, h->stream));
        return EAO_OK;
    }
    if (stride == g.L[0].pitch && (batch == 1 || frame_stride == fs0)) {     // contiguous frames: one linear copy
        EAO_HIP(hipMemcpyAsync(h->d_in.p, img, bytes, hipMemcpyHostToDevice, h->stream));
    } else {
        for (int f = 0; f < batch; f++)
            EAO_HIP(hipMemcpy2DAsync(h->d_in.p + f * fs0, g.L[0].pitch, img + (long long)f * frame_stride, stride, width, height, hipMemcpyHostToDevice, h->stream));
    }
    return EAO_OK;
}

}  // namespace orb
}  // namespace eao

extern "C" {


eao_status eao_orb_create(const eao_orb_cfg* cfg, eao_orb** out) {
    EAO_REQUIRE(cfg && out, "null argument");
    EAO_REQUIRE(cfg->nlevels >= 1 && cfg->nlevels <= kMaxLevels, "nlevels must be in 1..%d", kMaxLevels);
    EAO_REQUIRE(cfg->nfeatures >= 1 && cfg->scale_factor > 1.0f, "need nfeatures >= 1 and scale_factor > 1");
    eao_status st = eao::require_device();
    if (st) return st;
    eao_orb* h = new eao_orb();
    h->cfg = *cfg;
    const int nl = cfg->nlevels;
    const double scaleFactor = cfg->scale_factor;  // the reference keeps this member as a double (include/ORBextractor.h:97)
    h->scale.resize(nl); h->sigma2.resize(nl); h->invScale.resize(nl); h->invSigma2.resize(nl); h->quota.resize(nl);
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        h->invScale[i] = 1.0f / h->scale[i];
        h->invSigma2[i] = 1.0f / h->sigma2[i];
    }
    const float factor = (float)(1.0f / scaleFactor);
    float desired = cfg->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        h->quota[l] = cv_round(desired);
        sum += h->quota[l];
        desired *= factor;
    }
    h->quota[nl - 1] = std::max(cfg->nfeatures - sum, 0);
    {   // end of each row of the radius-15 disc (reference src/ORBextractor.cc:455-469)
        const int vmax = (int)std::floor(15 * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(15 * std::sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; ++v) h->umax[v] = cv_round(std::sqrt(225.0 - v * v));
        for (int v = 15, v0 = 0; v >= vmin; --v) {
            while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
            h->umax[v] = v0;
            ++v0;
        }
    }
    {
        const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
        for (int i = 0; i < 16; i++)
            if (h->umax[i] != expect[i]) { delete h; eao::set_error("umax table mismatch"); return EAO_ERR_INTERNAL; }
    }
    *out = h;
    return EAO_OK;
}

void eao_orb_destroy(eao_orb* h) {
    if (!h) return;
    for (hipEvent_t e : h->evs) if (e) (void)hipEventDestroy(e);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (int i = 0; i < eao_orb::kLanes; i++) {
        if (h->laneMain[i]) { (void)hipStreamSynchronize(h->laneMain[i]); (void)hipStreamDestroy(h->laneMain[i]); }
        if (h->laneSide[i]) { (void)hipStreamSynchronize(h->laneSide[i]); (void)hipStreamDestroy(h->laneSide[i]); }
        if (h->evFork[i]) (void)hipEventDestroy(h->evFork[i]);
        if (h->evFast0[i]) (void)hipEventDestroy(h->evFast0[i]);
        if (h->evMid[i]) (void)hipEventDestroy(h->evMid[i]);
        if (h->evJoin[i]) (void)hipEventDestroy(h->evJoin[i]);
        if (h->evDone[i]) (void)hipEventDestroy(h->evDone[i]);
    }
    if (h->pinOut) (void)hipHostFree(h->pinOut);
    if (h->pinUp) (void)hipHostFree(h->pinUp);
    if (h->pinPyr) (void)hipHostFree(h->pinPyr);
    stream_release(h);
    if (h->evStart) (void)hipEventDestroy(h->evStart);
    if (h->evLast) (void)hipEventDestroy(h->evLast);
    if (h->graphExec) (void)hipGraphExecDestroy(h->graphExec);
    delete h;
}

eao_status eao_orb_tables(const eao_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl) {
    EAO_REQUIRE(h, "null handle");
    for (int i = 0; i < h->cfg.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->invScale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->invSigma2[i];
        if (fpl) fpl[i] = h->quota[i];
    }
    return EAO_OK;
}

eao_status eao_orb_max_keypoints(eao_orb* h, int32_t width, int32_t height, int32_t* cap) {
    EAO_REQUIRE(h && cap, "null argument");
    eao_status st = ensure(h, width, height, 0);
    if (st) return st;
    *cap = h->geom.totalKpCap;
    return EAO_OK;
}

eao_status eao_orb_extract_batch_device(eao_orb* h, const uint8_t* d_img, int32_t width, int32_t height, int32_t stride,
                                        int64_t frame_stride, int32_t batch, eao_keypoint* d_kps, uint8_t* d_desc, int32_t cap,
                                        int32_t* d_n, void* stream) {
    EAO_REQUIRE(h && d_img && d_kps && d_desc && d_n, "null argument");
    EAO_REQUIRE(width > 0 && height > 0 && stride >= width && batch >= 1, "bad image geometry");
    eao_status st = ensure(h, width, height, batch, (hipStream_t)stream);
    if (st) return st;
    if (cap < h->geom.totalKpCap) {
        eao::set_error("cap %d < eao_orb_max_keypoints %d", cap, h->geom.totalKpCap);
        return EAO_ERR_CAPACITY;
    }
    // exactly the caller's stream: NULL is the (legacy) null stream, as in the Hamming entry points -- torch's default stream
    // among others; work the caller enqueues behind this call on that stream is ordered behind the extraction
    return enqueue(h, d_img, stride, frame_stride, batch, d_kps, d_desc, cap, d_n, (hipStream_t)stream);
}

eao_status eao_orb_extract_batch(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride, int64_t frame_stride,
                                 int32_t batch, eao_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n) {
    EAO_REQUIRE(h && n, "null argument");
    if (!img || width <= 0 || height <= 0) {  // empty image: outputs untouched (reference :1046-1047)
        for (int f = 0; f < std::max(batch, 0); f++) n[f] = 0;
        return EAO_OK;
    }
    EAO_REQUIRE(kps && desc && stride >= width && batch >= 1, "bad argument");
    eao_status st = ensure(h, width, height, batch);
    if (st) return st;
    const Geom& g = h->geom;
    if (cap < g.totalKpCap) {
        eao::set_error("cap %d < eao_orb_max_keypoints %d", cap, g.totalKpCap);
        return EAO_ERR_CAPACITY;
    }
    const size_t B = batch;
    if ((st = h->d_in.reserve(B * (size_t)g.L[0].pitch * height))) return st;
    if ((st = h->d_kps.reserve(B * (size_t)cap))) return st;
    if ((st = h->d_desc.reserve(B * (size_t)cap * 32))) return st;
    if ((st = h->d_nout.reserve(B))) return st;
    const long long fs0 = (long long)g.L[0].pitch * height;
    if ((st = upload_frames(h, img, width, height, stride, frame_stride, batch))) return st;
    // Small calls (the per-frame latency path): the last kernel writes keypoints, descriptors and counts straight into mapped
    // pinned host memory -- ~70 KB per frame over PCIe -- and the rows that exist are copied to the caller's (pageable) arrays
    // after the one synchronisation: three pageable device-to-host copies (~15 us each) gone.  Large batches keep the DMA path.
    const size_t outBytes = B * sizeof(int) + 64 + B * (size_t)cap * (sizeof(eao_keypoint) + 32);
    static const bool envNoPinned = getenv("EAO_ORB_PINNED_OUT") && !atoi(getenv("EAO_ORB_PINNED_OUT"));      // (A/B switch)
    if (outBytes <= kPinnedOutMax && !envNoPinned) {
        if (h->pinOutCap < outBytes) {
            if (h->pinOut) (void)hipHostFree(h->pinOut);
            h->pinOut = nullptr; h->pinOutCap = 0;
            EAO_HIP(hipHostMalloc((void**)&h->pinOut, outBytes, hipHostMallocMapped));
            h->pinOutCap = outBytes;
        }
        unsigned char* dv = nullptr;
        EAO_HIP(hipHostGetDevicePointer((void**)&dv, h->pinOut, 0));
        const size_t offK = (B * sizeof(int) + 63) & ~(size_t)63, offD = offK + B * (size_t)cap * sizeof(eao_keypoint);
        st = enqueue(h, h->d_in.p, g.L[0].pitch, fs0, batch, (eao_keypoint*)(dv + offK), dv + offD, cap, (int*)dv, h->stream);
        if (st) return st;
        EAO_HIP(eao::wait_latency(h->stream));
        h->lastComplete = true;
        const int* hn = (const int*)h->pinOut;
        for (int f = 0; f < batch; f++) {
            const int nf = std::min(std::max(hn[f], 0), cap);
            n[f] = nf;
            std::memcpy(kps + (size_t)f * cap, h->pinOut + offK + (size_t)f * cap * sizeof(eao_keypoint), (size_t)nf * sizeof(eao_keypoint));
            std::memcpy(desc + (size_t)f * cap * 32, h->pinOut + offD + (size_t)f * cap * 32, (size_t)nf * 32);
        }
        return EAO_OK;
    }
    st = enqueue(h, h->d_in.p, g.L[0].pitch, fs0, batch, h->d_kps.p, h->d_desc.p, cap, h->d_nout.p, h->stream);
    if (st) return st;
    EAO_HIP(hipMemcpyAsync(n, h->d_nout.p, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(hipMemcpyAsync(kps, h->d_kps.p, B * cap * sizeof(eao_keypoint), hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(hipMemcpyAsync(desc, h->d_desc.p, B * (size_t)cap * 32, hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    h->lastComplete = true;
    return EAO_OK;
}

eao_status eao_orb_extract(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride, eao_keypoint* kps,
                           uint8_t* desc, int32_t cap, int32_t* n) {
    return eao_orb_extract_batch(h, img, width, height, stride, 0, 1, kps, desc, cap, n);
}

// ---- streaming host API ------------------------------------------------------------------------------------------------
// What a sequence reader / Tracking thread feeds are HOST images (src/Frame.cc:616-622 behind the Frame constructors,
// src/Frame.cc:192-194).  eao_orb_extract_batch serves one call at a time from pageable memory: upload, extraction and download
// follow each other (0.83 ms per 64 frames against 0.26 ms of extraction).  Here the handle owns a ring of PINNED slots: the
// producer writes frames straight into a slot (a decoder's / camera driver's output buffer; cv::Mat can wrap it), submit() is
// asynchronous, and the upload of slot k + 1 and the download of slot k - 1 overlap the extraction of slot k on three streams.
// Results are identical to eao_orb_extract_batch's (tests/test_gpu_orb.py).
eao_status eao_orb_stream_create(eao_orb* h, int32_t width, int32_t height, int32_t batch, int32_t nslots) {
    EAO_REQUIRE(h && width > 0 && height > 0 && batch >= 1 && nslots >= 1 && nslots <= 8, "bad argument (1..8 slots)");
    eao_status st = ensure(h, width, height, batch);
    if (st) return st;
    stream_release(h);
    const Geom& g = h->geom;
    h->sW = width; h->sH = height; h->sB = batch; h->sCap = g.totalKpCap; h->sPitch = g.L[0].pitch;
    const size_t B = batch, cap = h->sCap;
    h->sInBytes = B * (size_t)h->sPitch * height;
    h->sOffK = (B * sizeof(int) + 255) & ~(size_t)255;
    h->sOffD = (h->sOffK + B * cap * sizeof(eao_keypoint) + 255) & ~(size_t)255;
    h->sOutBytes = (h->sOffD + B * cap * 32 + 15) & ~(size_t)15;
    // Three plain streams.  HIP maps streams onto a few hardware queues and streams that share one run in submission order: with the
    // handle's eight unused lane streams in the way the upload stream shared a queue with the extraction's side stream and the next
    // slot's upload started ~100 us into the current extraction (0.50 ms per 64 frames instead of 0.42; the lane streams are now
    // created on demand).  Stream PRIORITIES make it worse on this runtime (upload high: 0.44; download or extraction low: 0.83-0.88).
    EAO_HIP(eao::create_stream(&h->sUp, eao::StreamClass::Latency));
    EAO_HIP(eao::create_stream(&h->sRun, eao::StreamClass::Latency));
    EAO_HIP(eao::create_stream(&h->sDown, eao::StreamClass::Latency));
    h->slots.resize(nslots);
    for (eao_orb::StreamSlot& sl : h->slots) {
        EAO_HIP(hipHostMalloc((void**)&sl.pinIn, h->sInBytes, hipHostMallocDefault));
        EAO_HIP(hipHostMalloc((void**)&sl.pinOut, h->sOutBytes, hipHostMallocMapped));
        EAO_HIP(hipHostGetDevicePointer((void**)&sl.pinOutDev, sl.pinOut, 0));
        EAO_HIP(hipMalloc((void**)&sl.dIn, h->sInBytes));
        EAO_HIP(hipMalloc((void**)&sl.dOut, h->sOutBytes));
        EAO_HIP(hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming));
        EAO_HIP(hipEventCreateWithFlags(&sl.evOut, hipEventDisableTiming));
        EAO_HIP(hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming));
        std::memset(sl.pinOut, 0, h->sOutBytes);
    }
    return EAO_OK;
}

eao_status eao_orb_stream_slot(eao_orb* h, int32_t slot, eao_orb_slot* out) {
    EAO_REQUIRE(h && out && slot >= 0 && slot < (int)h->slots.size(), "no such slot (eao_orb_stream_create first)");
    const eao_orb::StreamSlot& sl = h->slots[slot];
    out->frames = sl.pinIn; out->stride = h->sPitch; out->frame_stride = (int64_t)h->sPitch * h->sH;
    out->n = (int32_t*)sl.pinOut; out->kps = (eao_keypoint*)(sl.pinOut + h->sOffK); out->desc = sl.pinOut + h->sOffD; out->cap = h->sCap;
    return EAO_OK;
}

eao_status eao_orb_stream_wait(eao_orb* h, int32_t slot) {
    EAO_REQUIRE(h && slot >= 0 && slot < (int)h->slots.size(), "no such slot (eao_orb_stream_create first)");
    eao_orb::StreamSlot& sl = h->slots[slot];
    EAO_REQUIRE(sl.submitted, "slot %d was not submitted", slot);
    EAO_HIP(hipEventSynchronize(sl.evOut));
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}

eao_status eao_orb_level(eao_orb* h, int32_t frame, int32_t level, int32_t which, int32_t* w, int32_t* hgt, uint8_t* dst) {
    EAO_REQUIRE(h && h->geomValid && h->lastBatch > 0, "no extraction has run on this handle");
    EAO_REQUIRE(frame >= 0 && frame < h->lastBatch && level >= 0 && level < h->geom.nlevels, "frame/level out of range");
    const LevelGeom& L = h->geom.L[level];
    if (w) *w = L.w;
    if (hgt) *hgt = L.h;
    if (!dst) return EAO_OK;
    const uint8_t* src;
    int pitch;
    if (which == 0 && level == 0) {
        src = h->lastSrc.img0 + (long long)frame * h->lastSrc.fs0;
        pitch = h->lastSrc.pitch0;
    } else {
        src = (which ? h->d_blur.p : h->d_pyr.p) + (long long)frame * h->geom.pyrFrameBytes + L.off;
        pitch = L.pitch;
    }
    eao_status st = wait_last_extraction(h);
    if (st) return st;
    EAO_HIP(hipMemcpy2DAsync(dst, L.w, src, pitch, L.w, L.h, hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    return EAO_OK;
}

eao_status eao_orb_pyramid(eao_orb* h, int32_t frame, int32_t border, eao_orb_level_view* levels) {
    EAO_REQUIRE(h && levels && h->geomValid && h->lastBatch > 0, "no extraction has run on this handle");
    EAO_REQUIRE(frame >= 0 && frame < h->lastBatch && border >= 0 && border <= 64, "frame / border out of range");
    if (!(h->lastComplete && h->pyrFrame == frame && h->pyrBorder == border)) {      // (else: exported behind the extraction itself, eao_orb_set_keep_pyramid)
        eao_status st = wait_last_extraction(h);
        if (!st) st = enqueue_pyramid_export(h, frame, border);
        if (st) return st;
        EAO_HIP(eao::wait_latency(h->stream));
        EAO_HIP(hipGetLastError());
    }
    for (int l = 0; l < h->geom.nlevels; l++) levels[l] = h->pyrViews[l];
    return EAO_OK;
}

eao_status eao_orb_set_keep_pyramid(eao_orb* h, int32_t border) {
    EAO_REQUIRE(h && border >= -1 && border <= 64, "border: -1 (off) .. 64");
    h->autoPyrBorder = border;
    return EAO_OK;
}

eao_status eao_orb_extract_ref(eao_orb* h, const uint8_t* img, int32_t width, int32_t height, int32_t stride, const eao_keypoint** kps,
                               const uint8_t** desc, int32_t* n) {
    EAO_REQUIRE(h && kps && desc && n, "null argument");
    *kps = nullptr; *desc = nullptr; *n = 0;
    if (!img || width <= 0 || height <= 0) return EAO_OK;      // empty image, as eao_orb_extract
    EAO_REQUIRE(stride >= width, "bad argument");
    eao_status st = ensure(h, width, height, 1);
    if (st) return st;
    const Geom& g = h->geom;
    const int cap = g.totalKpCap;
    const size_t offK = 64, offD = offK + (size_t)cap * sizeof(eao_keypoint), outBytes = offD + (size_t)cap * 32;
    if (h->pinOutCap < outBytes) {
        if (h->pinOut) (void)hipHostFree(h->pinOut);
        h->pinOut = nullptr; h->pinOutCap = 0;
        EAO_HIP(hipHostMalloc((void**)&h->pinOut, outBytes, hipHostMallocMapped));
        h->pinOutCap = outBytes;
    }
    unsigned char* dv = nullptr;
    EAO_HIP(hipHostGetDevicePointer((void**)&dv, h->pinOut, 0));
    if ((st = h->d_in.reserve((size_t)g.L[0].pitch * height))) return st;
    const long long fs0 = (long long)g.L[0].pitch * height;
    if ((st = upload_frames(h, img, width, height, stride, 0, 1))) return st;
    st = enqueue(h, h->d_in.p, g.L[0].pitch, fs0, 1, (eao_keypoint*)(dv + offK), dv + offD, cap, (int*)dv, h->stream);
    if (st) return st;
    if (h->autoPyrBorder >= 0 && (st = enqueue_pyramid_export(h, 0, h->autoPyrBorder))) return st;      // same stream, same synchronisation
    EAO_HIP(eao::wait_latency(h->stream));
    h->lastComplete = true;
    *n = std::min(std::max(*(const int*)h->pinOut, 0), cap);
    *kps = (const eao_keypoint*)(h->pinOut + offK);
    *desc = h->pinOut + offD;
    return EAO_OK;
}

int32_t eao_orb_lanes(int32_t batch) { return batch < eao_orb::kLanes ? (batch > 0 ? batch : 1) : eao_orb::kLanes; }

eao_status eao_orb_set_profiling(eao_orb* h, int32_t on) {
    EAO_REQUIRE(h, "null handle");
    h->profiling = on != 0;
    h->evUsed = 0;
    return EAO_OK;
}

eao_status eao_orb_last_timing(eao_orb* h, float ms[6]) {
    EAO_REQUIRE(h && ms && h->evUsed >= kProfEvents, "no profiled call since eao_orb_set_profiling(h, 1)");
    const size_t calls = h->evUsed / kProfEvents;
    EAO_HIP(hipEventSynchronize(h->evs[h->evUsed - kProfEvents + 8]));   // ev[8] of the last call
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (size_t c = 0; c < calls; c++) {
        hipEvent_t* ev = &h->evs[c * kProfEvents];
        float t;
        // stage intervals of slice 0 (one of the concurrently running sub-batches)
        EAO_HIP(hipEventElapsedTime(&t, ev[0], ev[1])); acc[0] += t;   // pyramid
        EAO_HIP(hipEventElapsedTime(&t, ev[9], ev[2])); acc[1] += t;   // FAST (after the blur of a profiled call)
        EAO_HIP(hipEventElapsedTime(&t, ev[2], ev[3])); acc[2] += t;   // quad-tree
        EAO_HIP(hipEventElapsedTime(&t, ev[6], ev[7])); acc[3] += t;   // blur, side stream
        EAO_HIP(hipEventElapsedTime(&t, ev[4], ev[5])); acc[4] += t;   // orientation + description
        EAO_HIP(hipEventElapsedTime(&t, ev[0], ev[8])); acc[5] += t;   // whole batch
    }
    for (int i = 0; i < 6; i++) ms[i] = (float)(acc[i] / calls);
    h->evUsed = 0;
    if (h->d_dbg) {
        long long st[16 * kMaxLevels];
        EAO_HIP(hipMemcpy(st, h->d_dbg, sizeof(st), hipMemcpyDeviceToHost));
        for (int l = 0; l < h->geom.nlevels; l++) {
            const long long* o = st + 16 * l;
            fprintf(stderr, "[eao quadtree stamps] level %d: M %lld S %lld passes %lld | setup %lld | multi-scan %lld hist %lld order %lld growth %lld rank %lld newlist %lld rehome %lld tail %lld | final %lld cycles\n",
                    l, o[10], o[11], o[9], o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], 0LL, o[8]);
        }
    }
    return EAO_OK;
}

}  // extern "C"
