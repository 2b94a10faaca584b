// orb.hip -- ORB extraction for MI355X (gfx950): pyramid, per-cell FAST-9/16 + NMS, quad-tree keypoint
// distribution, intensity-centroid orientation, 7x7 Gaussian, 256-bit steered BRIEF.
//
// Stands behind ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111, src/ORBextractor.cc:410-470,
// 765-853, 1043-1132).  Nothing here is translated from the reference: the work is re-decomposed for a
// 64-wide-wavefront machine and batched over frames so that one launch covers every level of every frame:
//
//   k_resize          level l from level l-1, 11-bit fixed-point bilinear, 4 px per lane: three aligned u32 loads per
//                     source row + v_alignbyte extraction, packed u32 stores
//   k_fast_cells      one WAVEFRONT per FAST cell: LDS-staged pixel tile (aligned u32 loads), compass pre-test with
//                     ballot compaction, branch-free arc score of the survivors (min3/max3 sliding windows), LDS NMS
//                     that is blind across the cell seam (as upstream's per-cell cv::FAST calls are), ballot-scan
//                     emission in row-major order, minThFAST retry
//   k_quadtree        one workgroup per (frame, level): level-synchronous restatement of DistributeOctTree --
//                     the std::list order is reproduced by prefix sums, the (size, pointer) sort by a rank
//   k_blur7           separable 7x7 fixed-point Gaussian: 4 px per lane, one dwordx3 per source row, v_dot4_u32_u8
//                     horizontal and v_dot2_u32_u16 vertical passes, window of row pairs in registers
//   k_orient_describe one wavefront per keypoint: integer moments reduced across lanes, fastAtan2, 256 point
//                     pairs (4 per lane) packed into the descriptor with 4 wave ballots
//
// Float semantics: this file is compiled with -ffp-contract=off (see Makefile): the reference's float
// expressions are rounded op by op.
#include "orb_internal.h"

using namespace eao::orb;

namespace {

__constant__ __align__(16) signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};

// ---------------------------------------------------------------------------------------------- resize
// Source bytes S[sx], S[sx+1] of the four output pixels of a lane all lie inside three aligned words of the source
// row (scale factors up to 2): three u32 loads per source row; one v_perm_b32 per pixel lifts its byte pair into two
// 16-bit halves, v_dot2_u32_u16 applies the column weights and v_mul_hi_u32 the row weights (all weights are >= 0 and
// sum to 2048, so the result needs no clamp).  A lane produces 4 px of kResizeRows output rows: the column set-up
// (coefficients, selectors) is shared between the rows.
constexpr int kResizeRows = 4;
__device__ __forceinline__ unsigned udot2w(unsigned a, unsigned b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), 0u, false);
}

// Everything the kernel needs about the two levels travels in the kernel arguments and the bilinear coefficients are
// computed in the kernel (the same float / double expressions OpenCV 3.3.x evaluates on the host: (float)((d + 0.5) *
// inv_scale - 0.5), floor, cvRound(w * 2048)), so a wave's only dependent memory accesses are its image loads: the
// seven launches of the pyramid chain are latency-, not bandwidth-bound.
struct ResizeLevel { int w, h, pitch, off; };
struct ResizeArgs {
    ResizeLevel D, S;
    double invX, invY;       // 1 / (D.w / S.w), 1 / (D.h / S.h)
    int srcIsInput;          // level l-1 is the caller's image
    int pyrFrameBytes;
    int G;                   // 4-pixel groups per output row, and its reciprocal (exact item / G by truncation, see QDIV)
    float invG;
    int frameAffinity;       // frame f is served by XCD f % 8 (batches that are a multiple of 8; see k_resize)
};

__device__ __forceinline__ void resize_coef(int d, double inv, int slimit, bool clampHi, int* ofs, unsigned* wpair) {
    float fr = (float)(((double)d + 0.5) * inv - 0.5);
    int o = (int)floorf(fr);
    fr -= (float)o;
    if (clampHi) {   // columns only (rows are clamped at the load, with their weights kept)
        if (o < 0) { fr = 0; o = 0; }
        if (o >= slimit - 1) { fr = 0; o = slimit - 1; }
    }
    const int w0 = min(max(__float2int_rn((1.f - fr) * 2048.f), -32768), 32767);
    const int w1 = min(max(__float2int_rn(fr * 2048.f), -32768), 32767);
    *ofs = o;
    *wpair = ((unsigned)w0 & 0xFFFFu) | ((unsigned)w1 << 16);
}


// one work item of a level: kResizeRows output rows x 4 output pixels of frame f
__device__ __forceinline__ void resize_item(const ResizeArgs& A, const ImgSrc& s, int f, int item) {
    const ResizeLevel D = A.D, S = A.S;
    const int rg = (int)(((float)item + 0.5f) * A.invG);
    const int dyb = rg * kResizeRows;
    const int dx0 = (item - rg * A.G) * 4;
    if (dyb >= D.h) return;
    const int spitch = A.srcIsInput ? s.pitch0 : S.pitch;
    const uint8_t* src = A.srcIsInput ? s.img0 + (long long)f * s.fs0 : s.pyr + (long long)f * A.pyrFrameBytes + S.off;
    uint8_t* dst = s.pyr + (long long)f * A.pyrFrameBytes + D.off;
    int sx[4];
    unsigned aa[4];
#pragma unroll
    for (int i = 0; i < 4; i++) resize_coef(min(dx0 + i, D.w - 1), A.invX, S.w, true, &sx[i], &aa[i]);
    const int wb = sx[0] >> 2, wmax = (S.w - 1) >> 2;
    const bool fast = ((((uintptr_t)src | (uintptr_t)spitch) & 3) == 0) && (sx[3] + 1 - 4 * wb < 12);
    if (fast) {
        const int i0 = 4 * wb, i1 = 4 * min(wb + 1, wmax), i2 = 4 * min(wb + 2, wmax);
        unsigned sel[4];
        bool up[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int o = sx[i] - 4 * wb;          // 0 .. 10: bytes o, o+1 of the 12 loaded
            up[i] = o >= 4;                        // pair lives in words (1, 2) instead of (0, 1)
            const unsigned oo = (unsigned)(up[i] ? o - 4 : o);
            sel[i] = oo | ((oo + 1) << 16) | 0x0c000c00u;   // v_perm: (S[sx], 0, S[sx+1], 0)
        }
        // All loads of the lane's rows are issued before any arithmetic (rows past the image repeat the last one: same
        // bytes to the same address), so a wave pays the memory latency once, not once per row.
        unsigned raw[kResizeRows][6], B0[kResizeRows], B1[kResizeRows];
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            int yo; unsigned yw;
            resize_coef(min(dyb + r, D.h - 1), A.invY, S.h, false, &yo, &yw);
            B0[r] = yw << 16; B1[r] = yw & 0xFFFF0000u;   // beta << 16
            const int sy0 = min(max(yo, 0), S.h - 1), sy1 = min(max(yo + 1, 0), S.h - 1);
            const unsigned r0 = __umul24(sy0, spitch), r1 = __umul24(sy1, spitch);   // 32-bit lane offsets from the wave-uniform base
            raw[r][0] = *reinterpret_cast<const unsigned*>(src + (r0 + i0)); raw[r][1] = *reinterpret_cast<const unsigned*>(src + (r0 + i1));
            raw[r][2] = *reinterpret_cast<const unsigned*>(src + (r0 + i2)); raw[r][3] = *reinterpret_cast<const unsigned*>(src + (r1 + i0));
            raw[r][4] = *reinterpret_cast<const unsigned*>(src + (r1 + i1)); raw[r][5] = *reinterpret_cast<const unsigned*>(src + (r1 + i2));
        }
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            const int dy = min(dyb + r, D.h - 1);
            unsigned packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned pa = __builtin_amdgcn_perm(up[i] ? raw[r][2] : raw[r][1], up[i] ? raw[r][1] : raw[r][0], sel[i]);
                const unsigned pc = __builtin_amdgcn_perm(up[i] ? raw[r][5] : raw[r][4], up[i] ? raw[r][4] : raw[r][3], sel[i]);
                const unsigned h0 = udot2w(pa, aa[i]), h1 = udot2w(pc, aa[i]);
                // ((beta0 * (h0 >> 4)) >> 16) + ((beta1 * (h1 >> 4)) >> 16), rounded: OpenCV's 11-bit fixed-point VResizeLinear
                const unsigned v = (__umulhi(B0[r], h0 >> 4) + __umulhi(B1[r], h1 >> 4) + 2) >> 2;
                packed |= min(v, 255u) << (8 * i);
            }
            // (the level's pitch is a multiple of 64: the tail of the last word lands in padding)
            *reinterpret_cast<unsigned*>(dst + (unsigned)(__umul24(dy, D.pitch) + dx0)) = packed;
        }
    } else {
        const int nvalid = min(4, D.w - dx0);
        for (int r = 0; r < kResizeRows; r++) {
            const int dy = dyb + r;
            if (dy >= D.h) break;
            int yo; unsigned yw;
            resize_coef(dy, A.invY, S.h, false, &yo, &yw);
            const int b0 = (short)(yw & 0xFFFF), b1 = (short)(yw >> 16);
            const int sy0 = min(max(yo, 0), S.h - 1), sy1 = min(max(yo + 1, 0), S.h - 1);
            const uint8_t* r0 = src + (long long)sy0 * spitch;
            const uint8_t* r1 = src + (long long)sy1 * spitch;
            uint8_t* o = dst + (long long)dy * D.pitch + dx0;
            for (int i = 0; i < nvalid; i++) {
                const int w0 = (short)(aa[i] & 0xFFFF), w1 = (short)(aa[i] >> 16);
                const int sx1 = min(sx[i] + 1, S.w - 1);
                const int h0 = r0[sx[i]] * w0 + r0[sx1] * w1;
                const int h1 = r1[sx[i]] * w0 + r1[sx1] * w1;
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                o[i] = (uint8_t)min(max(v, 0), 255);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_resize(ResizeArgs A, ImgSrc s, int f0) {
    // work items = (group of kResizeRows output rows) x (4-pixel group), dealt to the lanes in one flat sequence: with a
    // 256-pixel-wide block per row group the levels whose width is just above a multiple of 256 left up to half of the
    // lanes idle (257 px -> two blocks; measured 240 -> 181 us for the seven launches of a 256-frame batch)
    // Round 5: frame -> XCD affinity as in k_fast_cells.  Workgroups are dealt round-robin over the 8 XCDs and every XCD has its own L2; in plain order the
    // one-wave workgroups of adjacent row groups and of adjacent 256-pixel runs of a row land on different XCDs, and the source rows / cache lines they share
    // are fetched into two L2s: 2 x FETCH_SIZE + WRITE_SIZE = 151 MB per 64-frame step against 100.5 MB algorithmic (profiles/r05_pmc_traffic.json; the factor 2
    // holds for 4-byte loads too: profiles/r05_fetch_calib.txt).  With frame f on XCD f % 8 a line is fetched once.
    int f = blockIdx.z, bx = blockIdx.x;
    if (A.frameAffinity) {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.z, xcd = b & 7, slot = b >> 3;
        f = (int)(xcd + 8 * (slot / gridDim.x)); bx = (int)(slot % gridDim.x);
    }
    resize_item(A, s, f + f0, (bx * (int)blockDim.y + (int)threadIdx.y) * 64 + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------- fused pyramid
// The whole pyramid of a frame in ONE launch.  Level l is a resize of level l - 1, so the seven launches of k_resize are a
// dependent chain: 67 us for 100 MB at a 64-frame batch (latency, not bandwidth), 137 us with FAST running beside it.  Here a
// workgroup owns a tile of level 1 and everything above it: per level the pixels whose TOP-LEFT source pixel it owns on the
// level below (the boundaries follow from the source-offset function of the resize, so the owned ranges of the workgroups
// tile every level exactly), plus the one-pixel right / bottom fringe the next level's interpolation reaches into, which it
// recomputes instead of waiting for its neighbour.  The tiles of all levels live in LDS; only owned pixels go to HBM.
// Same arithmetic as k_resize, pixel by pixel (OpenCV's 11-bit fixed point; the coefficients come from resize_coef).
struct PyrRange { int x0, x1, y0, y1; };        // [x0, x1) x [y0, y1)
__global__ __launch_bounds__(kPyrThreads) void k_pyramid_fused(PyrArgs A, ImgSrc s, int f0) {
    extern __shared__ __align__(16) unsigned char psm[];
    __shared__ PyrRange s_own[kMaxLevels], s_req[kMaxLevels];
    const int t = threadIdx.x;
    const int tx = blockIdx.x % A.tilesX, ty = blockIdx.x / A.tilesX, f = blockIdx.z + f0;
    const int nl = A.nlevels;
    // owned / required ranges of this tile, per level: separable in x and y and the same for every frame -- the host ran the
    // range rules (pyr_first_at_least / pyr_src_ofs) once per tile column and tile row (PyrArgs::ranges); a thread deriving
    // them here took 30 us of serial double arithmetic per workgroup
    if (t < nl) {
        const int4 rx = A.ranges[(size_t)t * (A.tilesX + A.tilesY) + tx], ry = A.ranges[(size_t)t * (A.tilesX + A.tilesY) + A.tilesX + ty];
        s_own[t] = {rx.x, rx.y, ry.x, ry.y};
        s_req[t] = {rx.z, rx.w, ry.z, ry.w};
    }
    __syncthreads();
    // ---- stage the required region of the input image: aligned words when the image allows it (the tile's origin moves left
    //      to the word boundary), bytes otherwise
    int orgX0;       // image column of the staged tile's first byte
    {
        const PyrRange q = s_req[0];
        const uint8_t* src = s.img0 + (long long)f * s.fs0;
        const bool aligned = ((((uintptr_t)src | (uintptr_t)s.pitch0) & 3) == 0);
        orgX0 = aligned ? (q.x0 & ~3) : q.x0;
        const int rh = q.y1 - q.y0;
        unsigned char* dst = psm + A.L[0].ldsOff;
        const int ds = A.L[0].ldsW;
        if (q.x1 > q.x0 && rh > 0) {
            if (aligned) {
                const int nwr = (q.x1 - orgX0 + 3) >> 2;
                const float inv = __builtin_amdgcn_rcpf((float)nwr);
                for (int i = t; i < nwr * rh; i += kPyrThreads) {
                    const int y = (int)(((float)i + 0.5f) * inv), xw4 = i - y * nwr;
                    *reinterpret_cast<unsigned*>(dst + y * ds + 4 * xw4) = *reinterpret_cast<const unsigned*>(src + (long long)(q.y0 + y) * s.pitch0 + orgX0 + 4 * xw4);
                }
            } else {
                const int rw = q.x1 - q.x0;
                const float inv = __builtin_amdgcn_rcpf((float)rw);
                for (int i = t; i < rw * rh; i += kPyrThreads) {
                    const int y = (int)(((float)i + 0.5f) * inv), x = i - y * rw;
                    dst[y * ds + x] = src[(long long)(q.y0 + y) * s.pitch0 + q.x0 + x];
                }
            }
        }
    }
    int* xofs = reinterpret_cast<int*>(psm + A.coefOff);
    unsigned* xw = reinterpret_cast<unsigned*>(xofs + A.coefCap);
    int* yofs = reinterpret_cast<int*>(xw + A.coefCap);
    unsigned* yw = reinterpret_cast<unsigned*>(yofs + A.coefCap);
    constexpr int RR = 2;     // output rows per work item (the column set-up is shared)
    int srcOrg = orgX0;       // image column of the source tile's first byte
    for (int l = 1; l < nl; l++) {
        const PyrRange r = s_req[l], q = s_req[l - 1], own = s_own[l];
        // the tile's first column is the word boundary at or left of the required range (pitches are multiples of 64), so
        // that owned pixels leave as aligned words; the up to three extra columns are computed from whatever the source tile
        // holds there and are never read (the next level's sources start at its required range) nor written to HBM
        const int xa = r.x0 & ~3;
        const int rw = r.x1 - xa, rh = r.y1 - r.y0;
        __syncthreads();     // the source tile is complete; the coefficient tables of the previous level are no longer read
        if (r.x1 <= r.x0 || rh <= 0) { srcOrg = xa; continue; }
        const int sh = A.L[l - 1].h;
        const int rw4 = (rw + 3) & ~3;
        // bilinear coefficients of the level's columns and rows: built once per geometry by the host (the same float / double
        // expressions, resize_coef) -- every workgroup evaluating them per level was a third of the kernel's instructions
        const int2* cx = A.coefX + A.L[l].coefXOff;
        const int2* cy = A.coefY + A.L[l].coefYOff;
        for (int i = t; i < rw4; i += kPyrThreads) {   // (padded to a multiple of four with the last column)
            const int2 cw = cx[min(xa + i, r.x1 - 1)];
            xofs[i] = max(cw.x - srcOrg, 0); xw[i] = (unsigned)cw.y;
        }
        for (int i = t; i < rh; i += kPyrThreads) {
            const int2 cw = cy[r.y0 + i];
            const int yo = cw.x;
            const int sy0 = min(max(yo, 0), sh - 1), sy1 = min(max(yo + 1, 0), sh - 1);
            yofs[i] = (sy0 - q.y0) | ((sy1 - q.y0) << 16);
            yw[i] = (unsigned)cw.y;
        }
        __syncthreads();
        const unsigned* S32 = reinterpret_cast<const unsigned*>(psm + A.L[l - 1].ldsOff);
        unsigned* D32 = reinterpret_cast<unsigned*>(psm + A.L[l].ldsOff);
        const int ssw = A.L[l - 1].ldsW >> 2, dsw = A.L[l].ldsW >> 2;     // tile strides in words
        uint8_t* gdst = s.pyr + (long long)f * A.pyrFrameBytes + A.L[l].off;
        const int gp = A.L[l].pitch;
        const int G = rw4 >> 2, items = G * ((rh + RR - 1) / RR);
        const float invG = __builtin_amdgcn_rcpf((float)G);
        for (int it = t; it < items; it += kPyrThreads) {
            const int rg = (int)(((float)it + 0.5f) * invG), gx = it - rg * G, x = 4 * gx;
            const int4 xo = *reinterpret_cast<const int4*>(&xofs[x]);
            const uint4 xwv = *reinterpret_cast<const uint4*>(&xw[x]);
            const int sxv[4] = {xo.x, xo.y, xo.z, xo.w};
            const unsigned aa[4] = {xwv.x, xwv.y, xwv.z, xwv.w};
            const int wb = sxv[0] >> 2;
            // S[sx], S[sx + 1] of the four pixels lie inside three aligned words (scale factors up to 2, checked by the host)
            unsigned sel[4];
            bool up[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int o = min(sxv[i] - 4 * wb, 10);
                up[i] = o >= 4;
                const unsigned oo = (unsigned)(up[i] ? o - 4 : o);
                sel[i] = oo | ((oo + 1) << 16) | 0x0c000c00u;   // v_perm: (S[sx], 0, S[sx+1], 0)
            }
            const int gx0 = xa + x;
            const bool whole = gx0 >= own.x0 && gx0 + 4 <= own.x1;
#pragma unroll
            for (int rr = 0; rr < RR; rr++) {
                const int y = rg * RR + rr, yc = min(y, rh - 1);
                const int yy = yofs[yc];
                const unsigned wy = yw[yc];
                const unsigned B0 = wy << 16, B1 = wy & 0xFFFF0000u;
                const unsigned* p0 = S32 + (yy & 0xFFFF) * ssw + wb;
                const unsigned* p1 = S32 + (yy >> 16) * ssw + wb;
                const unsigned a0 = p0[0], a1 = p0[1], a2 = p0[2], c0 = p1[0], c1 = p1[1], c2 = p1[2];
                unsigned packed = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const unsigned pa = __builtin_amdgcn_perm(up[i] ? a2 : a1, up[i] ? a1 : a0, sel[i]);
                    const unsigned pc = __builtin_amdgcn_perm(up[i] ? c2 : c1, up[i] ? c1 : c0, sel[i]);
                    const unsigned h0 = udot2w(pa, aa[i]), h1 = udot2w(pc, aa[i]);
                    const unsigned v = (__umulhi(B0, h0 >> 4) + __umulhi(B1, h1 >> 4) + 2) >> 2;
                    packed |= min(v, 255u) << (8 * i);
                }
                if (y < rh) {
                    D32[y * dsw + gx] = packed;
                    const int gy = r.y0 + y;
                    if (gy >= own.y0 && gy < own.y1) {
                        uint8_t* o = gdst + gy * gp + gx0;
                        if (whole) *reinterpret_cast<unsigned*>(o) = packed;
                        else {
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (gx0 + i >= own.x0 && gx0 + i < own.x1) o[i] = (uint8_t)(packed >> (8 * i));
                        }
                    }
                }
            }
        }
        srcOrg = xa;
    }
}

// ---------------------------------------------------------------------------------------------- FAST
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// Arc value of ONE polarity.  With d_k = v - p_k the pixel is a FAST-9 corner at threshold t iff
//   A = max( max_arcs min_{k in arc} d_k ,  max_arcs min_{k in arc} -d_k ) > t          (OpenCV's corner score is A - 1)
// and min / max commute with the subtraction: the dark term is v - min_arcs max_{k in arc} p_k, the bright term the same
// expression on the complemented bytes 255 - p (= p ^ 0xFF), 255 - v.  Two 9-arcs of a 16-ring overlap, so at most one of
// the two terms is positive and a corner's score is the positive one: a lane evaluates the polarity `s` (0 = dark, 0xFF =
// bright) its candidate test allows -- 16 xor + 16 + 16 max3 + 8 min3 -- instead of both (the first version: 16 differences,
// 32 min3 / max3 for the triples and 64 for the arcs).  Pixels whose candidate test allows both polarities go through the
// bright one in a second, short pass (see k_fast_cells).
template <int T>
__device__ __forceinline__ int fast_arc_polar(const uint8_t* p, unsigned s, int* centre) {
    unsigned q[16];
    q[0] = p[3 * T];          q[1] = p[3 * T + 1];   q[2] = p[2 * T + 2];
    q[3] = p[T + 3];          q[4] = p[3];           q[5] = p[-T + 3];
    q[6] = p[-2 * T + 2];     q[7] = p[-3 * T + 1];  q[8] = p[-3 * T];
    q[9] = p[-3 * T - 1];     q[10] = p[-2 * T - 2]; q[11] = p[-T - 3];
    q[12] = p[-3];            q[13] = p[T - 3];      q[14] = p[2 * T - 2];
    q[15] = p[3 * T - 1];
    *centre = p[0];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] ^= s;
    // (v_max3 / v_min3 spelled out: left to itself the compiler re-associates the 32 three-input maxima into 24 + 15 two-input ones)
    auto mx3 = [](unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    auto mn3 = [](unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    unsigned hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) hi3[k] = mx3(q[k], q[(k + 1) & 15], q[(k + 2) & 15]);
    unsigned w9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w9[k] = mx3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    const unsigned m5 = mn3(w9[12], w9[13], w9[14]);
    const unsigned ma = mn3(mn3(w9[0], w9[1], w9[2]), mn3(w9[3], w9[4], w9[5]), mn3(w9[6], w9[7], w9[8]));
    const int m = (int)mn3(ma, mn3(w9[9], w9[10], w9[11]), min(m5, w9[15]));
    return (int)((unsigned)*centre ^ s) - m;
}

// A 9-arc of the 16-ring holds at least one pixel of every opposite pair (its complement is 7 contiguous positions): a
// pixel can only be a dark corner at threshold t if (p0 or p8) and (p4 or p12) are below v - t, a bright one if one of each
// pair is above v + t (positions 0 / 8 = three rows down / up, 4 / 12 = three columns right / left).  On the benchmark
// frames 15.6 % of the level-0 pixels pass (18.7 % passed the first version's "two of the four compass pixels" test).
template <int T>
__device__ __forceinline__ void fast_candidate(const uint8_t* p, int v, int t, bool* dark, bool* bright) {
    const int dn = p[3 * T], up = p[-3 * T], rt = p[3], lf = p[-3];
    *dark = max(min(dn, up), min(rt, lf)) < v - t;
    *bright = min(max(dn, up), max(rt, lf)) > v + t;
}

// ONE WAVEFRONT per FAST cell (64-thread workgroups): no cross-wave barriers, LDS sized by the largest cell of the
// current geometry.  Per threshold (iniThFAST, then minThFAST only if the cell stayed empty, as upstream :809-816):
//   1. candidate test of every pixel, four per lane on packed 16-bit halves; survivors are compacted into a work list in
//      row-major order (DPP scan of the lane counts, positions of a lane's survivors from a 16-entry LDS table)
//   2. arc value of the survivors in the polarity their candidate test allows; corners (arc > th) write their score into an
//      LDS score map that is zero everywhere else.  The few survivors that could be either polarity and were no dark corner
//      are appended to the list and evaluated as bright ones by later lanes of the same loop
//   3. the work list again: survivors with a non-zero score take the 3x3 strict NMS against the score map (blind across
//      the cell seam, like upstream's per-cell cv::FAST calls); kept corners leave in row-major order through a ballot scan
// WHOLE = true: the launch covers every cell of the frames (profiled calls, single-level pyramids); false: one share of the
// overlapped schedule.  Same code -- the parameter only gives the two kinds of launch different names in a rocprofv3 summary,
// so that the whole-stage duration bench.py reports can be read off the `k_fast_cells<true>` row directly.
// kTileStride = LDS row stride of the pixel tile: 48 bytes when every cell of the geometry is at most 45 px wide (the default
// 30-px cells are 37..43), 80 otherwise -- with the narrow tile and the shared list buffer a cell takes 6 KB of LDS instead
// of 9.7 KB, i.e. 6.5 instead of 4 resident wavefronts per SIMD for a kernel whose waves spend 3/4 of their life waiting.
template <bool WHOLE, int kTileStride>
__global__ __launch_bounds__(64) void k_fast_cells(const Geom* __restrict__ g, const CellDesc* __restrict__ cells, ImgSrc s,
                                                   unsigned* __restrict__ cellcand, int* __restrict__ cellcnt, int f0, int cellFirst,
                                                   int cellEnd, int frameAffinity) {
    extern __shared__ __align__(16) uint8_t fsm[];
    const int lane = threadIdx.x;
    // Workgroup -> cell placement.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2), so with
    // the plain mapping neighbouring cells (which share halo pixels and cache lines) land on 8 different L2s.  Measured
    // on MI355X, 64-frame batch (rocprofv3 --pmc FETCH_SIZE, per launch):   plain        134 MB   0.200 ms
    //    runs of 16 consecutive cells per XCD in rotation (kFastXcdRun = 16)               39 MB   0.228 ms
    //    one contiguous eighth of the cell table per XCD                                   30 MB   0.248 ms
    // The kernel is VALU/LDS-bound, not fabric-bound, and those cell orders cost more in load balance (level-0 cells are the
    // heavy ones) than the saved re-fetches return.  What does pay is FRAME affinity (below): frame f is served by XCD f % 8
    // only, every XCD walks its frames' cells in table order, so a cache line is fetched into one L2 and the load stays
    // balanced by construction -- 256-frame batch: FAST alone 0.540 -> 0.532 ms, the scheduled step 1.338 -> 1.301 ms (the
    // resize running beside FAST gets the fabric).  Speed only, never results.  EAO_FAST_AFFINITY=0 switches it off.
    // (bits 8.. of the last argument: the phase after which a CENSUS run leaves the kernel -- EAO_FAST_STOP_AFTER, tools/prof_fast_census.sh; 0 in every other run)
    const int stopAfter = frameAffinity >> 8;
    frameAffinity &= 1;
    int cell = blockIdx.x, fy = blockIdx.y;
    if constexpr (kFastXcdRun > 0) {
        const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3;
        cell = ((bslot / kFastXcdRun) * 8 + xcd) * kFastXcdRun + bslot % kFastXcdRun;
    }
    if (frameAffinity) {   // frame f on XCD f % 8 (batches that are a multiple of 8): all cells of a frame share one L2
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.y, xcd = b & 7, slot = b >> 3;
        fy = (int)(xcd + 8 * (slot / gridDim.x));
        cell = (int)(slot % gridDim.x);
    }
    cell += cellFirst;                                  // the launch covers cells [cellFirst, cellEnd)
    const int f = fy + f0;
    if (cell >= cellEnd) return;
    const CellDesc c = cells[cell];
    const int maxT = g->fastMaxTested;                 // max tested pixels of any cell
    uint8_t* tile = fsm;                                // fastTileBytes
    unsigned short* worklist = reinterpret_cast<unsigned short*>(fsm + g->fastTileBytes);
    // (survivors that wait for their bright evaluation are appended behind the list, with bit 14 set)
    unsigned* sc4 = reinterpret_cast<unsigned*>(fsm + g->fastTileBytes + ((2 * maxT + 15) & ~15));   // score map (tw+2) x (th+2) bytes, 16-byte aligned
    uint8_t* sc = reinterpret_cast<uint8_t*>(sc4);
    unsigned* lut = reinterpret_cast<unsigned*>(fsm + g->fastLdsBytes - 64);        // 16 words
    int pitch;
    const uint8_t* src = level_ptr(g, s, c.level, f, &pitch);
    const int sw = c.sw, sh = c.sh, tw = sw - 6, th_ = sh - 6;
    const long long slot = (long long)f * g->totalCells + cell;
    if (tw <= 0 || th_ <= 0) {
        if (lane == 0) cellcnt[slot] = 0;
        return;
    }
    if (lane < 16) {   // pixel positions of a word's survivors, one per byte, lowest first (index bits 0..3 = pixels 1, 0, 3, 2)
        unsigned e = 0, n = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (lane & (1 << (j ^ 1))) { e |= (unsigned)j << (8 * n); n++; }
        lut[lane] = e;
    }
    // i / d for 0 <= i < 8192, 1 <= d <= 80 without an integer divide: (i + 0.5) / d is at least 0.5/80 away from an
    // integer, the float product is off by < 1e-4, so truncation is exact
#define QDIV(i, inv) ((int)(((float)(i) + 0.5f) * (inv)))
    // (v_rcp_f32 is good to 1 ulp, far inside the 1e-4 the truncation argument needs; an IEEE division costs ten instructions)
    const float invSw = __builtin_amdgcn_rcpf((float)sw);
    const int x0a = c.x0 & ~3, ph = c.x0 - x0a;
    if ((((uintptr_t)src | (uintptr_t)pitch) & 3) == 0) {
        const int np = (c.x0 + sw + 7 - x0a) >> 3;   // word PAIRS per row (the odd word of a row lands in the row's padding)
        const float invNp = __builtin_amdgcn_rcpf((float)np);
        // the lane's pairs are 64 apart in the row-major pair sequence of the tile: (row, pair) advance by (64 / np, 64 % np)
        // with a carry, so one division per lane serves the whole loop and the addresses are running 32-bit offsets
        int y = QDIV(lane, invNp), xw = lane - y * np;
        const int q = QDIV(64, invNp), rem = 64 - q * np;
        unsigned goff = (unsigned)((c.y0 + y) * pitch + x0a + 8 * xw), loff = (unsigned)(y * kTileStride + 8 * xw);
        const unsigned gstep = (unsigned)(q * pitch + 8 * rem), lstep = (unsigned)(q * kTileStride + 8 * rem);
        const unsigned gwrap = (unsigned)(pitch - 8 * np), lwrap = (unsigned)(kTileStride - 8 * np);
        for (int i = lane; i < np * sh; i += 64) {
            const unsigned w0 = *reinterpret_cast<const unsigned*>(src + goff), w1 = *reinterpret_cast<const unsigned*>(src + goff + 4);
            *reinterpret_cast<unsigned*>(&tile[loff]) = w0;
            *reinterpret_cast<unsigned*>(&tile[loff + 4]) = w1;
            xw += rem; goff += gstep; loff += lstep;
            if (xw >= np) { xw -= np; goff += gwrap; loff += lwrap; }
        }
    } else {
        for (int i = lane; i < sw * sh; i += 64) {
            const int y = QDIV(i, invSw), x = i - y * sw;
            tile[y * kTileStride + x + ph] = src[(long long)(c.y0 + y) * pitch + c.x0 + x];
        }
    }
    if (stopAfter == 1) { if (lane == 0) cellcnt[slot] = 0; return; }      // census: launch + staging (an empty cell for the kernels behind)
    const int rw = tw + 2;
    const int scWords = (rw * (th_ + 2) + 3) >> 2;
    // (the geometry fields the loops need, read ONCE: the compiler re-loads g->x after every store it cannot tell apart)
    const int cellCap = g->cellCap, thIni = g->iniTh, thMin = g->minTh;
    unsigned* out = cellcand + slot * cellCap;
    for (int pass = 0; pass < 2; pass++) {
        const int th = min(max(pass ? thMin : thIni, 0), 255);
        for (int i = lane; i < ((scWords + 3) >> 2); i += 64) reinterpret_cast<uint4*>(sc4)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();   // tile staged (first pass) / previous pass done with sc
        // ---- 1. candidate test, EIGHT horizontally adjacent pixels per lane (two words): aligned LDS words of the rows three
        // up, three down and of the centre row (left / centre / right windows) shifted into place with v_alignbyte, bytes
        // widened to 16-bit pairs (even / odd pixels) with v_perm; on the raw bytes the test reads
        // max(min(up, down), min(left, right)) < v - th  (dark)  or  min(max(up, down), max(left, right)) > v + th  (bright):
        // six packed min / max and two packed subtractions whose sign bits are the answer.  Survivors keep row-major order: a
        // lane's eight pixels are consecutive, the lanes' counts are prefixed with a DPP scan, and the positions of a word's
        // survivors come from a 16-entry table.  Work-list entry = y << 7 | x (tested-region coordinates, both < 72).
        int nwork = 0;
        {
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const int G = (tw + 7) >> 3, items = G * th_;
            const float invG = __builtin_amdgcn_rcpf((float)G);
            const int a0 = 3 + ph;                       // byte offset of the first centre pixel inside its window
            const int shC = a0 & 3, wC = a0 >> 2;        // centre / up / down: words wC .. wC+2 shifted by shC
            const int shR = (a0 + 3) & 3, wR = (a0 + 3) >> 2;
            const unsigned* t32 = reinterpret_cast<const unsigned*>(tile);
            const short thS = (short)th;
            const s16x2 TH = {thS, thS};
            for (int it0 = 0; it0 < items; it0 += 64) {
                const int it = min(it0 + lane, items - 1);
                const int y = QDIV(it, invG), xg = it - (int)__umul24(y, G);
                const unsigned* tu = t32 + ((int)__umul24(y, kTileStride >> 2) + 2 * xg);   // (row y, byte 8 xg): the UP row
                const unsigned* tm = tu + 3 * (kTileStride >> 2);
                const unsigned* td = tu + 6 * (kTileStride >> 2);
                const unsigned u0 = tu[wC], u1 = tu[wC + 1], u2 = tu[wC + 2];
                const unsigned d0 = td[wC], d1 = td[wC + 1], d2 = td[wC + 2];
                const unsigned c0 = tm[wC], c1 = tm[wC + 1], c2 = tm[wC + 2];
                const unsigned l0 = tm[0], l1 = tm[1], l2 = tm[2];
                const unsigned r0 = tm[wR], r1 = tm[wR + 1], r2 = tm[wR + 2];
                unsigned m8 = 0;
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const unsigned C = __builtin_amdgcn_alignbyte(w ? c2 : c1, w ? c1 : c0, (unsigned)shC);
                    const unsigned U = __builtin_amdgcn_alignbyte(w ? u2 : u1, w ? u1 : u0, (unsigned)shC);
                    const unsigned D = __builtin_amdgcn_alignbyte(w ? d2 : d1, w ? d1 : d0, (unsigned)shC);
                    const unsigned Lf = __builtin_amdgcn_alignbyte(w ? l2 : l1, w ? l1 : l0, (unsigned)ph);
                    const unsigned R = __builtin_amdgcn_alignbyte(w ? r2 : r1, w ? r1 : r0, (unsigned)shR);
                    unsigned fl[2];
#pragma unroll
                    for (int par = 0; par < 2; par++) {
                        const unsigned sel = par ? 0x0c030c01u : 0x0c020c00u;   // (pixel 1, pixel 3) / (pixel 0, pixel 2) as 16-bit halves
                        const s16x2 v = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, C, sel));
                        const s16x2 pu = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, U, sel));
                        const s16x2 pd = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, D, sel));
                        const s16x2 pl = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, Lf, sel));
                        const s16x2 pr = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, R, sel));
                        const s16x2 dk = __builtin_elementwise_max(__builtin_elementwise_min(pu, pd), __builtin_elementwise_min(pl, pr));
                        const s16x2 br = __builtin_elementwise_min(__builtin_elementwise_max(pu, pd), __builtin_elementwise_max(pl, pr));
                        // sign bit of a half set <=> dk < v - th  or  br > v + th
                        fl[par] = __builtin_bit_cast(unsigned, dk - (v - TH)) | __builtin_bit_cast(unsigned, (v + TH) - br);
                    }
                    // signs: pixel 0 -> bit 15, pixel 1 -> 14, pixel 2 -> 31, pixel 3 -> 30; table index bits 0..3 = pixels 1, 0, 3, 2
                    const unsigned cw = (fl[0] & 0x80008000u) | ((fl[1] >> 1) & 0x40004000u);
                    const unsigned m4 = ((cw >> 14) & 3u) | ((cw >> 28) & 0xCu);
                    m8 |= m4 << (4 * w);
                }
                const int x0 = 8 * xg;
                // pixels beyond the row end / lanes beyond the last item masked off (mask bit order as above)
                const int nv = it0 + lane < items ? tw - x0 : 0;
                {
                    const unsigned okpx = (1u << min(max(nv, 0), 8)) - 1u;               // bit k = pixel k is a tested pixel
                    const unsigned sw1 = ((okpx & 0x55u) << 1) | ((okpx >> 1) & 0x55u);  // swap neighbours: table bit order
                    m8 &= sw1;
                }
                const unsigned mlo = m8 & 15u, mhi = m8 >> 4;
                const int cl = (int)__popc(mlo), cnt = (int)__popc(m8);
                // survivors of the lower lanes: an inclusive DPP scan of the lane counts (row shifts inside the rows of 16, then
                // the row broadcasts 15 / 31 -- six VALU adds)
                int incl = cnt;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1 and 3
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2 and 3
                unsigned short* dst = worklist + (nwork + incl - cnt);
                const unsigned e0 = lut[mlo], e1 = lut[mhi];      // byte k = position of the word's k-th survivor
                const int i0 = (y << 7) + x0;
                if (cl > 0) dst[0] = (unsigned short)(i0 + (int)(e0 & 0xFF));
                if (cl > 1) dst[1] = (unsigned short)(i0 + (int)((e0 >> 8) & 0xFF));
                if (cl > 2) dst[2] = (unsigned short)(i0 + (int)((e0 >> 16) & 0xFF));
                if (cl > 3) dst[3] = (unsigned short)(i0 + 3);
                dst += cl;
                const int ch = cnt - cl;
                if (ch > 0) dst[0] = (unsigned short)(i0 + 4 + (int)(e1 & 0xFF));
                if (ch > 1) dst[1] = (unsigned short)(i0 + 4 + (int)((e1 >> 8) & 0xFF));
                if (ch > 2) dst[2] = (unsigned short)(i0 + 4 + (int)((e1 >> 16) & 0xFF));
                if (ch > 3) dst[3] = (unsigned short)(i0 + 7);
                nwork += __builtin_amdgcn_readlane(incl, 63);
            }
        }
        __syncthreads();
        if (stopAfter == 2) { if (lane == 0) cellcnt[slot] = 0; return; }  // census: ... + candidate test
        // ---- 2. arc value of the survivors in one polarity; corners go to the score map.  A survivor whose candidate test allows
        // both polarities and that is no dark corner is APPENDED to the work list with kSecond set and meets the bright
        // evaluation in a later lane of the same loop -- normally one of the lanes the last 64-entry round leaves idle (a
        // separate loop over these few entries was a whole extra round of the arc code per cell)
        constexpr int kSecond = 0x4000;
        int nsecond = 0;
        for (int k0 = 0; k0 < nwork + nsecond;) {
            const int nlist = nwork + nsecond;
            const int k = k0 + lane;
            const bool live = k < nlist;
            const int ent = worklist[min(k, nlist - 1)];
            const bool second = ent & kSecond;
            const int i = ent & (kSecond - 1);
            const int y = i >> 7, x = i & 127;
            const uint8_t* p = &tile[__umul24(y + 3, kTileStride) + x + 3 + ph];
            bool dk, br;
            fast_candidate<kTileStride>(p, p[0], th, &dk, &br);
            int v;
            const int a = fast_arc_polar<kTileStride>(p, dk && !second ? 0u : 0xFFu, &v);
            const bool corner = live && a > th;
            if (corner) sc[__umul24(y + 1, rw) + x + 1] = (uint8_t)(a - 1);
            const bool again = live && !second && dk && br && !corner;     // could still be a bright corner
            const unsigned long long m = __ballot(again);
            if (m) {
                const int n2 = (int)__popcll(m);
                if (nlist + n2 <= maxT) {
                    if (again) worklist[nlist + (int)__popcll(m & ((1ull << lane) - 1))] = (unsigned short)(i | kSecond);
                    nsecond += n2;
                } else {   // (no room left behind the work list -- a cell of pure noise: the bright evaluation right here)
                    const int a2 = fast_arc_polar<kTileStride>(p, 0xFFu, &v);
                    if (again && a2 > th) sc[__umul24(y + 1, rw) + x + 1] = (uint8_t)(a2 - 1);
                }
            }
            k0 = min(k0 + 64, nlist);      // (entries appended by a partial round start right behind it)
        }
        __syncthreads();
        if (stopAfter == 3) { if (lane == 0) cellcnt[slot] = 0; return; }  // census: ... + arc values
        // ---- 3. NMS + ordered emission (a corner of score 0 is never kept: 0 > max(...) cannot hold, upstream the same)
        int total = 0;
        for (int k0 = 0; k0 < nwork; k0 += 64) {
            const int k = k0 + lane;
            const int i = worklist[min(k, nwork - 1)];
            const int y = i >> 7, x = i & 127;
            const uint8_t* q = &sc[__umul24(y + 1, rw) + x + 1];
            const int v = q[0];
            const unsigned long long cm = __ballot(k < nwork && v > 0);
            if (cm == 0) continue;
            const int m = max(max3i(q[-rw - 1], q[-rw], q[-rw + 1]), max(max(q[-1], q[1]), max3i(q[rw - 1], q[rw], q[rw + 1])));
            const bool keep = k < nwork && v > m;
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const int pos = total + __popcll(km & ((1ull << lane) - 1));
                if (pos < cellCap) out[pos] = (unsigned)(x + 3 + c.offX) | ((unsigned)(y + 3 + c.offY) << 12) | ((unsigned)v << 24);
            }
            total += __popcll(km);
        }
        if (total > 0 || pass == 1) {
            if (lane == 0) cellcnt[slot] = min(total, cellCap);
            return;
        }
    }
#undef QDIV
}

// ---------------------------------------------------------------------------------------------- blur
__device__ __forceinline__ int reflect101(int p, int len) {
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// Strip = 128 px (32 lanes x 4 px) wide, 16 rows high; a workgroup of 8 half-waves takes 8 consecutive strips of a level
// (x fastest), so only the last workgroup of a level and strips hanging over the right / bottom edge idle lanes.
// Per source row a lane loads the three aligned words around its 4 pixels (one dwordx3), the horizontal pass is two
// v_dot4_u32_u8 per pixel on v_alignbyte windows, two rows of 16-bit results are paired in one register and the
// vertical pass is four v_dot2_u32_u16 per pixel on a 4-deep window of row pairs; one packed store per row.  No LDS.
// Borders: in waves that touch the left / right image edge each lane carries three v_perm selectors (built once) that
// rebuild the reflect-101 window from the words it could load in bounds; rows reflect through their index.  An
// unaligned caller image (level 0 only) takes a byte-wise variant of the same loop.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned udot2(unsigned a, unsigned b, unsigned c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

struct BlurEdge {   // per-lane reconstruction of the 12-byte window x4-4 .. x4+7 (MODE 1)
    int off0;       // byte offset (inside the row) of the first loaded word
    bool left;      // window pools: W1, W2 come from (L0, L1) instead of (L1, L2)
    unsigned selA, selB, selC;
};

__device__ __forceinline__ BlurEdge blur_edge(int x4, int w) {
    BlurEdge e;
    const int wi = x4 >> 2, wmax = (w - 1) >> 2;
    const int wi0 = min(max(wi - 1, 0), wmax - 2);
    e.off0 = 4 * wi0;
    e.left = wi == 0;
    e.selA = e.selB = e.selC = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        int x = min(max(x4 - 4 + k, -3), w + 2);     // (bytes further out carry a zero tap or feed pixels beyond the row)
        x = x < 0 ? -x : x;
        x = min(x, 2 * w - 2 - x);                   // reflect-101, one fold is enough here
        const int m = min(max(x - 4 * wi0, 0), 11);  // loaded byte holding logical byte k
        if (k < 4) e.selA |= (unsigned)min(m, 7) << (8 * k);
        else {
            const unsigned q = (unsigned)min(max(e.left ? m : m - 4, 0), 7) << (8 * (k & 3));
            if (k < 8) e.selB |= q; else e.selC |= q;
        }
    }
    return e;
}

// MODE 0: interior wave (every lane's window is inside the row); 1: wave touching a border; 2: byte-wise loads
template <int MODE>
__device__ __forceinline__ void blur_load(const uint8_t* __restrict__ base, unsigned rowOff, int x4, int w, const BlurEdge& e, unsigned L[3]) {
    // (base is wave-uniform, the row and column enter as ONE 32-bit lane offset: SGPR-base addressing, no 64-bit VALU adds)
    const uint8_t* row = base + rowOff;
    if (MODE == 0) {
        const unsigned* q = reinterpret_cast<const unsigned*>(base + (rowOff + (unsigned)(x4 - 4)));
        L[0] = q[0]; L[1] = q[1]; L[2] = q[2];
    } else if (MODE == 1) {
        const unsigned* q = reinterpret_cast<const unsigned*>(base + (rowOff + (unsigned)e.off0));
        L[0] = q[0]; L[1] = q[1]; L[2] = q[2];
    } else {
        L[0] = L[1] = L[2] = 0;
#pragma unroll
        for (int k = 1; k < 11; k++) L[k >> 2] |= (unsigned)row[reflect101(x4 - 4 + k, w)] << (8 * (k & 3));
    }
}

template <int MODE>
__device__ __forceinline__ void blur_hpass(const unsigned L[3], const BlurEdge& e, unsigned hres[4]) {
    unsigned W0 = L[0], W1 = L[1], W2 = L[2];
    if (MODE == 1) {
        const unsigned a = e.left ? L[0] : L[1], b = e.left ? L[1] : L[2];
        W0 = __builtin_amdgcn_perm(L[1], L[0], e.selA);
        W1 = __builtin_amdgcn_perm(b, a, e.selB);
        W2 = __builtin_amdgcn_perm(b, a, e.selC);
    }
    // taps round(k*256) of getGaussianKernel(7, 2): 18 34 49 55 49 34 18 (host-verified at handle creation)
    constexpr unsigned T0 = 18u | 34u << 8 | 49u << 16 | 55u << 24, T1 = 49u | 34u << 8 | 18u << 16;
    hres[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 1), T0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W2, W1, 1), T1, 0, false), false);
    hres[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 2), T0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W2, W1, 2), T1, 0, false), false);
    hres[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 3), T0, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W2, W1, 3), T1, 0, false), false);
    hres[3] = __builtin_amdgcn_udot4(W1, T0, __builtin_amdgcn_udot4(W2, T1, 0, false), false);
}

// The loop is branch-free (rows below the image land in the padding rows every level of the blur buffer carries), so
// the loads of the next row pairs are issued before the arithmetic of the current one.
template <int MODE>
__device__ __forceinline__ void blur_strip(const uint8_t* __restrict__ src, int pitch, uint8_t* __restrict__ dst, int dpitch, int x4, int y0, int w, int h) {
    BlurEdge e = {};
    if (MODE == 1) e = blur_edge(x4, w);
    constexpr int NP = (kBlurRows + 6) / 2;   // source row pairs
    constexpr int PF = MODE == 2 ? 1 : 3;     // row pairs in flight
    unsigned raw[NP][2][3];
    auto fetch = [&](int m) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            int t = y0 - 3 + 2 * m + q;
            t = t < 0 ? -t : t;
            t = min(t, 2 * h - 2 - t);
            blur_load<MODE>(src, __umul24(t, pitch), x4, w, e, raw[m][q]);
        }
    };
#pragma unroll
    for (int m = 0; m < PF; m++) fetch(m);
    unsigned P[4][4];   // P[m & 3][i] = horizontal results of source rows 2m (low half) and 2m+1 (high half), pixel i
#pragma unroll
    for (int m = 0; m < NP; m++) {
        if (m + PF < NP) fetch(m + PF);
        unsigned hr[2][4];
        blur_hpass<MODE>(raw[m][0], e, hr[0]);
        blur_hpass<MODE>(raw[m][1], e, hr[1]);
#pragma unroll
        for (int i = 0; i < 4; i++) P[m & 3][i] = hr[0][i] | (hr[1][i] << 16);
        if (m >= 3) {
            // even output row 2(m-3): source rows 2(m-3) .. 2m hold taps 0..6; odd row 2(m-3)+1: rows 2(m-3)+1 .. 2m+1
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const unsigned K0 = q ? 18u << 16 : (18u | 34u << 16), K1 = q ? (34u | 49u << 16) : (49u | 55u << 16),
                               K2 = q ? (55u | 49u << 16) : (49u | 34u << 16), K3 = q ? (34u | 18u << 16) : 18u;
                unsigned sv[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    sv[i] = udot2(P[(m - 3) & 3][i], K0, udot2(P[(m - 2) & 3][i], K1, udot2(P[(m - 1) & 3][i], K2, udot2(P[m & 3][i], K3, 32768u))));
                // (sv >> 16) saturated to 255, four to a word
                const u16x2 lim = {255, 255};
                const u16x2 pa = __builtin_elementwise_min(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(sv[1], sv[0], 0x07060302u)), lim);
                const u16x2 pb = __builtin_elementwise_min(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(sv[3], sv[2], 0x07060302u)), lim);
                const unsigned packed = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, pb), __builtin_bit_cast(unsigned, pa), 0x06040200u);
                const int y = y0 + 2 * (m - 3) + q;
                // (pitch is a multiple of 64 and the level holds a multiple of kBlurRows rows: tails land in padding)
                *reinterpret_cast<unsigned*>(dst + (unsigned)(__umul24(y, dpitch) + x4)) = packed;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_blur7(const Geom* __restrict__ g, ImgSrc s, uint8_t* __restrict__ blur, int f0, int src0Aligned, int frameAffinity) {
    // (frame -> XCD affinity as in k_fast_cells was measured here and in k_resize in round 1: 0.152 vs 0.149 ms and 0.199 vs 0.194 ms at batch 256, plain order kept.
    //  Round 5 looks at the TRAFFIC: strips of 16 rows read 22, and a strip's neighbours run on other XCDs -- 149 MB per 64-frame step against 122 MB algorithmic.)
    const int lane = threadIdx.x & 31;
    int f = blockIdx.y, bxi = blockIdx.x;
    if (frameAffinity) {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.y, xcd = b & 7, slot = b >> 3;
        f = (int)(xcd + 8 * (slot / gridDim.x)); bxi = (int)(slot % gridDim.x);
    }
    f += f0;
    int l = 0;
    while (l + 1 < g->nlevels && bxi >= g->L[l + 1].tileBase) l++;
    // (upstream blurs only levels that hold keypoints, :1081-1082; blurring all of them changes no output and removes
    //  the dependency on the quad-tree, so this kernel can overlap it)
    const LevelGeom L = g->L[l];
    const int strip = (bxi - L.tileBase) * kBlurStripsPerWg + (int)(threadIdx.x >> 5);
    const int x4 = (strip % L.tilesX) * kBlurSegW + lane * 4;
    const int y0 = (strip / L.tilesX) * kBlurRows;
    if (x4 >= L.w || y0 >= L.h) return;
    int pitch;
    const uint8_t* src = level_ptr(g, s, l, f, &pitch);
    uint8_t* dst = blur + (long long)f * g->pyrFrameBytes + L.off;
    if (l == 0 && !src0Aligned) blur_strip<2>(src, pitch, dst, L.pitch, x4, y0, L.w, L.h);
    else if (__any(x4 < 4 || x4 + 7 > L.w)) blur_strip<1>(src, pitch, dst, L.pitch, x4, y0, L.w, L.h);
    else blur_strip<0>(src, pitch, dst, L.pitch, x4, y0, L.w, L.h);
}

// Sum over the 64 lanes on the VALU (DPP: two quad permutes, half-row and row mirrors, then the row broadcasts 15 / 31), the
// total read from lane 63 into an SGPR: six v_add with DPP + one v_readlane instead of six ds_bpermute round trips.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: every lane of a row holds the row sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// end of each row of the radius-15 disc by |v| (the table of reference src/ORBextractor.cc:455-469, verified against the
// computed one at handle creation); [16] closes the upper half-wave's sixteenth step (there is no row 16)
constexpr int kUmaxTab[17] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3, -1};

// ---------------------------------------------------------------------------------------------- orientation + rBRIEF
// sin/cos of a float angle in [0, 2*pi], evaluated in double (Cody-Waite reduction by pi/2 + the classic
// minimax kernels) and rounded to float: plain +,-,* only, so results do not depend on a device libm.
__device__ __forceinline__ void sincos_f32_via_f64(float angle, float* sn, float* cs) {
    const double x = (double)angle;
    const int k = (int)(x * 0.63661977236758134308 + 0.5);
    const double kd = (double)k;
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double sr = r + r * z * ps;
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double cr = 1.0 - 0.5 * z + z * z * pc;
    double sv, cv;
    switch (k & 3) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *sn = (float)sv;
    *cs = (float)cv;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ __launch_bounds__(256) void k_orient_describe(const Geom* __restrict__ g, ImgSrc s, const uint8_t* __restrict__ blur,
                                                         const unsigned* __restrict__ levelkps, const int* __restrict__ levelcnt,
                                                         eao_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                         int* __restrict__ nout, int cap, int f0, int nlevels, int lFirst, int lEnd, int writeN) {
    // Levels [lFirst, lEnd) of every frame: blockIdx.x * 4 + wave counts the keypoints of THOSE levels; a launch over the lower levels may run
    // while the upper levels' quad-trees are still at work (their counts are not read), the launch that covers the last level writes the
    // frame's keypoint count (writeN).
    // (the wave index through readfirstlane: the compiler then KNOWS that the keypoint -- index, key, position, level --
    //  is wave-uniform, keeps it in SGPRs, loads it with scalar loads and addresses the two windows as SGPR base + 32-bit
    //  lane offset instead of per-lane 64-bit pointer arithmetic)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the lane's four test pairs (xa, ya, xb, yb as int8) depend on nothing: fetch them first, under the other loads
    unsigned pat[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pat[k] = reinterpret_cast<const unsigned*>(c_pattern)[lane + 64 * k];
    // Workgroups are dealt round-robin over the 8 XCDs and every XCD has its own L2: with the plain (blockIdx.y = frame)
    // order the patches of one frame are pulled into all eight L2s.  When the batch is a multiple of 8, frame f is served
    // by XCD f % 8 only (speed only: any placement gives the same results).
    int bx = blockIdx.x, fy = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.y, xcd = b & 7, slot = b >> 3;
        fy = (int)(xcd + 8 * (slot / gridDim.x));
        bx = (int)(slot % gridDim.x);
    }
    int j = bx * 4 + wv;   // compact output index inside the frame (relative to the first keypoint of level lFirst until the counts are known)
    const int f = fy + f0;
    // level of compact index j: the level counts are wave-uniform -- independent scalar loads and a running sum on the
    // scalar unit (the first version scanned them across lanes: four ds_bpermute round trips before anything else could start)
    // (lanes 0 .. nlevels-1 hold the level counts, an inclusive DPP scan inside the 16-lane row gives the level ends, a ballot
    //  counts the levels that end at or before j: a dozen instructions.  As a scalar loop over the sixteen possible levels it
    //  was ~90 SALU instructions per wave, in a kernel that issues as many scalar as vector instructions -- one scalar unit per CU)
    int total, l, lbase;
    {
        int inc = lane < lEnd ? levelcnt[f * nlevels + lane] : 0;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);   // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);   // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);   // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);   // row_shr:8
        total = __builtin_amdgcn_readlane(inc, 15);
        if (lFirst > 0) j += __builtin_amdgcn_readlane(inc, lFirst - 1);
        l = (int)__popcll(__ballot(lane < nlevels && inc <= j));                // levels that end at or before j (levels >= lEnd: the flat tail of the scan)
        lbase = l > 0 ? __builtin_amdgcn_readlane(inc, min(l, kMaxLevels) - 1) : 0;
    }
    const int jout = j;
    if (writeN && bx == 0 && threadIdx.x == 0) nout[f] = min(total, cap);
    if (l >= nlevels || jout >= cap) return;
    j -= lbase;
    const LevelGeom L = g->L[l];
    const unsigned key = levelkps[(long long)f * g->totalKpCap + L.kpBase + j];
    const int cx = (int)(key & 0xFFF) + kMinBorder, cy = (int)((key >> 12) & 0xFFF) + kMinBorder;
    int pitch;
    const uint8_t* img = level_ptr(g, s, l, f, &pitch);
    // The 256 test pairs, rotated, stay inside radius 18.4 of the keypoint (|pattern| <= sqrt(338)): the 37 x 37 window of
    // the blurred level is staged in LDS with aligned word loads (6 per lane, issued before the moment loads) and the
    // 512 scattered byte reads of the wave go to LDS instead of the texture path.
    constexpr int kPR = 18, kPW = 10;   // window radius; words per staged row (37 + up to 3 bytes of alignment)
    __shared__ unsigned patch[4][(2 * kPR + 1) * kPW];
    const int ph = (cx - kPR) & 3;
    const uint8_t* bw = blur + (long long)f * g->pyrFrameBytes + L.off + (cy - kPR) * L.pitch + (cx - kPR - ph);
    unsigned pw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int i = min(lane + 64 * k, (2 * kPR + 1) * kPW - 1), r = (i * 6554) >> 16;   // i / 10
        pw[k] = *reinterpret_cast<const unsigned*>(bw + (unsigned)(r * L.pitch + 4 * (i - r * kPW)));
    }
    // the 31 x 31 window of the (unblurred) level for the intensity centroid goes through LDS the same way: 5 aligned word
    // loads per lane instead of 16 byte loads
    constexpr int kMR = 15, kMW = 9;    // 31 + up to 3 bytes of alignment = 9 words per row
    __shared__ unsigned mpatch[4][(2 * kMR + 1) * kMW];
    const bool imgAligned = ((((uintptr_t)img | (uintptr_t)pitch) & 3) == 0);   // (an unaligned caller image: byte loads)
    const int mph = (cx - kMR) & 3;
    if (imgAligned) {
        const uint8_t* mw = img + (cy - kMR) * pitch + (cx - kMR - mph);
        unsigned mv[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = min(lane + 64 * k, (2 * kMR + 1) * kMW - 1), r = (i * 7282) >> 16;   // i / 9
            mv[k] = *reinterpret_cast<const unsigned*>(mw + (unsigned)(r * pitch + 4 * (i - r * kMW)));
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (lane + 64 * k < (2 * kMR + 1) * kMW) mpatch[wv][lane + 64 * k] = mv[k];
    } else {
        uint8_t* mb = reinterpret_cast<uint8_t*>(mpatch[wv]);
        for (int i = lane; i < (2 * kMR + 1) * (2 * kMR + 1); i += 64) {
            const int r = i / (2 * kMR + 1), c = i - r * (2 * kMR + 1);
            mb[r * (4 * kMW) + c + mph] = img[(long long)(cy - kMR + r) * pitch + cx - kMR + c];
        }
    }
    eao::wave_sync();
    // intensity centroid over the radius-15 disc: lanes 0..30 / 32..62 take column u, the halves split the rows
    const int u = (lane & 31) - 15, half = lane >> 5;
    int m10 = 0, m01 = 0;
    if (u <= 15) {
        // rows v = -15..0 (lower half-wave) / 1..15 (upper); end of each row of the radius-15 disc = kUmaxTab[|v|]
        const uint8_t* c0 = reinterpret_cast<const uint8_t*>(mpatch[wv]) + kMR * (4 * kMW) + kMR + mph + u;
        const int au = u < 0 ? -u : u;
        // One row per step and half-wave; the row ends come into the code as IMMEDIATES (template constants).  Written as a
        // loop over a constexpr table the compiler selected between two table ADDRESSES and loaded the bound from constant
        // memory -- a dependent global load (plus s_waitcnt vmcnt(0)) in every one of the sixteen steps: 85 of the kernel's
        // 300 us.  Pixels outside the disc count as zero (the window holds every (u, v) with |u|, |v| <= 15).
        auto row = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int umLo = kUmaxTab[15 - k], umHi = kUmaxTab[k + 1];
            const int v = half ? k + 1 : k - 15;
            const int um = half ? umHi : umLo;
            const int pix = c0[v * (4 * kMW)];
            const int val = au <= um ? pix : 0;
            m10 += u * val;
            m01 += v * val;
        };
#define EAO_ROW(K) row(std::integral_constant<int, K>{});
        EAO_ROW(0) EAO_ROW(1) EAO_ROW(2) EAO_ROW(3) EAO_ROW(4) EAO_ROW(5) EAO_ROW(6) EAO_ROW(7)
        EAO_ROW(8) EAO_ROW(9) EAO_ROW(10) EAO_ROW(11) EAO_ROW(12) EAO_ROW(13) EAO_ROW(14) EAO_ROW(15)
#undef EAO_ROW
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // steered BRIEF on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    sincos_f32_via_f64(angle * factorPI, &b, &a);
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (lane + 64 * k < (2 * kPR + 1) * kPW) patch[wv][lane + 64 * k] = pw[k];
    eao::wave_sync();   // (one wave per keypoint: LDS writes of a wave are ordered before its later reads)
    const uint8_t* bc = reinterpret_cast<const uint8_t*>(patch[wv]) + kPR * (4 * kPW) + kPR + ph;
    unsigned long long words[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float xa = (float)(signed char)(pat[k] & 0xFF), ya = (float)(signed char)((pat[k] >> 8) & 0xFF);
        const float xb = (float)(signed char)((pat[k] >> 16) & 0xFF), yb = (float)(signed char)(pat[k] >> 24);
        const int ra = __float2int_rn(xa * b + ya * a), ca = __float2int_rn(xa * a - ya * b);
        const int rb = __float2int_rn(xb * b + yb * a), cb = __float2int_rn(xb * a - yb * b);
        const int t0 = bc[ra * (4 * kPW) + ca], t1 = bc[rb * (4 * kPW) + cb];
        words[k] = __ballot(t0 < t1);
    }
    if (lane == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((long long)f * cap + jout) * 32);
        d[0] = words[0]; d[1] = words[1]; d[2] = words[2]; d[3] = words[3];
        eao_keypoint kp;
        const float fxp = (float)cx, fyp = (float)cy;
        kp.x = l ? fxp * L.scale : fxp;
        kp.y = l ? fyp * L.scale : fyp;
        kp.size = (float)L.scaledPatch;
        kp.angle = angle;
        kp.response = (float)(key >> 24);
        kp.octave = l;
        kp.class_id = -1;
        kps[(long long)f * cap + jout] = kp;
    }
}

// unpack (x, y, response) of the gathered FAST candidates for the stage tap
__global__ void k_unpack_cand(const unsigned* __restrict__ keys, int n, float* __restrict__ xyr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned k = keys[i];
        xyr[3 * i] = (float)(k & 0xFFF);
        xyr[3 * i + 1] = (float)((k >> 12) & 0xFFF);
        xyr[3 * i + 2] = (float)(k >> 24);
    }
}


}  // namespace

// ---------------------------------------------------------------------------------------------- stereo matching
// Frame::ComputeStereoMatches (src/Frame.cc:841-1013): one wavefront per left keypoint.
//   1. descriptor search over the right keypoints whose row band (kp.y +- 2 scale[octave]) contains the left row, octave
//      within +-1, uR in [uL - maxD, uL + 3]: best Hamming distance below TH_HIGH, lowest index on ties (upstream walks
//      its per-row lists in index order);
//   2. 11x11 L1 patch correlation on the pyramid level of the left keypoint for the 11 shifts incR = -5 .. 5 (both patches
//      minus their centre pixel: integer arithmetic, exact), parabola through the best shift, disparity, depth.
// mvImagePyramid's 19-px reflect-101 border is not materialised on the device; the patch reads mirror the coordinate.
struct StereoArgs {
    int nl, nr, frame;
    const eao_keypoint* kl; const eao_keypoint* kr;
    const uint4* dl; const uint4* dr;
    const float* scale; const float* invScale;
    float mb, mbf;
    float* uRight; float* depth; int* sad;
};
__device__ __forceinline__ int px_reflect(const uint8_t* img, int pitch, int w, int h, int x, int y) {
    x = x < 0 ? -x : (x >= w ? 2 * w - 2 - x : x);
    y = y < 0 ? -y : (y >= h ? 2 * h - 2 - y : y);
    return img[(long long)y * pitch + x];
}
__global__ __launch_bounds__(256) void k_stereo_match(const Geom* __restrict__ g, ImgSrc sl, ImgSrc sr, StereoArgs A) {
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= A.nl) return;
    const eao_keypoint kpL = A.kl[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;                         // vRowIndices[vL]: float -> index truncation
    const float minZ = A.mb, minD = -3, maxD = A.mbf / minZ;
    const float minU = uL - maxD, maxU = uL - minD;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    if (!(maxU < 0)) {
        const uint4 a0 = A.dl[2 * iL], a1 = A.dl[2 * iL + 1];
        unsigned best = 0xFFFFFFFFu;                 // (distance << 16 | right index), smallest wins
        for (int iR = lane; iR < A.nr; iR += 64) {
            const eao_keypoint kr = A.kr[iR];
            const float r = 2.0f * A.scale[kr.octave];
            const int maxr = (int)ceilf(kr.y + r), minr = (int)floorf(kr.y - r);
            if (row < minr || row > maxr) continue;
            if (kr.octave < levelL - 1 || kr.octave > levelL + 1) continue;
            if (!(kr.x >= minU && kr.x <= maxU)) continue;
            const uint4 b0 = A.dr[2 * iR], b1 = A.dr[2 * iR + 1];
            const unsigned d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                               __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            best = min(best, (d << 16) | (unsigned)iR);
        }
        for (int dlt = 32; dlt >= 1; dlt >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, dlt));
        const int bestDist = (int)(best >> 16);
        if (best != 0xFFFFFFFFu && bestDist < refc::TH_HIGH) {
            const int bestIdxR = (int)(best & 0xFFFF);
            const float uR0 = A.kr[bestIdxR].x;
            const float sf = A.invScale[levelL];
            const int suL = (int)roundf(uL * sf), svL = (int)roundf(vL * sf), suR0 = (int)roundf(uR0 * sf);
            const int w = 5, Lw = 5;
            int pl, pr;
            const uint8_t* imgL = level_ptr(g, sl, levelL, A.frame, &pl);
            const uint8_t* imgR = level_ptr(g, sr, levelL, A.frame, &pr);
            const int lw = g->L[levelL].w, lh = g->L[levelL].h;
            const float iniu = (float)(suR0 + Lw - w), endu = (float)(suR0 + Lw + w + 1);
            if (!(iniu < 0 || endu >= (float)lw)) {
                const int cL = px_reflect(imgL, pl, lw, lh, suL, svL);
                // this lane's two patch pixels (121 = 64 + 57)
                int dL0 = 0, dL1 = 0, px0x = 0, px0y = 0, px1x = 0, px1y = 0;
                const bool has1 = lane + 64 < 121;
                px0y = lane / 11; px0x = lane - px0y * 11;
                dL0 = px_reflect(imgL, pl, lw, lh, suL - w + px0x, svL - w + px0y) - cL;
                if (has1) {
                    px1y = (lane + 64) / 11; px1x = lane + 64 - px1y * 11;
                    dL1 = px_reflect(imgL, pl, lw, lh, suL - w + px1x, svL - w + px1y) - cL;
                }
                int dists[11];
                int bestSad = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
                for (int inc = -5; inc <= 5; inc++) {
                    const int cR = px_reflect(imgR, pr, lw, lh, suR0 + inc, svL);
                    int sacc = abs(dL0 - (px_reflect(imgR, pr, lw, lh, suR0 + inc - w + px0x, svL - w + px0y) - cR));
                    if (has1) sacc += abs(dL1 - (px_reflect(imgR, pr, lw, lh, suR0 + inc - w + px1x, svL - w + px1y) - cR));
                    for (int dlt = 32; dlt >= 1; dlt >>= 1) sacc += __shfl_xor(sacc, dlt);
                    dists[inc + 5] = sacc;
                    if (sacc < bestSad) { bestSad = sacc; bestinc = inc; }
                }
                if (!(bestinc == -Lw || bestinc == Lw)) {
                    float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                    for (int k = 0; k < 11; k++) {
                        if (k == Lw + bestinc - 1) dist1 = (float)dists[k];
                        if (k == Lw + bestinc) dist2 = (float)dists[k];
                        if (k == Lw + bestinc + 1) dist3 = (float)dists[k];
                    }
                    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = A.scale[levelL] * ((float)suR0 + (float)bestinc + deltaR);
                        float disparity = uL - bestuR;
                        if (disparity >= 0 && disparity < maxD) {
                            if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
                            outD = A.mbf / disparity;
                            outU = bestuR;
                            outS = bestSad;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) { A.uRight[iL] = outU; A.depth[iL] = outD; A.sad[iL] = outS; }
}

namespace eao {
namespace orb {

// enqueue the whole pipeline for `batch` frames; level 0 is read from `src` (device memory)
eao_status enqueue_direct(eao_orb* h, const uint8_t* d_img, int pitch0, long long fs0, int batch, eao_keypoint* d_kps,
                          uint8_t* d_desc, int cap, int* d_n, hipStream_t st, int lanes) {
    const Geom& g = h->geom;
    eao::Range rAll("eao_orb_extract: enqueue");
    ImgSrc s;
    s.img0 = d_img; s.pitch0 = pitch0; s.fs0 = fs0; s.pyr = h->d_pyr.p;
    h->lastSrc = s; h->lastBatch = batch; h->lastComplete = false; h->pyrFrame = -1;
    const bool prof = h->profiling;
    hipEvent_t* ev = nullptr;
    if (prof) {
        if (h->evUsed + kProfEvents > h->evs.size()) {
            const size_t old = h->evs.size();
            h->evs.resize(old + kProfEvents * 16, nullptr);
            for (size_t i = old; i < h->evs.size(); i++) EAO_HIP(hipEventCreate(&h->evs[i]));
        }
        ev = &h->evs[h->evUsed];
        h->evUsed += kProfEvents;
        EAO_HIP(hipEventRecord(ev[0], st));
    }
    // split the batch into up to kLanes contiguous slices; slice i runs pyramid -> {FAST -> quad-tree | blur} ->
    // orientation + description on its own stream pair.  Profiling events bracket the stages of slice 0.
    lanes = std::max(1, std::min(std::min(eao_orb::kLanes, batch), lanes));
    // the handle's pyramid / candidate scratch is shared by its calls: a call enqueued on another stream than the previous
    // one waits for it (same stream: already ordered, the wait costs nothing)
    // (no event is recorded at the end of a call for this: an event record between two kernels of a stream costs ~5 us on this
    //  runtime, every call would pay it, and consecutive calls on one handle almost always share their stream -- the rare
    //  change of stream waits for the previous one on the host instead)
    //  (the previous stream may have been DESTROYED by its owner in the meantime -- a garbage-collected
    //  torch.cuda.Stream -- so its handle is never touched again: the rare change of stream drains the device instead)
    if (h->evLastValid && !h->capturing && h->lastStream != st) {
        eao_status so = order_behind_last_call(h, st);
        if (so) return so;
    }
    if (!h->capturing) h->lastStream = st;
    EAO_HIP(hipEventRecord(h->evStart, st));
    for (int i = 0; i < lanes; i++) {
        const int f0 = (int)((long long)batch * i / lanes), f1 = (int)((long long)batch * (i + 1) / lanes), nb = f1 - f0;
        // One lane (the default): the main chain runs on the CALLER's stream itself -- no hand-over to a private stream and
        // back, which put two cross-stream event hops (~25 us on PyTorch's bundled HIP 7.0 runtime) between consecutive calls.
        const bool onCaller = lanes == 1;
        hipStream_t ms = onCaller ? st : h->laneMain[i], ss = h->laneSide[i];

        const bool pe = prof && i == 0;
        if (!onCaller) EAO_HIP(hipStreamWaitEvent(ms, h->evStart, 0));
        // Level 0 IS the input image: its FAST cells (a third of all cells) do not wait for the pyramid.  Outside profiling
        // runs they start on the side stream at once and overlap the seven resize launches; the side stream then goes on
        // with the blur as before.  (Profiling runs keep the stages sequential so that each is timed alone.)
        const int cells0 = g.L[0].nCells;
        const bool early0 = !prof && g.nlevels > 1 && g.L[0].cellBase == 0 && cells0 < g.totalCells;
        // frame -> XCD placement of the resize chain and of the blur (round 5; EAO_RESIZE_AFFINITY=0 / EAO_BLUR_AFFINITY=0: plain order)
        static const int envRAff = getenv("EAO_RESIZE_AFFINITY") ? atoi(getenv("EAO_RESIZE_AFFINITY")) : 1;
        static const int envBAff = getenv("EAO_BLUR_AFFINITY") ? atoi(getenv("EAO_BLUR_AFFINITY")) : 1;
        const int resizeAff = envRAff && (nb & 7) == 0 ? 1 : 0, blurAff = envBAff && (nb & 7) == 0 ? 1 : 0;
        auto fast = [&](hipStream_t str, int first, int end) {
            eao::Range rg("orb: FAST cells");
            static const int envAff = getenv("EAO_FAST_AFFINITY") ? atoi(getenv("EAO_FAST_AFFINITY")) : 1;
            static const int envStop = getenv("EAO_FAST_STOP_AFTER") ? atoi(getenv("EAO_FAST_STOP_AFTER")) : 0;      // (census runs only: results are garbage)
            const int aff = (envAff && (nb & 7) == 0 ? 1 : 0) | envStop << 8;
            const bool whole = first == 0 && end == g.totalCells, narrow = g.fastStride == 48;
            auto* kern = whole ? (narrow ? k_fast_cells<true, 48> : k_fast_cells<true, kTileStrideWide>)
                               : (narrow ? k_fast_cells<false, 48> : k_fast_cells<false, kTileStrideWide>);
            hipLaunchKernelGGL(kern, dim3(128 * ((end - first + 127) / 128), nb), dim3(64), g.fastLdsBytes, str, h->d_geom.p, h->d_cells.p, s,
                               h->d_cellcand.p, h->d_cellcnt.p, f0, first, end, aff);
        };
        // Schedule of a (non-profiled) batch -- FAST and the blur are both VALU-bound, the quad-tree leaves the machine
        // almost idle, so the blur runs beside the quad-tree, not beside FAST:
        //   main:  resize 1..n-1 -> FAST(levels >= mid) -> quad-tree -> orientation + description
        //   side:  FAST(level 0) -> FAST(levels 1..mid-1) once they exist -> (all FAST done) blur
        // mid = 3: FAST of levels 1 and 2 joins level 0 on the side stream as soon as they exist, beside the rest of the pyramid
        // chain, whose small launches leave the machine idle (measured, ms per step, mid 0 / 3: batch 32 0.296 / 0.296, 64
        // 0.420 / 0.417, 128 0.717 / 0.706, 256 1.339 / 1.314; mid 4 = 1.314, mid 5 = 1.352).  EAO_ORB_MID=0 switches it off.
        // A single frame pays for the extra launch and event instead: 0.220 vs 0.211 ms per call -- small batches keep mid = 0.
        static const int envMid = getenv("EAO_ORB_MID") ? atoi(getenv("EAO_ORB_MID")) : -1;
        const int wantMid = envMid >= 0 ? envMid : (nb >= 48 ? 3 : 0);
        const int mid = early0 && wantMid >= 2 && wantMid < g.nlevels ? wantMid : 0;
        static const int envQtEarly = getenv("EAO_ORB_QT_EARLY") ? atoi(getenv("EAO_ORB_QT_EARLY")) : -1;
        const bool qtEarly = mid > 0 && (envQtEarly >= 0 ? envQtEarly != 0 : nb <= 96);
        // word loads need a 4-byte aligned source whose rows can be read up to the next multiple of 4
        const int src0Aligned = ((((uintptr_t)s.img0 | (uintptr_t)s.pitch0 | (uintptr_t)(batch > 1 ? s.fs0 : 0)) & 3) == 0) ? 1 : 0;
        auto quadtree = [&](hipStream_t str, int lFirst, int nLev) {
            eao::Range rg("orb: quad-tree");
            launch_quadtree(h, str, nb, f0, lFirst, nLev);
        };
        // Measured (device-resident step, ms): batch 1: 0.117 fused vs 0.142 chain, 8: 0.136 / 0.160, 32: 0.231 / 0.235, 64: 0.353 /
        // 0.309 -- small batches are bound by the chain of dependent launches, large ones by the VALU, where the fused kernel's
        // recomputed fringes (+35 % pixels) and its seven barrier-separated levels per workgroup lose.  EAO_ORB_PYRAMID=fused
        // forces it for any batch, =chain disables it.
        static const bool envFused = getenv("EAO_ORB_PYRAMID") && !strcmp(getenv("EAO_ORB_PYRAMID"), "fused");
        if (h->pyrFused && (envFused || nb <= 32)) {
            // ---- the pyramid is ONE launch (k_pyramid_fused): no chain to hide things behind, so the schedule is simply
            //   main:  pyramid -> FAST(levels >= 1) -> quad-trees(levels >= 1) ----------------------> orientation + description
            //   side:  FAST(level 0) -> quad-tree(level 0, the longest) -> (pyramid done) blur --^
            // (profiled calls: every stage alone, one after the other)
            const dim3 pgrid(h->pyr.tilesX * h->pyr.tilesY, 1, nb);
            if (prof) {
                { eao::Range rg("orb: pyramid"); hipLaunchKernelGGL(k_pyramid_fused, pgrid, dim3(kPyrThreads), h->pyrLds, ms, h->pyr, s, f0); }
                if (pe) EAO_HIP(hipEventRecord(ev[1], ms));
                EAO_HIP(hipEventRecord(h->evFork[i], ms));
                EAO_HIP(hipStreamWaitEvent(ss, h->evFork[i], 0));
                if (pe) EAO_HIP(hipEventRecord(ev[6], ss));
                { eao::Range rg("orb: blur"); hipLaunchKernelGGL(k_blur7, dim3(g.totalTiles, nb), dim3(256), 0, ss, h->d_geom.p, s, h->d_blur.p, f0, src0Aligned, blurAff); }
                if (pe) EAO_HIP(hipEventRecord(ev[7], ss));
                EAO_HIP(hipEventRecord(h->evJoin[i], ss));
                EAO_HIP(hipStreamWaitEvent(ms, h->evJoin[i], 0));
                if (pe) EAO_HIP(hipEventRecord(ev[9], ms));
                fast(ms, 0, g.totalCells);
                if (pe) EAO_HIP(hipEventRecord(ev[2], ms));
                quadtree(ms, 0, g.nlevels);
                if (pe) EAO_HIP(hipEventRecord(ev[3], ms));
                if (pe) EAO_HIP(hipEventRecord(ev[4], ms));
            } else {
                EAO_HIP(hipStreamWaitEvent(ss, h->evStart, 0));
                fast(ss, 0, cells0);
                quadtree(ss, 0, 1);
                { eao::Range rg("orb: pyramid"); hipLaunchKernelGGL(k_pyramid_fused, pgrid, dim3(kPyrThreads), h->pyrLds, ms, h->pyr, s, f0); }
                EAO_HIP(hipEventRecord(h->evFork[i], ms));
                EAO_HIP(hipStreamWaitEvent(ss, h->evFork[i], 0));
                { eao::Range rg("orb: blur"); hipLaunchKernelGGL(k_blur7, dim3(g.totalTiles, nb), dim3(256), 0, ss, h->d_geom.p, s, h->d_blur.p, f0, src0Aligned, blurAff); }
                EAO_HIP(hipEventRecord(h->evJoin[i], ss));
                fast(ms, cells0, g.totalCells);
                quadtree(ms, 1, g.nlevels - 1);
                EAO_HIP(hipStreamWaitEvent(ms, h->evJoin[i], 0));
            }
            { eao::Range rg("orb: orientation + description"); hipLaunchKernelGGL(k_orient_describe, dim3(eao::cdiv(std::min(cap, g.totalKpCap), 4), nb), dim3(256), 0, ms, h->d_geom.p, s,
                               h->d_blur.p, h->d_levelkps.p, h->d_levelcnt.p, d_kps, d_desc, d_n, cap, f0, g.nlevels, 0, g.nlevels, 1); }
            if (pe) EAO_HIP(hipEventRecord(ev[5], ms));
            if (!onCaller) {
                EAO_HIP(hipEventRecord(h->evDone[i], ms));
                EAO_HIP(hipStreamWaitEvent(st, h->evDone[i], 0));
            }
            continue;
        }
        if (early0) {
            EAO_HIP(hipStreamWaitEvent(ss, h->evStart, 0));
            fast(ss, 0, cells0);
        }
        // (Round 3 measured four more schedule variants and a multi-keypoint description kernel here -- the upper levels' FAST split over a third
        //  stream, the blur on its own stream with a split description, the top of the pyramid as one launch, several keypoints per wave -- none
        //  faster than this schedule; they were removed in round 4, DESIGN.md section 5 keeps the measurements and the commits that hold the code.)
        auto orient = [&](hipStream_t str, int lFirst, int lEnd, int writeN) {
            eao::Range rg("orb: orientation + description");
            const int rangeCap = (lEnd < g.nlevels ? g.L[lEnd].kpBase : g.totalKpCap) - g.L[lFirst].kpBase;
            hipLaunchKernelGGL(k_orient_describe, dim3(eao::cdiv(std::min(cap, rangeCap), 4), nb), dim3(256), 0, str, h->d_geom.p, s, h->d_blur.p, h->d_levelkps.p,
                               h->d_levelcnt.p, d_kps, d_desc, d_n, cap, f0, g.nlevels, lFirst, lEnd, writeN);
        };
        for (int l = 1; l < g.nlevels; l++) {
            const int G4 = eao::cdiv(g.L[l].w, 4);
            // ONE-WAVE workgroups: the chain's small launches run beside FAST launches that fill every CU with one-wave workgroups;
            // a 256-thread workgroup needs four wave slots free on ONE CU at the same moment and kept losing them to the next FAST
            // wave -- levels 2 and 4 took 32 / 37 us instead of 9, and the pyramid was the critical path of the step up to 121 us
            // (64 frames: 0.296 -> 0.279 ms per step; EAO_RESIZE_WAVES=4 restores the 256-thread workgroups)
            static const int envRW = getenv("EAO_RESIZE_WAVES") ? atoi(getenv("EAO_RESIZE_WAVES")) : 1;
            const int rw = envRW == 2 || envRW == 4 ? envRW : 1;
            dim3 grid(eao::cdiv(G4 * eao::cdiv(g.L[l].h, kResizeRows), 64 * rw), 1, nb), block(64, rw);
            ResizeArgs ra;
            ra.G = G4; ra.invG = 1.0f / (float)G4;
            ra.D = {g.L[l].w, g.L[l].h, g.L[l].pitch, g.L[l].off};
            ra.S = {g.L[l - 1].w, g.L[l - 1].h, g.L[l - 1].pitch, g.L[l - 1].off};
            ra.invX = 1. / ((double)g.L[l].w / g.L[l - 1].w); ra.invY = 1. / ((double)g.L[l].h / g.L[l - 1].h);
            ra.srcIsInput = l == 1; ra.pyrFrameBytes = g.pyrFrameBytes;
            ra.frameAffinity = resizeAff;
            { eao::Range rg("orb: pyramid"); hipLaunchKernelGGL(k_resize, grid, block, 0, ms, ra, s, f0); }
            if (l == mid - 1) {
                EAO_HIP(hipEventRecord(h->evMid[i], ms));
                EAO_HIP(hipStreamWaitEvent(ss, h->evMid[i], 0));
                fast(ss, cells0, g.L[mid].cellBase);
                // ... and, up to 96 frames, their quad-trees right behind, beside the main stream's FAST of the upper levels
                // instead of beside the blur afterwards: a quad-tree launch lasts as long as its level-0 workgroup (~60 us), and
                // at these batches that is the long pole of the last-but-one segment (64 frames: 0.403 -> 0.390 ms per step).
                // At 256 frames it is slower (1.258 vs 1.226 ms: 768 workgroups with 42 KB of LDS each take the occupancy FAST
                // needs), so large batches keep all eight levels after FAST.  EAO_ORB_QT_EARLY=0 / 1 overrides.
                if (qtEarly) quadtree(ss, 0, mid);
            }
        }
        if (early0) EAO_HIP(hipEventRecord(h->evFast0[i], ss));   // the side stream's share of FAST (and of the quad-trees)
        if (pe) EAO_HIP(hipEventRecord(ev[1], ms));
        if (prof) {   // profiled calls: every stage alone -- blur, then FAST
            EAO_HIP(hipEventRecord(h->evFork[i], ms));
            EAO_HIP(hipStreamWaitEvent(ss, h->evFork[i], 0));
            if (pe) EAO_HIP(hipEventRecord(ev[6], ss));
            { eao::Range rg("orb: blur"); hipLaunchKernelGGL(k_blur7, dim3(g.totalTiles, nb), dim3(256), 0, ss, h->d_geom.p, s, h->d_blur.p, f0, src0Aligned, blurAff); }
            if (pe) EAO_HIP(hipEventRecord(ev[7], ss));
            EAO_HIP(hipEventRecord(h->evJoin[i], ss));
            EAO_HIP(hipStreamWaitEvent(ms, h->evJoin[i], 0));
            if (pe) EAO_HIP(hipEventRecord(ev[9], ms));
            static const int envSplitProf = getenv("EAO_FAST_SPLIT_PROF") ? atoi(getenv("EAO_FAST_SPLIT_PROF")) : 0;      // (diagnostic: the three launches of the chain schedule, each alone)
            if (envSplitProf && g.nlevels > 3) { fast(ms, 0, cells0); fast(ms, cells0, g.L[3].cellBase); fast(ms, g.L[3].cellBase, g.totalCells); }
            else fast(ms, 0, g.totalCells);
        } else {
            // The blur's fork point is the END OF THE PYRAMID, not the end of the main stream's FAST: on the side stream the blur
            // follows the quad-trees of the lower levels in stream order, which end when the main stream's FAST does, so the
            // blur still runs beside the upper levels' quad-trees -- but without a cross-stream hand-over (~10 us) in front of
            // it (64 frames: 0.2784 -> 0.2762 ms, 256 frames: 1.092 -> 1.071; measured the other way round before the resize
            // launches became one-wave workgroups, when the pyramid ended 20 us later).  EAO_ORB_BLUR_EARLY=0 restores the old point.
            static const int envBlurEarly = getenv("EAO_ORB_BLUR_EARLY") ? atoi(getenv("EAO_ORB_BLUR_EARLY")) : 1;
            if (envBlurEarly) EAO_HIP(hipEventRecord(h->evFork[i], ms));
            fast(ms, early0 ? (mid ? g.L[mid].cellBase : cells0) : 0, g.totalCells);
            if (!envBlurEarly) EAO_HIP(hipEventRecord(h->evFork[i], ms));
            EAO_HIP(hipStreamWaitEvent(ss, h->evFork[i], 0));
            { eao::Range rg("orb: blur"); hipLaunchKernelGGL(k_blur7, dim3(g.totalTiles, nb), dim3(256), 0, ss, h->d_geom.p, s, h->d_blur.p, f0, src0Aligned, blurAff); }
            EAO_HIP(hipEventRecord(h->evJoin[i], ss));
            if (early0 && !qtEarly) EAO_HIP(hipStreamWaitEvent(ms, h->evFast0[i], 0));
        }
        if (pe) EAO_HIP(hipEventRecord(ev[2], ms));
        quadtree(ms, qtEarly ? mid : 0, g.nlevels - (qtEarly ? mid : 0));
        // (the lower levels' quad-trees ran on the side stream, ahead of the blur: the wait for evJoin below covers them)
        if (pe) EAO_HIP(hipEventRecord(ev[3], ms));
        EAO_HIP(hipStreamWaitEvent(ms, h->evJoin[i], 0));
        if (pe) EAO_HIP(hipEventRecord(ev[4], ms));
        orient(ms, 0, g.nlevels, 1);
        if (pe) EAO_HIP(hipEventRecord(ev[5], ms));
        if (!onCaller) {
            EAO_HIP(hipEventRecord(h->evDone[i], ms));
            EAO_HIP(hipStreamWaitEvent(st, h->evDone[i], 0));
        }
    }
    if (prof) EAO_HIP(hipEventRecord(ev[8], st));
    if (!h->capturing) {
        h->evLastValid = true;
        if (h->everyCallEvent) { EAO_HIP(hipEventRecord(h->evLast, st)); h->evLastRecorded = true; }
    }
    EAO_HIP(hipGetLastError());
    return EAO_OK;
}


// Optional (EAO_ORB_GRAPH=1): capture the launch sequence of one batch into a hipGraph per argument set and replay it.
// Measured on MI355X (tools/dbg_lanes.py): replay 0.736 ms vs direct launches 0.742 ms per 64-frame batch, and cutting
// the batch into 2 / 4 concurrent lanes (EAO_ORB_LANES) is SLOWER (0.75 / 0.94 ms): the kernels are throughput-bound,
// not latency-bound, so the defaults are one lane and direct launches.  (Graph replay also crashes inside the HIP 7.0
// runtime bundled with PyTorch 2.10; it works with the ROCm 7.2 runtime.)  Profiled calls always bypass the graph.
eao_status enqueue(eao_orb* h, const uint8_t* d_img, int pitch0, long long fs0, int batch, eao_keypoint* d_kps,
                   uint8_t* d_desc, int cap, int* d_n, hipStream_t st) {
    static const int envLanes = getenv("EAO_ORB_LANES") ? atoi(getenv("EAO_ORB_LANES")) : 1;
    static const bool envGraph = getenv("EAO_ORB_GRAPH") && atoi(getenv("EAO_ORB_GRAPH")) != 0;
    if (h->profiling || !envGraph) return enqueue_direct(h, d_img, pitch0, fs0, batch, d_kps, d_desc, cap, d_n, st, envLanes);
    const GraphKey key = {d_img, pitch0, fs0, batch, d_kps, d_desc, cap, d_n, envLanes};
    if (!h->graphExec || !(h->graphKey == key)) {
        if (h->graphExec) { (void)hipGraphExecDestroy(h->graphExec); h->graphExec = nullptr; }
        hipGraph_t graph = nullptr;
        EAO_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->capturing = true;
        const eao_status st2 = enqueue_direct(h, d_img, pitch0, fs0, batch, d_kps, d_desc, cap, d_n, h->stream, envLanes);
        h->capturing = false;
        const hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (st2) { if (graph) (void)hipGraphDestroy(graph); return st2; }
        if (e != hipSuccess) { eao::set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return EAO_ERR_NO_DEVICE; }
        EAO_HIP(hipGraphInstantiate(&h->graphExec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        h->graphKey = key;
    }
    h->lastBatch = batch; h->lastComplete = false; h->pyrFrame = -1;
    // the same ordering rule as the direct path: the replayed launches use the handle's scratch
    if (h->evLastValid && h->lastStream != st) {
        eao_status so = order_behind_last_call(h, st);
        if (so) return so;
    }
    h->lastStream = st;
    h->evLastValid = true;
    EAO_HIP(hipGraphLaunch(h->graphExec, st));
    if (h->everyCallEvent) { EAO_HIP(hipEventRecord(h->evLast, st)); h->evLastRecorded = true; }
    return EAO_OK;
}

// The download of a streaming slot: 16 bytes per lane from HBM into MAPPED PINNED host memory, as a kernel.
// (A hipMemcpyAsync on a third stream, waiting for the extraction's event, blocked the copy queue the next slot's upload sits in: with
// three slots in flight upload, extraction and download ran one after the other, 0.94 ms per 64 frames; two slots hid it by accident.)
__global__ __launch_bounds__(256) void k_stream_download(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// eao_orb_pyramid: every level of one frame WITH its BORDER_REFLECT_101 frame (what upstream keeps in mvImagePyramid, src/ORBextractor.cc:1113-1128), built by
// ONE launch straight into mapped pinned host memory: a lane assembles 16 bytes of a bordered row (reflected coordinates, byte gathers out of the L2-resident
// level) and stores them over PCIe.  ~1.3 MB per 640 x 480 frame; replaces 16 eao_orb_level calls + a per-pixel host loop in the adapter.
struct PyrExportLevel { int w, h, srcPitch, srcOff, dstPitch, rowChunks, chunkBase; long long dstOff; };
struct PyrExportArgs { int nlevels, border, totalChunks, pyrFrameBytes; PyrExportLevel L[kMaxLevels]; };
__global__ __launch_bounds__(256) void k_pyramid_export(PyrExportArgs A, ImgSrc s, int frame, uint8_t* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= A.totalChunks) return;
    int l = 0;
    while (l + 1 < A.nlevels && idx >= A.L[l + 1].chunkBase) l++;
    const PyrExportLevel& L = A.L[l];
    const int r = idx - L.chunkBase, row = r / L.rowChunks, ch = r - row * L.rowChunks;
    const uint8_t* src = l == 0 ? s.img0 + (long long)frame * s.fs0 : s.pyr + (long long)frame * A.pyrFrameBytes + L.srcOff;
    const int pitch = l == 0 ? s.pitch0 : L.srcPitch;
    int y = row - A.border;
    y = y < 0 ? -y : (y >= L.h ? 2 * L.h - 2 - y : y);
    y = min(max(y, 0), L.h - 1);
    const uint8_t* srow = src + (long long)y * pitch;
    unsigned w4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int x = ch * 16 + q * 4 + b - A.border;
            x = x < 0 ? -x : (x >= L.w ? 2 * L.w - 2 - x : x);
            x = min(max(x, 0), L.w - 1);       // (bytes of the row pitch beyond the right border)
            v |= (unsigned)srow[x] << (8 * b);
        }
        w4[q] = v;
    }
    *reinterpret_cast<uint4*>(dst + L.dstOff + (long long)row * L.dstPitch + ch * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
}

// enqueue the export of frame `frame`'s bordered levels on the handle's stream (no synchronisation); fills h->pyrViews
eao_status enqueue_pyramid_export(eao_orb* h, int frame, int border) {
    const Geom& g = h->geom;
    for (int l = 0; l < g.nlevels; l++) EAO_REQUIRE(border < g.L[l].w && border < g.L[l].h, "level %d is smaller than the border", l);
    PyrExportArgs A;
    A.nlevels = g.nlevels; A.border = border; A.pyrFrameBytes = g.pyrFrameBytes;
    size_t off = 0;
    int chunks = 0;
    for (int l = 0; l < g.nlevels; l++) {
        PyrExportLevel& L = A.L[l];
        L.w = g.L[l].w; L.h = g.L[l].h; L.srcPitch = g.L[l].pitch; L.srcOff = g.L[l].off;
        L.dstPitch = (L.w + 2 * border + 63) & ~63;
        L.rowChunks = L.dstPitch / 16; L.chunkBase = chunks; L.dstOff = (long long)off;
        chunks += L.rowChunks * (L.h + 2 * border);
        off = (off + (size_t)L.dstPitch * (L.h + 2 * border) + 255) & ~(size_t)255;
    }
    A.totalChunks = chunks;
    if (h->pinPyrCap < off) {
        EAO_HIP(eao::wait_latency(h->stream));      // (a previous export may still be writing the old block)
        if (h->pinPyr) (void)hipHostFree(h->pinPyr);
        h->pinPyr = nullptr; h->pinPyrCap = 0;
        EAO_HIP(hipHostMalloc((void**)&h->pinPyr, off, hipHostMallocMapped));
        h->pinPyrCap = off;
    }
    unsigned char* dv = nullptr;
    EAO_HIP(hipHostGetDevicePointer((void**)&dv, h->pinPyr, 0));
    hipLaunchKernelGGL(k_pyramid_export, dim3(eao::cdiv(chunks, 256)), dim3(256), 0, h->stream, A, h->lastSrc, frame, dv);
    for (int l = 0; l < g.nlevels; l++) {
        h->pyrViews[l].data = h->pinPyr + A.L[l].dstOff + (size_t)border * A.L[l].dstPitch + border;
        h->pyrViews[l].width = A.L[l].w; h->pyrViews[l].height = A.L[l].h; h->pyrViews[l].step = A.L[l].dstPitch;
    }
    h->pyrFrame = frame; h->pyrBorder = border;
    return EAO_OK;
}

}  // namespace orb
}  // namespace eao

extern "C" {


eao_status eao_orb_stream_submit(eao_orb* h, int32_t slot, int32_t batch) {
    EAO_REQUIRE(h && slot >= 0 && slot < (int)h->slots.size(), "no such slot (eao_orb_stream_create first)");
    EAO_REQUIRE(batch >= 1 && batch <= h->sB, "a slot holds 1..%d frames", h->sB);
    eao_orb::StreamSlot& sl = h->slots[slot];
    if (sl.submitted) EAO_HIP(hipEventSynchronize(sl.evOut));      // the slot's previous results must have left the device buffers
    eao_status st = ensure(h, h->sW, h->sH, h->sB);
    if (st) return st;
    const size_t B = batch;
    static const int envSkip = getenv("EAO_STREAM_SKIP") ? atoi(getenv("EAO_STREAM_SKIP")) : 0;      // diagnostic: 1 upload, 2 extraction, 4 download left out
    if (!(envSkip & 1)) EAO_HIP(hipMemcpyAsync(sl.dIn, sl.pinIn, B * (size_t)h->sPitch * h->sH, hipMemcpyHostToDevice, h->sUp));
    EAO_HIP(hipEventRecord(sl.evIn, h->sUp));
    EAO_HIP(hipStreamWaitEvent(h->sRun, sl.evIn, 0));
    if (!(envSkip & 2)) st = enqueue(h, sl.dIn, h->sPitch, (long long)h->sPitch * h->sH, batch, (eao_keypoint*)(sl.dOut + h->sOffK), sl.dOut + h->sOffD, h->sCap, (int*)sl.dOut, h->sRun);
    if (st) return st;
    // the download is a KERNEL (see k_stream_download) on a stream of its own behind the extraction's event: a few workgroups that
    // store over PCIe beside the next slot's extraction; counts, keypoints and descriptors of the frames submitted lie in three ranges
    EAO_HIP(hipEventRecord(sl.evDone, h->sRun));
    EAO_HIP(hipStreamWaitEvent(h->sDown, sl.evDone, 0));
    const size_t parts[3][2] = {{0, (B * sizeof(int) + 15) & ~(size_t)15}, {h->sOffK, B * (size_t)h->sCap * sizeof(eao_keypoint)}, {h->sOffD, B * (size_t)h->sCap * 32}};
    for (int q = 0; q < 3 && !(envSkip & 4); q++) {
        const size_t n16 = (parts[q][1] + 15) / 16;
        const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 64);
        hipLaunchKernelGGL(k_stream_download, dim3(grid), dim3(256), 0, h->sDown, reinterpret_cast<const uint4*>(sl.dOut + parts[q][0]),
                           reinterpret_cast<uint4*>(sl.pinOutDev + parts[q][0]), n16);
    }
    EAO_HIP(hipEventRecord(sl.evOut, h->sDown));
    sl.batch = batch; sl.submitted = true;
    return EAO_OK;
}

eao_status eao_orb_level_candidates(eao_orb* h, int32_t frame, int32_t level, float* xyr, int32_t cap, int32_t* n) {
    EAO_REQUIRE(h && n && h->geomValid && h->lastBatch > 0, "no extraction has run on this handle");
    EAO_REQUIRE(frame >= 0 && frame < h->lastBatch && level >= 0 && level < h->geom.nlevels, "frame/level out of range");
    const Geom& g = h->geom;
    eao_status st0 = wait_last_extraction(h);
    if (st0) return st0;
    int cnt = 0;
    EAO_HIP(hipMemcpyAsync(&cnt, h->d_candcnt.p + frame * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    *n = cnt;
    if (!xyr || cnt == 0) return EAO_OK;
    const int m = std::min(cnt, cap);
    eao_status st = h->d_xyr.reserve((size_t)m * 3);
    if (st) return st;
    hipLaunchKernelGGL(k_unpack_cand, dim3(eao::cdiv(m, 256)), dim3(256), 0, h->stream,
                       h->d_cand.p + (long long)frame * g.totalCandCap + g.L[level].candBase, m, h->d_xyr.p);
    EAO_HIP(hipMemcpyAsync(xyr, h->d_xyr.p, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    EAO_HIP(eao::wait_latency(h->stream));
    return EAO_OK;
}


eao_status eao_compute_stereo_matches(eao_orb* left, eao_orb* right, int32_t frame, int32_t nl, const eao_keypoint* kps_l,
                                      const uint8_t* desc_l, int32_t nr, const eao_keypoint* kps_r, const uint8_t* desc_r,
                                      float mb, float mbf, float* u_right, float* depth) {
    EAO_REQUIRE(left && right && left->geomValid && right->geomValid && left->lastBatch > 0 && right->lastBatch > 0,
                "both extractors must have run on the stereo pair");
    EAO_REQUIRE(frame >= 0 && frame < left->lastBatch && frame < right->lastBatch, "frame out of range");
    const Geom& g = left->geom;
    EAO_REQUIRE(g.nlevels == right->geom.nlevels && g.L[0].w == right->geom.L[0].w && g.L[0].h == right->geom.L[0].h &&
                left->cfg.scale_factor == right->cfg.scale_factor, "the two extractors must share image size and pyramid");
    EAO_REQUIRE(nl >= 0 && nr >= 0 && nr < 65536 && (nl == 0 || (kps_l && desc_l && u_right && depth)) && (nr == 0 || (kps_r && desc_r)), "bad argument");
    for (int i = 0; i < nl; i++) {
        u_right[i] = -1.0f; depth[i] = -1.0f;
        EAO_REQUIRE(kps_l[i].octave >= 0 && kps_l[i].octave < g.nlevels, "left keypoint %d: octave out of range", i);
    }
    for (int i = 0; i < nr; i++) EAO_REQUIRE(kps_r[i].octave >= 0 && kps_r[i].octave < g.nlevels, "right keypoint %d: octave out of range", i);
    if (nl == 0 || nr == 0) return EAO_OK;
    eao_status stw = wait_last_extraction(left);     // (host-API extractions, what Frame's stereo constructor makes on its two threads, are final already)
    if (!stw) stw = wait_last_extraction(right);
    if (stw) return stw;
    eao::DevBuf<unsigned char>& buf = left->d_stereo;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t oKl = off; off = al(off + sizeof(eao_keypoint) * (size_t)nl);
    const size_t oKr = off; off = al(off + sizeof(eao_keypoint) * (size_t)nr);
    const size_t oDl = off; off = al(off + 32 * (size_t)nl);
    const size_t oDr = off; off = al(off + 32 * (size_t)nr);
    const size_t oSc = off; off = al(off + 8 * (size_t)g.nlevels);
    const size_t oOut = off; off = al(off + 12 * (size_t)nl);
    eao_status st = buf.reserve(off);
    if (st) return st;
    hipStream_t s = left->stream;
    EAO_HIP(hipMemcpyAsync(buf.p + oKl, kps_l, sizeof(eao_keypoint) * (size_t)nl, hipMemcpyHostToDevice, s));
    EAO_HIP(hipMemcpyAsync(buf.p + oKr, kps_r, sizeof(eao_keypoint) * (size_t)nr, hipMemcpyHostToDevice, s));
    EAO_HIP(hipMemcpyAsync(buf.p + oDl, desc_l, 32 * (size_t)nl, hipMemcpyHostToDevice, s));
    EAO_HIP(hipMemcpyAsync(buf.p + oDr, desc_r, 32 * (size_t)nr, hipMemcpyHostToDevice, s));
    std::vector<float> sc(2 * (size_t)g.nlevels);
    for (int l = 0; l < g.nlevels; l++) { sc[l] = left->scale[l]; sc[g.nlevels + l] = left->invScale[l]; }
    EAO_HIP(hipMemcpyAsync(buf.p + oSc, sc.data(), 8 * (size_t)g.nlevels, hipMemcpyHostToDevice, s));
    StereoArgs A;
    A.nl = nl; A.nr = nr; A.frame = frame;
    A.kl = (const eao_keypoint*)(buf.p + oKl); A.kr = (const eao_keypoint*)(buf.p + oKr);
    A.dl = (const uint4*)(buf.p + oDl); A.dr = (const uint4*)(buf.p + oDr);
    A.scale = (const float*)(buf.p + oSc); A.invScale = A.scale + g.nlevels;
    A.mb = mb; A.mbf = mbf;
    A.uRight = (float*)(buf.p + oOut); A.depth = A.uRight + nl; A.sad = (int*)(A.depth + nl);
    hipLaunchKernelGGL(k_stereo_match, dim3(eao::cdiv(nl, 4)), dim3(256), 0, s, left->d_geom.p, left->lastSrc, right->lastSrc, A);
    std::vector<int> sad(nl);
    EAO_HIP(hipMemcpyAsync(u_right, A.uRight, 4 * (size_t)nl, hipMemcpyDeviceToHost, s));
    EAO_HIP(hipMemcpyAsync(depth, A.depth, 4 * (size_t)nl, hipMemcpyDeviceToHost, s));
    EAO_HIP(hipMemcpyAsync(sad.data(), A.sad, 4 * (size_t)nl, hipMemcpyDeviceToHost, s));
    EAO_HIP(eao::wait_latency(s));
    EAO_HIP(hipGetLastError());
    // median-based rejection (src/Frame.cc:995-1012)
    std::vector<std::pair<int, int> > vDistIdx;
    for (int i = 0; i < nl; i++) if (sad[i] >= 0) vDistIdx.push_back(std::make_pair(sad[i], i));
    if (!vDistIdx.empty()) {
        std::sort(vDistIdx.begin(), vDistIdx.end());
        const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
            if (vDistIdx[i].first < thDist) break;
            u_right[vDistIdx[i].second] = -1; depth[vDistIdx[i].second] = -1;
        }
    }
    return EAO_OK;
}
}  // extern "C"
