// oracle/orb_cpu.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into or called by the product).
//
// Scalar restatement of the reference's ORB extraction path:
//   ORBextractor ctor            reference src/ORBextractor.cc:410-470
//   ComputePyramid               :1107-1132   (+ OpenCV 3.3.x resize INTER_LINEAR, generic u8 path)
//   ComputeKeyPointsOctTree      :765-853     (+ OpenCV 3.3.x cv::FAST 9/16 with NMS)
//   ExtractorNode::DivideNode    :481-537,  DistributeOctTree :539-763
//   IC_Angle / computeOrientation:77-104, 472-479 (+ OpenCV 3.x fastAtan2)
//   GaussianBlur 7x7 sigma 2     call site :1085-1086 (OpenCV 3.3.x u8 fixed-point separable filter)
//   computeOrbDescriptor         :108-147,  operator() :1043-1105
//
// PARITY UNPINNED: the reference ships no tests / golden vectors for this path, and its own sources do
// not compile here (OpenCV absent; writing stand-in headers is not allowed), so this restatement is
// pinned only by (a) the rBRIEF pattern table, compared number-for-number with the reference file by
// tests/test_oracle_orb.py, (b) constants derivable from the reference text (umax, quotas, level sizes),
// (c) independent numpy/scipy re-derivations of the OpenCV primitives in the tests.
// OpenCV itself is a third-party dependency that is NOT vendored in /root/reference (README.md:46 pins
// "opencv: 3.3.1" in prose only); FAST / resize / GaussianBlur / fastAtan2 follow the published generic
// (non-IPP, non-OpenCL) 3.3.x algorithms.
//
// Deliberate, documented choices where the reference is build- or run-dependent:
//  * no FMA contraction in float expressions (compile with -ffp-contract=off);
//  * cos/sin of the keypoint angle: libm double cos/sin rounded to float (the reference resolves to
//    cosf/sinf of whatever libm it was linked with);
//  * DistributeOctTree sorts (size, node*) pairs (:684): ties on size are broken by heap address upstream;
//    here by node creation sequence (later-created node compares greater).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>

#include "ref_constants.inc"   // GENERATED from the reference text (tools/gen_ref_constants.py): namespace refc

namespace {

constexpr int kPatch = refc::PATCH_SIZE;
constexpr int kHalfPatch = refc::HALF_PATCH_SIZE;
constexpr int kEdge = refc::EDGE_THRESHOLD;

const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

inline int cv_round(double v) { return (int)std::lrint(v); }  // round-half-even (default FP env)

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // stride == w
    const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
    uint8_t* row(int y) { return px.data() + (size_t)y * w; }
};

struct Cand {  // FAST candidate in (minBorder-relative) level coordinates
    float x, y, response;
};

struct KeyPoint {  // cv::KeyPoint POD mirror, 28 bytes
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

// ---------------------------------------------------------------- resize (INTER_LINEAR, u8, 1 channel)
void resize_linear_u8(const Image& src, Image& dst, int dw, int dh) {
    dst.w = dw; dst.h = dh; dst.px.assign((size_t)dw * dh, 0);
    const int sw = src.w, sh = src.h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        int a0 = cv_round(c0 * 2048.f), a1 = cv_round(c1 * 2048.f);
        alpha[2 * dx] = (short)std::min(std::max(a0, -32768), 32767);
        alpha[2 * dx + 1] = (short)std::min(std::max(a1, -32768), 32767);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        beta[2 * dy] = (short)std::min(std::max(cv_round(c0 * 2048.f), -32768), 32767);
        beta[2 * dy + 1] = (short)std::min(std::max(cv_round(c1 * 2048.f), -32768), 32767);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hpass = [&](int sy, std::vector<int>& out) {
        sy = std::min(std::max(sy, 0), sh - 1);
        const uint8_t* S = src.row(sy);
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int s1 = (sx + 1 < sw) ? S[sx + 1] : 0;  // weight is 0 whenever sx+1 is out of range
            out[dx] = S[sx] * alpha[2 * dx] + s1 * alpha[2 * dx + 1];
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        hpass(yofs[dy], r0);
        hpass(yofs[dy] + 1, r1);
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst.row(dy);
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ---------------------------------------------------------------- FAST 9/16 with 3x3 NMS on a sub-image
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// Returns true when (x,y) passes the segment test at `th`; *score = the stored u8 corner score.
// (An antipodal ring pair that is entirely within +-th rules out every 9-arc: cheap early exit.)
inline bool has_run9(uint32_t m16) {
    uint32_t x = m16 | (m16 << 16);
    x &= x >> 1; x &= x >> 2; x &= x >> 4; x &= x >> 1;  // bit i survives iff bits i..i+8 were all set
    return x != 0;
}
bool fast_test(const uint8_t* p, int stride, int th, int* score) {
    const int v = p[0];
    {
        int d0 = v - p[3 * stride], d8 = v - p[-3 * stride];
        if (d0 <= th && d0 >= -th && d8 <= th && d8 >= -th) return false;
        int d4 = v - p[3], d12 = v - p[-3];
        if (d4 <= th && d4 >= -th && d12 <= th && d12 >= -th) return false;
    }
    int d[16];
    uint32_t dark = 0, bright = 0;
    for (int k = 0; k < 16; k++) {
        d[k] = v - p[kRingDy[k] * stride + kRingDx[k]];
        dark |= (uint32_t)(d[k] > th) << k;
        bright |= (uint32_t)(-d[k] > th) << k;
    }
    if (!has_run9(dark) && !has_run9(bright)) return false;
    int best_dark = -256, best_bright = -256;  // max over 9-arcs of min(d) and of min(-d)
    for (int s = 0; s < 16; s++) {
        int mn = 255, mx = -255;
        for (int j = 0; j < 9; j++) {
            int dv = d[(s + j) & 15];
            mn = std::min(mn, dv);
            mx = std::max(mx, dv);
        }
        best_dark = std::max(best_dark, mn);
        best_bright = std::max(best_bright, -mx);
    }
    // OpenCV cornerScore<16>: a0 = max(th, dark arcs); b0 = min(-a0, -bright arcs); score = -b0 - 1
    int a0 = std::max(th, best_dark);
    int sc = std::max(a0, best_bright) - 1;
    *score = sc & 0xFF;  // stored as uchar
    return true;
}

// sub-image [x0,x0+w) x [y0,y0+h) of img; emits (x,y,score) relative to the sub-image, row-major.
void fast_cell(const Image& img, int x0, int y0, int w, int h, int th, std::vector<Cand>& out) {
    th = std::min(std::max(th, 0), 255);
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> sc((size_t)w * h, 0), is((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s;
            if (fast_test(img.row(y0 + y) + x0 + x, img.w, th, &s)) {
                is[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = (uint8_t)s;
            }
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!is[(size_t)y * w + x]) continue;
            int s = sc[(size_t)y * w + x];
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (!(s > sc[(size_t)(y + dy) * w + x + dx])) { keep = false; break; }
                }
            if (keep) out.push_back({(float)x, (float)y, (float)s});
        }
}

// ---------------------------------------------------------------- quad-tree distribution
struct Node {
    std::vector<int> keys;  // candidate indices, reference order preserved
    int x0, y0, x1, y1;     // UL.x, UL.y, UR.x(=BR.x), BL.y(=BR.y)
    bool leaf = false;      // bNoMore
    uint64_t seq = 0;       // creation sequence: stands in for the heap address in the (size, ptr) sort
    std::list<Node>::iterator self;
};

void split_node(const Node& p, const std::vector<Cand>& c, Node out[4]) {
    const int halfX = (int)std::ceil((float)(p.x1 - p.x0) / 2);
    const int halfY = (int)std::ceil((float)(p.y1 - p.y0) / 2);
    const int mx = p.x0 + halfX, my = p.y0 + halfY;
    out[0].x0 = p.x0; out[0].y0 = p.y0; out[0].x1 = mx;   out[0].y1 = my;
    out[1].x0 = mx;   out[1].y0 = p.y0; out[1].x1 = p.x1; out[1].y1 = my;
    out[2].x0 = p.x0; out[2].y0 = my;   out[2].x1 = mx;   out[2].y1 = p.y1;
    out[3].x0 = mx;   out[3].y0 = my;   out[3].x1 = p.x1; out[3].y1 = p.y1;
    for (int k : p.keys) {
        const Cand& kp = c[k];
        int q = (kp.x < (float)mx) ? ((kp.y < (float)my) ? 0 : 2) : ((kp.y < (float)my) ? 1 : 3);
        out[q].keys.push_back(k);
    }
    for (int q = 0; q < 4; q++) out[q].leaf = (out[q].keys.size() == 1);
}

std::vector<int> distribute_octree(const std::vector<Cand>& c, int minX, int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni <= 0) return {};  // reference divides by zero here (portrait images with h > 2w); unsupported
    const float hX = (float)(maxX - minX) / (float)nIni;
    std::list<Node> nodes;
    uint64_t seq = 0;
    std::vector<Node*> ini(nIni > 0 ? nIni : 0);
    for (int i = 0; i < nIni; i++) {
        Node n;
        n.x0 = (int)(hX * (float)i); n.x1 = (int)(hX * (float)(i + 1));
        n.y0 = 0; n.y1 = maxY - minY;
        n.seq = seq++;
        nodes.push_back(n);
        ini[i] = &nodes.back();
    }
    for (int i = 0; i < (int)c.size(); i++) ini[(size_t)(c[i].x / hX)]->keys.push_back(i);
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->leaf = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    typedef std::pair<int, Node*> SizeNode;
    auto by_size_then_seq = [](const SizeNode& a, const SizeNode& b) {
        return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq;
    };
    std::vector<SizeNode> expandable;
    // children are pushed to the FRONT in quadrant order 0..3; multi-key children are remembered
    auto push_children = [&](Node kids[4], int* nToExpand) {
        for (int q = 0; q < 4; q++) {
            if (kids[q].keys.empty()) continue;
            kids[q].seq = seq++;
            nodes.push_front(kids[q]);
            if (kids[q].keys.size() > 1) {
                if (nToExpand) ++*nToExpand;
                expandable.push_back(SizeNode((int)kids[q].keys.size(), &nodes.front()));
                nodes.front().self = nodes.begin();
            }
        }
    };
    bool finish = false;
    while (!finish) {
        int prevSize = (int)nodes.size();
        int nToExpand = 0;
        expandable.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->leaf) { ++it; continue; }
            Node kids[4];
            split_node(*it, c, kids);
            push_children(kids, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SizeNode> prev = expandable;
                expandable.clear();
                std::sort(prev.begin(), prev.end(), by_size_then_seq);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node kids[4];
                    split_node(*prev[j].second, c, kids);
                    push_children(kids, nullptr);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    std::vector<int> result;
    result.reserve(nodes.size());
    for (const Node& n : nodes) {
        int best = n.keys[0];
        float br = c[best].response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (c[n.keys[k]].response > br) { best = n.keys[k]; br = c[best].response; }
        result.push_back(best);
    }
    return result;
}

// ---------------------------------------------------------------- fastAtan2 (OpenCV 3.x scalar path)
float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
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

// ---------------------------------------------------------------- Gaussian 7x7 sigma 2, u8 fixed point
void gaussian_taps(int taps[7]) {
    // getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256)
    float cf[7];
    double sum = 0;
    const double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3;
        float t = (float)std::exp(scale2X * x * x);
        cf[i] = t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = cv_round((double)cf[i] * 256.0);
    }
}

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void gaussian_blur7(const Image& src, Image& dst) {
    int k[7];
    gaussian_taps(k);
    const int w = src.w, h = src.h;
    dst.w = w; dst.h = h; dst.px.assign((size_t)w * h, 0);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * S[reflect101(x + t - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * tmp[(size_t)reflect101(y + t - 3, h) * w + x];
            int v = (s + 32768) >> 16;
            dst.px[(size_t)y * w + x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
}

// ---------------------------------------------------------------- extractor state
struct Extractor {
    int nfeatures, nlevels, iniTh, minTh;
    float scaleFactorF;
    double scaleFactor;  // the reference member is a double initialised from the float ctor argument
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[kHalfPatch + 1];
    // per-run products
    std::vector<Image> pyr, blurred;
    std::vector<std::vector<Cand>> cands;     // per level, before distribution (minBorder-relative)
    std::vector<std::vector<KeyPoint>> kps;   // per level, level coordinates, with angle
    std::vector<KeyPoint> out_kps;
    std::vector<uint8_t> out_desc;

    Extractor(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactorF(sf), scaleFactor(sf) {
        scale.resize(nl); sigma2.resize(nl); invScale.resize(nl); invSigma2.resize(nl); quota.resize(nl);
        scale[0] = 1.0f; sigma2[0] = 1.0f;
        for (int i = 1; i < nl; i++) {
            scale[i] = (float)(scale[i - 1] * scaleFactor);
            sigma2[i] = scale[i] * scale[i];
        }
        for (int i = 0; i < nl; i++) {
            invScale[i] = 1.0f / scale[i];
            invSigma2[i] = 1.0f / sigma2[i];
        }
        float factor = (float)(1.0f / scaleFactor);
        float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) {
            quota[l] = cv_round(desired);
            sum += quota[l];
            desired *= factor;
        }
        quota[nl - 1] = std::max(nfeatures - sum, 0);
        // circular patch row ends
        int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
        for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    void compute_pyramid(const uint8_t* img, int w, int h, int stride) {
        pyr.assign(nlevels, Image());
        for (int l = 0; l < nlevels; l++) {
            float s = invScale[l];
            int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
            if (l == 0) {
                pyr[0].w = w; pyr[0].h = h; pyr[0].px.resize((size_t)w * h);
                for (int y = 0; y < h; y++) std::memcpy(pyr[0].row(y), img + (size_t)y * stride, w);
            } else {
                resize_linear_u8(pyr[l - 1], pyr[l], lw, lh);
            }
        }
    }

    float ic_angle(const Image& im, float px, float py) const {
        int m01 = 0, m10 = 0;
        const int cx = cv_round(px), cy = cv_round(py);
        const uint8_t* center = im.row(cy) + cx;
        for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
        const int step = im.w;
        for (int v = 1; v <= kHalfPatch; ++v) {
            int vsum = 0, d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int vp = center[u + v * step], vm = center[u - v * step];
                vsum += (vp - vm);
                m10 += u * (vp + vm);
            }
            m01 += v * vsum;
        }
        return fast_atan2_deg((float)m01, (float)m10);
    }

    void describe(const Image& im, const KeyPoint& kp, uint8_t* desc) const {
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float angle = kp.angle * factorPI;
        float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
        const int cx = cv_round(kp.x), cy = cv_round(kp.y);
        const uint8_t* center = im.row(cy) + cx;
        const int step = im.w;
        auto sample = [&](int idx) -> int {
            float fx = (float)kPattern[2 * idx], fy = (float)kPattern[2 * idx + 1];
            float ry = fx * b + fy * a;   // two rounded products, one rounded sum (no FMA)
            float rx = fx * a - fy * b;
            return center[cv_round(ry) * step + cv_round(rx)];
        };
        for (int i = 0; i < 32; i++) {
            int val = 0;
            for (int j = 0; j < 8; j++) {
                int t0 = sample(16 * i + 2 * j), t1 = sample(16 * i + 2 * j + 1);
                val |= (t0 < t1) << j;
            }
            desc[i] = (uint8_t)val;
        }
    }

    int run(const uint8_t* img, int w, int h, int stride) {
        out_kps.clear(); out_desc.clear();
        cands.assign(nlevels, {}); kps.assign(nlevels, {}); blurred.assign(nlevels, Image());
        if (!img || w <= 0 || h <= 0) return 0;
        compute_pyramid(img, w, h, stride);
        const float W = refc::FAST_CELL;
        for (int l = 0; l < nlevels; l++) {
            const Image& im = pyr[l];
            const int minBX = kEdge - 3, minBY = minBX;
            const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
            const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
            const int nCols = (int)(width / W), nRows = (int)(height / W);
            if (nCols <= 0 || nRows <= 0) return -1;
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            std::vector<Cand>& vc = cands[l];
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    std::vector<Cand> cell;
                    fast_cell(im, (int)iniX, (int)iniY, (int)maxX - (int)iniX, (int)maxY - (int)iniY, iniTh, cell);
                    if (cell.empty())
                        fast_cell(im, (int)iniX, (int)iniY, (int)maxX - (int)iniX, (int)maxY - (int)iniY, minTh, cell);
                    for (Cand& k : cell) {
                        k.x += j * wCell;
                        k.y += i * hCell;
                        vc.push_back(k);
                    }
                }
            }
            std::vector<int> sel = distribute_octree(vc, minBX, maxBX, minBY, maxBY, quota[l]);
            const int scaledPatch = (int)(kPatch * scale[l]);
            for (int idx : sel) {
                KeyPoint kp;
                kp.x = vc[idx].x + minBX; kp.y = vc[idx].y + minBY;
                kp.size = (float)scaledPatch; kp.angle = -1; kp.response = vc[idx].response;
                kp.octave = l; kp.class_id = -1;
                kps[l].push_back(kp);
            }
        }
        for (int l = 0; l < nlevels; l++)
            for (KeyPoint& kp : kps[l]) kp.angle = ic_angle(pyr[l], kp.x, kp.y);
        int total = 0;
        for (int l = 0; l < nlevels; l++) total += (int)kps[l].size();
        out_desc.assign((size_t)total * 32, 0);
        int off = 0;
        for (int l = 0; l < nlevels; l++) {
            if (kps[l].empty()) continue;
            gaussian_blur7(pyr[l], blurred[l]);
            for (size_t i = 0; i < kps[l].size(); i++) describe(blurred[l], kps[l][i], &out_desc[(size_t)(off + i) * 32]);
            off += (int)kps[l].size();
            for (KeyPoint kp : kps[l]) {
                if (l != 0) { kp.x *= scale[l]; kp.y *= scale[l]; }
                out_kps.push_back(kp);
            }
        }
        return total;
    }
};

}  // namespace

// -------------------------------------------------------------------------------- C surface (ctypes)
extern "C" {

void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orc_orb_destroy(void* h) { delete (Extractor*)h; }

// returns number of keypoints (or <0 on unsupported geometry)
int orc_orb_run(void* h, const uint8_t* img, int w, int hgt, int stride) {
    return ((Extractor*)h)->run(img, w, hgt, stride);
}
int orc_orb_result(void* h, void* kps28, uint8_t* desc, int cap) {
    Extractor* e = (Extractor*)h;
    int n = std::min((int)e->out_kps.size(), cap);
    if (kps28) std::memcpy(kps28, e->out_kps.data(), (size_t)n * sizeof(KeyPoint));
    if (desc) std::memcpy(desc, e->out_desc.data(), (size_t)n * 32);
    return (int)e->out_kps.size();
}
void orc_orb_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* quota, int* umax16) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->scale[i];
        if (invScale) invScale[i] = e->invScale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (invSigma2) invSigma2[i] = e->invSigma2[i];
        if (quota) quota[i] = e->quota[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}
int orc_orb_level_dims(void* h, int level, int* w, int* hgt) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= (int)e->pyr.size()) return -1;
    *w = e->pyr[level].w; *hgt = e->pyr[level].h;
    return 0;
}
// which: 0 = pyramid level, 1 = blurred level (empty if the level had no keypoints)
int orc_orb_level_image(void* h, int level, int which, uint8_t* out) {
    Extractor* e = (Extractor*)h;
    const Image& im = which ? e->blurred[level] : e->pyr[level];
    if (im.px.empty()) return 0;
    std::memcpy(out, im.px.data(), im.px.size());
    return (int)im.px.size();
}
// FAST candidates of a level before distribution: (x, y, response) float triples, minBorder-relative
int orc_orb_level_candidates(void* h, int level, float* xyr, int cap) {
    Extractor* e = (Extractor*)h;
    int n = (int)e->cands[level].size();
    for (int i = 0; i < std::min(n, cap); i++) {
        xyr[3 * i] = e->cands[level][i].x; xyr[3 * i + 1] = e->cands[level][i].y; xyr[3 * i + 2] = e->cands[level][i].response;
    }
    return n;
}
int orc_orb_level_keypoints(void* h, int level, void* kps28, int cap) {
    Extractor* e = (Extractor*)h;
    int n = (int)e->kps[level].size();
    const int m = std::min(n, cap);
    if (m > 0) std::memcpy(kps28, e->kps[level].data(), (size_t)m * sizeof(KeyPoint));   // (a count query passes a null destination: UBSan, r03)
    return n;
}

// stand-alone primitives for unit tests
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Image s, d; s.w = sw; s.h = sh; s.px.assign(src, src + (size_t)sw * sh);
    resize_linear_u8(s, d, dw, dh);
    std::memcpy(dst, d.px.data(), d.px.size());
}
void orc_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
    Image s, d; s.w = w; s.h = h; s.px.assign(src, src + (size_t)w * h);
    gaussian_blur7(s, d);
    std::memcpy(dst, d.px.data(), d.px.size());
}
void orc_gaussian_taps(int* taps7) { gaussian_taps(taps7); }
float orc_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
// FAST on a whole (sub-)image; returns count, writes (x,y,score) float triples
int orc_fast(const uint8_t* img, int w, int h, int th, float* xyr, int cap) {
    Image s; s.w = w; s.h = h; s.px.assign(img, img + (size_t)w * h);
    std::vector<Cand> out;
    fast_cell(s, 0, 0, w, h, th, out);
    for (int i = 0; i < std::min((int)out.size(), cap); i++) {
        xyr[3 * i] = out[i].x; xyr[3 * i + 1] = out[i].y; xyr[3 * i + 2] = out[i].response;
    }
    return (int)out.size();
}
// quad-tree distribution alone: xyr = candidates (minBorder-relative); returns selected indices in output order
int orc_distribute(const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, int* sel, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; i++) c[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    std::vector<int> r = distribute_octree(c, minX, maxX, minY, maxY, N);
    for (int i = 0; i < std::min((int)r.size(), cap); i++) sel[i] = r[i];
    return (int)r.size();
}
const int8_t* orc_orb_pattern() { return kPattern; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------- stereo matching
// Frame::ComputeStereoMatches (src/Frame.cc:841-1013) over the pyramids of two extractors that just ran on the stereo
// pair.  mvImagePyramid[l] is a view into a buffer with a 19-px BORDER_REFLECT_101 frame (src/ORBextractor.cc:1113-1128):
// patch reads that leave the level are answered with the mirrored pixel.  The patches minus their centre pixel and their
// L1 distance are integers, hence exact in float.
namespace {
int refl_px(const Image& im, int x, int y) {
    x = x < 0 ? -x : (x >= im.w ? 2 * im.w - 2 - x : x);
    y = y < 0 ? -y : (y >= im.h ? 2 * im.h - 2 - y : y);
    return im.row(y)[x];
}
}  // namespace

extern "C" int orc_stereo_matches(void* hl, void* hr, int nl, const void* kpsL28, const uint8_t* descL, int nr, const void* kpsR28,
                                  const uint8_t* descR, float mb, float mbf, float* uRight, float* depth) {
    const Extractor* L = (const Extractor*)hl;
    const Extractor* R = (const Extractor*)hr;
    struct KP { float x, y, size, angle, response; int32_t octave, class_id; };
    const KP* kl = (const KP*)kpsL28;
    const KP* kr = (const KP*)kpsR28;
    for (int i = 0; i < nl; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int nRows = L->pyr[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * L->scale[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);
    }
    const float minZ = mb, minD = -3, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < nl; iL++) {
        const KP& kpL = kl[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        if ((int)vL < 0 || (int)vL >= nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = refc::TH_HIGH;   // ORBmatcher::TH_HIGH
        size_t bestIdxR = 0;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                uint32_t a[8], b[8];
                std::memcpy(a, descL + 32 * (size_t)iL, 32); std::memcpy(b, descR + 32 * iR, 32);
                int dist = 0;
                for (int w = 0; w < 8; w++) dist += __builtin_popcount(a[w] ^ b[w]);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < refc::TH_HIGH) {
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = L->invScale[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Image& imL = L->pyr[kpL.octave];
            const Image& imR = R->pyr[kpL.octave];
            float IL[11][11];
            const float cL = (float)refl_px(imL, (int)scaleduL, (int)scaledvL);
            for (int py = 0; py < 11; py++)
                for (int px = 0; px < 11; px++) IL[py][px] = (float)refl_px(imL, (int)(scaleduL - w) + px, (int)(scaledvL - w) + py) - cL;
            int bestDistSad = INT_MAX, bestincR = 0;
            const int Lw = 5;
            std::vector<float> vDists(2 * Lw + 1);
            const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            for (int incR = -Lw; incR <= +Lw; incR++) {
                const float cR = (float)refl_px(imR, (int)(scaleduR0 + incR), (int)scaledvL);
                float dist = 0;
                for (int py = 0; py < 11; py++)
                    for (int px = 0; px < 11; px++)
                        dist += std::fabs(IL[py][px] - ((float)refl_px(imR, (int)(scaleduR0 + incR - w) + px, (int)(scaledvL - w) + py) - cR));
                if (dist < bestDistSad) { bestDistSad = (int)dist; bestincR = incR; }
                vDists[Lw + incR] = dist;
            }
            if (bestincR == -Lw || bestincR == Lw) continue;
            const float dist1 = vDists[Lw + bestincR - 1], dist2 = vDists[Lw + bestincR], dist3 = vDists[Lw + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = L->scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= 0 && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistSad, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
    }
    return (int)vDistIdx.size();
}
