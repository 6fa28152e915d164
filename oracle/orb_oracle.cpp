// CPU oracle for the ORB extractor hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// See orb_oracle.h for the scope, the pinned decisions and the "parity unpinned" note.
//
// Every function cites the reference lines (src/ORBextractor.cc unless another file is
// named) it restates.  OpenCV primitives are restated from the published OpenCV 3.4.5
// algorithms (the version the reference's author linked: cmake-build-debug/CMakeFiles/
// myvideo.dir/link.txt:1); OpenCV itself is an un-vendored dependency of the reference.
//
// Build with -ffp-contract=off: every fused multiply-add below is an explicit fma()/fmaf().

#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <map>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

const int PATCH_SIZE = 31;        // :73
const int HALF_PATCH_SIZE = 15;   // :74
const int EDGE_THRESHOLD = 19;    // :75

const int8_t kPattern[1024] = {
#include "../include/orb_pattern_31.inc"
};

// cvRound on x86-64 is cvtss2si / cvtsd2si under the default rounding mode: half to even.
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_roundf(float v) { return (int)std::nearbyintf(v); }

inline int reflect101(int p, int len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Constructor arithmetic: src/ORBextractor.cc:416-494.  scaleFactor is a double member
// initialised from a float argument (include/ORBextractor.h:98).
extern "C" int orc_params_init(orc_params* p, int nfeatures, float scale_factor, int nlevels,
                               int ini_th, int min_th) {
    if (!p || nlevels < 1 || nlevels > ORC_MAX_LEVELS) return -1;
    std::memset(p, 0, sizeof(*p));
    p->nfeatures = nfeatures;
    p->scale_factor = scale_factor;
    p->nlevels = nlevels;
    p->ini_th_fast = ini_th;
    p->min_th_fast = min_th;
    const double scaleFactor = (double)scale_factor;
    p->scale[0] = 1.0f;
    p->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        p->scale[i] = (float)((double)p->scale[i - 1] * scaleFactor);  // :432
        p->sigma2[i] = p->scale[i] * p->scale[i];                      // :433
    }
    for (int i = 0; i < nlevels; i++) {
        p->inv_scale[i] = 1.0f / p->scale[i];    // :441
        p->inv_sigma2[i] = 1.0f / p->sigma2[i];  // :442
    }
    // :448-463
    float factor = (float)(1.0 / scaleFactor);
    float nDesired = (float)nfeatures * (1.0f - factor) /
                     (1.0f - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        p->quota[level] = cv_roundf(nDesired);
        sum += p->quota[level];
        nDesired *= factor;
    }
    p->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    // :473-493
    int v, v0;
    int vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) p->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (p->umax[v0] == p->umax[v0 + 1]) ++v0;
        p->umax[v] = v0;
        ++v0;
    }
    return 0;
}

// Level size: :1220-1221 (float multiply, then cvRound).
extern "C" void orc_level_size(const orc_params* p, int w0, int h0, int level, int* w, int* h) {
    float scale = p->inv_scale[level];
    *w = cv_roundf((float)w0 * scale);
    *h = cv_roundf((float)h0 * scale);
}

// ---------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1, OpenCV 3.4.5 fixed-point path
// (call site :1237).  11-bit coefficients; horizontal pass into int32, vertical pass
// ((b0*(T0>>4))>>16) + ((b1*(T1>>4))>>16) + 2) >> 2.
extern "C" void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                                  uint8_t* dst, int dw, int dh, int dstride) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * (size_t)dw), beta(2 * (size_t)dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = (short)cv_roundf((1.f - fx) * 2048.f);
        alpha[2 * dx + 1] = (short)cv_roundf(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = (short)cv_roundf((1.f - fy) * 2048.f);
        beta[2 * dy + 1] = (short)cv_roundf(fy * 2048.f);
    }
    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](int sy, std::vector<int>& out) {
        const uint8_t* S = src + (size_t)sy * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (sx >= sw - 1)
                out[dx] = S[sx] * 2048;
            else
                out[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
        }
    };
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        hresize(sy0, row0);
        hresize(sy1, row1);
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ---------------------------------------------------------------------------------------
// cv::GaussianBlur(m, m, Size(7,7), 2, 2, BORDER_REFLECT_101) on a continuous 8UC1 image
// (call site :1188; the input is a clone, :1182, so the border reflects the level's own
// pixels).  OpenCV 3.4.5 8-bit fixed-point path: taps rint(256*g) = 18,34,49,55,49,34,18.
extern "C" void orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst,
                                   int dstride) {
    static const int K[7] = {18, 34, 49, 55, 49, 34, 18};
    std::vector<uint16_t> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        uint16_t* T = &tmp[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            unsigned s = 0;
            if (x >= 3 && x < w - 3) {
                for (int i = 0; i < 7; i++) s += K[i] * S[x + i - 3];
            } else {
                for (int i = 0; i < 7; i++) s += K[i] * S[reflect101(x + i - 3, w)];
            }
            T[x] = (uint16_t)s;  // <= 255*257 = 65535, exact
        }
    }
    for (int y = 0; y < h; y++) {
        const uint16_t* R[7];
        for (int j = 0; j < 7; j++) R[j] = &tmp[(size_t)reflect101(y + j - 3, h) * w];
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int j = 0; j < 7; j++) s += (uint32_t)K[j] * R[j][x];
            s = (s + 32768u) >> 16;
            D[x] = (uint8_t)std::min(s, 255u);
        }
    }
}

// ---------------------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, true): FAST-9/16 with 3x3 non-max suppression, OpenCV
// 3.4.5 FAST_t<16> and cornerScore<16> (call sites :900, :905).
namespace {

const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16>: d[k] = v - ring[k], k = 0..24 (cyclic).
int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int N = 25;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

struct FastScratch {
    std::vector<uint8_t> score;
};

int fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, orc_cand* out, int cap,
             FastScratch& scratch) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kRingDx[k] + kRingDy[k] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return 0;
    scratch.score.assign((size_t)w * h, 0);
    uint8_t* score = scratch.score.data();
    for (int i = 3; i < h - 3; i++) {
        const uint8_t* row = img + (size_t)i * stride;
        for (int j = 3; j < w - 3; j++) {
            const uint8_t* ptr = row + j;
            int v = ptr[0];
            int lo = v - threshold, hi = v + threshold;
            // quick rejection on the opposite pairs (0,8) (2,10) (4,12) (6,14) then the odd ones
            auto cls = [&](int k) { int x = ptr[pixel[k]]; return (x < lo ? 1 : 0) | (x > hi ? 2 : 0); };
            int d = cls(0) | cls(8);
            if (d == 0) continue;
            d &= cls(2) | cls(10);
            d &= cls(4) | cls(12);
            d &= cls(6) | cls(14);
            if (d == 0) continue;
            d &= cls(1) | cls(9);
            d &= cls(3) | cls(11);
            d &= cls(5) | cls(13);
            d &= cls(7) | cls(15);
            bool corner = false;
            if (d & 1) {
                int count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] < lo) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (!corner && (d & 2)) {
                int count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] > hi) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (corner) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    }
    // 3x3 strict non-max suppression; rows/cols outside [3, dim-3) hold score 0.
    // A corner always has score >= threshold >= 0; score 0 corners can only arise at
    // threshold 0 and are never strictly greater than their (zero) neighbours... except
    // that OpenCV still tests them; with all-zero neighbours 0 > 0 is false, so they drop.
    int n = 0;
    for (int i = 3; i < h - 3; i++) {
        for (int j = 3; j < w - 3; j++) {
            const uint8_t* s = score + (size_t)i * w + j;
            int sc = s[0];
            if (sc == 0) continue;
            if (sc > s[-1] && sc > s[1] && sc > s[-w - 1] && sc > s[-w] && sc > s[-w + 1] &&
                sc > s[w - 1] && sc > s[w] && sc > s[w + 1]) {
                if (n < cap) {
                    out[n].x = j;
                    out[n].y = i;
                    out[n].response = sc;
                }
                n++;
            }
        }
    }
    return n;
}

}  // namespace

extern "C" int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold,
                        orc_cand* out, int cap) {
    FastScratch scratch;
    return fast9_16(img, w, h, stride, threshold, out, cap, scratch);
}

// ComputeKeyPointsOctTree, cell-grid part: :862-920.  Output coordinates are relative to
// (minBorderX, minBorderY) = (16,16), in (cell row, cell col, y, x) order.
extern "C" int orc_level_candidates(const uint8_t* img, int w, int h, int stride, int ini_th,
                                    int min_th, orc_cand* out, int cap) {
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return 0;  // the reference divides by zero here
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    FastScratch scratch;
    std::vector<orc_cand> cell((size_t)(wCell + 6) * (hCell + 6));
    int n = 0;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
            const uint8_t* sub = img + (size_t)y0 * stride + x0;
            int m = fast9_16(sub, cw, ch, stride, ini_th, cell.data(), (int)cell.size(), scratch);
            if (m == 0) m = fast9_16(sub, cw, ch, stride, min_th, cell.data(), (int)cell.size(), scratch);
            for (int k = 0; k < m; k++) {
                if (n < cap) {
                    out[n].x = cell[k].x + j * wCell;
                    out[n].y = cell[k].y + i * hCell;
                    out[n].response = cell[k].response;
                }
                n++;
            }
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------
// DistributeOctTree (:572-852) and ExtractorNode::DivideNode (:507-570).  Points are kept as
// indices into the candidate array.  Sort ties: decision D2 (creation sequence).
namespace {

struct Node {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::vector<int> keys;
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0;
};

void divide_node(const Node& p, const orc_cand* c, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil((float)(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil((float)(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (size_t i = 0; i < p.keys.size(); i++) {
        const orc_cand& kp = c[p.keys[i]];
        if ((float)kp.x < (float)n1.URx) {
            if ((float)kp.y < (float)n1.BRy)
                n1.keys.push_back(p.keys[i]);
            else
                n3.keys.push_back(p.keys[i]);
        } else if ((float)kp.y < (float)n1.BRy)
            n2.keys.push_back(p.keys[i]);
        else
            n4.keys.push_back(p.keys[i]);
    }
    if (n1.keys.size() == 1) n1.bNoMore = true;
    if (n2.keys.size() == 1) n2.bNoMore = true;
    if (n3.keys.size() == 1) n3.bNoMore = true;
    if (n4.keys.size() == 1) n4.bNoMore = true;
}

}  // namespace

// Tie order among equal-size nodes in the sort at :770.  0 = decision D2 (the pinned order, the
// only one the product is compared with); 1 = the reverse; 2 = a fixed pseudo-random order.  The
// alternatives exist to measure how much of the output hangs on the tie-break
// (tools/tolerance_study.py): the reference orders such nodes by their heap addresses.
static int g_d2_mode = 0;
extern "C" void orc_set_d2_mode(int mode) { g_d2_mode = mode; }

// d2_sensitive (may be null) receives 1 when the one-at-a-time phase stopped (:816) with the next
// node in line as large as the last one split: the pinned tie-break D2 then chose which of several
// equal-size nodes was split, and a build of the reference with another heap layout may retain a
// different set; 0 when the retained SET cannot depend on the tie-break.
extern "C" int orc_distribute_octree_ex(const orc_cand* cands, int n, int minX, int maxX, int minY,
                                        int maxY, int N, int32_t* out_idx, int cap, int* d2_sensitive) {
    if (d2_sensitive) *d2_sensitive = 0;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));  // :576
    if (nIni < 1) return -1;  // undefined behaviour in the reference (tall images)
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * (float)i); ni.ULy = 0;
        ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)cands[i].x / hX);  // :620
        if (r < 0 || r >= nIni) return -2;      // out-of-range write in the reference
        vpIniNodes[r]->keys.push_back(i);
    }
    std::list<Node>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->keys.size() == 1) {
            lit->bNoMore = true;
            lit++;
        } else if (lit->keys.empty())
            lit = lNodes.erase(lit);
        else
            lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, Node*> SizeNode;
    // decision D2: the reference compares (size, pointer); pointers are replaced by seq
    const int d2_mode = g_d2_mode;
    auto less_d2 = [d2_mode](const SizeNode& a, const SizeNode& b) {
        if (a.first != b.first) return a.first < b.first;
        if (d2_mode == 1) return a.second->seq > b.second->seq;
        if (d2_mode == 2) {
            const unsigned long ha = (unsigned long)a.second->seq * 2654435761ul % 1000003ul;
            const unsigned long hb = (unsigned long)b.second->seq * 2654435761ul % 1000003ul;
            if (ha != hb) return ha < hb;
        }
        return a.second->seq < b.second->seq;
    };
    std::vector<SizeNode> vSizeAndPointerToNode;
    auto push_child = [&](Node& c, bool record, int* nToExpand) {
        if (c.keys.size() > 0) {
            c.seq = seq++;
            lNodes.push_front(c);
            if (c.keys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                if (record) vSizeAndPointerToNode.push_back(std::make_pair((int)c.keys.size(), &lNodes.front()));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };
    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) {
                lit++;
                continue;
            }
            Node n1, n2, n3, n4;
            divide_node(*lit, cands, n1, n2, n3, n4);
            push_child(n1, true, &nToExpand);
            push_child(n2, true, &nToExpand);
            push_child(n3, true, &nToExpand);
            push_child(n4, true, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeNode> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(), less_d2);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*vPrev[j].second, cands, n1, n2, n3, n4);
                    push_child(n1, true, nullptr);
                    push_child(n2, true, nullptr);
                    push_child(n3, true, nullptr);
                    push_child(n4, true, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) {
                        if (d2_sensitive && j > 0 && vPrev[j - 1].first == vPrev[j].first) *d2_sensitive = 1;
                        break;
                    }
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    int m = 0;
    for (lit = lNodes.begin(); lit != lNodes.end(); lit++) {
        const std::vector<int>& keys = lit->keys;
        int best = keys[0];
        float maxResponse = (float)cands[best].response;
        for (size_t k = 1; k < keys.size(); k++) {
            if ((float)cands[keys[k]].response > maxResponse) {
                best = keys[k];
                maxResponse = (float)cands[keys[k]].response;
            }
        }
        if (m < cap) out_idx[m] = best;
        m++;
    }
    return m;
}

extern "C" int orc_distribute_octree(const orc_cand* cands, int n, int minX, int maxX, int minY,
                                     int maxY, int N, int32_t* out_idx, int cap) {
    return orc_distribute_octree_ex(cands, n, minX, maxX, minY, maxY, N, out_idx, cap, nullptr);
}

// ---------------------------------------------------------------------------------------
// cv::fastAtan2 scalar form, OpenCV 3.4.5 (call site :105).  Decision D3.
extern "C" float orc_fast_atan2(float y, float x) {
    static const float rad2deg = (float)(180 / 3.1415926535897932384626433832795);
    static const float p1 = 0.9997878412794807f * rad2deg;
    static const float p3 = -0.3258083974640975f * rad2deg;
    static const float p5 = 0.1555786518463281f * rad2deg;
    static const float p7 = -0.04432655554792128f * rad2deg;
    static const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
    float ax = std::fabs(x), ay = std::fabs(y);
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

// IC_Angle: :78-106.  (x, y) is the already rounded centre in level coordinates.
extern "C" float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

// Decision D6: cos/sin of the float32 radian angle, evaluated in double with a fixed
// operation sequence (so that any IEEE-754 machine reproduces it bit for bit) and rounded
// once to float32.  Range reduction by multiples of pi/2 (two-term constant), then the
// fdlibm __kernel_sin/__kernel_cos minimax polynomials on |r| <= pi/4.
namespace {
inline double ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = std::fma(z, S6, S5);
    p = std::fma(z, p, S4);
    p = std::fma(z, p, S3);
    p = std::fma(z, p, S2);
    p = std::fma(z, p, S1);
    return std::fma(r * z, p, r);
}
inline double kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = std::fma(z, C6, C5);
    p = std::fma(z, p, C4);
    p = std::fma(z, p, C3);
    p = std::fma(z, p, C2);
    p = std::fma(z, p, C1);
    double zz = z * z;
    return std::fma(zz, p, std::fma(z, -0.5, 1.0));
}
void sincos_pinned(float angf, float* c_out, float* s_out) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
    double x = (double)angf;
    double kd = std::floor(std::fma(x, TWO_OVER_PI, 0.5));
    double r = std::fma(-kd, PIO2_HI, x);
    r = std::fma(-kd, PIO2_LO, r);
    double s = ksin(r), c = kcos(r);
    int k = (int)kd & 3;
    double ss, cc;
    switch (k) {
        case 0: ss = s; cc = c; break;
        case 1: ss = c; cc = -s; break;
        case 2: ss = -s; cc = -c; break;
        default: ss = -c; cc = s; break;
    }
    *c_out = (float)cc;
    *s_out = (float)ss;
}
}  // namespace

extern "C" void orc_sincos_deg(float angle_deg, float* cos_out, float* sin_out) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);  // :109
    float angle = angle_deg * factorPI;                                          // :116
    sincos_pinned(angle, cos_out, sin_out);
}

// computeOrbDescriptor: :112-152.  Decision D1 (FMA form) and D6 (a, b).
extern "C" void orc_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                               uint8_t desc[32]) {
    float a, b;
    orc_sincos_deg(angle_deg, &a, &b);
    const uint8_t* center = blurred + (size_t)y * stride + x;
    const int8_t* pat = kPattern;
    auto get = [&](int idx) -> int {
        float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        int row = cv_roundf(std::fmaf(px, b, py * a));
        int col = cv_roundf(std::fmaf(px, a, -(py * b)));
        return center[row * stride + col];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = get(2 * j), t1 = get(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// The descriptor under the alternatives to decisions D1 and D6, for measuring how far a real
// build of the reference can be from the pinned form (tools/tolerance_study.py):
//   rotation 0: the FMA form (D1, the author's -march=native build)
//            1: :123-125 as written, every product and sum rounded (a build without FMA contraction)
//   sincos   0: the pinned a = cos, b = sin (D6)
//            1: this host's libm cosf / sinf (:118)
//            2..5: the pinned values with a + 1 ulp, a - 1 ulp, b + 1 ulp, b - 1 ulp (what a libm
//               within one unit in the last place can return)
extern "C" void orc_descriptor_variant(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                                       int rotation, int sincos, uint8_t desc[32]) {
    float a, b;
    if (sincos == 1) {
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float angle = angle_deg * factorPI;
        a = cosf(angle);
        b = sinf(angle);
    } else {
        orc_sincos_deg(angle_deg, &a, &b);
        if (sincos == 2) a = std::nextafterf(a, 2.f);
        if (sincos == 3) a = std::nextafterf(a, -2.f);
        if (sincos == 4) b = std::nextafterf(b, 2.f);
        if (sincos == 5) b = std::nextafterf(b, -2.f);
    }
    const uint8_t* center = blurred + (size_t)y * stride + x;
    const int8_t* pat = kPattern;
    auto get = [&](int idx) -> int {
        const float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        int row, col;
        if (rotation == 0) {
            row = cv_roundf(std::fmaf(px, b, py * a));
            col = cv_roundf(std::fmaf(px, a, -(py * b)));
        } else {
            volatile float xb = px * b, ya = py * a, xa = px * a, yb = py * b;  // each product rounded to float32
            row = cv_roundf(xb + ya);
            col = cv_roundf(xa - yb);
        }
        return center[row * stride + col];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = get(2 * j), t1 = get(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

extern "C" const int8_t* orc_pattern(void) { return kPattern; }

// ---------------------------------------------------------------------------------------
// Whole path: ORBextractor::operator() (:1137-1211), ComputePyramid (:1213-1250),
// ComputeKeyPointsOctTree (:854-947).
struct orc_ctx {
    orc_params p;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    std::vector<uint8_t> level[ORC_MAX_LEVELS], blurred[ORC_MAX_LEVELS];
    std::vector<orc_cand> cands[ORC_MAX_LEVELS];
    int nkp_level[ORC_MAX_LEVELS];
    int d2_sensitive[ORC_MAX_LEVELS];
    std::vector<orc_kp> kps;
    std::vector<uint8_t> desc;
};

extern "C" orc_ctx* orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    orc_ctx* c = new orc_ctx();
    if (orc_params_init(&c->p, nfeatures, scale_factor, nlevels, ini_th, min_th) != 0) {
        delete c;
        return nullptr;
    }
    std::memset(c->w, 0, sizeof(c->w));
    std::memset(c->h, 0, sizeof(c->h));
    std::memset(c->nkp_level, 0, sizeof(c->nkp_level));
    return c;
}
extern "C" void orc_destroy(orc_ctx* c) { delete c; }
extern "C" const orc_params* orc_get_params(const orc_ctx* c) { return &c->p; }

extern "C" int orc_extract(orc_ctx* c, const uint8_t* img, int w, int h, int stride) {
    const orc_params& p = c->p;
    c->kps.clear();
    c->desc.clear();
    if (!img || w <= 0 || h <= 0) return 0;  // :1140 (empty image: outputs untouched)
    // ComputePyramid
    for (int l = 0; l < p.nlevels; l++) {
        orc_level_size(&p, w, h, l, &c->w[l], &c->h[l]);
        if (c->w[l] < 1 || c->h[l] < 1) return -1;
        c->level[l].resize((size_t)c->w[l] * c->h[l]);
        if (l == 0) {
            for (int y = 0; y < h; y++) std::memcpy(&c->level[0][(size_t)y * w], img + (size_t)y * stride, w);
        } else {
            orc_resize_linear(c->level[l - 1].data(), c->w[l - 1], c->h[l - 1], c->w[l - 1],
                              c->level[l].data(), c->w[l], c->h[l], c->w[l]);
        }
    }
    // ComputeKeyPointsOctTree
    std::vector<std::vector<orc_kp>> all(p.nlevels);
    for (int l = 0; l < p.nlevels; l++) {
        const int lw = c->w[l], lh = c->h[l];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = lw - EDGE_THRESHOLD + 3, maxBorderY = lh - EDGE_THRESHOLD + 3;
        std::vector<orc_cand>& cd = c->cands[l];
        cd.resize((size_t)((lw + 1) / 2) * ((lh + 1) / 2) + 16);
        int n = orc_level_candidates(c->level[l].data(), lw, lh, lw, p.ini_th_fast, p.min_th_fast,
                                     cd.data(), (int)cd.size());
        if (n > (int)cd.size()) return -2;
        cd.resize(n);
        std::vector<int32_t> keep((size_t)n + 4);
        int m = 0;
        c->d2_sensitive[l] = 0;
        if (n > 0) {
            m = orc_distribute_octree_ex(cd.data(), n, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                         p.quota[l], keep.data(), (int)keep.size(), &c->d2_sensitive[l]);
            if (m < 0) return -3;
        }
        const int scaledPatchSize = (int)(PATCH_SIZE * p.scale[l]);  // :930
        all[l].resize(m);
        for (int i = 0; i < m; i++) {
            const orc_cand& k = cd[keep[i]];
            orc_kp& o = all[l][i];
            o.x = (float)k.x + minBorderX;
            o.y = (float)k.y + minBorderY;
            o.size = (float)scaledPatchSize;
            o.response = (float)k.response;
            o.octave = l;
            o.class_id = -1;
            o.angle = -1.f;
        }
        c->nkp_level[l] = m;
    }
    for (int l = 0; l < p.nlevels; l++)
        for (orc_kp& k : all[l])
            k.angle = orc_ic_angle(c->level[l].data(), c->w[l], cv_roundf(k.x), cv_roundf(k.y), p.umax);
    // descriptors, :1172-1210
    int nkeypoints = 0;
    for (int l = 0; l < p.nlevels; l++) nkeypoints += (int)all[l].size();
    c->desc.resize((size_t)nkeypoints * 32);
    c->kps.reserve(nkeypoints);
    int offset = 0;
    for (int l = 0; l < p.nlevels; l++) {
        // the reference blurs only levels that hold keypoints; the blurred image of an empty
        // level is never observable, so it is produced here unconditionally for the tests
        c->blurred[l].resize((size_t)c->w[l] * c->h[l]);
        orc_gaussian_blur7(c->level[l].data(), c->w[l], c->h[l], c->w[l], c->blurred[l].data(), c->w[l]);
        std::vector<orc_kp>& kl = all[l];
        if (kl.empty()) continue;
        for (size_t i = 0; i < kl.size(); i++)
            orc_descriptor(c->blurred[l].data(), c->w[l], cv_roundf(kl[i].x), cv_roundf(kl[i].y), kl[i].angle,
                           &c->desc[(size_t)(offset + i) * 32]);
        offset += (int)kl.size();
        if (l != 0) {
            float scale = p.scale[l];
            for (orc_kp& k : kl) {
                k.x *= scale;
                k.y *= scale;
            }
        }
        c->kps.insert(c->kps.end(), kl.begin(), kl.end());
    }
    return nkeypoints;
}

extern "C" int orc_num_keypoints(const orc_ctx* c) { return (int)c->kps.size(); }
extern "C" void orc_get_keypoints(const orc_ctx* c, orc_kp* out) {
    if (!c->kps.empty()) std::memcpy(out, c->kps.data(), c->kps.size() * sizeof(orc_kp));
}
extern "C" void orc_get_descriptors(const orc_ctx* c, uint8_t* out) {
    if (!c->desc.empty()) std::memcpy(out, c->desc.data(), c->desc.size());
}
extern "C" void orc_get_level_dims(const orc_ctx* c, int level, int* w, int* h) {
    *w = c->w[level];
    *h = c->h[level];
}
extern "C" void orc_get_level(const orc_ctx* c, int level, uint8_t* out) {
    std::memcpy(out, c->level[level].data(), c->level[level].size());
}
extern "C" void orc_get_blurred(const orc_ctx* c, int level, uint8_t* out) {
    std::memcpy(out, c->blurred[level].data(), c->blurred[level].size());
}
extern "C" int orc_get_num_candidates(const orc_ctx* c, int level) { return (int)c->cands[level].size(); }
extern "C" void orc_get_candidates(const orc_ctx* c, int level, orc_cand* out) {
    if (!c->cands[level].empty())
        std::memcpy(out, c->cands[level].data(), c->cands[level].size() * sizeof(orc_cand));
}
extern "C" int orc_get_num_level_keypoints(const orc_ctx* c, int level) { return c->nkp_level[level]; }
extern "C" int orc_get_d2_sensitive(const orc_ctx* c, int level) { return c->d2_sensitive[level]; }

// copyMakeBorder(..., 19,19,19,19, BORDER_REFLECT_101): :1239-1245
extern "C" void orc_get_level_padded(const orc_ctx* c, int level, uint8_t* out) {
    const int w = c->w[level], h = c->h[level], E = EDGE_THRESHOLD, pw = w + 2 * E;
    const uint8_t* src = c->level[level].data();
    for (int y = -E; y < h + E; y++) {
        const uint8_t* S = src + (size_t)reflect101(y, h) * w;
        uint8_t* D = out + (size_t)(y + E) * pw;
        for (int x = -E; x < w + E; x++) D[x + E] = S[reflect101(x, w)];
    }
}

// ---------------------------------------------------------------------------------------------
// Stereo association: Frame::ComputeStereoMatches, reference src/Frame.cc:485-659.
// ---------------------------------------------------------------------------------------------

// src/ORBmatcher.cc:1684-1700 (the bit trick there is a population count per 32-bit word)
extern "C" int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32]) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

// Pixel of mvImagePyramid[level] at ROI coordinates that may lie in the 19-px border
// (the border holds the reflect-101 continuation, src/ORBextractor.cc:1239-1245).
static inline int pyr_px(const orc_ctx* c, int level, int x, int y) {
    return c->level[level][(size_t)reflect101(y, c->h[level]) * c->w[level] + reflect101(x, c->w[level])];
}

extern "C" int orc_stereo_match(const orc_ctx* left, const orc_ctx* right, float mb, float mbf,
                                float* u_right, float* depth, int32_t* sad) {
    const int TH_HIGH = 100, TH_LOW = 50;  // src/ORBmatcher.cc:37-38
    const int N = (int)left->kps.size();
    for (int i = 0; i < N; i++) {  // :487-488
        u_right[i] = -1.0f;
        depth[i] = -1.0f;
        if (sad) sad[i] = -1;
    }
    if (N == 0 || left->h[0] <= 0) return 0;
    const float* scale = left->p.scale;
    const float* inv_scale = left->p.inv_scale;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;  // :490
    const int nRows = left->h[0];                  // :492

    // :494-512 right keypoints by image row
    std::vector<std::vector<size_t> > vRowIndices(nRows);
    const int Nr = (int)right->kps.size();
    for (int iR = 0; iR < Nr; iR++) {
        const orc_kp& kp = right->kps[iR];
        const float kpY = kp.y;
        const float r = 2.0f * scale[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back((size_t)iR);  // D7
    }

    // :514-517
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;

    std::vector<std::pair<int, int> > vDistIdx;
    vDistIdx.reserve(N);
    for (int iL = 0; iL < N; iL++) {
        const orc_kp& kpL = left->kps[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y;
        const float uL = kpL.x;
        const size_t row = (size_t)vL;  // :530 (float index, truncated)
        if (row >= (size_t)nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[row];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;

        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = &left->desc[(size_t)iL * 32];
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {  // :547-568
            const size_t iR = vCandidates[iC];
            const orc_kp& kpR = right->kps[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL, &right->desc[iR * 32]);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }

        if (bestDist < thOrbDist) {  // :571
            const float uR0 = right->kps[bestIdxR].x;
            const float scaleFactor = inv_scale[kpL.octave];
            const float scaleduL = roundf(kpL.x * scaleFactor);
            const float scaledvL = roundf(kpL.y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5;
            const int lv = kpL.octave;
            const int xL = (int)(scaleduL - w), yL = (int)(scaledvL - w);  // window origin, :582
            const int cL = pyr_px(left, lv, xL + w, yL + w);                // IL.at(w,w), :584

            int bestDistW = INT_MAX;
            int bestincR = 0;
            const int L = 5;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= right->w[lv]) continue;  // :594

            for (int incR = -L; incR <= +L; incR++) {  // :597-611
                const int xR = (int)(scaleduR0 + incR - w);
                const int cR = pyr_px(right, lv, xR + w, yL + w);
                // cv::norm(IL, IR, NORM_L1) of the two centred float windows: integers, exact
                double acc = 0.0;
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++) {
                        const float a = (float)pyr_px(left, lv, xL + xx, yL + yy) - (float)cL;
                        const float b = (float)pyr_px(right, lv, xR + xx, yL + yy) - (float)cR;
                        acc += (double)std::fabs(a - b);
                    }
                const float dist = (float)acc;
                if (dist < bestDistW) {
                    bestDistW = (int)dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;  // :613

            // :616-624 parabola through the three distances around the minimum
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;

            float bestuR = scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);  // :627
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {  // :633-637, double constants
                    disparity = (float)0.01;
                    bestuR = (float)((double)uL - 0.01);
                }
                depth[iL] = mbf / disparity;
                u_right[iL] = bestuR;
                if (sad) sad[iL] = bestDistW;
                vDistIdx.push_back(std::pair<int, int>(bestDistW, iL));
            }
        }
    }

    if (vDistIdx.empty()) return 0;  // D7
    std::sort(vDistIdx.begin(), vDistIdx.end());  // :645
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {  // :649-658
        if (vDistIdx[i].first < thDist) break;
        u_right[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}

// ---------------------------------------------------------------------------------------------
// Input staging: gray conversion and rectification (reference src/Tracking.cc:175-199,
// Examples/Stereo/stereo_euroc.cc:137-138), OpenCV 3.4.5 8-bit paths.
// ---------------------------------------------------------------------------------------------

extern "C" void orc_cvt_gray(const uint8_t* src, int w, int h, int sstride, int channels, int blue_idx,
                             uint8_t* dst, int dstride) {
    // RGB2Gray<uchar>: table form of (b*B2Y + g*G2Y + r*R2Y + (1 << 13)) >> 14
    const int B2Y = 1868, G2Y = 9617, R2Y = 4899, shift = 14;
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, S += channels) {
            const int b = S[blue_idx], g = S[1], r = S[blue_idx ^ 2];
            D[x] = (uint8_t)((b * B2Y + g * G2Y + r * R2Y + (1 << (shift - 1))) >> shift);
        }
    }
}

static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

extern "C" void orc_remap_linear(const uint8_t* src, int sw, int sh, int sstride, const float* mapx,
                                 const float* mapy, int map_stride, uint8_t* dst, int dw, int dh, int dstride) {
    const int INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS;
    for (int y = 0; y < dh; y++) {
        const float* MX = mapx + (size_t)y * map_stride;
        const float* MY = mapy + (size_t)y * map_stride;
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            // remap(): float maps -> fixed point, 5 fractional bits
            const int fsx = cv_round((double)(MX[x] * (float)INTER_TAB_SIZE));
            const int fsy = cv_round((double)(MY[x] * (float)INTER_TAB_SIZE));
            const int sx = sat_short(fsx >> INTER_BITS), sy = sat_short(fsy >> INTER_BITS);
            const int fx = fsx & (INTER_TAB_SIZE - 1), fy = fsy & (INTER_TAB_SIZE - 1);
            // BilinearTab_i: exact for bilinear weights, sum 32768
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int v;
            if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
                const uint8_t* S = src + (size_t)sy * sstride + sx;
                v = S[0] * w00 + S[1] * w01 + S[sstride] * w10 + S[sstride + 1] * w11;
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                D[x] = 0;  // BORDER_CONSTANT, borderValue 0
                continue;
            } else {
                const int sx1 = sx + 1, sy1 = sy + 1;
                const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)sx1 < (unsigned)sw;
                const bool y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)sy1 < (unsigned)sh;
                const int v0 = (x0 && y0) ? src[(size_t)sy * sstride + sx] : 0;
                const int v1 = (x1 && y0) ? src[(size_t)sy * sstride + sx1] : 0;
                const int v2 = (x0 && y1) ? src[(size_t)sy1 * sstride + sx] : 0;
                const int v3 = (x1 && y1) ? src[(size_t)sy1 * sstride + sx1] : 0;
                v = v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11;
            }
            v = (v + (1 << 14)) >> 15;  // FixedPtCast<int, uchar, 15>
            D[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Undistorted keypoints and the frame grid (reference src/Frame.cc:421-451, 232-247, 398-408).
// ---------------------------------------------------------------------------------------------

// cvUndistortPointsInternal of OpenCV 3.4.5 with R = I, P = K, distortion (k1,k2,p1,p2,k3), no
// tilt, criteria (MAX_ITER, 5): all arithmetic in double, statements in OpenCV's order.
extern "C" void orc_undistort_points(const float* xy_in, int n, const float K[4], const float dist[5], float* xy_out) {
    if (dist[0] == 0.0f) {  // src/Frame.cc:423-427
        std::memcpy(xy_out, xy_in, (size_t)n * 2 * sizeof(float));
        return;
    }
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            // k[5..7] (rational model) and k[8..11] (thin prism) are zero
            const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0. * r2 + 0. * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        // RR = K * I: xx = fx*x + 0*y + cx, yy = 0*x + fy*y + cy, ww = 1/(0*x + 0*y + 1)
        const double xx = fx * x + 0. * y + cx;
        const double yy = 0. * x + fy * y + cy;
        const double ww = 1. / (0. * x + 0. * y + 1.);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

extern "C" void orc_assign_grid(const float* xy, int n, float min_x, float min_y, float inv_w, float inv_h,
                                int32_t* cell_start, int32_t* idx) {
    const int COLS = 64, ROWS = 48;  // include/Frame.h:37-38
    std::vector<std::vector<int> > grid((size_t)COLS * ROWS);
    for (int i = 0; i < n; i++) {
        const int posX = (int)roundf((xy[2 * i] - min_x) * inv_w);  // :400-401
        const int posY = (int)roundf((xy[2 * i + 1] - min_y) * inv_h);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[(size_t)posX * ROWS + posY].push_back(i);
    }
    int at = 0;
    for (int c = 0; c < COLS * ROWS; c++) {
        cell_start[c] = at;
        for (size_t k = 0; k < grid[c].size(); k++) idx[at++] = grid[c][k];
    }
    cell_start[COLS * ROWS] = at;
}

// ---------------------------------------------------------------------------------------------
// Bag-of-words transform (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260).
struct orc_voc {
    struct Node {
        std::vector<uint32_t> children;
        uint8_t desc[32];
        double weight;
        uint32_t word_id;
    };
    int k, L, scoring, weighting;
    std::vector<Node> nodes;
    int n_words;
};

extern "C" orc_voc* orc_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                                   const uint8_t* is_leaf, const uint8_t* desc, const double* weight) {
    if (n_nodes < 1 || L < 1 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return nullptr;
    orc_voc* v = new orc_voc;
    v->k = k;
    v->L = L;
    v->scoring = scoring;
    v->weighting = weighting;
    v->nodes.resize((size_t)n_nodes);
    v->n_words = 0;
    for (int i = 0; i < n_nodes; ++i) {
        orc_voc::Node& nd = v->nodes[(size_t)i];
        std::memcpy(nd.desc, desc + (size_t)i * 32, 32);
        nd.weight = weight[i];
        nd.word_id = 0;
        if (i == 0) continue;  // the root has no line in the file
        if (parent[i] < 0 || parent[i] >= i) {
            delete v;
            return nullptr;
        }
        v->nodes[(size_t)parent[i]].children.push_back((uint32_t)i);  // :1390-1391
        if (is_leaf[i]) nd.word_id = (uint32_t)v->n_words++;           // :1408-1415
    }
    return v;
}
extern "C" void orc_voc_destroy(orc_voc* v) { delete v; }
extern "C" int orc_voc_num_words(const orc_voc* v) { return v->n_words; }

extern "C" void orc_voc_transform_feature(const orc_voc* v, const uint8_t desc[32], int levelsup, uint32_t* word,
                                          uint32_t* node, double* weight) {
    const int nid_level = v->L - levelsup;  // :1226
    uint32_t nid = 0, final_id = 0;
    int current_level = 0;
    if (!v->nodes[0].children.empty()) {
        do {  // :1232-1254
            ++current_level;
            const std::vector<uint32_t>& ch = v->nodes[final_id].children;
            final_id = ch[0];
            double best_d = (double)orc_descriptor_distance(desc, v->nodes[final_id].desc);
            for (size_t c = 1; c < ch.size(); ++c) {
                const double d = (double)orc_descriptor_distance(desc, v->nodes[ch[c]].desc);
                if (d < best_d) {
                    best_d = d;
                    final_id = ch[c];
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!v->nodes[final_id].children.empty());
    }
    *word = v->nodes[final_id].word_id;  // :1257-1258
    *weight = v->nodes[final_id].weight;
    *node = nid;
}

extern "C" void orc_voc_transform(const orc_voc* v, const uint8_t* descs, int n, int levelsup, uint32_t* bow_id,
                                  double* bow_val, int* n_bow, uint32_t* fv_node, uint32_t* fv_idx, int* n_fv) {
    std::map<uint32_t, double> bow;
    std::map<uint32_t, std::vector<uint32_t>> fv;
    // L1 for every scoring but L2_NORM; DOT_PRODUCT does not normalise (ScoringObject.h:73-90)
    const bool must = v->scoring != 5, l2 = v->scoring == 1;
    const bool sums = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    if (!v->nodes[0].children.empty()) {  // empty(): nothing is produced (:1133-1136)
        for (int i = 0; i < n; ++i) {
            uint32_t id, nid;
            double w;
            orc_voc_transform_feature(v, descs + (size_t)i * 32, levelsup, &id, &nid, &w);
            if (w > 0) {  // :1157, :1185
                auto it = bow.lower_bound(id);
                if (it != bow.end() && it->first == id) {
                    if (sums) it->second += w;
                } else {
                    bow.insert(it, std::make_pair(id, w));
                }
                fv[nid].push_back((uint32_t)i);
            }
        }
        if (sums && !bow.empty() && !must) {  // :1164-1170
            const double nd = (double)bow.size();
            for (auto& e : bow) e.second /= nd;
        }
        if (must) {  // BowVector::normalize, BowVector.cpp:62-84
            double norm = 0.0;
            if (!l2) {
                for (auto& e : bow) norm += std::fabs(e.second);
            } else {
                for (auto& e : bow) norm += e.second * e.second;
                norm = std::sqrt(norm);
            }
            if (norm > 0.0)
                for (auto& e : bow) e.second /= norm;
        }
    }
    int nb = 0;
    for (auto& e : bow) {
        bow_id[nb] = e.first;
        bow_val[nb] = e.second;
        ++nb;
    }
    *n_bow = nb;
    int nf = 0;
    for (auto& e : fv)
        for (uint32_t i : e.second) {
            fv_node[nf] = e.first;
            fv_idx[nf] = i;
            ++nf;
        }
    *n_fv = nf;
}

// ---------------------------------------------------------------------------------------------
// Frame::GetFeaturesInArea (reference src/Frame.cc:338-396) + the best-distance loop of
// ORBmatcher::SearchByProjection (reference src/ORBmatcher.cc:1427-1455).
extern "C" int orc_search_window(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                                 const uint8_t* taken, const int32_t* cell_start, const int32_t* cell_idx, float min_x,
                                 float min_y, float inv_w, float inv_h, float x, float y, float r, int minLevel,
                                 int maxLevel, float ur, const uint8_t q_desc[32], int* best_dist) {
    int bestDist = 256, bestIdx2 = -1;
    *best_dist = bestDist;
    const int COLS = 64, ROWS = 48;
    const int nMinCellX = std::max(0, (int)std::floor((x - min_x - r) * inv_w));  // :343
    if (nMinCellX >= COLS) return -1;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - min_x + r) * inv_w));
    if (nMaxCellX < 0) return -1;
    const int nMinCellY = std::max(0, (int)std::floor((y - min_y - r) * inv_h));
    if (nMinCellY >= ROWS) return -1;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - min_y + r) * inv_h));
    if (nMaxCellY < 0) return -1;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);  // :359
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * ROWS + iy;
            for (int j = cell_start[c]; j < cell_start[c + 1]; j++) {
                const int i2 = cell_idx[j];
                if (bCheckLevels) {
                    if (octave[i2] < minLevel) continue;
                    if (maxLevel >= 0)
                        if (octave[i2] > maxLevel) continue;
                }
                const float distx = kps_xy[2 * i2] - x;
                const float disty = kps_xy[2 * i2 + 1] - y;
                if (!(std::fabs(distx) < r && std::fabs(disty) < r)) continue;  // :389
                // the candidate loop of SearchByProjection (src/ORBmatcher.cc:1432-1455)
                if (taken && taken[i2]) continue;
                if (u_right && u_right[i2] > 0) {
                    const float er = std::fabs(ur - u_right[i2]);
                    if (er > r) continue;
                }
                const int dist = orc_descriptor_distance(q_desc, desc + (size_t)i2 * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdx2 = i2;
                }
            }
        }
    *best_dist = bestDist;
    return bestIdx2;
}


// ORBmatcher::ComputeThreeMaxima (reference src/ORBmatcher.cc:1638-1679) on the bin sizes.
static void three_maxima(const int* size, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = size[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (reference src/ORBmatcher.cc:1365-1507).
extern "C" int orc_search_by_projection(const float* kps_xy, const int32_t* octave, const float* angle, const uint8_t* desc,
                                        const float* u_right, const uint8_t* taken0, int n_kp, const int32_t* cell_start,
                                        const int32_t* cell_idx, const float* scale_factors, const orc_proj_params* P,
                                        const float Tcw[12], const float Tlw[12], const orc_map_query* q, int nq,
                                        int32_t* match, int32_t* query_match) {
    const int HISTO_LENGTH = 30, TH_HIGH = 100;
    int nmatches = 0;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;  // :1373
    std::vector<uint8_t> blocked((size_t)std::max(n_kp, 1), 0);
    for (int i = 0; i < n_kp; i++) {
        blocked[i] = taken0 ? taken0[i] : 0;
        match[i] = -1;
    }
    // twc = -Rcw.t()*tcw (:1378): cv::gemm with a transposed operand, sums kept in double, scaled by -1
    float twc[3];
    for (int k = 0; k < 3; k++) {
        double s = 0.0;
        for (int j = 0; j < 3; j++) s += (double)Tcw[4 * j + k] * (double)Tcw[4 * j + 3];
        twc[k] = (float)(s * -1.0);
    }
    // tlc = Rlw*twc + tlw (:1383): cv::gemm's 3x3 float path, products summed left to right in float
    float tlc[3];
    for (int r = 0; r < 3; r++) {
        const float t = Tlw[4 * r] * twc[0] + Tlw[4 * r + 1] * twc[1] + Tlw[4 * r + 2] * twc[2];
        tlc[r] = t + Tlw[4 * r + 3];
    }
    const bool bForward = tlc[2] > P->mb && !P->mono;    // :1385
    const bool bBackward = -tlc[2] > P->mb && !P->mono;  // :1386
    for (int i = 0; i < nq; i++) {
        query_match[i] = -1;
        if (q[i].flags & 2) continue;  // no map point, or an outlier (:1392-1394)
        float c3[3];
        for (int r = 0; r < 3; r++) {  // x3Dc = Rcw*x3Dw + tcw (:1398)
            const float t = Tcw[4 * r] * q[i].x + Tcw[4 * r + 1] * q[i].y + Tcw[4 * r + 2] * q[i].z;
            c3[r] = t + Tcw[4 * r + 3];
        }
        const float xc = c3[0], yc = c3[1];
        const float invzc = (float)(1.0 / (double)c3[2]);
        if (invzc < 0) continue;
        const float u = P->fx * xc * invzc + P->cx;
        const float v = P->fy * yc * invzc + P->cy;
        if (u < P->min_x || u > P->max_x) continue;
        if (v < P->min_y || v > P->max_y) continue;
        const int nLastOctave = q[i].octave;
        const float radius = P->th * scale_factors[nLastOctave];
        int minLevel, maxLevel;
        if (bForward) {
            minLevel = nLastOctave; maxLevel = -1;
        } else if (bBackward) {
            minLevel = 0; maxLevel = nLastOctave;
        } else {
            minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1;
        }
        const float ur = u - P->mbf * invzc;  // :1446
        int bestDist = 256;
        const int bestIdx2 = orc_search_window(kps_xy, octave, desc, u_right, blocked.data(), cell_start, cell_idx, P->min_x,
                                               P->min_y, P->inv_w, P->inv_h, u, v, radius, minLevel, maxLevel, ur, q[i].desc,
                                               &bestDist);
        if (bestDist <= TH_HIGH) {
            match[bestIdx2] = i;
            query_match[i] = bestIdx2;
            if (q[i].flags & 1) blocked[bestIdx2] = 1;  // Observations() > 0: later points skip it (:1440-1442)
            nmatches++;
            if (P->check_orientation) {
                float rot = q[i].angle - angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (P->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (reference src/ORBmatcher.cc:45-129,
// RadiusByViewingCos :131-137), the local-map search of Tracking::SearchLocalPoints.  A query is a map
// point as Frame::isInFrustum left it: mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos;
// flags bit 0: Observations() > 0, bit 1: skipped (!mbTrackInView or isBad()).  match[k] = index of
// the map point feature k holds after the call (-1: the call wrote none), query_match[i] = the
// feature point i was written to (-1: none); returns nmatches.
extern "C" int orc_search_local_map(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                                    const uint8_t* taken0, int n_kp, const int32_t* cell_start, const int32_t* cell_idx,
                                    const float* scale_factors, float min_x, float min_y, float inv_w, float inv_h, float th,
                                    float nn_ratio, const orc_local_query* q, int nq, int32_t* match, int32_t* query_match) {
    const int TH_HIGH = 100, COLS = 64, ROWS = 48;
    int nmatches = 0;
    std::vector<uint8_t> blocked((size_t)std::max(n_kp, 1), 0);
    for (int i = 0; i < n_kp; i++) {
        blocked[i] = taken0 ? taken0[i] : 0;
        match[i] = -1;
    }
    const bool bFactor = th != 1.0;  // :49
    for (int iMP = 0; iMP < nq; iMP++) {
        query_match[iMP] = -1;
        if (q[iMP].flags & 2) continue;  // :54-58
        const int nPredictedLevel = q[iMP].level;
        float r = (double)q[iMP].view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos
        if (bFactor) r *= th;
        // GetFeaturesInArea(mTrackProjX, mTrackProjY, r * mvScaleFactors[level], level - 1, level), src/Frame.cc:338-396
        const float x = q[iMP].u, y = q[iMP].v, rr = r * scale_factors[nPredictedLevel];
        const int minLevel = nPredictedLevel - 1, maxLevel = nPredictedLevel;
        const int nMinCellX = std::max(0, (int)std::floor((x - min_x - rr) * inv_w));
        if (nMinCellX >= COLS) continue;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - min_x + rr) * inv_w));
        if (nMaxCellX < 0) continue;
        const int nMinCellY = std::max(0, (int)std::floor((y - min_y - rr) * inv_h));
        if (nMinCellY >= ROWS) continue;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - min_y + rr) * inv_h));
        if (nMaxCellY < 0) continue;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;  // :76-80
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c = ix * ROWS + iy;
                for (int j = cell_start[c]; j < cell_start[c + 1]; j++) {
                    const int idx = cell_idx[j];
                    if (bCheckLevels) {
                        if (octave[idx] < minLevel) continue;
                        if (maxLevel >= 0)
                            if (octave[idx] > maxLevel) continue;
                    }
                    const float distx = kps_xy[2 * idx] - x;
                    const float disty = kps_xy[2 * idx + 1] - y;
                    if (!(std::fabs(distx) < rr && std::fabs(disty) < rr)) continue;
                    if (blocked[idx]) continue;  // :87-89
                    if (u_right && u_right[idx] > 0) {  // :91-96
                        const float er = std::fabs(q[iMP].ur - u_right[idx]);
                        if (er > r * scale_factors[nPredictedLevel]) continue;
                    }
                    const int dist = orc_descriptor_distance(q[iMP].desc, desc + (size_t)idx * 32);
                    if (dist < bestDist) {  // :102-114
                        bestDist2 = bestDist;
                        bestDist = dist;
                        bestLevel2 = bestLevel;
                        bestLevel = octave[idx];
                        bestIdx = idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = octave[idx];
                        bestDist2 = dist;
                    }
                }
            }
        if (bestDist <= TH_HIGH) {  // :118-125
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            match[bestIdx] = iMP;
            query_match[iMP] = bestIdx;
            if (q[iMP].flags & 1) blocked[bestIdx] = 1;
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (reference src/ORBmatcher.cc:159-290):
// features of the keyframe and of the frame that fall into the same vocabulary node are matched, node by
// node.  A FeatureVector comes as (node, feature index) pairs, nodes ascending and indices ascending inside
// a node (what std::map<NodeId, vector<unsigned>> iterates as); valid_kf[i] != 0: the keyframe's feature i
// holds a map point that is not bad (:193-197).  match[k] = index of the keyframe feature whose map point
// frame feature k ends with (-1: none); returns nmatches.
extern "C" int orc_search_by_bow(const uint32_t* fv_node_kf, const uint32_t* fv_idx_kf, int fv_n_kf, const uint8_t* desc_kf,
                                 const float* angle_kf, const uint8_t* valid_kf, const uint32_t* fv_node_f,
                                 const uint32_t* fv_idx_f, int fv_n_f, const uint8_t* desc_f, const float* angle_f, int n_f,
                                 float nn_ratio, int check_orientation, int32_t* match) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    for (int i = 0; i < n_f; i++) match[i] = -1;  // :163
    int nmatches = 0;
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;  // :172
    int kf = 0, f = 0;  // first pair of the current node on each side (the map iterators KFit, Fit)
    while (kf < fv_n_kf && f < fv_n_f) {  // :180
        const uint32_t node_kf = fv_node_kf[kf], node_f = fv_node_f[f];
        if (node_kf == node_f) {  // :182
            int kf_end = kf, f_end = f;
            while (kf_end < fv_n_kf && fv_node_kf[kf_end] == node_kf) kf_end++;
            while (f_end < fv_n_f && fv_node_f[f_end] == node_f) f_end++;
            for (int iKF = kf; iKF < kf_end; iKF++) {  // :187
                const unsigned realIdxKF = fv_idx_kf[iKF];
                if (!valid_kf[realIdxKF]) continue;  // :193-197
                const uint8_t* dKF = desc_kf + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;  // :201-203
                for (int iF = f; iF < f_end; iF++) {
                    const unsigned realIdxF = fv_idx_f[iF];
                    if (match[realIdxF] >= 0) continue;  // :209-210
                    const int dist = orc_descriptor_distance(dKF, desc_f + (size_t)realIdxF * 32);
                    if (dist < bestDist1) {  // :216-225
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = (int)realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= TH_LOW) {  // :228
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {  // :230
                        match[bestIdxF] = (int)realIdxKF;
                        if (check_orientation) {  // :236-248
                            float rot = angle_kf[realIdxKF] - angle_f[bestIdxF];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            kf = kf_end;  // :253-254
            f = f_end;
        } else if (node_kf < node_f) {  // :256-259: lower_bound(Fit->first)
            while (kf < fv_n_kf && fv_node_kf[kf] < node_f) kf++;
        } else {  // :260-263
            while (f < fv_n_f && fv_node_f[f] < node_kf) f++;
        }
    }
    if (check_orientation) {  // :267-288
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (reference src/ORBmatcher.cc:551-684), the
// matcher of the loop closer: as above between two keyframes.  valid_1 / valid_2: the feature holds a map point
// that is not bad; match12[i] = the feature of the second keyframe whose map point vpMatches12[i] becomes
// (-1: NULL); returns nmatches.
extern "C" int orc_search_by_bow_keyframes(const uint32_t* fv_node_1, const uint32_t* fv_idx_1, int fv_n_1,
                                           const uint8_t* desc_1, const float* angle_1, const uint8_t* valid_1, int n_1,
                                           const uint32_t* fv_node_2, const uint32_t* fv_idx_2, int fv_n_2,
                                           const uint8_t* desc_2, const float* angle_2, const uint8_t* valid_2, int n_2,
                                           float nn_ratio, int check_orientation, int32_t* match12) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    for (int i = 0; i < n_1; i++) match12[i] = -1;  // :563
    std::vector<uint8_t> vbMatched2((size_t)std::max(n_2, 1), 0);  // :564
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;  // :570
    int nmatches = 0;
    int a = 0, b = 0;  // first pair of the current node on each side (f1it, f2it)
    while (a < fv_n_1 && b < fv_n_2) {  // :579
        const uint32_t node_1 = fv_node_1[a], node_2 = fv_node_2[b];
        if (node_1 == node_2) {  // :581
            int a_end = a, b_end = b;
            while (a_end < fv_n_1 && fv_node_1[a_end] == node_1) a_end++;
            while (b_end < fv_n_2 && fv_node_2[b_end] == node_2) b_end++;
            for (int i1 = a; i1 < a_end; i1++) {  // :583
                const unsigned idx1 = fv_idx_1[i1];
                if (!valid_1[idx1]) continue;  // :587-591
                const uint8_t* d1 = desc_1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;  // :595-597
                for (int i2 = b; i2 < b_end; i2++) {
                    const unsigned idx2 = fv_idx_2[i2];
                    if (vbMatched2[idx2] || !valid_2[idx2]) continue;  // :603-609
                    const int dist = orc_descriptor_distance(d1, desc_2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) {  // :615-624
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = (int)idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < TH_LOW) {  // :627
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {  // :629
                        match12[idx1] = bestIdx2;  // :631-632
                        vbMatched2[bestIdx2] = 1;
                        if (check_orientation) {  // :634-645
                            float rot = angle_1[idx1] - angle_2[bestIdx2];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            a = a_end;  // :650-651
            b = b_end;
        } else if (node_1 < node_2) {  // :653-656
            while (a < fv_n_1 && fv_node_1[a] < node_2) a++;
        } else {  // :657-660
            while (b < fv_n_2 && fv_node_2[b] < node_1) b++;
        }
    }
    if (check_orientation) {  // :663-682
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match12[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

// Frame::ComputeStereoFromRGBD (reference src/Frame.cc:662-682) on a depth map that Tracking::GrabImageRGBD
// would convert with imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) (src/Tracking.cc:230-231; OpenCV 3.4.5's
// 16u -> 32f and 32f -> 32f scale conversions compute sample * (float)alpha + 0 in float32).  kps_xy: mvKeys
// (x, y) pairs; un_x: mvKeysUn[i].pt.x.
extern "C" void orc_stereo_from_rgbd(const void* depth, int is_f32, int width, int height, int stride_bytes, float factor,
                                     const float* kps_xy, const float* un_x, int n, float mbf, float* u_right,
                                     float* depth_out) {
    for (int i = 0; i < n; i++) {
        u_right[i] = -1.0f;  // :664-665
        depth_out[i] = -1.0f;
        const float u = kps_xy[2 * i], v = kps_xy[2 * i + 1];  // :672-673
        const int x = (int)u, y = (int)v;                      // at<float>(v, u): float -> int truncates
        if (x < 0 || x >= width || y < 0 || y >= height) continue;
        const uint8_t* row = (const uint8_t*)depth + (size_t)y * stride_bytes;
        const float raw = is_f32 ? ((const float*)row)[x] : (float)((const uint16_t*)row)[x];
        const float d = raw * factor;
        if (d > 0) {  // :677-681
            depth_out[i] = d;
            u_right[i] = un_x[i] - mbf / d;
        }
    }
}

// ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:418-549), the matcher of the monocular
// initialisation (Tracking::MonocularInitialization, src/Tracking.cc:617): every level-0 keypoint of the first
// frame searches a square window of the second frame around the place it was last matched at.  octave1 / angle1 /
// desc1: F1.mvKeysUn and descriptors; prev_matched (in and out): vbPrevMatched as (x, y) pairs; the second frame's
// undistorted keypoints with its grid in CSR form (cell = column * 48 + row).  match12 = vnMatches12; returns
// nmatches.
extern "C" int orc_search_for_initialization(const int32_t* octave1, const float* angle1, const uint8_t* desc1, int n1,
                                             float* prev_matched, const float* kps2_xy, const int32_t* octave2,
                                             const float* angle2, const uint8_t* desc2, int n2, const int32_t* cell_start2,
                                             const int32_t* cell_idx2, float min_x, float min_y, float inv_w, float inv_h,
                                             int window_size, float nn_ratio, int check_orientation, int32_t* match12) {
    const int HISTO_LENGTH = 30, TH_LOW = 50, COLS = 64, ROWS = 48;
    int nmatches = 0;                                    // :420
    for (int i = 0; i < n1; i++) match12[i] = -1;        // :422
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;            // :428
    std::vector<int> vMatchedDistance((size_t)std::max(n2, 1), INT_MAX), vnMatches21((size_t)std::max(n2, 1), -1);  // :430-431
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = octave1[i1];
        if (level1 > 0) continue;  // :438
        // F2.GetFeaturesInArea(vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, level1, level1), src/Frame.cc:338-396
        const float x = prev_matched[2 * i1], y = prev_matched[2 * i1 + 1], r = (float)window_size;
        const int minLevel = level1, maxLevel = level1;
        const int nMinCellX = std::max(0, (int)std::floor((x - min_x - r) * inv_w));
        if (nMinCellX >= COLS) continue;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - min_x + r) * inv_w));
        if (nMaxCellX < 0) continue;
        const int nMinCellY = std::max(0, (int)std::floor((y - min_y - r) * inv_h));
        if (nMinCellY >= ROWS) continue;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - min_y + r) * inv_h));
        if (nMaxCellY < 0) continue;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        const uint8_t* d1 = desc1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;  // :450-452
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c = ix * ROWS + iy;
                for (int j = cell_start2[c]; j < cell_start2[c + 1]; j++) {
                    const int i2 = cell_idx2[j];
                    if (bCheckLevels) {
                        if (octave2[i2] < minLevel) continue;
                        if (maxLevel >= 0)
                            if (octave2[i2] > maxLevel) continue;
                    }
                    const float distx = kps2_xy[2 * i2] - x;
                    const float disty = kps2_xy[2 * i2 + 1] - y;
                    if (!(std::fabs(distx) < r && std::fabs(disty) < r)) continue;
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)i2 * 32);
                    if (vMatchedDistance[i2] <= dist) continue;  // :465-466
                    if (dist < bestDist) {                       // :468-477
                        bestDist2 = bestDist;
                        bestDist = dist;
                        bestIdx2 = i2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
            }
        if (bestDist <= TH_LOW) {                                // :482
            if (bestDist < (float)bestDist2 * nn_ratio) {        // :484
                if (vnMatches21[bestIdx2] >= 0) {                // :486-490
                    match12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                match12[i1] = bestIdx2;                          // :491-495
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {                         // :497-509
                    float rot = angle1[i1] - angle2[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {  // :514-540
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    const int idx1 = rotHist[i][j];
                    if (match12[idx1] >= 0) {  // :532-536
                        match12[idx1] = -1;
                        nmatches--;
                    }
                }
    }
    for (int i1 = 0; i1 < n1; i1++)  // :544-546
        if (match12[i1] >= 0) {
            prev_matched[2 * i1] = kps2_xy[2 * match12[i1]];
            prev_matched[2 * i1 + 1] = kps2_xy[2 * match12[i1] + 1];
        }
    return nmatches;
}

// ORBmatcher::CheckDistEpipolarLine (reference src/ORBmatcher.cc:140-157): the squared distance of kp2 from the
// epipolar line x1' F12 against 3.84 * mvLevelSigma2[kp2.octave] (a double product).  f12: row-major 3x3.
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* f12, float sigma2) {
    const float a = x1 * f12[0] + y1 * f12[3] + f12[6];
    const float b = x1 * f12[1] + y1 * f12[4] + f12[7];
    const float c = x1 * f12[2] + y1 * f12[5] + f12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

// ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:686-852), the matcher of
// LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:329): between the features of two keyframes that hold no map
// point yet, node by node of the FeatureVectors, the smallest descriptor distance <= TH_LOW among the candidates that
// lie far enough from the epipole (monocular pairs) and close enough to the epipolar line.  The reference never sets
// vbMatched2 (:706 is its only write), so a feature of the second keyframe may serve several of the first.
// kps*_xy: mvKeysUn positions; has_mp*: GetMapPoint(idx) != NULL; u_right*: mvuRight; f12: row-major; (ex, ey): the
// epipole in the second image (:693-699, computed by the caller).  match12 = vMatches12, from which vMatchedPairs is
// every (i, match12[i]) with match12[i] >= 0 in index order (:841-849); returns nmatches.
extern "C" int orc_search_for_triangulation(const uint32_t* fv_node_1, const uint32_t* fv_idx_1, int fv_n_1,
                                            const uint8_t* desc_1, const float* kps1_xy, const float* angle_1,
                                            const uint8_t* has_mp_1, const float* u_right_1, int n_1,
                                            const uint32_t* fv_node_2, const uint32_t* fv_idx_2, int fv_n_2,
                                            const uint8_t* desc_2, const float* kps2_xy, const float* angle_2,
                                            const int32_t* octave_2, const uint8_t* has_mp_2, const float* u_right_2,
                                            int n_2, const float* f12, float ex, float ey, const float* scale_factors,
                                            const float* level_sigma2, int only_stereo, int check_orientation,
                                            int32_t* match12) {
    (void)n_2;
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    int nmatches = 0;                                  // :705
    for (int i = 0; i < n_1; i++) match12[i] = -1;     // :707
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    const float factor = 1.0f / HISTO_LENGTH;          // :713
    int a = 0, b = 0;
    while (a < fv_n_1 && b < fv_n_2) {  // :720
        const uint32_t node_1 = fv_node_1[a], node_2 = fv_node_2[b];
        if (node_1 == node_2) {
            int a_end = a, b_end = b;
            while (a_end < fv_n_1 && fv_node_1[a_end] == node_1) a_end++;
            while (b_end < fv_n_2 && fv_node_2[b_end] == node_2) b_end++;
            for (int i1 = a; i1 < a_end; i1++) {  // :724
                const unsigned idx1 = fv_idx_1[i1];
                if (has_mp_1[idx1]) continue;  // :728-732
                const bool bStereo1 = u_right_1[idx1] >= 0;  // :734
                if (only_stereo && !bStereo1) continue;      // :736-738
                const uint8_t* d1 = desc_1 + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;  // :744-745
                for (int i2 = b; i2 < b_end; i2++) {
                    const unsigned idx2 = fv_idx_2[i2];
                    if (has_mp_2[idx2]) continue;  // :751-755 (vbMatched2 stays false)
                    const bool bStereo2 = u_right_2[idx2] >= 0;  // :757
                    if (only_stereo && !bStereo2) continue;      // :759-761
                    const int dist = orc_descriptor_distance(d1, desc_2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;  // :767
                    const float x2 = kps2_xy[2 * idx2], y2 = kps2_xy[2 * idx2 + 1];
                    if (!bStereo1 && !bStereo2) {  // :772-778
                        const float distex = ex - x2, distey = ey - y2;
                        if (distex * distex + distey * distey < 100 * scale_factors[octave_2[idx2]]) continue;
                    }
                    if (check_dist_epipolar_line(kps1_xy[2 * idx1], kps1_xy[2 * idx1 + 1], x2, y2, f12,
                                                 level_sigma2[octave_2[idx2]])) {  // :780-784
                        bestIdx2 = (int)idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {  // :787-804
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = angle_1[idx1] - angle_2[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            a = a_end;  // :807-808
            b = b_end;
        } else if (node_1 < node_2) {  // :810-813
            while (a < fv_n_1 && fv_node_1[a] < node_2) a++;
        } else {  // :814-817
            while (b < fv_n_2 && fv_node_2[b] < node_1) b++;
        }
    }
    if (check_orientation) {  // :820-839
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match12[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

// The search of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) for one map point that passed the tests
// of :873-922 (reference src/ORBmatcher.cc:929-986; KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:575-614): u, v
// (:892-893), ur (:899), radius (:927), nPredictedLevel (:924).  u_right: pKF->mvuRight (null: all -1);
// inv_level_sigma2: pKF->mvInvLevelSigma2.  Returns bestIdx (-1: none) and bestDist (256: none); the caller applies
// bestDist <= TH_LOW (:989).
extern "C" int orc_fuse_search(const float* kps_xy, const int32_t* octave, const uint8_t* desc, const float* u_right,
                               const int32_t* cell_start, const int32_t* cell_idx, float min_x, float min_y, float inv_w,
                               float inv_h, float u, float v, float ur, float radius, int nPredictedLevel,
                               const float* inv_level_sigma2, const uint8_t q_desc[32], int* best_dist) {
    int bestDist = 256, bestIdx = -1;  // :938-939
    *best_dist = bestDist;
    const int COLS = 64, ROWS = 48;
    const float x = u, y = v, r = radius;
    const int nMinCellX = std::max(0, (int)std::floor((x - min_x - r) * inv_w));  // KeyFrame.cc:580
    if (nMinCellX >= COLS) return -1;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - min_x + r) * inv_w));
    if (nMaxCellX < 0) return -1;
    const int nMinCellY = std::max(0, (int)std::floor((y - min_y - r) * inv_h));
    if (nMinCellY >= ROWS) return -1;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - min_y + r) * inv_h));
    if (nMaxCellY < 0) return -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * ROWS + iy;
            for (int j = cell_start[c]; j < cell_start[c + 1]; j++) {
                const int idx = cell_idx[j];
                const float kpx = kps_xy[2 * idx], kpy = kps_xy[2 * idx + 1];
                if (!(std::fabs(kpx - x) < r && std::fabs(kpy - y) < r)) continue;  // KeyFrame.cc:604-608
                const int kpLevel = octave[idx];
                if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;  // :948
                if (u_right && u_right[idx] >= 0) {  // :951-964
                    const float kpr = u_right[idx];
                    const float ex = u - kpx;
                    const float ey = v - kpy;
                    const float er = ur - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {  // :965-975
                    const float ex = u - kpx;
                    const float ey = v - kpy;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
                const int dist = orc_descriptor_distance(q_desc, desc + (size_t)idx * 32);
                if (dist < bestDist) {  // :981-985
                    bestDist = dist;
                    bestIdx = idx;
                }
            }
        }
    *best_dist = bestDist;
    return bestIdx;
}

// The matching loops of ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (reference
// src/ORBmatcher.cc:1565-1632: level_up = 1, max_dist = ORBdist, rotation histogram) and of
// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:362-398: level_up = 0, max_dist = TH_LOW,
// no histogram) on map points the caller has projected (:1534-1563, :321-360).  q: orc_local_query records read as
// (u, v, angle, predicted level, radius, flags, descriptor); flags bit 1: the reference skips the point.  taken0:
// features that hold a map point before the call (:1578, :375).  match[k] = the point feature k receives,
// query_match[i] = the feature of point i; returns nmatches.
extern "C" int orc_search_guided(const float* kps_xy, const int32_t* octave, const float* angle, const uint8_t* desc, int n_kp,
                                 const uint8_t* taken0, const int32_t* cell_start, const int32_t* cell_idx, float min_x,
                                 float min_y, float inv_w, float inv_h, const orc_local_query* q, int n_q, int level_up,
                                 int max_dist, int check_orientation, int32_t* match, int32_t* query_match) {
    const int HISTO_LENGTH = 30;
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> blocked((size_t)std::max(n_kp, 1), 0);
    for (int k = 0; k < n_kp; k++) {
        match[k] = -1;
        blocked[k] = taken0 ? taken0[k] : 0;
    }
    std::vector<std::vector<int>> rotHist(HISTO_LENGTH);
    int nmatches = 0;
    for (int i = 0; i < n_q; i++) {
        query_match[i] = -1;
        if (q[i].flags & 2) continue;
        const int nPredictedLevel = q[i].level;
        const float radius = q[i].view_cos;  // the record's fifth field: th * mvScaleFactors[nPredictedLevel]
        int bestDist = 256;
        const int bestIdx2 = orc_search_window(kps_xy, octave, desc, nullptr, blocked.data(), cell_start, cell_idx, min_x, min_y,
                                               inv_w, inv_h, q[i].u, q[i].v, radius, nPredictedLevel - 1,
                                               nPredictedLevel + level_up, 0.0f, q[i].desc, &bestDist);
        if (bestIdx2 >= 0 && bestDist <= max_dist) {  // :1592, :394
            match[bestIdx2] = i;
            query_match[i] = bestIdx2;
            blocked[bestIdx2] = 1;
            nmatches++;
            if (check_orientation) {  // :1597-1607; the record's third field: pKF->mvKeysUn[i].angle
                float rot = q[i].ur - angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {  // :1614-1633
        int ind1 = -1, ind2 = -1, ind3 = -1, size[30];
        for (int i = 0; i < HISTO_LENGTH; i++) size[i] = (int)rotHist[i].size();
        three_maxima(size, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    query_match[match[rotHist[i][j]]] = -1;
                    match[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

// DBoW2 L1Scoring::score (reference Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68), what ORBVocabulary::score computes
// for ORB-SLAM2's vocabulary (L1_NORM; called from KeyFrameDatabase::DetectLoopCandidates, src/KeyFrameDatabase.cc:133,
// DetectRelocalizationCandidates, :249, and LoopClosing::DetectLoop, src/LoopClosing.cc:137): both BowVectors as
// ascending (word id, value) arrays.
extern "C" double orc_bow_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2) {
    int a = 0, b = 0;
    double score = 0;
    while (a < n1 && b < n2) {
        const double vi = v1[a], wi = v2[b];
        if (id1[a] == id2[b]) {
            score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);  // :41
            ++a;
            ++b;
        } else if (id1[a] < id2[b]) {
            while (a < n1 && id1[a] < id2[b]) ++a;  // lower_bound (:50)
        } else {
            while (b < n2 && id2[b] < id1[a]) ++b;  // :56
        }
    }
    score = -score / 2.0;  // :65
    return score;
}

// Natural logarithm of a positive finite float in double, from IEEE additions, multiplications and one division
// (no libm call, so any IEEE machine computes the same bits): x = m * 2^e with m in (sqrt(1/2), sqrt(2)],
// s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + ... + s^24/25).  Narrowed to float it is the correctly rounded
// logf(x), which is what the reference's log(float) (src/MapPoint.cc:410) returns from a C library up to that
// library's rounding of arguments that sit on a rounding boundary.
extern "C" double orc_log_pinned(float x) {
    int e = 0;
    double m = 2.0 * std::frexp((double)x, &e);  // [1, 2), x = m * 2^(e - 1); both steps are exact
    e -= 1;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    for (int k = 23; k >= 1; k -= 2) {
        p = p * z;
        p = p + 1.0 / (double)k;
    }
    const double hi = (double)e * 0.6931471805599453;
    const double lo = (2.0 * s) * p;
    return hi + lo;
}

// MapPoint::PredictScale (reference src/MapPoint.cc:402-417).  On x86-64 the conversion of a float that is not a
// finite number of int range yields INT_MIN, which the clamp turns into level 0.
static int predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels) {
    const float ratio = max_distance / current_dist;
    if (!(ratio > 0.0f) || std::isinf(ratio)) return 0;  // logf: -inf, NaN or +inf
    const float q = std::ceil((float)orc_log_pinned(ratio) / log_scale_factor);
    int nScale;
    if (!(q >= -2147483648.0f && q < 2147483648.0f)) nScale = INT32_MIN;
    else nScale = (int)q;
    if (nScale < 0) nScale = 0;
    else if (nScale >= n_levels) nScale = n_levels - 1;
    return nScale;
}

// The steps Frame::isInFrustum (reference src/Frame.cc:271-327) and the first half of ORBmatcher::Fuse
// (src/ORBmatcher.cc:881-927) share, with cv::Mat's arithmetic: gemm of a 3x3 by a 3x1 float matrix sums its
// products left to right in float; cv::norm(NORM_L2) and cv::Mat::dot of three floats accumulate in double.
struct Projected {
    float pc[3], po[3], dist;
    double dot;
};
static Projected project_point(const float T[12], const float Ow[3], const float P[3], const float Pn[3]) {
    Projected r;
    for (int k = 0; k < 3; k++) {
        float acc = T[4 * k] * P[0];
        acc = acc + T[4 * k + 1] * P[1];
        acc = acc + T[4 * k + 2] * P[2];
        r.pc[k] = acc + T[4 * k + 3];
        r.po[k] = P[k] - Ow[k];
    }
    double n2 = 0.0, d = 0.0;
    for (int k = 0; k < 3; k++) {
        n2 += (double)r.po[k] * (double)r.po[k];
        d += (double)r.po[k] * (double)Pn[k];
    }
    r.dist = (float)std::sqrt(n2);
    r.dot = d;
    return r;
}

// Frame::isInFrustum (reference src/Frame.cc:271-327) for one map point; cam = fx, fy, cx, cy, mbf; bounds = mnMinX,
// mnMaxX, mnMinY, mnMaxY.  Returns mbTrackInView and fills mTrackProjX, mTrackProjY, mTrackProjXR,
// mnTrackScaleLevel, mTrackViewCos.
extern "C" int orc_is_in_frustum(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4],
                                 float viewing_cos_limit, float log_scale_factor, int n_levels, const float P[3],
                                 const float Pn[3], float min_distance, float max_distance, float* proj_x, float* proj_y,
                                 float* proj_xr, int* scale_level, float* view_cos) {
    const Projected g = project_point(Tcw, Ow, P, Pn);
    const float PcX = g.pc[0], PcY = g.pc[1], PcZ = g.pc[2];
    if (PcZ < 0.0f) return 0;  // :285
    const float invz = 1.0f / PcZ;
    const float u = cam[0] * PcX * invz + cam[2];
    const float v = cam[1] * PcY * invz + cam[3];
    if (u < bounds[0] || u > bounds[1]) return 0;  // :293
    if (v < bounds[2] || v > bounds[3]) return 0;
    const float maxDistance = 1.2f * max_distance;  // MapPoint::GetMaxDistanceInvariance, src/MapPoint.cc:379-383
    const float minDistance = 0.8f * min_distance;
    const float dist = g.dist;
    if (dist < minDistance || dist > maxDistance) return 0;  // :304
    const float viewCos = (float)(g.dot / (double)dist);     // :310, double / float
    if (viewCos < viewing_cos_limit) return 0;
    *scale_level = predict_scale(max_distance, dist, log_scale_factor, n_levels);
    *proj_x = u;
    *proj_xr = u - cam[4] * invz;
    *proj_y = v;
    *view_cos = viewCos;
    return 1;
}

// The tests ORBmatcher::Fuse(pKF, vpMapPoints, th) applies to a map point before it searches (reference
// src/ORBmatcher.cc:881-927; KeyFrame::IsInImage, src/KeyFrame.cc:616-619).  Returns 0 when the point is dropped.
extern "C" int orc_fuse_project(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4],
                                float log_scale_factor, int n_levels, const float* scale_factors, float th, const float P[3],
                                const float Pn[3], float min_distance, float max_distance, float* u_out, float* v_out,
                                float* ur_out, float* radius, int* predicted_level) {
    const Projected g = project_point(Tcw, Ow, P, Pn);
    if (g.pc[2] < 0.0f) return 0;  // :885
    const float invz = 1 / g.pc[2];
    const float x = g.pc[0] * invz;
    const float y = g.pc[1] * invz;
    const float u = cam[0] * x + cam[2];
    const float v = cam[1] * y + cam[3];
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) return 0;  // IsInImage
    const float ur = u - cam[4] * invz;
    const float maxDistance = 1.2f * max_distance;
    const float minDistance = 0.8f * min_distance;
    const float dist3D = g.dist;
    if (dist3D < minDistance || dist3D > maxDistance) return 0;  // :915
    if (g.dot < 0.5 * dist3D) return 0;                           // :921
    const int nPredictedLevel = predict_scale(max_distance, dist3D, log_scale_factor, n_levels);
    *radius = th * scale_factors[nPredictedLevel];  // :927
    *u_out = u;
    *v_out = v;
    *ur_out = ur;
    *predicted_level = nPredictedLevel;
    return 1;
}

// The projection of ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (reference
// src/ORBmatcher.cc:1534-1563) for one map point: no depth test and no viewing test.  Returns 0 when the point is dropped.
extern "C" int orc_reloc_project(const float Tcw[12], const float Ow[3], const float cam[5], const float bounds[4],
                                 float log_scale_factor, int n_levels, const float* scale_factors, float th, const float P[3],
                                 float min_distance, float max_distance, float* u_out, float* v_out, float* radius,
                                 int* predicted_level) {
    const float none[3] = {0.0f, 0.0f, 0.0f};
    const Projected g = project_point(Tcw, Ow, P, none);
    const float xc = g.pc[0];
    const float yc = g.pc[1];
    const float invzc = (float)(1.0 / (double)g.pc[2]);  // :1539
    const float u = cam[0] * xc * invzc + cam[2];
    const float v = cam[1] * yc * invzc + cam[3];
    if (u < bounds[0] || u > bounds[1]) return 0;  // :1544
    if (v < bounds[2] || v > bounds[3]) return 0;
    const float dist3D = g.dist;
    const float maxDistance = 1.2f * max_distance;
    const float minDistance = 0.8f * min_distance;
    if (dist3D < minDistance || dist3D > maxDistance) return 0;  // :1557
    const int nPredictedLevel = predict_scale(max_distance, dist3D, log_scale_factor, n_levels);
    *radius = th * scale_factors[nPredictedLevel];  // :1563
    *u_out = u;
    *v_out = v;
    *predicted_level = nPredictedLevel;
    return 1;
}

// The projection of one direction of ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1195-1226) for one map
// point: T1w = [R1w | t1w], S21 = [sR21 | t21].  Returns 0 when the point is dropped.
extern "C" int orc_sim3_project(const float T1w[12], const float S21[12], const float cam[5], const float bounds[4],
                                float log_scale_factor, int n_levels, const float* scale_factors, float th, const float P[3],
                                float min_distance, float max_distance, float* u_out, float* v_out, float* radius,
                                int* predicted_level) {
    float c1[3], c2[3];
    for (int k = 0; k < 3; k++) {  // p3Dc1 = R1w * p3Dw + t1w (:1196)
        float acc = T1w[4 * k] * P[0];
        acc = acc + T1w[4 * k + 1] * P[1];
        acc = acc + T1w[4 * k + 2] * P[2];
        c1[k] = acc + T1w[4 * k + 3];
    }
    for (int k = 0; k < 3; k++) {  // p3Dc2 = sR21 * p3Dc1 + t21 (:1197)
        float acc = S21[4 * k] * c1[0];
        acc = acc + S21[4 * k + 1] * c1[1];
        acc = acc + S21[4 * k + 2] * c1[2];
        c2[k] = acc + S21[4 * k + 3];
    }
    if (c2[2] < 0.0) return 0;  // :1200
    const float invz = (float)(1.0 / (double)c2[2]);
    const float x = c2[0] * invz;
    const float y = c2[1] * invz;
    const float u = cam[0] * x + cam[2];
    const float v = cam[1] * y + cam[3];
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) return 0;  // pKF2->IsInImage
    double n2 = 0.0;
    for (int k = 0; k < 3; k++) n2 += (double)c2[k] * (double)c2[k];
    const float dist3D = (float)std::sqrt(n2);  // cv::norm(p3Dc2) (:1216)
    const float maxDistance = 1.2f * max_distance;
    const float minDistance = 0.8f * min_distance;
    if (dist3D < minDistance || dist3D > maxDistance) return 0;  // :1219
    const int nPredictedLevel = predict_scale(max_distance, dist3D, log_scale_factor, n_levels);
    *radius = th * scale_factors[nPredictedLevel];  // :1226
    *u_out = u;
    *v_out = v;
    *predicted_level = nPredictedLevel;
    return 1;
}
