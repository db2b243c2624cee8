// CPU ORACLE — TEST INFRASTRUCTURE (see orb_oracle.hpp).  PARITY UNPINNED.
//
// Restatement of the OpenCV 4.2 primitives the reference calls but does not
// vendor.  Call sites in the reference: resize src/ORBextractor.cc:926-931,
// GaussianBlur :493, FAST :847-858, fastAtan2 :62, cvRound :26,72,75,409,918.
// Build with -ffp-contract=off: every float operation here is rounded on its own.
#include "orb_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace orbo {

// ---------------------------------------------------------------------------
// Rounding helpers: cvRound = round half to even (lrint in the default
// rounding mode), cvFloor / cvCeil = mathematical floor / ceil.
int cvRound(float v) { return (int)lrintf(v); }
int cvRound(double v) { return (int)lrint(v); }
int cvFloor(double v) {
    int i = (int)v;
    return i - (i > v);
}
int cvCeil(double v) {
    int i = (int)v;
    return i + (i < v);
}
static inline uint8_t satU8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline short satS16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// ---------------------------------------------------------------------------
// cv::resize(INTER_AREA), 8UC1, non-integer shrink: the generic area path.
// One table per axis lists, for every destination index, the source cells it
// covers and the covered fraction of each (float).  Rows are accumulated in
// table order: buf[dx] = sum_k src*alpha (float), then sum[dx] (+)= beta*buf.
struct AreaTap {
    int di, si;
    float alpha;
};

static std::vector<AreaTap> areaTable(int ssize, int dsize, double scale) {
    std::vector<AreaTap> tab;
    for (int d = 0; d < dsize; d++) {
        double f1 = d * scale;
        double f2 = f1 + scale;
        double cell = std::min(scale, ssize - f1);
        int s1 = cvCeil(f1), s2 = cvFloor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) tab.push_back({d, s1 - 1, (float)((s1 - f1) / cell)});
        for (int s = s1; s < s2; s++) tab.push_back({d, s, (float)(1.0 / cell)});
        if (f2 - s2 > 1e-3) tab.push_back({d, s2, (float)(std::min(std::min(f2 - s2, 1.), cell) / cell)});
    }
    return tab;
}

void resizeArea(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const double scale_x = 1. / ((double)dw / sw);
    const double scale_y = 1. / ((double)dh / sh);
    assert(scale_x >= 1 && scale_y >= 1);
    std::vector<AreaTap> xtab = areaTable(sw, dw, scale_x);
    std::vector<AreaTap> ytab = areaTable(sh, dh, scale_y);
    std::vector<float> buf(dw), sum(dw, 0.f);
    int prev_dy = ytab.empty() ? 0 : ytab[0].di;
    for (size_t j = 0; j < ytab.size(); j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di;
        const uint8_t* S = src + (size_t)ytab[j].si * sstride;
        std::fill(buf.begin(), buf.end(), 0.f);
        for (const AreaTap& t : xtab) buf[t.di] += S[t.si] * t.alpha;
        if (dy != prev_dy) {
            uint8_t* D = dst + (size_t)prev_dy * dstride;
            for (int dx = 0; dx < dw; dx++) {
                D[dx] = satU8(cvRound(sum[dx]));
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
        }
    }
    uint8_t* D = dst + (size_t)prev_dy * dstride;
    for (int dx = 0; dx < dw; dx++) D[dx] = satU8(cvRound(sum[dx]));
}

// ---------------------------------------------------------------------------
// cv::resize(INTER_LINEAR), 8UC1: 11-bit fixed-point weights per axis, 32-bit
// horizontal sums, vertical pass ((b0*(H0>>4))>>16 + (b1*(H1>>4))>>16 + 2) >> 2.
void resizeLinear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const double scale_x = 1. / ((double)dw / sw);
    const double scale_y = 1. / ((double)dh / sh);
    std::vector<int> xofs(dw);
    std::vector<short> xa(2 * (size_t)dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) fx = 0, sx = 0;
        if (sx >= sw - 1) fx = 0, sx = sw - 1;
        xofs[dx] = sx;
        xa[2 * dx] = satS16(cvRound((1.f - fx) * 2048));
        xa[2 * dx + 1] = satS16(cvRound(fx * 2048));
    }
    std::vector<int> H0(dw), H1(dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        const int r0 = std::min(std::max(sy, 0), sh - 1);
        const int r1 = std::min(std::max(sy + 1, 0), sh - 1);
        const short b0 = satS16(cvRound((1.f - fy) * 2048));
        const short b1 = satS16(cvRound(fy * 2048));
        const uint8_t* S0 = src + (size_t)r0 * sstride;
        const uint8_t* S1 = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            H0[dx] = S0[sx] * xa[2 * dx] + S0[sx1] * xa[2 * dx + 1];
            H1[dx] = S1[sx] * xa[2 * dx] + S1[sx1] * xa[2 * dx + 1];
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (H0[dx] >> 4)) >> 16) + ((b1 * (H1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------
// cv::GaussianBlur(8UC1, Size(7,7), 2, 2, BORDER_REFLECT_101) on an unpadded
// image: 8.8 fixed-point separable filter, exact integers,
// dst = (sum_j k_j * (sum_i k_i * p) + 2^15) >> 16.
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

void gaussianBlur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, bool legacy) {
    static const int kNew[7] = {18, 34, 48, 56, 48, 34, 18};
    static const int kOld[7] = {18, 34, 49, 55, 49, 34, 18};
    const int* k = legacy ? kOld : kNew;
    std::vector<uint32_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            uint32_t a = 0;
            for (int i = 0; i < 7; i++) a += (uint32_t)k[i] * S[reflect101(x + i - 3, w)];
            hbuf[(size_t)y * w + x] = a;
        }
    }
    std::vector<uint8_t> out((size_t)w * h);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint32_t a = 0;
            for (int j = 0; j < 7; j++) a += (uint32_t)k[j] * hbuf[(size_t)reflect101(y + j - 3, h) * w + x];
            uint32_t v = (a + 32768u) >> 16;
            out[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstride, out.data() + (size_t)y * w, w);
}

// ---------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmaxSuppression), TYPE_9_16.
// Ring of 16 at radius 3; index k and k+16 name the same pixel (25 entries so
// that every arc of 9 is a contiguous index range).
static const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16>: largest threshold at which the pixel is still a corner,
// never less than (threshold - 1).
int fastCornerScore(const uint8_t* p, int stride, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[kRingDy[k & 15] * stride + kRingDx[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int i = 4; i <= 8; i++) a = std::min(a, d[k + i]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        for (int i = 3; i <= 5; i++) b = std::max(b, d[k + i]);
        if (b >= b0) continue;
        for (int i = 6; i <= 8; i++) b = std::max(b, d[k + i]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// Corner test: at least 9 contiguous ring pixels all brighter than v+t or all
// darker than v-t.
static bool fastIsCorner(const uint8_t* p, int stride, int t) {
    const int v = p[0];
    int cls[25];  // 1 = darker than v-t, 2 = brighter than v+t
    for (int k = 0; k < 16; k++) {
        const int r = p[kRingDy[k] * stride + kRingDx[k]];
        cls[k] = r < v - t ? 1 : (r > v + t ? 2 : 0);
    }
    for (int k = 16; k < 25; k++) cls[k] = cls[k - 16];
    for (int want = 1; want <= 2; want++) {
        int run = 0;
        for (int k = 0; k < 25; k++) {
            if (cls[k] == want) {
                if (++run >= 9) return true;
            } else {
                run = 0;
            }
        }
    }
    return false;
}

void fastDetect(const uint8_t* img, int stride, int w, int h, int threshold, bool nms, std::vector<KeyPoint>& out) {
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return;
    // Three rolling rows of scores (0 = not a corner) and of corner columns.
    std::vector<uint8_t> sbuf((size_t)3 * w, 0);
    std::vector<int> cbuf((size_t)3 * (w + 1), 0);
    uint8_t* buf[3] = {sbuf.data(), sbuf.data() + w, sbuf.data() + 2 * (size_t)w};
    int* cpbuf[3] = {cbuf.data() + 1, cbuf.data() + (w + 1) + 1, cbuf.data() + 2 * (size_t)(w + 1) + 1};
    for (int i = 3; i < h - 2; i++) {
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            const uint8_t* rowp = img + (size_t)i * stride;
            for (int j = 3; j < w - 3; j++) {
                if (fastIsCorner(rowp + j, stride, threshold)) {
                    cornerpos[ncorners++] = j;
                    if (nms) curr[j] = (uint8_t)fastCornerScore(rowp + j, stride, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            const int j = cornerpos[k];
            const int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                         score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                out.push_back(KeyPoint{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1});
            }
        }
    }
}

// ---------------------------------------------------------------------------
// cv::fastAtan2(y, x): degree-valued odd polynomial in min/max ratio, float.
float fastAtan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------------------
// Portable sine/cosine for the descriptor rotation (src/ORBextractor.cc:69-70
// calls libm, whose last-bit behaviour differs between C libraries and between
// FMA and non-FMA builds of one library).  Specification: reduce by multiples
// of pi/2 in double (two-constant Cody-Waite), evaluate the fdlibm kernel
// polynomials in double with every operation rounded on its own, round the
// result once to float.  |x| < 2^20.
void sincosPortable(float xf, float* s, float* c) {
    const double x = (double)xf;
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double pio2_1t = 6.07710050650619224932e-11;  // pi/2 - pio2_1
    const double big = 6755399441055744.0;              // 1.5 * 2^52
    double fn = x * invpio2 + big;
    fn = fn - big;  // nearest integer, ties to even
    const int n = (int)fn;
    double r = x - fn * pio2_1;
    r = r - fn * pio2_1t;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6;
    ps = ps * z + S5;
    ps = ps * z + S4;
    ps = ps * z + S3;
    ps = ps * z + S2;
    ps = ps * z + S1;
    const double sn = r + r * (z * ps);
    double pc = C6;
    pc = pc * z + C5;
    pc = pc * z + C4;
    pc = pc * z + C3;
    pc = pc * z + C2;
    pc = pc * z + C1;
    const double cs = (1.0 - 0.5 * z) + z * (z * pc);
    double so, co;
    switch (n & 3) {
        case 0: so = sn; co = cs; break;
        case 1: so = cs; co = -sn; break;
        case 2: so = -sn; co = -cs; break;
        default: so = -cs; co = sn; break;
    }
    *s = (float)so;
    *c = (float)co;
}

}  // namespace orbo
