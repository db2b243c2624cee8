// Sanitizer driver (AddressSanitizer + UndefinedBehaviorSanitizer, host code only): runs the CPU oracle's
// extraction, matching, vocabulary and brute-force routines, and the parts of the drop-in C++ layer that need
// no GPU (ORBextractor construction and tables, DescriptorDistance, Frame pose members, cv_compat containers),
// on small synthetic inputs.  Any report from a sanitizer fails the run (-fno-sanitize-recover).
// Built and run by tests/test_sanitizers_cpu.py; `make -C oracle san` builds it by hand.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../oracle/orb_oracle.hpp"
#include "../../orb_slam2_amd/cpp/Frame.h"
#include "../../orb_slam2_amd/cpp/ORBextractor.h"
#include "../../orb_slam2_amd/cpp/ORBmatcher.h"

static uint64_t s_state = 0x0B5EED00ull;
static uint64_t rnd() {  // SplitMix64
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static orbo::Image scene(int w, int h) {  // rectangles on gray plus noise: corners, flat cells, borders
    orbo::Image im;
    im.w = w;
    im.h = h;
    im.px.assign((size_t)w * h, 128);
    for (int k = 0; k < w * h / 600; k++) {
        const int x0 = (int)(rnd() % w), y0 = (int)(rnd() % h), rw = 6 + (int)(rnd() % 43), rh = 6 + (int)(rnd() % 43);
        const uint8_t g = (uint8_t)(16 + rnd() % 225);
        for (int y = y0; y < std::min(h, y0 + rh); y++)
            for (int x = x0; x < std::min(w, x0 + rw); x++) im.px[(size_t)y * w + x] = g;
    }
    for (auto& p : im.px) p = (uint8_t)std::min(255, std::max(0, (int)p + (int)(rnd() % 8 + rnd() % 8) - 7));
    return im;
}

#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main() {
    using namespace orbo;
    // ---- oracle: extraction with both presets, odd sizes, an image with no corners ----------------------
    std::vector<KeyPoint> k1, k2;
    std::vector<uint8_t> d1, d2;
    for (unsigned quirks : {(unsigned)QUIRKS_FORK, (unsigned)QUIRKS_CANONICAL, (unsigned)(QUIRKS_FORK | Q_GAUSS_LEGACY | Q_LIBM_TRIG)}) {
        Extractor e(500, 1.2f, 8, 20, 7, quirks);
        const Image a = scene(421, 317);
        e.extract(a.px.data(), a.w, a.h, a.w, k1, d1);
        CHECK(k1.size() > 100 && d1.size() == k1.size() * 32);
        Image flat;
        flat.w = 300;
        flat.h = 260;
        flat.px.assign(300 * 260, 90);
        std::vector<KeyPoint> kf;
        std::vector<uint8_t> df;
        e.extract(flat.px.data(), flat.w, flat.h, flat.w, kf, df);
        CHECK(kf.empty());
    }
    Extractor e(1000, 1.2f, 8, 20, 7, QUIRKS_FORK);
    const Image a = scene(640, 480);
    Image b = a;  // second frame: shifted by (3, 2) with new noise
    for (int y = 479; y >= 2; y--)
        for (int x = 639; x >= 3; x--) b.px[(size_t)y * 640 + x] = a.px[(size_t)(y - 2) * 640 + x - 3];
    e.extract(a.px.data(), 640, 480, 640, k1, d1);
    e.extract(b.px.data(), 640, 480, 640, k2, d2);
    CHECK(k1.size() > 800 && k2.size() > 800);
    // primitives on degenerate shapes
    {
        std::vector<uint8_t> s(9, 200), o(9);
        gaussianBlur7(s.data(), 9, 1, 9, o.data(), 9, false);
        gaussianBlur7(s.data(), 1, 9, 1, o.data(), 1, true);
        std::vector<uint8_t> big(97 * 61, 7), small(80 * 50);
        resizeArea(big.data(), 97, 61, 97, small.data(), 80, 50, 80);
        resizeLinear(big.data(), 97, 61, 97, small.data(), 80, 50, 80);
        CHECK(small[0] == 7);
        CHECK(std::fabs(fastAtan2(1.f, 0.f) - 90.f) < 1e-3f);
    }
    // ---- oracle: matching ---------------------------------------------------------------------------
    FrameGrid F1, F2;
    F1.build(k1.data(), d1.data(), (int)k1.size(), 640, 480);
    F2.build(k2.data(), d2.data(), (int)k2.size(), 640, 480);
    std::vector<float> prev(k1.size() * 2);
    for (size_t i = 0; i < k1.size(); i++) {
        prev[2 * i] = k1[i].x;
        prev[2 * i + 1] = k1[i].y;
    }
    std::vector<int> m12(k1.size());
    for (unsigned quirks : {(unsigned)QUIRKS_FORK, (unsigned)QUIRKS_CANONICAL}) {
        std::vector<float> p = prev;
        const int n = searchForInitialization(F1, F2, p.data(), m12.data(), 100, 0.9f, true, quirks);
        CHECK(n > 20);
        std::vector<float> far(prev.size(), 1.0e6f);  // windows entirely outside the grid
        CHECK(searchForInitialization(F1, F2, far.data(), m12.data(), 100, 0.9f, true, quirks) == 0);
    }
    {
        const int nq = (int)k1.size();
        std::vector<float> u(nq), v(nq), r(nq), ang(nq);
        std::vector<int> l0(nq), l1(nq), match2(k2.size());
        std::vector<uint8_t> obs(nq, 1), occ(k2.size(), 0);
        for (int i = 0; i < nq; i++) {
            u[i] = k1[i].x + 3;
            v[i] = k1[i].y + 2;
            r[i] = 15.f * e.mvScaleFactor[k1[i].octave];
            ang[i] = k1[i].angle;
            l0[i] = k1[i].octave - 1;
            l1[i] = k1[i].octave + 1;
        }
        u[0] = -50.f;
        const int n = searchByProjection(u.data(), v.data(), r.data(), l0.data(), l1.data(), ang.data(), d1.data(), obs.data(), nq,
                                         F2, occ.data(), match2.data(), 100, true, QUIRKS_FORK);
        CHECK(n > 20);
        const int nl = searchLocalMap(u.data(), v.data(), r.data(), l0.data(), l1.data(), d1.data(), obs.data(), nq, F2, occ.data(),
                                      match2.data(), 100, 0.8f, QUIRKS_FORK);
        CHECK(nl > 20);
        std::vector<int> bi(nq), bd(nq);
        searchBest(u.data(), v.data(), r.data(), l0.data(), l1.data(), d1.data(), nq, F2, e.mvInvLevelSigma2.data(), 8, 5.99, 50,
                   QUIRKS_FORK, bi.data(), bd.data());
        int med = 0;
        CHECK(distinctiveDescriptor(d1.data(), 40, false, &med) >= 0);
        CHECK(distinctiveDescriptor(d1.data(), 0, false, &med) == -1);
    }
    // ---- oracle: brute force, ragged sizes ---------------------------------------------------------------
    for (int nd : {0, 1, 2, 3, 257, (int)k2.size()}) {
        std::vector<int32_t> idx(50);
        std::vector<uint16_t> b1(50), b2(50);
        bruteForceTop2(d1.data(), 50, d2.data(), nd, idx.data(), b1.data(), b2.data());
        CHECK(nd == 0 ? idx[0] == -1 : idx[0] >= 0);
    }
    // ---- oracle: vocabulary (k = 3, L = 3 built from the descriptors at hand) --------------------------------
    {
        std::vector<int> parent;
        std::vector<uint8_t> leaf, desc;
        std::vector<double> weight;
        int next = 1;
        std::vector<int> level = {0};  // node ids of the current level (0 = root)
        for (int l = 1; l <= 3; l++) {
            std::vector<int> nl;
            for (int p : level)
                for (int c = 0; c < 3; c++) {
                    parent.push_back(p);
                    leaf.push_back(l == 3);
                    const uint8_t* src = d1.data() + (size_t)((next * 7) % (int)k1.size()) * 32;
                    desc.insert(desc.end(), src, src + 32);
                    weight.push_back(l == 3 ? 1.0 + 0.1 * c : 0.0);
                    nl.push_back(next++);
                }
            level = nl;
        }
        Vocabulary voc;
        voc.fromArrays(3, 3, 0, 0, parent.data(), leaf.data(), desc.data(), weight.data(), (int)parent.size());
        std::map<int, double> bow1, bow2;
        std::map<int, std::vector<unsigned> > fv1, fv2;
        voc.transform(d1.data(), (int)k1.size(), 2, bow1, fv1);
        voc.transform(d2.data(), (int)k2.size(), 2, bow2, fv2);
        CHECK(!bow1.empty() && scoreL1(bow1, bow1) > 0.99);
        (void)scoreL1(bow1, bow2);
        std::vector<uint8_t> valid(k1.size(), 1);
        std::vector<int> mk(k2.size());
        (void)searchByBoW(k1.data(), d1.data(), valid.data(), fv1, k2.data(), d2.data(), (int)k2.size(), fv2, 0.7f, true, mk.data());
    }
    // ---- drop-in host layer without a GPU ---------------------------------------------------------------
    {
        using namespace ORB_SLAM2;
        ORBextractor ext(1000, 1.2f, 8, 20, 7);  // tables from host arithmetic alone
        const std::vector<float> sf = ext.GetScaleFactors(), isf = ext.GetInverseScaleFactors();
        CHECK(ext.GetLevels() == 8 && sf.size() == 8 && std::fabs(sf[1] - 1.2f) < 1e-6f && std::fabs(isf[7] * sf[7] - 1.f) < 1e-5f);
        for (int l = 0; l < 8; l++) CHECK(sf[l] == e.mvScaleFactor[l] && ext.GetScaleSigmaSquares()[l] == e.mvLevelSigma2[l]);
        cv::Mat da(1, 32, CV_8U), db(1, 32, CV_8U);
        memcpy(da.data, d1.data(), 32);
        memcpy(db.data, d2.data(), 32);
        CHECK(ORBmatcher::DescriptorDistance(da, db) == hammingPopcount(d1.data(), d2.data()));
        CHECK(ORBmatcher::DescriptorDistance(da, da) == 0);
        Frame F;
        cv::Mat T(4, 4, CV_32F);
        for (int i = 0; i < 16; i++) T.at<float>(i / 4, i % 4) = (i / 4 == i % 4) ? 1.f : 0.f;
        T.at<float>(0, 3) = 1.f;
        T.at<float>(1, 3) = -2.f;
        T.at<float>(2, 3) = 0.5f;
        F.SetPose(T);
        const cv::Mat Ow = F.GetCameraCenter(), Rwc = F.GetRotationInverse();
        CHECK(Ow.at<float>(0, 0) == -1.f && Ow.at<float>(1, 0) == 2.f && Ow.at<float>(2, 0) == -0.5f && Rwc.at<float>(1, 1) == 1.f);
        Frame G(F);  // copy constructor: pose matrices follow
        CHECK(G.GetCameraCenter().at<float>(1, 0) == 2.f);
        // the protected host helper the reference's subclasses may use, against the oracle's restatement
        struct Open : ORBmatcher {
            using ORBmatcher::ComputeThreeMaxima;
        } om;
        unsigned rs = 12345u;
        for (int rep = 0; rep < 2000; rep++) {
            std::vector<int> histo[30];
            int counts[30];
            for (int i = 0; i < 30; i++) {
                rs = rs * 1664525u + 1013904223u;
                counts[i] = (rep % 3 == 0) ? (int)((rs >> 24) % 4) : (int)((rs >> 20) % 50) * (int)((rs >> 12) % 2);
                histo[i].assign((size_t)counts[i], 0);
            }
            int a1 = -1, a2 = -1, a3 = -1, b1 = -1, b2 = -1, b3 = -1;
            om.ComputeThreeMaxima(histo, 30, a1, a2, a3);
            computeThreeMaxima(counts, 30, b1, b2, b3);
            CHECK(a1 == b1 && a2 == b2 && a3 == b3);
        }
    }
    printf("sanitizer driver: ok (%zu + %zu keypoints)\n", k1.size(), k2.size());
    return 0;
}
