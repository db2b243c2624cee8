// CPU ORACLE — TEST INFRASTRUCTURE (see orb_oracle.hpp).  PARITY UNPINNED.
//
// Plain-C entry points over the oracle so that pytest (ctypes) and bench.py's
// cpu_baseline leg can drive it.  Never linked into the product library.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "orb_oracle.hpp"

using namespace orbo;

extern "C" {

void* oro_create(int nfeatures, float scale, int nlevels, int iniTh, int minTh, unsigned quirks) {
    return new Extractor(nfeatures, scale, nlevels, iniTh, minTh, quirks);
}
void oro_destroy(void* h) { delete (Extractor*)h; }

// returns the keypoint count (even when larger than cap; only cap entries are written)
int oro_extract(void* h, const uint8_t* img, int w, int hgt, int stride, KeyPoint* kps, uint8_t* desc, int cap) {
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    e->extract(img, w, hgt, stride, k, d);
    const int n = (int)k.size(), m = std::min(n, cap);
    if (kps && m > 0) memcpy(kps, k.data(), (size_t)m * sizeof(KeyPoint));
    if (desc && m > 0) memcpy(desc, d.data(), (size_t)m * 32);
    return n;
}

void oro_get_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* perLevel, int* umax) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->mvScaleFactor[i];
        if (invScale) invScale[i] = e->mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = e->mvLevelSigma2[i];
        if (invSigma2) invSigma2[i] = e->mvInvLevelSigma2[i];
        if (perLevel) perLevel[i] = e->mnFeaturesPerLevel[i];
    }
    if (umax)
        for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
}

int oro_level_size(void* h, int level, int* w, int* hgt) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->pyramid[level].w;
    *hgt = e->pyramid[level].h;
    return 0;
}
// which: 0 = pyramid level, 1 = blurred level.  Returns 0, or -1 when absent.
int oro_get_level(void* h, int which, int level, uint8_t* dst) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    const Image& im = which == 0 ? e->pyramid[level] : e->blurred[level];
    if (im.px.empty()) return -1;
    memcpy(dst, im.px.data(), im.px.size());
    return 0;
}
// which: 0 = keys fed to the quadtree (cell-major order), 1 = per-level result with angle (level coordinates)
int oro_get_level_keys(void* h, int which, int level, KeyPoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    const std::vector<KeyPoint>& v = which == 0 ? e->fastKeys[level] : e->levelKeys[level];
    const int m = std::min((int)v.size(), cap);
    if (out && m > 0) memcpy(out, v.data(), (size_t)m * sizeof(KeyPoint));
    return (int)v.size();
}

void oro_compute_pyramid(void* h, const uint8_t* img, int w, int hgt, int stride) {
    ((Extractor*)h)->computePyramid(img, w, hgt, stride);
}

int oro_distribute(void* h, const KeyPoint* keys, int n, int minX, int maxX, int minY, int maxY, int N, KeyPoint* out,
                   int cap) {
    std::vector<KeyPoint> in(keys, keys + n);
    std::vector<KeyPoint> r = ((Extractor*)h)->distributeOctTree(in, minX, maxX, minY, maxY, N);
    const int m = std::min((int)r.size(), cap);
    if (m > 0) memcpy(out, r.data(), (size_t)m * sizeof(KeyPoint));
    return (int)r.size();
}

// ---- primitives ---------------------------------------------------------------
void oro_resize(int area, const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    if (area)
        resizeArea(src, sw, sh, sstride, dst, dw, dh, dstride);
    else
        resizeLinear(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void oro_gaussian(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int legacy) {
    gaussianBlur7(src, w, h, sstride, dst, dstride, legacy != 0);
}
int oro_fast(const uint8_t* img, int stride, int w, int h, int threshold, int nms, KeyPoint* out, int cap) {
    std::vector<KeyPoint> v;
    fastDetect(img, stride, w, h, threshold, nms != 0, v);
    const int m = std::min((int)v.size(), cap);
    if (m > 0) memcpy(out, v.data(), (size_t)m * sizeof(KeyPoint));
    return (int)v.size();
}
int oro_fast_score(const uint8_t* p, int stride, int threshold) { return fastCornerScore(p, stride, threshold); }
float oro_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void oro_sincos(float x, float* s, float* c) { sincosPortable(x, s, c); }
void oro_sincos_libm(float x, float* s, float* c) {
    *s = sinf(x);
    *c = cosf(x);
}
// count inputs in [lo, hi) stepping by `stride` ulps whose portable result differs from libm (sin or cos)
long oro_sincos_mismatch(float lo, float hi, int strideUlps, long* total) {
    long bad = 0, tot = 0;
    uint32_t a, b;
    memcpy(&a, &lo, 4);
    memcpy(&b, &hi, 4);
    for (uint32_t u = a; u < b; u += (uint32_t)strideUlps) {
        float x, s, c;
        memcpy(&x, &u, 4);
        sincosPortable(x, &s, &c);
        if (s != sinf(x) || c != cosf(x)) bad++;
        tot++;
    }
    if (total) *total = tot;
    return bad;
}
float oro_ic_angle(const uint8_t* img, int w, int h, float x, float y, const int* umax) {
    Image im;
    im.w = w;
    im.h = h;
    im.px.assign(img, img + (size_t)w * h);
    std::vector<int> um(umax, umax + 16);
    return icAngle(im, x, y, um);
}
void oro_descriptor(const uint8_t* blurred, int w, int h, float x, float y, float angle, int libm, uint8_t* desc) {
    Image im;
    im.w = w;
    im.h = h;
    im.px.assign(blurred, blurred + (size_t)w * h);
    orbDescriptor(im, x, y, angle, libm != 0, desc);
}

// ---- matcher ---------------------------------------------------------------------
int oro_hamming_bithack(const uint8_t* a, const uint8_t* b) { return hammingBitHack(a, b); }
int oro_hamming_popcount(const uint8_t* a, const uint8_t* b) { return hammingPopcount(a, b); }
void oro_three_maxima(const int* counts, int L, int* ind) { computeThreeMaxima(counts, L, ind[0], ind[1], ind[2]); }

// Frame::mnMinX .. mnMaxY are static members set from the first frame; nullptr = image rectangle.
static float g_bounds[4];
static bool g_haveBounds = false;
void oro_set_image_bounds(const float* b) {
    g_haveBounds = b != nullptr;
    if (b) memcpy(g_bounds, b, sizeof(g_bounds));
}
static const float* bounds() { return g_haveBounds ? g_bounds : nullptr; }

void oro_undistort_keypoints(const KeyPoint* in, KeyPoint* out, int n, const float* K, const float* dist) {
    undistortKeyPoints(in, out, n, K, dist);
}
void oro_undistort_points(const float* in, float* out, int n, const float* K, const float* dist) {
    undistortPoints(in, out, n, K, dist);
}
void oro_compute_image_bounds(int cols, int rows, const float* K, const float* dist, float* b) {
    computeImageBounds(cols, rows, K, dist, b);
}

int oro_features_in_area(const KeyPoint* k, int n, int imgW, int imgH, float x, float y, float r, int minLevel,
                         int maxLevel, unsigned quirks, int* out, int cap) {
    FrameGrid g;
    g.build(k, nullptr, n, imgW, imgH, bounds());
    std::vector<int> v = g.featuresInArea(x, y, r, minLevel, maxLevel, quirks);
    const int m = std::min((int)v.size(), cap);
    if (m > 0) memcpy(out, v.data(), (size_t)m * sizeof(int));
    return (int)v.size();
}

int oro_search_for_initialization(const KeyPoint* k1, const uint8_t* d1, int n1, const KeyPoint* k2, const uint8_t* d2,
                                  int n2, int imgW, int imgH, float* prevXY, int* matches12, int window, float nnratio,
                                  int checkOri, unsigned quirks) {
    FrameGrid F1, F2;
    F1.build(k1, d1, n1, imgW, imgH, bounds());
    F2.build(k2, d2, n2, imgW, imgH, bounds());
    return searchForInitialization(F1, F2, prevXY, matches12, window, nnratio, checkOri != 0, quirks);
}

int oro_search_by_projection(const float* u, const float* v, const float* radius, const int* minLevel,
                             const int* maxLevel, const float* qangle, const uint8_t* qdesc, const uint8_t* qobs,
                             int nq, const KeyPoint* k2, const uint8_t* d2, int n2, int imgW, int imgH,
                             const uint8_t* occupied2, int* match2, int thHigh, int checkOri, unsigned quirks) {
    FrameGrid F2;
    F2.build(k2, d2, n2, imgW, imgH, bounds());
    return searchByProjection(u, v, radius, minLevel, maxLevel, qangle, qdesc, qobs, nq, F2, occupied2, match2, thHigh,
                              checkOri != 0, quirks);
}

int oro_search_local_map(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                         const uint8_t* qdesc, const uint8_t* qobs, int nq, const KeyPoint* k2, const uint8_t* d2, int n2,
                         int imgW, int imgH, const uint8_t* occupied2, int* match2, int thHigh, float nnratio,
                         unsigned quirks) {
    FrameGrid F2;
    F2.build(k2, d2, n2, imgW, imgH, bounds());
    return searchLocalMap(u, v, radius, minLevel, maxLevel, qdesc, qobs, nq, F2, occupied2, match2, thHigh, nnratio, quirks);
}

void oro_search_best(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                     const uint8_t* qdesc, int nq, const KeyPoint* k2, const uint8_t* d2, int n2, int imgW, int imgH,
                     const float* invLevelSigma2, int nlevels, double chi2Max, int th, unsigned quirks, int* bestIdx,
                     int* bestDist) {
    FrameGrid F2;
    F2.build(k2, d2, n2, imgW, imgH, bounds());
    searchBest(u, v, radius, minLevel, maxLevel, qdesc, nq, F2, invLevelSigma2, nlevels, chi2Max, th, quirks, bestIdx,
               bestDist);
}

void oro_bf_top2(const uint8_t* q, int nq, const uint8_t* db, long long nd, int32_t* idx, uint16_t* d1, uint16_t* d2) {
    bruteForceTop2(q, nq, db, nd, idx, d1, d2);
}

int oro_distinctive(const uint8_t* desc, int n, int selfDistance, int* median) {
    return distinctiveDescriptor(desc, n, selfDistance != 0, median);
}

// ---- vocabulary / bag of words ----------------------------------------------------------
void* oro_voc_from_arrays(int k, int L, int scoring, int weighting, const int* parent, const uint8_t* leaf,
                          const uint8_t* desc, const double* weight, int nnodes) {
    Vocabulary* v = new Vocabulary();
    v->fromArrays(k, L, scoring, weighting, parent, leaf, desc, weight, nnodes);
    return v;
}
void* oro_voc_load(const char* path) {
    Vocabulary* v = new Vocabulary();
    if (!v->loadFromTextFile(path)) {
        delete v;
        return nullptr;
    }
    return v;
}
void oro_voc_free(void* h) { delete (Vocabulary*)h; }
void oro_voc_info(void* h, int* out6) {
    const Vocabulary* v = (const Vocabulary*)h;
    out6[0] = v->k;
    out6[1] = v->L;
    out6[2] = v->scoring;
    out6[3] = v->weighting;
    out6[4] = (int)v->nodes.size() - 1;
    out6[5] = v->nwords;
}
void oro_voc_transform_features(void* h, const uint8_t* desc, int n, int levelsup, int* word, double* weight, int* node) {
    const Vocabulary* v = (const Vocabulary*)h;
    for (int i = 0; i < n; i++) v->transformOne(desc + (size_t)i * 32, levelsup, &word[i], &weight[i], &node[i]);
}
static std::map<int, std::vector<unsigned> > fv_from_csr(const int* nodes, const int* offsets, const int* idx, int nn) {
    std::map<int, std::vector<unsigned> > fv;
    for (int a = 0; a < nn; a++) fv[nodes[a]].assign(idx + offsets[a], idx + offsets[a + 1]);
    return fv;
}
// Bag-of-words vector (ids ascending, values) and feature vector (CSR: node ids ascending, offsets, feature
// indices) of n descriptors; the output arrays hold up to n (offsets n + 1) entries.
void oro_voc_bow(void* h, const uint8_t* desc, int n, int levelsup, int* bowIds, double* bowVals, int* nBow, int* fvNodes,
                 int* fvOffsets, int* fvIdx, int* nFvNodes) {
    const Vocabulary* v = (const Vocabulary*)h;
    std::map<int, double> bow;
    std::map<int, std::vector<unsigned> > fv;
    v->transform(desc, n, levelsup, bow, fv);
    int i = 0;
    for (auto& e : bow) {
        bowIds[i] = e.first;
        bowVals[i] = e.second;
        i++;
    }
    *nBow = i;
    int a = 0, o = 0;
    fvOffsets[0] = 0;
    for (auto& e : fv) {
        fvNodes[a] = e.first;
        for (unsigned f : e.second) fvIdx[o++] = (int)f;
        fvOffsets[++a] = o;
    }
    *nFvNodes = a;
}
// The same outputs from a stream of (word id, weight, node id) triples, one per feature, for a vocabulary of
// the given scoring (0 L1 .. 5 dot product) and weighting (0 TF-IDF, 1 TF, 2 IDF, 3 binary) types.
void oro_bow_accumulate(const int* ids, const double* weights, const int* nids, int n, int scoring, int weighting,
                        int* bowIds, double* bowVals, int* nBow, int* fvNodes, int* fvOffsets, int* fvIdx, int* nFvNodes) {
    std::map<int, double> bow;
    std::map<int, std::vector<unsigned> > fv;
    bowAccumulate(ids, weights, nids, n, scoring, weighting, bow, fv);
    int i = 0;
    for (auto& e : bow) {
        bowIds[i] = e.first;
        bowVals[i] = e.second;
        i++;
    }
    *nBow = i;
    int a = 0, o = 0;
    fvOffsets[0] = 0;
    for (auto& e : fv) {
        fvNodes[a] = e.first;
        for (unsigned f : e.second) fvIdx[o++] = (int)f;
        fvOffsets[++a] = o;
    }
    *nFvNodes = a;
}
double oro_score_l1(const int* ids1, const double* vals1, int n1, const int* ids2, const double* vals2, int n2) {
    std::map<int, double> a, b;
    for (int i = 0; i < n1; i++) a[ids1[i]] = vals1[i];
    for (int i = 0; i < n2; i++) b[ids2[i]] = vals2[i];
    return scoreL1(a, b);
}
int oro_search_by_bow(const KeyPoint* kfKeysUn, const uint8_t* kfDesc, const uint8_t* kfValid, const int* kfNodes,
                      const int* kfOffsets, const int* kfIdx, int kfNN, const KeyPoint* fKeys, const uint8_t* fDesc, int nF,
                      const int* fNodes, const int* fOffsets, const int* fIdx, int fNN, float nnratio, int checkOri,
                      int* matchKF) {
    return searchByBoW(kfKeysUn, kfDesc, kfValid, fv_from_csr(kfNodes, kfOffsets, kfIdx, kfNN), fKeys, fDesc, nF,
                       fv_from_csr(fNodes, fOffsets, fIdx, fNN), nnratio, checkOri != 0, matchKF);
}

int oro_search_by_bow_keyframes(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* valid1, int n1,
                                const int* nodes1, const int* offsets1, const int* idx1, int nn1, const KeyPoint* keysUn2,
                                const uint8_t* desc2, const uint8_t* valid2, int n2, const int* nodes2, const int* offsets2,
                                const int* idx2, int nn2, float nnratio, int checkOri, int* matches12) {
    return searchByBoWKeyFrames(keysUn1, desc1, valid1, n1, fv_from_csr(nodes1, offsets1, idx1, nn1), keysUn2, desc2,
                                valid2, n2, fv_from_csr(nodes2, offsets2, idx2, nn2), nnratio, checkOri != 0, matches12);
}

int oro_search_for_triangulation(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* free1, int n1,
                                 const int* nodes1, const int* offsets1, const int* idx1, int nn1, const KeyPoint* keysUn2,
                                 const uint8_t* desc2, const uint8_t* free2, int n2, const int* nodes2, const int* offsets2,
                                 const int* idx2, int nn2, const float* F12, const float* epipole,
                                 const float* scaleFactors, const float* levelSigma2, int nlevels, int markMatched,
                                 int checkOri, int* matches12) {
    return searchForTriangulation(keysUn1, desc1, free1, n1, fv_from_csr(nodes1, offsets1, idx1, nn1), keysUn2, desc2,
                                  free2, n2, fv_from_csr(nodes2, offsets2, idx2, nn2), F12, epipole, scaleFactors,
                                  levelSigma2, nlevels, markMatched != 0, checkOri != 0, matches12);
}

}  // extern "C"
