// CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar restatement of the reference's ORB front end (ORBextractor, the
// Frame grid query and the ORBmatcher Hamming routines) used only as the
// checker in tests/, in __graft_entry__.smoke() and as bench.py's timed CPU
// baseline.  Nothing in the product library links or loads it.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
// for this path, its three hot-path sources do not compile as written, and
// their pixel arithmetic lives in OpenCV 4.2, which is neither vendored nor
// installed.  The OpenCV primitives (resize, GaussianBlur, FAST, fastAtan2,
// cvRound) are therefore restated here from their published algorithms and
// anchored on the reference's call sites (cited per function).  The only
// vendored cross-check is DBoW2's FORB::distance for the Hamming distance.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace orbo {

// Behaviour switches.  Each bit selects the text of the fork where it differs
// (compile-clean, deterministic) from upstream ORB-SLAM2.
enum : unsigned {
    Q_RESIZE_AREA     = 1u << 0,  // src/ORBextractor.cc:931 INTER_AREA (upstream INTER_LINEAR)
    Q_UMAX_FORK       = 1u << 1,  // src/ORBextractor.cc:431 compares umax[v] (upstream umax[v0])
    Q_DIVIDE_N3       = 1u << 2,  // src/ORBextractor.cc:567-568 stray ';' : n3 always bNoMore, n4 never
    Q_GRID_EXCLUSIVE  = 1u << 3,  // src/Frame.cc:101,110,112 '<' loop bounds and '>' row test
    Q_WINDOW_CIRCULAR = 1u << 4,  // src/Frame.cc:133 dx^2+dy^2 < r^2 (upstream |dx|<r && |dy|<r)
    Q_GAUSS_LEGACY    = 1u << 5,  // OpenCV <= 4.1 8-bit Gaussian taps {18,34,49,55,49,34,18}
    Q_LIBM_TRIG       = 1u << 6,  // oracle only: cosf/sinf from libm instead of the portable routine
    QUIRKS_FORK       = Q_RESIZE_AREA | Q_UMAX_FORK | Q_DIVIDE_N3 | Q_GRID_EXCLUSIVE | Q_WINDOW_CIRCULAR,
    QUIRKS_CANONICAL  = 0u,
};

// Same layout as cv::KeyPoint (28 bytes).
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // row-major, stride == w
    uint8_t* row(int y) { return px.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

// ---- rounding helpers (OpenCV semantics) ----------------------------------
int cvRound(float v);
int cvRound(double v);
int cvFloor(double v);
int cvCeil(double v);

// ---- OpenCV 4.2 primitives restated ----------------------------------------
void resizeArea(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void resizeLinear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void gaussianBlur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, bool legacy);
void fastDetect(const uint8_t* img, int stride, int w, int h, int threshold, bool nms, std::vector<KeyPoint>& out);
int fastCornerScore(const uint8_t* p, int stride, int threshold);
float fastAtan2(float y, float x);
void sincosPortable(float x, float* s, float* c);

// ---- ORBextractor -----------------------------------------------------------
class Extractor {
public:
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, unsigned quirks);

    // operator(): src/ORBextractor.cc:448-509 with the truncated tail restored.
    void extract(const uint8_t* img, int w, int h, int stride, std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

    void computePyramid(const uint8_t* img, int w, int h, int stride);
    void computeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& all);
    std::vector<KeyPoint> distributeOctTree(const std::vector<KeyPoint>& keys, int minX, int maxX, int minY, int maxY, int N) const;

    int nfeatures;
    double scaleFactor;  // double member, as include/ORBextractor.h:88
    int nlevels, iniThFAST, minThFAST;
    unsigned quirks;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Image> pyramid;                 // level images (the ROI part of mvImagePyramid)
    // per-stage dumps of the last extract() for stage-wise parity tests
    std::vector<Image> blurred;                 // GaussianBlur output per level (empty if level had no keypoints)
    std::vector<std::vector<KeyPoint>> fastKeys;   // vToDistributeKeys per level
    std::vector<std::vector<KeyPoint>> levelKeys;  // per level after quadtree + orientation (level coordinates)
};

float icAngle(const Image& im, float px, float py, const std::vector<int>& umax);
void orbDescriptor(const Image& blurred, float px, float py, float angleDeg, bool libmTrig, uint8_t* desc);

// ---- Frame grid + ORBmatcher --------------------------------------------------
int hammingBitHack(const uint8_t* a, const uint8_t* b);   // src/ORBmatcher.cc:460-476 literal
int hammingPopcount(const uint8_t* a, const uint8_t* b);  // 4 x popcount64

struct FrameGrid {  // the part of Frame the matcher reads
    int N = 0;
    float minX = 0, maxX = 0, minY = 0, maxY = 0, wInv = 0, hInv = 0;
    const KeyPoint* keys = nullptr;
    const uint8_t* desc = nullptr;
    std::vector<std::vector<int>> cells;  // [ix*48+iy]
    // bounds: mnMinX, mnMaxX, mnMinY, mnMaxY (static members of Frame, set from the first frame);
    // nullptr = the image rectangle (no lens distortion)
    void build(const KeyPoint* k, const uint8_t* d, int n, int imgW, int imgH, const float* bounds = nullptr);
    std::vector<int> featuresInArea(float x, float y, float r, int minLevel, int maxLevel, unsigned quirks) const;
};

// cv::undistortPoints(src, dst, K, distCoef, noArray(), K) as Frame::UndistortKeyPoints and
// Frame::ComputeImageBounds call it (declared at include/Frame.h:121-123, bodies as in upstream
// ORB-SLAM2): five fixed iterations of the inverse radial-tangential model in double, result
// re-projected with K and stored as float.  K = fx, fy, cx, cy; dist = k1, k2, p1, p2, k3.
void undistortPoints(const float* xyIn, float* xyOut, int n, const float K[4], const float dist[5]);
// Frame::UndistortKeyPoints: positions replaced, everything else copied; k1 == 0 copies the keypoints.
void undistortKeyPoints(const KeyPoint* in, KeyPoint* out, int n, const float K[4], const float dist[5]);
// Frame::ComputeImageBounds: bounds = mnMinX, mnMaxX, mnMinY, mnMaxY.
void computeImageBounds(int cols, int rows, const float K[4], const float dist[5], float bounds[4]);

void computeThreeMaxima(const int* counts, int L, int& i1, int& i2, int& i3);

int searchForInitialization(const FrameGrid& F1, const FrameGrid& F2, float* prevXY, int* matches12,
                            int windowSize, float nnratio, bool checkOri, unsigned quirks);

// Kernel-level form of SearchByProjection(Frame&,const Frame&,...): the projection is the caller's.
int searchByProjection(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                       const float* qangle, const uint8_t* qdesc, const uint8_t* qobs, int nq, const FrameGrid& F2,
                       const uint8_t* occupied2, int* match2 /*n2, -1 or query idx*/, int thHigh, bool checkOri,
                       unsigned quirks);

// Kernel-level forms of overloads the fork only declares (include/ORBmatcher.h:39,146-184); loops of upstream ORB-SLAM2.
int searchLocalMap(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                   const uint8_t* qdesc, const uint8_t* qobs, int nq, const FrameGrid& F, const uint8_t* occupied2,
                   int* match2, int thHigh, float nnratio, unsigned quirks);
void searchBest(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                const uint8_t* qdesc, int nq, const FrameGrid& F, const float* invLevelSigma2, int nlevels,
                double chi2Max, int th, unsigned quirks, int* bestIdxOut, int* bestDistOut);

// MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:186-214: index of the descriptor with the least median
// distance to the others (first on ties); *median receives that median.  selfDistance reproduces the fork's
// DescriptorDistance(d[i], d[i]) at :195 (every distance 0).  Returns -1 for n == 0.
int distinctiveDescriptor(const uint8_t* desc, int n, bool selfDistance, int* median);

// ---- DBoW2 vocabulary tree and SearchByBoW (orb_oracle_bow.cc) -------------------------
struct VocNode {
    int parent = 0;
    std::vector<int> children;  // in file order
    uint8_t desc[32] = {0};
    double weight = 0;
    int wordId = 0;
};
struct Vocabulary {  // TemplatedVocabulary<FORB::TDescriptor, FORB>
    int k = 0, L = 0, scoring = 0, weighting = 0, nwords = 0;
    std::vector<VocNode> nodes;  // nodes[0] = root
    bool loadFromTextFile(const std::string& filename);
    // nodes 1..nnodes in file order: parent id, leaf flag, descriptor, weight
    void fromArrays(int k, int L, int scoring, int weighting, const int* parent, const uint8_t* leaf, const uint8_t* desc,
                    const double* weight, int nnodes);
    void transformOne(const uint8_t* feature, int levelsup, int* wordId, double* weight, int* nid) const;
    void transform(const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                   std::map<int, std::vector<unsigned> >& fv) const;
};
double scoreL1(const std::map<int, double>& v1, const std::map<int, double>& v2);
void bowAccumulate(const int* ids, const double* weights, const int* nids, int n, int scoring, int weighting,
                   std::map<int, double>& bow, std::map<int, std::vector<unsigned> >& fv);
int searchByBoW(const KeyPoint* kfKeysUn, const uint8_t* kfDesc, const uint8_t* kfValid,
                const std::map<int, std::vector<unsigned> >& kfFeat, const KeyPoint* fKeys, const uint8_t* fDesc, int nF,
                const std::map<int, std::vector<unsigned> >& fFeat, float nnratio, bool checkOri, int* matchKF);

int searchByBoWKeyFrames(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* valid1, int n1,
                         const std::map<int, std::vector<unsigned> >& feat1, const KeyPoint* keysUn2,
                         const uint8_t* desc2, const uint8_t* valid2, int n2,
                         const std::map<int, std::vector<unsigned> >& feat2, float nnratio, bool checkOri,
                         int* matches12);

int searchForTriangulation(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* free1, int n1,
                           const std::map<int, std::vector<unsigned> >& feat1, const KeyPoint* keysUn2,
                           const uint8_t* desc2, const uint8_t* free2, int n2,
                           const std::map<int, std::vector<unsigned> >& feat2, const float* F12, const float* epipole,
                           const float* scaleFactors, const float* levelSigma2, int nlevels, bool markMatched,
                           bool checkOri, int* vMatches12);

void bruteForceTop2(const uint8_t* q, int nq, const uint8_t* db, int64_t nd, int32_t* bestIdx, uint16_t* bestDist,
                    uint16_t* secondDist);

}  // namespace orbo
