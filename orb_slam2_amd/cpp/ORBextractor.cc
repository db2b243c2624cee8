// Drop-in ORBextractor over the C ABI of liborbhip (see ORBextractor.h).
#include "ORBextractor.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/orb_hip.h"

namespace ORB_SLAM2 {

static unsigned gQuirks = ORB_QUIRKS_FORK;
static int gDevice = 0;
static bool gEnvRead = false;

static void readEnv() {
    if (gEnvRead) return;
    gEnvRead = true;
    if (const char* q = getenv("ORBHIP_QUIRKS")) gQuirks = (unsigned)strtoul(q, nullptr, 0);
    if (const char* d = getenv("ORBHIP_DEVICE")) gDevice = atoi(d);
}

void ORBextractor::SetDefaults(unsigned quirks, int device) {
    gEnvRead = true;
    gQuirks = quirks;
    gDevice = device;
}

static void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + orb_last_error()); }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    readEnv();
    mQuirks = gQuirks;
    mDevice = gDevice;
    // The tables depend on the constructor arguments alone (host arithmetic); the device context is
    // created by the first frame, sized for that frame.
    orb_params p;
    memset(&p, 0, sizeof(p));
    p.nfeatures = nfeatures;
    p.scale_factor = (float)scaleFactor;
    p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST;
    p.min_th_fast = minThFAST;
    p.quirks = mQuirks;
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    if (orb_compute_tables(&p, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                           mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data()))
        fail("orb_compute_tables");
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() {
    if (mpCtx) orb_destroy(mpCtx);
}

orb_ctx* ORBextractor::EnsureContext(int width, int height) {
    if (mpCtx && width <= mCtxW && height <= mCtxH) return mpCtx;
    if (mpCtx) orb_destroy(mpCtx);
    mpCtx = nullptr;
    orb_params p;
    p.nfeatures = nfeatures;
    p.scale_factor = (float)scaleFactor;
    p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST;
    p.min_th_fast = minThFAST;
    p.max_width = width;
    p.max_height = height;
    p.max_batch = 1;
    p.device = mDevice;
    p.quirks = mQuirks;
    if (orb_create(&p, &mpCtx)) fail("orb_create");
    mCtxW = width;
    mCtxH = height;
    return mpCtx;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;  // src/ORBextractor.cc:451-452
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    EnsureContext(image.cols, image.rows);
    const int cap = orb_max_keypoints_per_frame(mpCtx);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
    mKpBuf.resize((size_t)cap * sizeof(orb_keypoint));
    mDescBuf.resize((size_t)cap * 32);
    int n = 0;
    if (orb_extract(mpCtx, image.data, image.cols, image.rows, (int)image.step, (orb_keypoint*)mKpBuf.data(),
                    mDescBuf.data(), cap, &n))
        fail("orb_extract");
    _keypoints.clear();
    _keypoints.resize(n);
    if (n) memcpy((void*)_keypoints.data(), mKpBuf.data(), (size_t)n * sizeof(orb_keypoint));
    if (n == 0) {
        _descriptors.release();  // :468-469
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) memcpy(d.ptr<unsigned char>(i), mDescBuf.data() + (size_t)i * 32, 32);
    }
    mbPyramidCurrent = false;
    if (mbKeepPyramid) DownloadPyramid();
}

void ORBextractor::ExtractUndistorted(const cv::Mat& image, const float K[4], const float dist[5],
                                      std::vector<cv::KeyPoint>& keypoints, std::vector<cv::KeyPoint>& keypointsUn,
                                      cv::Mat& descriptors) {
    keypoints.clear();
    keypointsUn.clear();
    if (image.empty()) return;
    assert(image.type() == CV_8UC1);
    EnsureContext(image.cols, image.rows);
    const int cap = orb_max_keypoints_per_frame(mpCtx);
    mKpBuf.resize((size_t)cap * sizeof(orb_keypoint));
    mKpUnBuf.resize((size_t)cap * sizeof(orb_keypoint));
    mDescBuf.resize((size_t)cap * 32);
    int n = 0;
    if (orb_extract_undistorted(mpCtx, image.data, image.cols, image.rows, (int)image.step, K, dist, (orb_keypoint*)mKpBuf.data(),
                                (orb_keypoint*)mKpUnBuf.data(), mDescBuf.data(), cap, &n))
        fail("orb_extract_undistorted");
    keypoints.resize(n);
    keypointsUn.resize(n);
    if (n) {
        memcpy((void*)keypoints.data(), mKpBuf.data(), (size_t)n * sizeof(orb_keypoint));
        memcpy((void*)keypointsUn.data(), mKpUnBuf.data(), (size_t)n * sizeof(orb_keypoint));
    }
    if (n == 0) {
        descriptors.release();
    } else {
        descriptors.create(n, 32, CV_8U);
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr<unsigned char>(i), mDescBuf.data() + (size_t)i * 32, 32);
    }
    mbPyramidCurrent = false;
    if (mbKeepPyramid) DownloadPyramid();
}

// src/ORBextractor.cc:913-949: every level is the centre of a buffer EDGE_THRESHOLD = 19 pixels larger on each side,
// the rim filled by copyMakeBorder(BORDER_REFLECT_101) from the level's own pixels (level 0 likewise: the library sees
// the caller's image through a pointer and a row stride, never what lies around it).  mvImagePyramid[level] is the
// view of the centre, as there; the buffers belong to the extractor.
void ORBextractor::DownloadPyramid() {
    const int E = 19;
    std::vector<unsigned char*> dst(nlevels);
    std::vector<int> step(nlevels), ws(nlevels), hs(nlevels);
    mvPaddedLevels.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0;
        if (orb_get_pyramid_level(mpCtx, 0, l, nullptr, 0, &w, &h)) fail("orb_get_pyramid_level");
        ws[l] = w;
        hs[l] = h;
        step[l] = w + 2 * E;
        mvPaddedLevels[l].resize((size_t)step[l] * (h + 2 * E));
        dst[l] = mvPaddedLevels[l].data() + (size_t)E * step[l] + E;
    }
    if (orb_get_pyramid(mpCtx, 0, dst.data(), step.data())) fail("orb_get_pyramid");  // one synchronisation for all levels
    for (int l = 0; l < nlevels; l++) {
        const int w = ws[l], h = hs[l], W = step[l];
        unsigned char* c = dst[l];
        if (w > E && h > E) {  // (reflection needs E pixels to mirror; every level the library accepts has them)
            for (int y = 0; y < h; y++) {
                unsigned char* r = c + (size_t)y * W;
                for (int k = 1; k <= E; k++) {
                    r[-k] = r[k];
                    r[w - 1 + k] = r[w - 1 - k];
                }
            }
            for (int k = 1; k <= E; k++) {
                memcpy(c - (size_t)k * W - E, c + (size_t)k * W - E, (size_t)W);
                memcpy(c + (size_t)(h - 1 + k) * W - E, c + (size_t)(h - 1 - k) * W - E, (size_t)W);
            }
        }
        mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, c, (size_t)W);
    }
    mbPyramidCurrent = true;
}

const std::vector<cv::Mat>& ORBextractor::GetImagePyramid() {
    if (!mpCtx) throw std::runtime_error("GetImagePyramid: no frame has been extracted yet");
    if (!mbPyramidCurrent) DownloadPyramid();
    return mvImagePyramid;
}

}  // namespace ORB_SLAM2
