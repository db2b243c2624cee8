// Drop-in for include/ORBextractor.h of the reference: same class name,
// namespace, constructor, operator() and getters.  The work is done by the
// HIP kernels of liborbhip through the C ABI (include/orb_hip.h); this class
// only converts between the OpenCV-style containers and plain buffers.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#include "cv_compat.h"

struct orb_ctx;

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // include/ORBextractor.h:36, src/ORBextractor.cc:369-436
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // include/ORBextractor.h:39, src/ORBextractor.cc:448-509.  The mask is ignored, as in the reference.
    // An empty image returns silently; a failure inside the GPU library throws std::runtime_error.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Level images of the last frame (include/ORBextractor.h:71): as in the reference each is the centre
    // view of a buffer 19 pixels larger on every side whose rim is the level reflected (BORDER_REFLECT_101,
    // src/ORBextractor.cc:919-943).  Nothing on the front-end path reads them, and copying eight
    // levels back from the device costs more than the extraction of a 640x480 frame, so the member
    // is filled on request: GetImagePyramid() downloads the levels of the last frame (once per
    // frame); code that reads the member directly calls SetKeepPyramid(true) first, after which
    // every operator() fills it as the reference does.  The views point into buffers this object owns.
    std::vector<cv::Mat> mvImagePyramid;

    // ---- additions (no counterpart in the reference) -------------------------------
    void SetKeepPyramid(bool keep) { mbKeepPyramid = keep; }
    const std::vector<cv::Mat>& GetImagePyramid();
    // operator() and the undistortion of its keypoints (the Frame constructor's ExtractORB + UndistortKeyPoints) as
    // one call into the library: K = fx, fy, cx, cy; dist = k1, k2, p1, p2, k3.
    void ExtractUndistorted(const cv::Mat& image, const float K[4], const float dist[5], std::vector<cv::KeyPoint>& keypoints,
                            std::vector<cv::KeyPoint>& keypointsUn, cv::Mat& descriptors);
    // Behaviour preset and device for extractors constructed afterwards
    // (ORB_QUIRKS_FORK by default; also read from ORBHIP_QUIRKS / ORBHIP_DEVICE).
    static void SetDefaults(unsigned quirks, int device);
    // for Frame and ORBmatcher, which share the device state (created on the first frame, sized for it)
    orb_ctx* Context() { return mpCtx ? mpCtx : EnsureContext(640, 480); }
    unsigned Quirks() const { return mQuirks; }
    orb_ctx* EnsureContext(int width, int height);

protected:
    int nfeatures;
    double scaleFactor;  // double member as in the reference
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

    orb_ctx* mpCtx = nullptr;
    int mCtxW = 0, mCtxH = 0;
    bool mbKeepPyramid = false;
    bool mbPyramidCurrent = false;  // mvImagePyramid holds the last frame's levels
    void DownloadPyramid();
    unsigned mQuirks;
    int mDevice;
    std::vector<unsigned char> mKpBuf, mKpUnBuf, mDescBuf;
    std::vector<std::vector<unsigned char> > mvPaddedLevels;  // the buffers mvImagePyramid's views lie in
};

}  // namespace ORB_SLAM2

#endif
