// Drop-in for the part of include/Frame.h of the reference that the ORB front
// end touches: the monocular constructor, ExtractORB, the 64x48 bucket grid
// with GetFeaturesInArea, ComputeBoW, and the public members ORBmatcher reads.
// Pose members (SetPose, UpdatePoseMatrices, GetCameraCenter, GetRotationInverse and
// the matrices behind them) are implemented as in src/Frame.cc:80-84,146-166;
// isInFrustum is declared and, as in the fork, defined nowhere; stereo members
// are kept as data only.
#ifndef FRAME_H
#define FRAME_H

#include <cstddef>
#include <vector>

#include "ORBextractor.h"
#include "cv_compat.h"

#include "ORBVocabulary.h"
#ifdef ORB_SLAM2_FULL_TREE
#include "KeyFrame.h"
#include "MapPoint.h"
#else
#include "MapPointCompat.h"
namespace ORB_SLAM2 {
class KeyFrame;
}  // namespace ORB_SLAM2
#endif

namespace ORB_SLAM2 {

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

class Frame {
public:
    Frame();
    Frame(const Frame& frame);
    // Monocular constructor, include/Frame.h:31, src/Frame.cc:53-69 (with the tail the fork lost:
    // keypoint count, undistorted copy, map-point slots, image bounds, grid).
    Frame(const cv::Mat& imGray, const double& timeStamp, ORBextractor* extractor, ORBVocabulary* voc, cv::Mat& K,
          cv::Mat& distCoef, const float& bf, const float& thDepth);

    void ExtractORB(int flag, const cv::Mat& im);  // src/Frame.cc:71-77
    // include/Frame.h:36 (declared only in the fork; body as KeyFrame::ComputeBoW, src/KeyFrame.cc:145-152)
    void ComputeBoW();
    void SetPose(cv::Mat Tcw);                       // src/Frame.cc:80-84
    void UpdatePoseMatrices();                       // src/Frame.cc:146-166
    inline cv::Mat GetCameraCenter() { return mOw.clone(); }      // include/Frame.h:42-45
    inline cv::Mat GetRotationInverse() { return mRwc.clone(); }  // include/Frame.h:47-50
    bool isInFrustum(MapPoint* pMP, float viewingCosLimit);       // include/Frame.h:52 (no body in the fork either)
    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY);
    // src/Frame.cc:86-144.  Host-side accessor for callers that want the index list itself;
    // the Search* kernels run the same query on the device.
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1,
                                          const int maxLevel = -1) const;

public:
    ORBVocabulary* mpORBvocabulary;
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    double mTimeStamp;
    cv::Mat mK;
    static float fx, fy, cx, cy, invfx, invfy;
    cv::Mat mDistCoef;
    float mbf, mb, mThDepth;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::BowVector mBowVec;        // include/Frame.h:87
    DBoW2::FeatureVector mFeatVec;   // include/Frame.h:88
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    static long unsigned int nNextId;
    long unsigned int mnId;
    KeyFrame* mpReferenceKF;
    int mnScaleLevels;
    float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static bool mbInitialComputations;
    static unsigned msGridQuirks;  // ORB_Q_GRID_EXCLUSIVE | ORB_Q_WINDOW_CIRCULAR as selected (fork by default)
    int mnImgWidth, mnImgHeight;

private:
    void UndistortKeyPoints();
    void ComputeImageBounds(const cv::Mat& imLeft);
    void AssignFeaturesToGrid();

    cv::Mat mRcw;  // rotation world -> camera, include/Frame.h:127-130
    cv::Mat mtcw;  // translation world -> camera
    cv::Mat mRwc;  // rotation camera -> world
    cv::Mat mOw;   // camera centre in world coordinates
};

}  // namespace ORB_SLAM2
#endif
