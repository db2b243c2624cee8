// Drop-in Frame (see Frame.h).
#include "Frame.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/orb_hip.h"

namespace ORB_SLAM2 {

long unsigned int Frame::nNextId = 0;
bool Frame::mbInitialComputations = true;
float Frame::cx, Frame::cy, Frame::fx, Frame::fy, Frame::invfx, Frame::invfy;
float Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;
unsigned Frame::msGridQuirks = ORB_Q_GRID_EXCLUSIVE | ORB_Q_WINDOW_CIRCULAR;

Frame::Frame()
    : mpORBvocabulary(nullptr), mpORBextractorLeft(nullptr), mpORBextractorRight(nullptr), mTimeStamp(0), mbf(0), mb(0),
      mThDepth(0), N(0), mnId(0), mpReferenceKF(nullptr), mnScaleLevels(0), mfScaleFactor(0), mfLogScaleFactor(0),
      mnImgWidth(0), mnImgHeight(0) {}

// src/Frame.cc:11-51
Frame::Frame(const Frame& f)
    : mpORBvocabulary(f.mpORBvocabulary), mpORBextractorLeft(f.mpORBextractorLeft),
      mpORBextractorRight(f.mpORBextractorRight), mTimeStamp(f.mTimeStamp), mK(f.mK.clone()),
      mDistCoef(f.mDistCoef.clone()), mbf(f.mbf), mb(f.mb), mThDepth(f.mThDepth), N(f.N), mvKeys(f.mvKeys),
      mvKeysRight(f.mvKeysRight), mvKeysUn(f.mvKeysUn), mvuRight(f.mvuRight), mvDepth(f.mvDepth),
      mBowVec(f.mBowVec), mFeatVec(f.mFeatVec), mDescriptors(f.mDescriptors.clone()), mDescriptorsRight(f.mDescriptorsRight.clone()),
      mvpMapPoints(f.mvpMapPoints), mvbOutlier(f.mvbOutlier), mnId(f.mnId), mpReferenceKF(f.mpReferenceKF),
      mnScaleLevels(f.mnScaleLevels), mfScaleFactor(f.mfScaleFactor), mfLogScaleFactor(f.mfLogScaleFactor),
      mvScaleFactors(f.mvScaleFactors), mvInvScaleFactors(f.mvInvScaleFactors), mvLevelSigma2(f.mvLevelSigma2),
      mvInvLevelSigma2(f.mvInvLevelSigma2), mnImgWidth(f.mnImgWidth), mnImgHeight(f.mnImgHeight) {
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j] = f.mGrid[i][j];
    if (!f.mTcw.empty()) SetPose(f.mTcw);
}

// Body as KeyFrame::ComputeBoW of the reference (src/KeyFrame.cc:145-152): four levels up from the leaves.
void Frame::ComputeBoW() {
    if (mBowVec.empty() && mpORBvocabulary) mpORBvocabulary->transform(mDescriptors, mBowVec, mFeatVec, 4);
}

// Camera matrix and distortion coefficients in the plain form the C ABI takes.  mDistCoef holds
// k1, k2, p1, p2 and, when non-zero, k3 (src/Tracking.cc:76-88).
static bool CameraArrays(const cv::Mat& K, const cv::Mat& D, float k[4], float d[5]) {
    for (int i = 0; i < 5; i++) d[i] = 0.f;
    if (K.empty()) return false;
    k[0] = K.at<float>(0, 0);
    k[1] = K.at<float>(1, 1);
    k[2] = K.at<float>(0, 2);
    k[3] = K.at<float>(1, 2);
    const int nd = D.empty() ? 0 : std::min(5, D.rows * D.cols);
    for (int i = 0; i < nd; i++) d[i] = D.at<float>(i);
    return true;
}

Frame::Frame(const cv::Mat& imGray, const double& timeStamp, ORBextractor* extractor, ORBVocabulary* voc, cv::Mat& K,
             cv::Mat& distCoef, const float& bf, const float& thDepth)
    : mpORBvocabulary(voc), mpORBextractorLeft(extractor), mpORBextractorRight(static_cast<ORBextractor*>(NULL)),
      mTimeStamp(timeStamp), mK(K.clone()), mDistCoef(distCoef.clone()), mbf(bf), mThDepth(thDepth),
      mpReferenceKF(nullptr) {
    mnId = nNextId++;
    msGridQuirks = extractor->Quirks() & (ORB_Q_GRID_EXCLUSIVE | ORB_Q_WINDOW_CIRCULAR);  // same preset as the kernels
    mnScaleLevels = mpORBextractorLeft->GetLevels();
    mfScaleFactor = mpORBextractorLeft->GetScaleFactor();
    mfLogScaleFactor = (float)log(mfScaleFactor);
    mvScaleFactors = mpORBextractorLeft->GetScaleFactors();
    mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();
    mvLevelSigma2 = mpORBextractorLeft->GetScaleSigmaSquares();  // (the fork copies the inverse scale factors here)
    mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
    mnImgWidth = imGray.cols;
    mnImgHeight = imGray.rows;

    // ExtractORB(0, imGray) and UndistortKeyPoints() (src/Frame.cc:67-74): with a distorted camera both in one call
    // into the library (one wait for the device instead of two); otherwise as there
    {
        float k[4], d[5];
        if (CameraArrays(mK, mDistCoef, k, d) && d[0] != 0.0f) {
            mpORBextractorLeft->ExtractUndistorted(imGray, k, d, mvKeys, mvKeysUn, mDescriptors);
            N = (int)mvKeys.size();
            if (mvKeys.empty()) return;
        } else {
            ExtractORB(0, imGray);
            N = (int)mvKeys.size();
            if (mvKeys.empty()) return;
            UndistortKeyPoints();
        }
    }
    mvuRight = std::vector<float>(N, -1);
    mvDepth = std::vector<float>(N, -1);
    mvpMapPoints = std::vector<MapPoint*>(N, static_cast<MapPoint*>(NULL));
    mvbOutlier = std::vector<bool>(N, false);
    if (mbInitialComputations) {
        ComputeImageBounds(imGray);
        mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (mnMaxX - mnMinX);
        mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (mnMaxY - mnMinY);
        if (!K.empty()) {
            fx = K.at<float>(0, 0);
            fy = K.at<float>(1, 1);
            cx = K.at<float>(0, 2);
            cy = K.at<float>(1, 2);
            invfx = 1.0f / fx;
            invfy = 1.0f / fy;
        }
        mbInitialComputations = false;
    }
    mb = fx != 0.f ? mbf / fx : 0.f;
    {   // the matcher kernels lay their grid over the same (static) bounds, whichever extractor made this frame
        const float b[4] = {mnMinX, mnMaxX, mnMinY, mnMaxY};
        if (orb_set_image_bounds(mpORBextractorLeft->Context(), b) != ORB_OK)
            throw std::runtime_error(std::string("Frame: ") + orb_last_error());
    }
    AssignFeaturesToGrid();
}

void Frame::ExtractORB(int flag, const cv::Mat& im) {
    if (flag == 0)
        (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    else
        (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
}

void Frame::SetPose(cv::Mat Tcw) {
    mTcw = Tcw.clone();
    UpdatePoseMatrices();
}

// src/Frame.cc:146-166: mRcw = R, mRwc = R^T, mtcw = t, mOw = -R^T t of the 4x4 float pose.
void Frame::UpdatePoseMatrices() {
    if (mTcw.empty()) return;
    mRcw.create(3, 3, CV_32F);
    mRwc.create(3, 3, CV_32F);
    mtcw.create(3, 1, CV_32F);
    mOw.create(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            mRcw.at<float>(r, c) = mTcw.at<float>(r, c);
            mRwc.at<float>(c, r) = mTcw.at<float>(r, c);
        }
        mtcw.at<float>(r, 0) = mTcw.at<float>(r, 3);
    }
    for (int r = 0; r < 3; r++) {
        float s = 0.f;
        for (int k = 0; k < 3; k++) s += mRwc.at<float>(r, k) * mtcw.at<float>(k, 0);
        mOw.at<float>(r, 0) = -s;
    }
}

// Upstream ORB-SLAM2 Frame::UndistortKeyPoints (the fork only declares it, include/Frame.h:121).
void Frame::UndistortKeyPoints() {
    float k[4], d[5];
    if (!CameraArrays(mK, mDistCoef, k, d) || d[0] == 0.0f) {
        mvKeysUn = mvKeys;
        return;
    }
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint must be the 28-byte OpenCV layout");
    mvKeysUn.resize(N);
    if (orb_undistort_keypoints(mpORBextractorLeft->Context(), k, d, reinterpret_cast<const orb_keypoint*>(mvKeys.data()),
                                reinterpret_cast<orb_keypoint*>(mvKeysUn.data()), N) != ORB_OK)
        throw std::runtime_error(std::string("Frame::UndistortKeyPoints: ") + orb_last_error());
}

// Upstream ORB-SLAM2 Frame::ComputeImageBounds (include/Frame.h:123).
void Frame::ComputeImageBounds(const cv::Mat& imLeft) {
    float k[4], d[5], b[4] = {0.0f, (float)imLeft.cols, 0.0f, (float)imLeft.rows};
    if (CameraArrays(mK, mDistCoef, k, d) && d[0] != 0.0f) {
        if (orb_compute_image_bounds(mpORBextractorLeft->Context(), imLeft.cols, imLeft.rows, k, d, b) != ORB_OK)
            throw std::runtime_error(std::string("Frame::ComputeImageBounds: ") + orb_last_error());
    }
    mnMinX = b[0];
    mnMaxX = b[1];
    mnMinY = b[2];
    mnMaxY = b[3];
}

bool Frame::PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY) {
    posX = (int)round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
    posY = (int)round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
    return true;
}

void Frame::AssignFeaturesToGrid() {
    int nReserve = (int)(0.5f * N / (FRAME_GRID_COLS * FRAME_GRID_ROWS));
    for (unsigned int i = 0; i < FRAME_GRID_COLS; i++)
        for (unsigned int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j].reserve(nReserve);
    for (int i = 0; i < N; i++) {
        const cv::KeyPoint& kp = mvKeysUn[i];
        int nGridPosX, nGridPosY;
        if (PosInGrid(kp, nGridPosX, nGridPosY)) mGrid[nGridPosX][nGridPosY].push_back(i);
    }
}

std::vector<size_t> Frame::GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                             const int maxLevel) const {
    std::vector<size_t> vIndices;
    vIndices.reserve(N);
    const bool excl = (msGridQuirks & ORB_Q_GRID_EXCLUSIVE) != 0;
    const bool circ = (msGridQuirks & ORB_Q_WINDOW_CIRCULAR) != 0;
    const int nMinCellX = std::max(0, (int)floor((x - mnMinX - r) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + r) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - mnMinY - r) * mfGridElementHeightInv));
    if (excl ? nMinCellY > FRAME_GRID_ROWS : nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + r) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel >= 0) || (maxLevel >= 0);
    const int endX = excl ? nMaxCellX : nMaxCellX + 1, endY = excl ? nMaxCellY : nMaxCellY + 1;
    for (int ix = nMinCellX; ix < endX; ix++) {
        for (int iy = nMinCellY; iy < endY; iy++) {
            const std::vector<size_t>& vCell = mGrid[ix][iy];
            for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                const cv::KeyPoint& kpUn = mvKeysUn[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.pt.x - x;
                const float disty = kpUn.pt.y - y;
                if (circ ? (distx * distx + disty * disty < r * r) : (fabs(distx) < r && fabs(disty) < r))
                    vIndices.push_back(vCell[j]);
            }
        }
    }
    return vIndices;
}

}  // namespace ORB_SLAM2
