// Stand-alone stand-in for the MapPoint accessors that ORBmatcher reads
// (GetWorldPos, GetDescriptor, Observations: src/ORBmatcher.cc:310,337,346;
// isBad: :175).  When
// the drop-in is compiled inside the reference tree, define
// ORB_SLAM2_FULL_TREE and the real MapPoint.h is used instead.
#pragma once
#include "cv_compat.h"

namespace ORB_SLAM2 {
class MapPoint {
public:
    MapPoint(float x, float y, float z, const unsigned char desc[32], int nObs) : mnObs(nObs), mbBad(false) {
        mWorldPos.create(3, 1, CV_32F);
        mWorldPos.at<float>(0) = x;
        mWorldPos.at<float>(1) = y;
        mWorldPos.at<float>(2) = z;
        mDescriptor.create(1, 32, CV_8U);
        memcpy(mDescriptor.data, desc, 32);
    }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return mnObs; }
    bool isBad() { return mbBad; }
    void SetBadFlag() { mbBad = true; }

    // variables used by the tracking of the local map (include/MapPoint.h:87-93); Frame::isInFrustum fills them
    float mTrackProjX = 0.f;
    float mTrackProjY = 0.f;
    float mTrackProjXR = 0.f;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0.f;
    bool mbTrackInView = false;

private:
    cv::Mat mWorldPos, mDescriptor;
    int mnObs;
    bool mbBad;
};
}  // namespace ORB_SLAM2
