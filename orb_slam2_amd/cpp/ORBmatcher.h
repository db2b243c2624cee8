// Drop-in for include/ORBmatcher.h of the reference.  The searches that the
// fork defines (SearchForInitialization, SearchByProjection(Frame&, const
// Frame&, th, bMono), DescriptorDistance) run on the GPU through liborbhip.
// The overloads the fork only declares (include/ORBmatcher.h:39,75,90,110,143,
// 160,171,184) or defines with an empty body stay declared here, unimplemented,
// exactly as callers of the fork see them.  SearchByBoW(KeyFrame*, Frame&, ...)
// (src/ORBmatcher.cc:138-268) runs on the GPU as well; outside the reference's
// source tree, where KeyFrame is not defined, the same search is offered with a
// Frame in the keyframe's place (it reads the members both classes share:
// map points, undistorted keypoints, descriptors, feature vector).
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#include "Frame.h"
#include "cv_compat.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);  // include/ORBmatcher.h:19

    // src/ORBmatcher.cc:460-476
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

    // src/ORBmatcher.cc:26-136
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10);

    // src/ORBmatcher.cc:279-408 (with the window query the fork lost at :332-335:
    // octave range [nLastOctave-1, nLastOctave+1] for the monocular case, [nLastOctave, ..) when
    // moving forward, [0, nLastOctave] when moving backward)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

    // src/ORBmatcher.cc:138-268
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(const Frame& KF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);

    // include/ORBmatcher.h:24-39 (tracking of the local map).  The reference declares it with its five steps and
    // holds no body; the loop is upstream ORB-SLAM2's and runs on the GPU (orb_match_local_map).
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);

    // Declared by the reference, defined nowhere in it: kept for source compatibility.  Their candidate
    // loops exist in liborbhip as array-level entry points (orb_match_local_map, orb_match_projected,
    // orb_match_best, orb_match_bow_keyframes, orb_match_triangulation in orb_hip.h); the pose algebra and the map surgery around
    // them need KeyFrame / MapPoint / Sim3 objects that live in the reference's source tree.
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound,
                           const float th, const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                           std::vector<MapPoint*>& vpMatched, int th);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                               std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*> vpMatches12, const float& s12,
                     const cv::Mat& R12, const cv::Mat& t12, const float th);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th,
             std::vector<MapPoint*>& vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:
    int SearchByBoWImpl(const std::vector<MapPoint*>& vpMapPointsKF, const std::vector<cv::KeyPoint>& vKeysUnKF,
                        const cv::Mat& descriptorsKF, const DBoW2::FeatureVector& vFeatVecKF, Frame& F,
                        std::vector<MapPoint*>& vpMapPointMatches);
    float RadiusByViewingCos(const float& viewCos);  // include/ORBmatcher.h:197
    // include/ORBmatcher.h:198, src/ORBmatcher.cc:413-454.  The searches of this class do not call it (the rotation
    // histogram and its three maxima are part of the matching kernels); it is kept as the host helper it is in
    // the reference for subclasses and friends that use it.
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2
#endif
