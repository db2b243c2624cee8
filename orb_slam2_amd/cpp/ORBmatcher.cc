// Drop-in ORBmatcher over the C ABI of liborbhip (see ORBmatcher.h).
#include "ORBmatcher.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/orb_hip.h"

namespace ORB_SLAM2 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + orb_last_error()); }

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return orb_hamming(a.ptr<unsigned char>(), b.ptr<unsigned char>());
}

// Descriptors of a frame as one contiguous n x 32 block (cv::Mat rows may be strided).
static const unsigned char* packed(const cv::Mat& d, std::vector<unsigned char>& tmp) {
    if (d.empty()) return nullptr;
    if (d.step == 32) return d.data;
    tmp.resize((size_t)d.rows * 32);
    for (int i = 0; i < d.rows; i++) memcpy(tmp.data() + (size_t)i * 32, d.ptr<unsigned char>(i), 32);
    return tmp.data();
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                        std::vector<int>& vnMatches12, int windowSize) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    if (n1 == 0) return 0;
    if ((int)vbPrevMatched.size() < n1) throw std::invalid_argument("SearchForInitialization: vbPrevMatched too short");
    orb_ctx* ctx = F2.mpORBextractorLeft ? F2.mpORBextractorLeft->Context() : F1.mpORBextractorLeft->Context();
    std::vector<unsigned char> t1, t2;
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
    static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f layout");
    int nmatches = 0;
    if (orb_match_window(ctx, (const orb_keypoint*)F1.mvKeysUn.data(), packed(F1.mDescriptors, t1), n1,
                         (const orb_keypoint*)F2.mvKeysUn.data(), packed(F2.mDescriptors, t2), n2, F2.mnImgWidth,
                         F2.mnImgHeight, (float*)vbPrevMatched.data(), vnMatches12.data(), windowSize, mfNNratio,
                         mbCheckOrientation ? 1 : 0, &nmatches))
        fail("orb_match_window");
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    // Pose algebra of src/ORBmatcher.cc:289-300 on 4x4 float matrices (sums accumulated in double and
    // rounded once, as cv::gemm does for CV_32F).
    auto T = [](const cv::Mat& m, int r, int c) { return (double)m.at<float>(r, c); };
    const cv::Mat& Tc = CurrentFrame.mTcw;
    const cv::Mat& Tl = LastFrame.mTcw;
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += -T(Tc, k, i) * T(Tc, k, 3);  // -Rcw^T * tcw
        twc[i] = (float)s;
    }
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += T(Tl, i, k) * (double)twc[k];
        tlc[i] = (float)(s + T(Tl, i, 3));
    }
    const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;

    std::vector<float> u, v, radius, qangle;
    std::vector<int> lmin, lmax, src;
    std::vector<unsigned char> qdesc, qobs;
    const size_t most = (size_t)std::max(LastFrame.N, 0);
    u.reserve(most), v.reserve(most), radius.reserve(most), qangle.reserve(most);
    lmin.reserve(most), lmax.reserve(most), src.reserve(most), qobs.reserve(most), qdesc.reserve(most * 32);
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        float xc3[3];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += T(Tc, r, k) * (double)x3Dw.at<float>(k);
            xc3[r] = (float)(s + T(Tc, r, 3));
        }
        const float xc = xc3[0], yc = xc3[1];
        const float invzc = (float)(1.0 / xc3[2]);
        if (invzc < 0) continue;
        const float uu = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float vv = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX) continue;
        if (vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        u.push_back(uu);
        v.push_back(vv);
        radius.push_back(th * CurrentFrame.mvScaleFactors[nLastOctave]);
        if (bForward) {
            lmin.push_back(nLastOctave);
            lmax.push_back(-1);
        } else if (bBackward) {
            lmin.push_back(0);
            lmax.push_back(nLastOctave);
        } else {
            lmin.push_back(nLastOctave - 1);
            lmax.push_back(nLastOctave + 1);
        }
        qangle.push_back(LastFrame.mvKeysUn[i].angle);
        cv::Mat dMP = pMP->GetDescriptor();
        qdesc.insert(qdesc.end(), dMP.ptr<unsigned char>(), dMP.ptr<unsigned char>() + 32);
        qobs.push_back(pMP->Observations() > 0 ? 1 : 0);
        src.push_back(i);
    }
    const int nq = (int)u.size(), n2 = CurrentFrame.N;
    if (nq == 0 || n2 == 0) return 0;
    std::vector<unsigned char> occ(n2, 0), t2;
    for (int i2 = 0; i2 < n2; i2++)
        if (CurrentFrame.mvpMapPoints[i2] && CurrentFrame.mvpMapPoints[i2]->Observations() > 0) occ[i2] = 1;
    std::vector<int> match2(n2, -1);
    int nmatches = 0;
    orb_ctx* ctx = CurrentFrame.mpORBextractorLeft->Context();
    if (orb_match_projected(ctx, u.data(), v.data(), radius.data(), lmin.data(), lmax.data(), qangle.data(),
                            qdesc.data(), qobs.data(), nq, (const orb_keypoint*)CurrentFrame.mvKeysUn.data(),
                            packed(CurrentFrame.mDescriptors, t2), n2, CurrentFrame.mnImgWidth,
                            CurrentFrame.mnImgHeight, occ.data(), match2.data(), TH_HIGH, mbCheckOrientation ? 1 : 0,
                            &nmatches))
        fail("orb_match_projected");
    for (int i2 = 0; i2 < n2; i2++) {
        if (match2[i2] >= 0)
            CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[src[match2[i2]]];
        else if (match2[i2] == -2)  // assigned, then rejected by the rotation histogram (:400)
            CurrentFrame.mvpMapPoints[i2] = static_cast<MapPoint*>(NULL);
    }
    return nmatches;
}

// include/ORBmatcher.h:197, declared only; upstream ORB-SLAM2: a smaller window when the point is seen
// almost along its mean viewing direction.
// src/ORBmatcher.cc:413-454: the three fullest bins, an earlier bin winning among equals (a bin moves up only
// past strictly smaller ones); the second and third are dropped when they hold less than a tenth of the first.
// As in the reference, ind1..ind3 are only written when a bin takes the place: callers pass them in as -1.
void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int top[3] = {0, 0, 0};
    int* const ind[3] = {&ind1, &ind2, &ind3};
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        int place = 3;
        while (place > 0 && s > top[place - 1]) place--;
        if (place == 3) continue;
        for (int k = 2; k > place; k--) {
            top[k] = top[k - 1];
            *ind[k] = *ind[k - 1];
        }
        top[place] = s;
        *ind[place] = i;
    }
    if ((float)top[1] < 0.1f * (float)top[0]) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)top[2] < 0.1f * (float)top[0]) {
        ind3 = -1;
    }
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) {
    if (viewCos > 0.998) return 2.5;
    return 4.0;
}

// include/ORBmatcher.h:24-39, declared only; upstream ORB-SLAM2's loop.  The points Frame::isInFrustum left in
// view carry their projection, predicted level and viewing cosine; the search itself is orb_match_local_map.
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    const bool bFactor = th != 1.0;
    std::vector<float> u, v, radius;
    std::vector<int> lmin, lmax, src;
    std::vector<unsigned char> qdesc, qobs;
    const size_t most = vpMapPoints.size();
    u.reserve(most), v.reserve(most), radius.reserve(most);
    lmin.reserve(most), lmax.reserve(most), src.reserve(most), qobs.reserve(most), qdesc.reserve(most * 32);
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;
        if (pMP->isBad()) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);  // the window depends on the viewing direction
        if (bFactor) r *= th;
        u.push_back(pMP->mTrackProjX);
        v.push_back(pMP->mTrackProjY);
        radius.push_back(r * F.mvScaleFactors[nPredictedLevel]);
        lmin.push_back(nPredictedLevel - 1);
        lmax.push_back(nPredictedLevel);
        cv::Mat dMP = pMP->GetDescriptor();
        qdesc.insert(qdesc.end(), dMP.ptr<unsigned char>(), dMP.ptr<unsigned char>() + 32);
        qobs.push_back(pMP->Observations() > 0 ? 1 : 0);
        src.push_back((int)iMP);
    }
    const int nq = (int)u.size(), n2 = F.N;
    if (nq == 0 || n2 == 0) return 0;
    std::vector<unsigned char> occ(n2, 0), t2;
    for (int i2 = 0; i2 < n2; i2++)
        if (F.mvpMapPoints[i2] && F.mvpMapPoints[i2]->Observations() > 0) occ[i2] = 1;
    std::vector<int> match2(n2, -1);
    int nmatches = 0;
    orb_ctx* ctx = F.mpORBextractorLeft->Context();
    if (orb_match_local_map(ctx, u.data(), v.data(), radius.data(), lmin.data(), lmax.data(), qdesc.data(), qobs.data(),
                            nq, (const orb_keypoint*)F.mvKeysUn.data(), packed(F.mDescriptors, t2), n2, F.mnImgWidth,
                            F.mnImgHeight, occ.data(), match2.data(), TH_HIGH, mfNNratio, &nmatches))
        fail("orb_match_local_map");
    for (int i2 = 0; i2 < n2; i2++)
        if (match2[i2] >= 0) F.mvpMapPoints[i2] = vpMapPoints[src[match2[i2]]];
    return nmatches;
}

// FeatureVector -> CSR arrays (node ids ascend in map order)
static void flatten(const DBoW2::FeatureVector& fv, std::vector<int>& node, std::vector<int>& off, std::vector<int>& idx) {
    node.clear();
    idx.clear();
    off.assign(1, 0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        node.push_back((int)it->first);
        idx.insert(idx.end(), it->second.begin(), it->second.end());
        off.push_back((int)idx.size());
    }
}

// src/ORBmatcher.cc:138-268; MapPoint::isBad (:175) is asked on the host, the search runs on the device.
int ORBmatcher::SearchByBoWImpl(const std::vector<MapPoint*>& vpMapPointsKF, const std::vector<cv::KeyPoint>& vKeysUnKF,
                                const cv::Mat& descriptorsKF, const DBoW2::FeatureVector& vFeatVecKF, Frame& F,
                                std::vector<MapPoint*>& vpMapPointMatches) {
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    const int nkf = (int)vKeysUnKF.size();
    if (nkf == 0 || F.N == 0) return 0;
    std::vector<unsigned char> valid(nkf, 0), tk, tf;
    for (int i = 0; i < nkf; i++) {
        MapPoint* pMP = i < (int)vpMapPointsKF.size() ? vpMapPointsKF[i] : static_cast<MapPoint*>(NULL);
        valid[i] = (pMP && !pMP->isBad()) ? 1 : 0;
    }
    std::vector<int> kn, ko, ki, fn, fo, fi, match(F.N, -1);
    flatten(vFeatVecKF, kn, ko, ki);
    flatten(F.mFeatVec, fn, fo, fi);
    int nmatches = 0;
    orb_ctx* ctx = F.mpORBextractorLeft->Context();
    if (orb_match_bow(ctx, (const orb_keypoint*)vKeysUnKF.data(), packed(descriptorsKF, tk), valid.data(), nkf, kn.data(),
                      ko.data(), ki.data(), (int)kn.size(), (const orb_keypoint*)F.mvKeys.data(), packed(F.mDescriptors, tf),
                      F.N, fn.data(), fo.data(), fi.data(), (int)fn.size(), mfNNratio, mbCheckOrientation ? 1 : 0,
                      match.data(), &nmatches))
        fail("orb_match_bow");
    for (int j = 0; j < F.N; j++)
        if (match[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[match[j]];
    return nmatches;
}

int ORBmatcher::SearchByBoW(const Frame& KF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    return SearchByBoWImpl(KF.mvpMapPoints, KF.mvKeysUn, KF.mDescriptors, KF.mFeatVec, F, vpMapPointMatches);
}

#ifdef ORB_SLAM2_FULL_TREE
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    return SearchByBoWImpl(pKF->GetMapPointMatches(), pKF->mvKeysUn, pKF->mDescriptors, pKF->mFeatVec, F, vpMapPointMatches);
}
#endif

}  // namespace ORB_SLAM2
