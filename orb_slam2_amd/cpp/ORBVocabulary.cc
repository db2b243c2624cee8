// Drop-in ORBVocabulary over the C ABI of liborbhip (see ORBVocabulary.h).
#include "ORBVocabulary.h"

#include <cstring>
#include <stdexcept>

#include "../../include/orb_hip.h"

namespace ORB_SLAM2 {

ORBVocabulary::ORBVocabulary(int device) : mVoc(nullptr), mDevice(device) {}

ORBVocabulary::~ORBVocabulary() { orb_voc_destroy(mVoc); }

bool ORBVocabulary::loadFromTextFile(const std::string& filename) {
    orb_voc_destroy(mVoc);
    mVoc = nullptr;
    return orb_voc_load_text(filename.c_str(), mDevice, &mVoc) == ORB_OK;  // false on any failure, as the reference
}

unsigned int ORBVocabulary::size() const {
    int info[6] = {0};
    if (!mVoc || orb_voc_info(mVoc, info)) return 0;
    return (unsigned int)info[5];
}

void ORBVocabulary::transformPacked(const unsigned char* desc, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv,
                                    int levelsup) const {
    v.clear();
    fv.clear();
    if (!mVoc || n == 0) return;
    std::vector<int> bowId(n), fvNode(n), fvOff(n + 1), fvIdx(n);
    std::vector<double> bowVal(n);
    int nBow = 0, nFv = 0;
    if (orb_voc_bow(mVoc, desc, n, levelsup, bowId.data(), bowVal.data(), &nBow, fvNode.data(), fvOff.data(), fvIdx.data(), &nFv))
        throw std::runtime_error(std::string("orb_voc_bow: ") + orb_last_error());
    for (int i = 0; i < nBow; i++) v.insert(v.end(), std::make_pair((DBoW2::WordId)bowId[i], (DBoW2::WordValue)bowVal[i]));
    for (int a = 0; a < nFv; a++) {
        std::vector<unsigned int>& dst = fv[(DBoW2::NodeId)fvNode[a]];
        dst.assign(fvIdx.begin() + fvOff[a], fvIdx.begin() + fvOff[a + 1]);
    }
}

void ORBVocabulary::transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv,
                              int levelsup) const {
    std::vector<unsigned char> packed(features.size() * 32);
    for (size_t i = 0; i < features.size(); i++) memcpy(packed.data() + i * 32, features[i].ptr<unsigned char>(0), 32);
    transformPacked(packed.data(), (int)features.size(), v, fv, levelsup);
}

void ORBVocabulary::transform(const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    std::vector<unsigned char> packed((size_t)descriptors.rows * 32);
    for (int i = 0; i < descriptors.rows; i++) memcpy(packed.data() + (size_t)i * 32, descriptors.ptr<unsigned char>(i), 32);
    transformPacked(packed.data(), descriptors.rows, v, fv, levelsup);
}

double ORBVocabulary::score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
    int info[6] = {0};
    if (mVoc && orb_voc_info(mVoc, info) == ORB_OK && info[2] != 0)
        throw std::runtime_error("ORBVocabulary::score: only L1_NORM scoring (the one ORBvoc.txt uses) is provided");
    std::vector<int> ia, ib;
    std::vector<double> va, vb;
    for (DBoW2::BowVector::const_iterator it = a.begin(); it != a.end(); ++it) {
        ia.push_back((int)it->first);
        va.push_back(it->second);
    }
    for (DBoW2::BowVector::const_iterator it = b.begin(); it != b.end(); ++it) {
        ib.push_back((int)it->first);
        vb.push_back(it->second);
    }
    return orb_bow_score_l1(ia.data(), va.data(), (int)ia.size(), ib.data(), vb.data(), (int)ib.size());
}

}  // namespace ORB_SLAM2
