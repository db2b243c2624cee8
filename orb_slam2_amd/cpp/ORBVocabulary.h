// Drop-in for include/ORBVocabulary.h of the reference (a typedef of
// DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>): the
// members its callers use — loadFromTextFile, transform(features, BowVector&,
// FeatureVector&, levelsup), score, size, empty — over the vocabulary tree that
// liborbhip keeps on the GPU (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:241,
// 145-147, 160-161).
#ifndef ORBVOCABULARY_H
#define ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>

#include "cv_compat.h"

#ifdef ORB_SLAM2_FULL_TREE
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#else
namespace DBoW2 {  // Thirdparty/DBoW2/DBoW2/BowVector.h:17-21,54, FeatureVector.h:21
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2
#endif

struct orb_voc;

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0);
    ~ORBVocabulary();
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;

    bool loadFromTextFile(const std::string& filename);
    // one 1 x 32 CV_8U row per feature, as Converter::toDescriptorVector hands them over
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    // all rows of an N x 32 descriptor matrix (Frame::mDescriptors)
    void transform(const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const;
    unsigned int size() const;  // number of words
    bool empty() const { return size() == 0; }

    orb_voc* Handle() const { return mVoc; }

private:
    void transformPacked(const unsigned char* desc, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const;
    orb_voc* mVoc;
    int mDevice;
};

}  // namespace ORB_SLAM2
#endif
