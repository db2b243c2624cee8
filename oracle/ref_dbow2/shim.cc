// TEST INFRASTRUCTURE.  C entry points over the reference's own DBoW2::BowVector and DBoW2::FeatureVector,
// which are compiled from /root/reference/Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp as they lie
// (they include only the standard library).  The calls are made in the order
// TemplatedVocabulary<>::transform makes them (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1206);
// that template itself needs OpenCV and is not compiled.  Scoring types that normalise and their norm:
// ScoringObject.h:85-93 (L1, L2, chi-square, KL, Bhattacharyya: yes - L2 for L2_NORM, else L1; dot product: no).
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" void ref_bow_accumulate(const int* ids, const double* weights, const int* nids, int n, int scoring,
                                   int weighting, int* bowIds, double* bowVals, int* nBow, int* fvNodes,
                                   int* fvOffsets, int* fvIdx, int* nFvNodes) {
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    const bool must = scoring != DBoW2::DOT_PRODUCT;
    const DBoW2::LNorm norm = scoring == DBoW2::L2_NORM ? DBoW2::L2 : DBoW2::L1;
    if (weighting == DBoW2::TF || weighting == DBoW2::TF_IDF) {
        for (int i = 0; i < n; i++)
            if (weights[i] > 0) {
                v.addWeight((DBoW2::WordId)ids[i], weights[i]);
                fv.addFeature((DBoW2::NodeId)nids[i], (unsigned)i);
            }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); it++) it->second /= nd;
        }
    } else {
        for (int i = 0; i < n; i++)
            if (weights[i] > 0) {
                v.addIfNotExist((DBoW2::WordId)ids[i], weights[i]);
                fv.addFeature((DBoW2::NodeId)nids[i], (unsigned)i);
            }
    }
    if (must) v.normalize(norm);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) {
        bowIds[k] = (int)it->first;
        bowVals[k] = it->second;
    }
    *nBow = k;
    int a = 0, o = 0;
    fvOffsets[0] = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        fvNodes[a] = (int)it->first;
        for (size_t j = 0; j < it->second.size(); j++) fvIdx[o++] = (int)it->second[j];
        fvOffsets[++a] = o;
    }
    *nFvNodes = a;
}
