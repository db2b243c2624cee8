// CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Bag-of-words side of the matcher: the vocabulary tree of the vendored DBoW2
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h, BowVector.cpp,
// FeatureVector.cpp, ScoringObject.cpp) and ORBmatcher::SearchByBoW(KeyFrame*,
// Frame&, ...) (src/ORBmatcher.cc:138-268), restated with std::map / std::vector
// exactly as the reference keeps them.
//
// PARITY UNPINNED: the reference ships neither a vocabulary file nor any test
// vector for this part; the tests build small vocabularies of their own.
//
// Ruling on an unspecified case: TemplatedVocabulary::transform
// (TemplatedVocabulary.h:1240-1289) leaves *nid unwritten when the descent
// reaches a leaf above level L - levelsup (the caller's variable is
// uninitialised, :1162); here the leaf itself is reported.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "orb_oracle.hpp"

namespace orbo {

// TemplatedVocabulary.h:1368-1470.  Lines of the file: "k L scoring weighting", then one line per
// node (id = line order, from 1): parent, leaf flag, 32 descriptor bytes, weight.
bool Vocabulary::loadFromTextFile(const std::string& filename) {
    std::ifstream f(filename.c_str());
    if (!f.is_open()) return false;
    std::string s;
    getline(f, s);
    std::stringstream ss;
    ss << s;
    int n1 = -1, n2 = -1;
    k = L = -1;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return false;
    scoring = n1;
    weighting = n2;
    nodes.clear();
    nodes.resize(1);
    nwords = 0;
    while (!f.eof()) {
        std::string snode;
        getline(f, snode);
        if (snode.empty()) continue;  // (the reference would append a node for a trailing empty line; files end without one)
        std::stringstream ssnode;
        ssnode << snode;
        const int nid = (int)nodes.size();
        nodes.resize(nodes.size() + 1);
        int pid = 0, leaf = 0;
        ssnode >> pid >> leaf;
        if (pid < 0 || pid >= nid) return false;
        nodes[nid].parent = pid;
        nodes[pid].children.push_back(nid);
        for (int i = 0; i < 32; i++) {
            int v = 0;
            ssnode >> v;
            nodes[nid].desc[i] = (uint8_t)v;
        }
        ssnode >> nodes[nid].weight;
        if (leaf > 0) nodes[nid].wordId = nwords++;
    }
    return true;
}

void Vocabulary::fromArrays(int k_, int L_, int scoring_, int weighting_, const int* parent, const uint8_t* leaf,
                            const uint8_t* desc, const double* weight, int nnodes) {
    k = k_;
    L = L_;
    scoring = scoring_;
    weighting = weighting_;
    nodes.clear();
    nodes.resize((size_t)nnodes + 1);
    nwords = 0;
    for (int i = 1; i <= nnodes; i++) {
        nodes[i].parent = parent[i - 1];
        nodes[parent[i - 1]].children.push_back(i);
        memcpy(nodes[i].desc, desc + (size_t)(i - 1) * 32, 32);
        nodes[i].weight = weight[i - 1];
        if (leaf[i - 1]) nodes[i].wordId = nwords++;
    }
}

// TemplatedVocabulary.h:1240-1289
void Vocabulary::transformOne(const uint8_t* feature, int levelsup, int* wordId, double* weight, int* nid) const {
    const int nidLevel = L - levelsup;
    int node = -1;
    if (nidLevel <= 0) node = 0;
    int finalId = 0, level = 0;
    do {
        ++level;
        const std::vector<int>& ch = nodes[finalId].children;
        finalId = ch[0];
        double best = (double)hammingBitHack(feature, nodes[finalId].desc);
        for (size_t i = 1; i < ch.size(); i++) {
            const double d = (double)hammingBitHack(feature, nodes[ch[i]].desc);
            if (d < best) {
                best = d;
                finalId = ch[i];
            }
        }
        if (level == nidLevel) node = finalId;
    } while (!nodes[finalId].children.empty());
    if (node < 0) node = finalId;  // ruling, see the file header
    *wordId = nodes[finalId].wordId;
    *weight = nodes[finalId].weight;
    *nid = node;
}

static bool mustNormalize(int scoring, int* norm) {  // ScoringObject.h:85-93 (L1 = 0, L2 = 1)
    switch (scoring) {
        case 0: *norm = 0; return true;   // L1_NORM
        case 1: *norm = 1; return true;   // L2_NORM
        case 2: *norm = 0; return true;   // CHI_SQUARE
        case 3: *norm = 0; return true;   // KL
        case 4: *norm = 0; return true;   // BHATTACHARYYA
        default: *norm = 0; return false; // DOT_PRODUCT
    }
}

// The bookkeeping half of TemplatedVocabulary.h:1136-1206: what transform() does with the (word, weight, node)
// triple of each feature - BowVector::addWeight / addIfNotExist / normalize (BowVector.cpp:40-95) and
// FeatureVector::addFeature (FeatureVector.cpp:37-52).  tests/test_oracle_ref_dbow2.py runs the same streams
// through those two reference files compiled as they are (oracle/_ref/libdbow2_ref.so).
void bowAccumulate(const int* ids, const double* weights, const int* nids, int n, int scoring, int weighting,
                   std::map<int, double>& bow, std::map<int, std::vector<unsigned> >& fv) {
    bow.clear();
    fv.clear();
    int norm = 0;
    const bool must = mustNormalize(scoring, &norm);
    const bool tf = weighting == 0 /*TF_IDF*/ || weighting == 1 /*TF*/;
    for (int i = 0; i < n; i++) {
        const double w = weights[i];
        if (w > 0) {
            std::map<int, double>::iterator it = bow.find(ids[i]);
            if (it == bow.end()) bow[ids[i]] = w;
            else if (tf) it->second += w;
            fv[nids[i]].push_back((unsigned)i);
        }
    }
    if (tf && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (auto& e : bow) e.second /= nd;
    }
    if (must) {
        double s = 0.0;
        if (norm == 0) {
            for (auto& e : bow) s += fabs(e.second);
        } else {
            for (auto& e : bow) s += e.second * e.second;
            s = sqrt(s);
        }
        if (s > 0.0)
            for (auto& e : bow) e.second /= s;
    }
}

// TemplatedVocabulary.h:1136-1206
void Vocabulary::transform(const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                           std::map<int, std::vector<unsigned> >& fv) const {
    bow.clear();
    fv.clear();
    if (nwords == 0) return;
    std::vector<int> ids((size_t)n), nids((size_t)n);
    std::vector<double> ws((size_t)n);
    for (int i = 0; i < n; i++) transformOne(desc + (size_t)i * 32, levelsup, &ids[i], &ws[i], &nids[i]);
    bowAccumulate(ids.data(), ws.data(), nids.data(), n, scoring, weighting, bow, fv);
}

// L1Scoring::score, ScoringObject.cpp:23-68
double scoreL1(const std::map<int, double>& v1, const std::map<int, double>& v2) {
    auto a = v1.begin(), b = v2.begin();
    double score = 0;
    while (a != v1.end() && b != v2.end()) {
        if (a->first == b->first) {
            score += fabs(a->second - b->second) - fabs(a->second) - fabs(b->second);
            ++a;
            ++b;
        } else if (a->first < b->first) {
            a = v1.lower_bound(b->first);
        } else {
            b = v2.lower_bound(a->first);
        }
    }
    return -score / 2.0;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), src/ORBmatcher.cc:138-268.
// kfValid[i] != 0 stands for "vpMapPointsKF[i] is set and not bad" (:170-176); matchKF[j] receives the
// keyframe feature whose map point frame feature j takes, or -1 (vpMapPointMatches).
int searchByBoW(const KeyPoint* kfKeysUn, const uint8_t* kfDesc, const uint8_t* kfValid,
                const std::map<int, std::vector<unsigned> >& kfFeat, const KeyPoint* fKeys, const uint8_t* fDesc, int nF,
                const std::map<int, std::vector<unsigned> >& fFeat, float nnratio, bool checkOri, int* matchKF) {
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int j = 0; j < nF; j++) matchKF[j] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    auto KFit = kfFeat.begin();
    auto Fit = fFeat.begin();
    while (KFit != kfFeat.end() && Fit != fFeat.end()) {
        if (KFit->first == Fit->first) {
            const std::vector<unsigned>& vIndicesKF = KFit->second;
            const std::vector<unsigned>& vIndicesF = Fit->second;
            for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
                const unsigned realIdxKF = vIndicesKF[iKF];
                if (!kfValid[realIdxKF]) continue;
                const uint8_t* dKF = kfDesc + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
                    const unsigned realIdxF = vIndicesF[iF];
                    if (matchKF[realIdxF] >= 0) continue;
                    const int dist = hammingBitHack(dKF, fDesc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = (int)realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matchKF[bestIdxF] = (int)realIdxKF;
                        if (checkOri) {
                            float rot = kfKeysUn[realIdxKF].angle - fKeys[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++;
            Fit++;
        } else if (KFit->first < Fit->first) {
            KFit = kfFeat.lower_bound(Fit->first);
        } else {
            Fit = fFeat.lower_bound(KFit->first);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        computeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                matchKF[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&), include/ORBmatcher.h:110.  The fork
// declares it and defines it nowhere; this is the loop of upstream ORB-SLAM2 (of which the fork is a
// copy), which differs from the keyframe-to-frame search above in three places: second-side features need a
// good map point as well (valid2) and are given away once (vbMatched2), the distance test is the strict
// bestDist1 < TH_LOW, and the result is indexed by the first keyframe's features.
int searchByBoWKeyFrames(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* valid1, int n1,
                         const std::map<int, std::vector<unsigned> >& feat1, const KeyPoint* keysUn2,
                         const uint8_t* desc2, const uint8_t* valid2, int n2,
                         const std::map<int, std::vector<unsigned> >& feat2, float nnratio, bool checkOri,
                         int* matches12) {
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<bool> vbMatched2((size_t)n2, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    int nmatches = 0;
    auto f1it = feat1.begin();
    auto f2it = feat2.begin();
    while (f1it != feat1.end() && f2it != feat2.end()) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
                const size_t idx1 = f1it->second[i1];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
                    const size_t idx2 = f2it->second[i2];
                    if (vbMatched2[idx2] || !valid2[idx2]) continue;
                    const int dist = hammingBitHack(d1, desc2 + idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = (int)idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[(size_t)bestIdx2] = true;
                        if (checkOri) {
                            float rot = keysUn1[idx1].angle - keysUn2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = feat1.lower_bound(f2it->first);
        } else {
            f2it = feat2.lower_bound(f1it->first);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        computeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                matches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// ORBmatcher::CheckDistEpipolarLine, include/ORBmatcher.h:196 (declared only; upstream ORB-SLAM2's body).
static bool checkDistEpipolarLine(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* levelSigma2) {
    // epipolar line in second image l = x1'F12 = [a b c]
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[kp2.octave];
}

// ORBmatcher::SearchForTriangulation, include/ORBmatcher.h:133-144 (declared only; upstream ORB-SLAM2's loops for
// the monocular case: no right-image coordinates).  free1 / free2 = "the feature has no map point".  Upstream
// declares vbMatched2 and never sets it; markMatched turns the marking on.
int searchForTriangulation(const KeyPoint* keysUn1, const uint8_t* desc1, const uint8_t* free1, int n1,
                           const std::map<int, std::vector<unsigned> >& feat1, const KeyPoint* keysUn2,
                           const uint8_t* desc2, const uint8_t* free2, int n2,
                           const std::map<int, std::vector<unsigned> >& feat2, const float* F12, const float* epipole,
                           const float* scaleFactors, const float* levelSigma2, int nlevels, bool markMatched,
                           bool checkOri, int* vMatches12) {
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    int nmatches = 0;
    std::vector<bool> vbMatched2((size_t)n2, false);
    for (int i = 0; i < n1; i++) vMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;  // the fork's factor, src/ORBmatcher.cc:35
    auto f1it = feat1.begin();
    auto f2it = feat2.begin();
    while (f1it != feat1.end() && f2it != feat2.end()) {
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
                const size_t idx1 = f1it->second[i1];
                if (!free1[idx1]) continue;  // there is already a MapPoint
                const KeyPoint& kp1 = keysUn1[idx1];
                const uint8_t* d1 = desc1 + idx1 * 32;
                int bestDist = TH_LOW;
                int bestIdx2 = -1;
                for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
                    const size_t idx2 = f2it->second[i2];
                    if (vbMatched2[idx2] || !free2[idx2]) continue;
                    const int dist = hammingBitHack(d1, desc2 + idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = keysUn2[idx2];
                    if (kp2.octave < 0 || kp2.octave >= nlevels) continue;  // the reference would read outside its tables
                    if (epipole) {
                        const float distex = epipole[0] - kp2.x;
                        const float distey = epipole[1] - kp2.y;
                        if (distex * distex + distey * distey < 100 * scaleFactors[kp2.octave]) continue;
                    }
                    if (checkDistEpipolarLine(kp1, kp2, F12, levelSigma2)) {
                        bestIdx2 = (int)idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    const KeyPoint& kp2 = keysUn2[bestIdx2];
                    vMatches12[idx1] = bestIdx2;
                    if (markMatched) vbMatched2[(size_t)bestIdx2] = true;
                    nmatches++;
                    if (checkOri) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = feat1.lower_bound(f2it->first);
        } else {
            f2it = feat2.lower_bound(f1it->first);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        computeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                vMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

}  // namespace orbo
