// CPU ORACLE — TEST INFRASTRUCTURE (see orb_oracle.hpp).  PARITY UNPINNED.
//
// Restatement of the reference's ORBextractor (src/ORBextractor.cc:19-103,
// 369-949, include/ORBextractor.h:12-102).  Where the fork's text does not
// compile, is truncated or is undefined behaviour it is repaired to upstream
// ORB-SLAM2 semantics (noted per function); where it compiles and merely
// differs from upstream the fork's behaviour sits behind a Q_* switch.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>

#include "orb_oracle.hpp"

namespace orbo {

static const int PATCH_SIZE = 31;        // src/ORBextractor.cc:19
static const int HALF_PATCH_SIZE = 15;   // :20
static const int EDGE_THRESHOLD = 19;    // :21

static const int kPattern[256 * 4] = {
#include "orb_pattern.inc"
};

// src/ORBextractor.cc:369-436.  Repairs: the inverse tables divide by the
// forward tables (the fork divides a zero vector by itself).
Extractor::Extractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_, unsigned quirks_)
    : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_),
      minThFAST(minThFAST_), quirks(quirks_) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);  // float * double member
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    pyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = (float)(nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        mnFeaturesPerLevel[level] = cvRound(nDesired);
        sumFeatures += mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    // Patch row half-widths, :420-435.
    umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0;
    const int vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    const int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        if (quirks & Q_UMAX_FORK) {
            while (umax[v] == umax[v0 + 1]) ++v0;
        } else {
            while (umax[v0] == umax[v0 + 1]) ++v0;
        }
        umax[v] = v0;
        ++v0;
    }
}

// src/ORBextractor.cc:913-949.  Level sizes derive from the input size, each
// level is resized from the previous one.  The 19-pixel padding of the
// reference's buffers is never read by later stages and is not modelled.
void Extractor::computePyramid(const uint8_t* img, int w, int h, int stride) {
    for (int level = 0; level < nlevels; level++) {
        const float scale = mvInvScaleFactor[level];
        Image& L = pyramid[level];
        L.w = cvRound((float)w * scale);
        L.h = cvRound((float)h * scale);
        L.px.assign((size_t)L.w * L.h, 0);
        if (level != 0) {
            const Image& P = pyramid[level - 1];
            if (quirks & Q_RESIZE_AREA)
                resizeArea(P.px.data(), P.w, P.h, P.w, L.px.data(), L.w, L.h, L.w);
            else
                resizeLinear(P.px.data(), P.w, P.h, P.w, L.px.data(), L.w, L.h, L.w);
        } else {
            for (int y = 0; y < h; y++) memcpy(L.row(y), img + (size_t)y * stride, w);
        }
    }
}

// ---------------------------------------------------------------------------
// Quadtree (include/ORBextractor.h:12-26, src/ORBextractor.cc:511-796).
namespace {
struct Pt {
    int x, y;
};
struct Node {
    std::vector<KeyPoint> vKeys;
    Pt UL{0, 0}, UR{0, 0}, BL{0, 0}, BR{0, 0};
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0;  // creation number; replaces the heap address in the size sort
    void divide(Node& n1, Node& n2, Node& n3, Node& n4, bool n3quirk) const;
};

// src/ORBextractor.cc:511-569.
void Node::divide(Node& n1, Node& n2, Node& n3, Node& n4, bool n3quirk) const {
    const int halfX = (int)ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = (int)ceil(static_cast<float>(BR.y - UL.y) / 2);
    n1.UL = UL;
    n1.UR = Pt{UL.x + halfX, UL.y};
    n1.BL = Pt{UL.x, UL.y + halfY};
    n1.BR = Pt{UL.x + halfX, UL.y + halfY};
    n2.UL = n1.UR;
    n2.UR = UR;
    n2.BL = n1.BR;
    n2.BR = Pt{UR.x, UL.y + halfY};
    n3.UL = n1.BL;
    n3.UR = n1.BR;
    n3.BL = BL;
    n3.BR = Pt{n1.BR.x, BL.y};
    n4.UL = n3.UR;
    n4.UR = n2.BR;
    n4.BL = n3.BR;
    n4.BR = BR;
    for (size_t i = 0; i < vKeys.size(); i++) {
        const KeyPoint& kp = vKeys[i];
        if (kp.x < n1.UR.x) {
            if (kp.y < n1.BR.y)
                n1.vKeys.push_back(kp);
            else
                n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BR.y)
            n2.vKeys.push_back(kp);
        else
            n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3quirk) {
        // fork :565-568: "if(n4.size()==1); n3.bNoMore = true;" — n3 always, n4 never
        n3.bNoMore = true;
    } else {
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
}
}  // namespace

// src/ORBextractor.cc:571-796.  Repairs: `N` is the quota parameter; all four
// children are pushed to the FRONT of the list (the fork pushes n2..n4 to the
// back but records the front, leaving dangling iterators); the size sort
// breaks ties by creation number instead of heap address.
std::vector<KeyPoint> Extractor::distributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX,
                                                   int minY, int maxY, int N) const {
    const bool n3quirk = (quirks & Q_DIVIDE_N3) != 0;
    const int nIni = (int)round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    long seqCounter = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.UL = Pt{(int)(hX * static_cast<float>(i)), 0};
        ni.UR = Pt{(int)(hX * static_cast<float>(i + 1)), 0};
        ni.BL = Pt{ni.UL.x, maxY - minY};
        ni.BR = Pt{ni.UR.x, maxY - minY};
        ni.seq = seqCounter++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint& kp = vToDistributeKeys[i];
        vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
    }
    std::list<Node>::iterator lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) {
            lit->bNoMore = true;
            lit++;
        } else if (lit->vKeys.empty())
            lit = lNodes.erase(lit);
        else
            lit++;
    }

    bool bFinish = false;
    typedef std::pair<int, Node*> SizeNode;
    auto bySizeThenSeq = [](const SizeNode& a, const SizeNode& b) {
        return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq;
    };
    std::vector<SizeNode> vSizeAndPointerToNode;
    auto pushChildren = [&](Node* c[4], int& nToExpand) {
        for (int q = 0; q < 4; q++) {
            if (c[q]->vKeys.size() > 0) {
                c[q]->seq = seqCounter++;
                lNodes.push_front(*c[q]);
                if (c[q]->vKeys.size() > 1) {
                    nToExpand++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)c[q]->vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) {
                lit++;
                continue;
            }
            Node n1, n2, n3, n4;
            lit->divide(n1, n2, n3, n4, n3quirk);
            Node* c[4] = {&n1, &n2, &n3, &n4};
            pushChildren(c, nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeNode> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(), bySizeThenSeq);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    vPrev[j].second->divide(n1, n2, n3, n4, n3quirk);
                    Node* c[4] = {&n1, &n2, &n3, &n4};
                    pushChildren(c, nToExpand);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }

    std::vector<KeyPoint> vResultKeys;
    vResultKeys.reserve(lNodes.size());
    for (lit = lNodes.begin(); lit != lNodes.end(); lit++) {
        const std::vector<KeyPoint>& vNodeKeys = lit->vKeys;
        const KeyPoint* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

// src/ORBextractor.cc:23-63.
float icAngle(const Image& im, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const int step = im.w;
    const uint8_t* center = im.px.data() + (size_t)cvRound(pty) * step + cvRound(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; u++) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
}

// src/ORBextractor.cc:65-103.  px*b + py*a is two float products and one float
// sum, each rounded (no fused multiply-add).
void orbDescriptor(const Image& img, float ptx, float pty, float angleDeg, bool libmTrig, uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    const float angle = angleDeg * factorPI;
    float a, b;
    if (libmTrig) {
        a = cosf(angle);
        b = sinf(angle);
    } else {
        sincosPortable(angle, &b, &a);
    }
    const int step = img.w;
    const uint8_t* center = img.px.data() + (size_t)cvRound(pty) * step + cvRound(ptx);
    const int* pattern = kPattern;
    auto value = [&](int idx) -> int {
        const float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
        const int dy = cvRound(px * b + py * a);
        const int dx = cvRound(px * a - py * b);
        return center[dy * step + dx];
    };
    for (int i = 0; i < 32; i++, pattern += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const int t0 = value(2 * j), t1 = value(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// src/ORBextractor.cc:799-898.
void Extractor::computeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.assign(nlevels, {});
    fastKeys.assign(nlevels, {});
    const float W = 30;
    for (int level = 0; level < nlevels; level++) {
        const Image& im = pyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KeyPoint> vToDistributeKeys;
        vToDistributeKeys.reserve(nfeatures * 10);
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        const int wCell = (int)ceil(width / nCols);
        const int hCell = (int)ceil(height / nRows);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
                std::vector<KeyPoint> vKeysCell;
                fastDetect(im.row(y0) + x0, im.w, x1 - x0, y1 - y0, iniThFAST, true, vKeysCell);
                if (vKeysCell.empty()) fastDetect(im.row(y0) + x0, im.w, x1 - x0, y1 - y0, minThFAST, true, vKeysCell);
                for (KeyPoint& kp : vKeysCell) {
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    vToDistributeKeys.push_back(kp);
                }
            }
        }
        fastKeys[level] = vToDistributeKeys;
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        keypoints = distributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                      mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
        for (KeyPoint& kp : keypoints) {
            kp.x += minBorderX;
            kp.y += minBorderY;
            kp.octave = level;
            kp.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; level++)
        for (KeyPoint& kp : allKeypoints[level]) kp.angle = icAngle(pyramid[level], kp.x, kp.y, umax);
}

// src/ORBextractor.cc:448-509 plus the tail the fork lost: after the level's
// descriptors, advance the row offset, scale level>0 keypoints by the level's
// scale factor and append them to the output.
void Extractor::extract(const uint8_t* img, int w, int h, int stride, std::vector<KeyPoint>& kps,
                        std::vector<uint8_t>& desc) {
    kps.clear();
    desc.clear();
    if (!img || w <= 0 || h <= 0) return;
    computePyramid(img, w, h, stride);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    computeKeyPointsOctTree(allKeypoints);
    levelKeys = allKeypoints;
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; level++) nkeypoints += (int)allKeypoints[level].size();
    desc.assign((size_t)nkeypoints * 32, 0);
    kps.reserve(nkeypoints);
    blurred.assign(nlevels, Image());
    int offset = 0;
    for (int level = 0; level < nlevels; level++) {
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        const int n = (int)keypoints.size();
        if (n == 0) continue;
        Image& B = blurred[level];
        B.w = pyramid[level].w;
        B.h = pyramid[level].h;
        B.px.resize((size_t)B.w * B.h);
        gaussianBlur7(pyramid[level].px.data(), B.w, B.h, B.w, B.px.data(), B.w, (quirks & Q_GAUSS_LEGACY) != 0);
        for (int i = 0; i < n; i++)
            orbDescriptor(B, keypoints[i].x, keypoints[i].y, keypoints[i].angle, (quirks & Q_LIBM_TRIG) != 0,
                          desc.data() + (size_t)(offset + i) * 32);
        offset += n;
        if (level != 0) {
            const float scale = mvScaleFactor[level];
            for (KeyPoint& kp : keypoints) {
                kp.x *= scale;
                kp.y *= scale;
            }
        }
        kps.insert(kps.end(), keypoints.begin(), keypoints.end());
    }
}

}  // namespace orbo
