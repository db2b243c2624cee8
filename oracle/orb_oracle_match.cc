// CPU ORACLE — TEST INFRASTRUCTURE (see orb_oracle.hpp).  PARITY UNPINNED.
//
// Restatement of the matcher side: ORBmatcher::DescriptorDistance,
// SearchForInitialization, SearchByProjection(Frame&,const Frame&,..),
// ComputeThreeMaxima (src/ORBmatcher.cc:16-136, 279-476) and the Frame grid
// query Frame::GetFeaturesInArea (src/Frame.cc:86-144, include/Frame.h:16-17).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

#include "orb_oracle.hpp"

namespace orbo {

static const int TH_HIGH = 100;      // src/ORBmatcher.cc:16
static const int TH_LOW = 50;        // :17
static const int HISTO_LENGTH = 30;  // :18
static const int GRID_ROWS = 48;     // include/Frame.h:16
static const int GRID_COLS = 64;     // include/Frame.h:17

// src/ORBmatcher.cc:460-476, literal (including the 9-digit mask literal,
// which behaves as 0x55555555 because v>>1 has no bits above bit 30).
int hammingBitHack(const uint8_t* a, const uint8_t* b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x555555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

int hammingPopcount(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

// The Frame constructor tail that the fork lost (src/Frame.cc:53-69): no lens
// distortion, so mvKeysUn == mvKeys and the bounds are the image rectangle;
// grid element inverses are 64/(maxX-minX), 48/(maxY-minY); a keypoint goes to
// cell (round((x-minX)*wInv), round((y-minY)*hInv)) unless that falls outside
// the grid; cell lists hold indices in increasing order.
void FrameGrid::build(const KeyPoint* k, const uint8_t* d, int n, int imgW, int imgH, const float* bounds) {
    keys = k;
    desc = d;
    N = n;
    minX = bounds ? bounds[0] : 0.f;
    maxX = bounds ? bounds[1] : (float)imgW;
    minY = bounds ? bounds[2] : 0.f;
    maxY = bounds ? bounds[3] : (float)imgH;
    wInv = static_cast<float>(GRID_COLS) / (maxX - minX);
    hInv = static_cast<float>(GRID_ROWS) / (maxY - minY);
    cells.assign(GRID_COLS * GRID_ROWS, {});
    for (int i = 0; i < n; i++) {
        const int posX = (int)round((k[i].x - minX) * wInv);
        const int posY = (int)round((k[i].y - minY) * hInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        cells[posX * GRID_ROWS + posY].push_back(i);
    }
}

// cvUndistortPointsInternal of OpenCV 4.2 for the call undistortPoints(src, dst, K, D, noArray(), K):
// criteria (MAX_ITER, 5), no tilt, R = I, P = K; all arithmetic in double, operation order as there.
void undistortPoints(const float* xyIn, float* xyOut, int n, const float K[4], const float dist[5]) {
    double k[14] = {0};
    k[0] = dist[0];
    k[1] = dist[1];
    k[2] = dist[2];
    k[3] = dist[3];
    k[4] = dist[4];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        double x = xyIn[2 * i], y = xyIn[2 * i + 1];
        const double u = x, v = y;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) {  // the model is not invertible there: the undistorted point is the normalised input
                x = (u - cx) * ifx;
                y = (v - cy) * ify;
                break;
            }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        // RR = P * R = K: xx = fx*x + 0*y + cx, yy = 0*x + fy*y + cy, ww = 1 / (0*x + 0*y + 1)
        const double xx = fx * x + 0. * y + cx;
        const double yy = 0. * x + fy * y + cy;
        const double ww = 1. / (0. * x + 0. * y + 1.);
        xyOut[2 * i] = (float)(xx * ww);
        xyOut[2 * i + 1] = (float)(yy * ww);
    }
}

void undistortKeyPoints(const KeyPoint* in, KeyPoint* out, int n, const float K[4], const float dist[5]) {
    if (dist[0] == 0.0f) {  // if(mDistCoef.at<float>(0)==0.0) mvKeysUn = mvKeys
        for (int i = 0; i < n; i++) out[i] = in[i];
        return;
    }
    std::vector<float> mat((size_t)2 * n);
    for (int i = 0; i < n; i++) {
        mat[2 * i] = in[i].x;
        mat[2 * i + 1] = in[i].y;
    }
    undistortPoints(mat.data(), mat.data(), n, K, dist);
    for (int i = 0; i < n; i++) {
        KeyPoint kp = in[i];
        kp.x = mat[2 * i];
        kp.y = mat[2 * i + 1];
        out[i] = kp;
    }
}

void computeImageBounds(int cols, int rows, const float K[4], const float dist[5], float bounds[4]) {
    if (dist[0] != 0.0f) {
        float mat[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        undistortPoints(mat, mat, 4, K, dist);
        bounds[0] = std::min(mat[0], mat[4]);  // mnMinX = min(mat(0,0), mat(2,0))
        bounds[1] = std::max(mat[2], mat[6]);  // mnMaxX = max(mat(1,0), mat(3,0))
        bounds[2] = std::min(mat[1], mat[3]);  // mnMinY = min(mat(0,1), mat(1,1))
        bounds[3] = std::max(mat[5], mat[7]);  // mnMaxY = max(mat(2,1), mat(3,1))
    } else {
        bounds[0] = 0.0f;
        bounds[1] = (float)cols;
        bounds[2] = 0.0f;
        bounds[3] = (float)rows;
    }
}

// src/Frame.cc:86-144.
std::vector<int> FrameGrid::featuresInArea(float x, float y, float r, int minLevel, int maxLevel,
                                           unsigned quirks) const {
    std::vector<int> vIndices;
    const bool excl = (quirks & Q_GRID_EXCLUSIVE) != 0;
    const bool circ = (quirks & Q_WINDOW_CIRCULAR) != 0;
    const int nMinCellX = std::max(0, (int)floor((x - minX - r) * wInv));
    if (nMinCellX >= GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)GRID_COLS - 1, (int)ceil((x - minX + r) * wInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - minY - r) * hInv));
    if (excl ? nMinCellY > GRID_ROWS : nMinCellY >= GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)GRID_ROWS - 1, (int)ceil((y - minY + r) * hInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel >= 0) || (maxLevel >= 0);
    const int endX = excl ? nMaxCellX : nMaxCellX + 1;
    const int endY = excl ? nMaxCellY : nMaxCellY + 1;
    for (int ix = nMinCellX; ix < endX; ix++) {
        for (int iy = nMinCellY; iy < endY; iy++) {
            const std::vector<int>& vCell = cells[ix * GRID_ROWS + iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const KeyPoint& kpUn = keys[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x;
                const float disty = kpUn.y - y;
                if (circ) {
                    if (distx * distx + disty * disty < r * r) vIndices.push_back(vCell[j]);
                } else {
                    if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        }
    }
    return vIndices;
}

// src/ORBmatcher.cc:413-454.
void computeThreeMaxima(const int* counts, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = counts[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

static inline int rotBin(float a1, float a2) {
    const float factor = HISTO_LENGTH / 360.0f;  // src/ORBmatcher.cc:35
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// src/ORBmatcher.cc:26-136.
int searchForInitialization(const FrameGrid& F1, const FrameGrid& F2, float* prevXY, int* vnMatches12, int windowSize,
                            float nnratio, bool checkOri, unsigned quirks) {
    int nmatches = 0;
    for (int i = 0; i < F1.N; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(F2.N, INT_MAX);
    std::vector<int> vnMatches21(F2.N, -1);
    for (int i1 = 0; i1 < F1.N; i1++) {
        const KeyPoint kp1 = F1.keys[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<int> vIndices2 =
            F2.featuresInArea(prevXY[2 * i1], prevXY[2 * i1 + 1], (float)windowSize, level1, level1, quirks);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = F1.desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const int dist = hammingBitHack(d1, F2.desc + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) rotHist[rotBin(F1.keys[i1].angle, F2.keys[bestIdx2].angle)].push_back(i1);
            }
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH], ind1, ind2, ind3;
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        computeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) {
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i1 = 0; i1 < F1.N; i1++)
        if (vnMatches12[i1] >= 0) {
            prevXY[2 * i1] = F2.keys[vnMatches12[i1]].x;
            prevXY[2 * i1 + 1] = F2.keys[vnMatches12[i1]].y;
        }
    return nmatches;
}

// src/ORBmatcher.cc:302-405 with the missing window query restored by the
// caller (minLevel/maxLevel per query).  Queries are the last frame's map
// points already projected to (u,v).  A current-frame keypoint is skipped while
// it holds a map point with Observations() > 0 (:345-347): `occupied2` gives
// that state on entry and `qobs[i]` says whether query i's map point has
// observations, because an accepted query stores its map point into the
// keypoint (:366) and thereby changes what later queries see.  match2[i2]
// receives the index of the query whose map point keypoint i2 holds at the end.
int searchByProjection(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                       const float* qangle, const uint8_t* qdesc, const uint8_t* qobs, int nq, const FrameGrid& F2,
                       const uint8_t* occupied2, int* match2, int thHigh, bool checkOri, unsigned quirks) {
    int nmatches = 0;
    std::vector<uint8_t> occ(F2.N, 0);
    for (int i = 0; i < F2.N; i++) {
        match2[i] = -1;
        if (occupied2) occ[i] = occupied2[i];
    }
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < nq; i++) {
        if (u[i] < F2.minX || u[i] > F2.maxX) continue;
        if (v[i] < F2.minY || v[i] > F2.maxY) continue;
        std::vector<int> vIndices2 = F2.featuresInArea(u[i], v[i], radius[i], minLevel[i], maxLevel[i], quirks);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = qdesc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (occ[i2]) continue;
            const int dist = hammingBitHack(dMP, F2.desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= thHigh && bestIdx2 >= 0) {
            match2[bestIdx2] = i;
            occ[bestIdx2] = qobs ? qobs[i] : 1;
            nmatches++;
            if (checkOri) rotHist[rotBin(qangle[i], F2.keys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH], ind1, ind2, ind3;
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        computeThreeMaxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int i2 : rotHist[i]) {
                    match2[i2] = -2;  // the reference stores NULL here; -2 tells "cleared" from "never assigned"
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), include/ORBmatcher.h:24-39
// (tracking of the local map).  The fork declares it with its five steps and defines it nowhere; the loop
// is upstream ORB-SLAM2's.  Queries are the local map points that isInFrustum left in view: (u,v) =
// (mTrackProjX, mTrackProjY), radius = r * scale factor of the predicted level, levels [predicted - 1,
// predicted].  A keypoint that holds a map point with observations is skipped, and an accepted query
// stores its own map point there, so later queries see it (qobs as in searchByProjection above).
int searchLocalMap(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                   const uint8_t* qdesc, const uint8_t* qobs, int nq, const FrameGrid& F, const uint8_t* occupied2,
                   int* match2, int thHigh, float nnratio, unsigned quirks) {
    int nmatches = 0;
    std::vector<uint8_t> occ(F.N, 0);
    for (int i = 0; i < F.N; i++) {
        match2[i] = -1;
        if (occupied2) occ[i] = occupied2[i] ? 1 : 0;
    }
    for (int iMP = 0; iMP < nq; iMP++) {
        // isInFrustum (upstream Frame.cc) has rejected projections outside the image bounds
        if (u[iMP] < F.minX || u[iMP] > F.maxX || v[iMP] < F.minY || v[iMP] > F.maxY) continue;
        const std::vector<int> vIndices = F.featuresInArea(u[iMP], v[iMP], radius[iMP], minLevel[iMP], maxLevel[iMP], quirks);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = qdesc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (occ[idx]) continue;
            const int dist = hammingBitHack(MPdescriptor, F.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F.keys[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.keys[idx].octave;
                bestDist2 = dist;
            }
        }
        // the ratio applies only when best and second best lie on the same scale level
        if (bestDist <= thHigh) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match2[bestIdx] = iMP;
            occ[bestIdx] = qobs ? (qobs[iMP] ? 1 : 0) : 1;
            nmatches++;
        }
    }
    return nmatches;
}

// Candidate loop of ORBmatcher::Fuse (include/ORBmatcher.h:163-184, both overloads) and of
// ORBmatcher::SearchBySim3 (:146-160), declared by the fork and defined nowhere in it; upstream ORB-SLAM2's
// loops.  Every query on its own: window, level range [predicted - 1, predicted], for the first Fuse
// overload the monocular reprojection gate (squared error times the level's inverse sigma^2 above 5.99
// rejects the candidate), smallest distance, first candidate among equals; kept when <= th.
void searchBest(const float* u, const float* v, const float* radius, const int* minLevel, const int* maxLevel,
                const uint8_t* qdesc, int nq, const FrameGrid& F, const float* invLevelSigma2, int nlevels,
                double chi2Max, int th, unsigned quirks, int* bestIdxOut, int* bestDistOut) {
    for (int i = 0; i < nq; i++) {
        const std::vector<int> vIndices = F.featuresInArea(u[i], v[i], radius[i], minLevel[i], maxLevel[i], quirks);
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            const KeyPoint& kp = F.keys[idx];
            if (invLevelSigma2) {
                const int kpLevel = kp.octave;
                if (kpLevel < 0 || kpLevel >= nlevels) continue;  // the reference would read outside mvInvLevelSigma2
                const float ex = u[i] - kp.x;
                const float ey = v[i] - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * invLevelSigma2[kpLevel] > chi2Max) continue;
            }
            const int dist = hammingBitHack(qdesc + (size_t)i * 32, F.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        bestIdxOut[i] = (bestDist <= th) ? bestIdx : -1;
        bestDistOut[i] = bestDist;
    }
}

// Dense scan with the best / second-best rule of src/ORBmatcher.cc:53-76.
void bruteForceTop2(const uint8_t* q, int nq, const uint8_t* db, int64_t nd, int32_t* bestIdx, uint16_t* bestDist,
                    uint16_t* secondDist) {
    for (int i = 0; i < nq; i++) {
        uint64_t a[4];
        memcpy(a, q + (size_t)i * 32, 32);
        int best = INT_MAX, second = INT_MAX, idx = -1;
        for (int64_t j = 0; j < nd; j++) {
            uint64_t b[4];
            memcpy(b, db + (size_t)j * 32, 32);
            const int dist = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                             __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (dist < best) {
                second = best;
                best = dist;
                idx = (int)j;
            } else if (dist < second) {
                second = dist;
            }
        }
        bestIdx[i] = idx;
        bestDist[i] = (uint16_t)std::min(best, 65535);
        secondDist[i] = (uint16_t)std::min(second, 65535);
    }
}

// src/MapPoint.cc:186-214.  Distances are kept as float and sorted as int, as the reference does.
int distinctiveDescriptor(const uint8_t* desc, int n, bool selfDistance, int* median) {
    if (n <= 0) return -1;
    const std::size_t N = (std::size_t)n;
    std::vector<std::vector<float>> Distances(N, std::vector<float>(N, 0));
    for (std::size_t i = 0; i < N; i++) {
        Distances[i][i] = 0;
        for (std::size_t j = i + 1; j < N; j++) {
            const int distij = hammingBitHack(desc + 32 * i, desc + 32 * (selfDistance ? i : j));  // :195
            Distances[i][j] = (float)distij;
            Distances[j][i] = (float)distij;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (std::size_t i = 0; i < N; i++) {
        std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
        std::sort(vDists.begin(), vDists.end());
        const int med = vDists[(std::size_t)(0.5 * (N - 1))];
        if (med < BestMedian) {
            BestMedian = med;
            BestIdx = (int)i;
        }
    }
    if (median) *median = BestMedian;
    return BestIdx;
}

}  // namespace orbo
