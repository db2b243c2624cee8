// Windowed Hamming matching for gfx950: ORBmatcher::SearchForInitialization
// (src/ORBmatcher.cc:26-136) and the search loop of
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (:302-405),
// both on top of Frame::GetFeaturesInArea (src/Frame.cc:86-144) and the 64x48
// bucket grid of include/Frame.h:16-17,96.
//
// Three kernels per call:
//  1. k_grid_build (one workgroup per searched frame) sorts the frame's
//     keypoints by (grid column * 48 + grid row, keypoint index).
//     GetFeaturesInArea visits columns outermost, rows inside, indices inside a
//     bucket in increasing order, which is exactly ascending position in that
//     array; a window query is one contiguous slice of it with a row filter,
//     and "first candidate wins on equal distance" is "smallest position wins".
//  2. k_match_lists (one wavefront per query, all queries of all pairs in
//     parallel) runs the window query and writes the query's candidates as
//     (Hamming distance << 16 | position), in window order.  This is all of the
//     descriptor work and does not depend on the order queries are resolved in.
//  3. k_match_resolve (one workgroup per pair) walks the queries in index order,
//     because each accepted match changes what later queries may take
//     (vMatchedDistance / map-point occupancy): it filters the query's list by
//     that state, reduces (distance, position) over the lanes and updates the
//     state, all in LDS.  Queries with an empty list are skipped; three wavefronts
//     stage the lists of the next 64 queries while one resolves the current 64.
//     The rotation histogram is filled afterwards from the recorded acceptances
//     (the reference never removes a stolen match from rotHist, so every
//     acceptance counts once).
//
// The same grid serves the overloads the reference only declares
// (include/ORBmatcher.h:39,146-184): k_match_resolve<2> applies the local-map
// rule (best and second best with their pyramid levels), and k_match_best takes
// every query's best candidate on its own (Fuse, SearchBySim3), with no lists
// and no sequential stage.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstring>
#include <vector>

#include "match_common.h"
#include "orb_internal.h"

#define M_MAXKP 8192          // keypoints per frame the LDS state is sized for
#define M_CELLS (ORB_GRID_COLS * ORB_GRID_ROWS)



// Frame::mnMinX .. mnMaxY and the grid element inverses (upstream Frame.cc constructor tail).
struct GridGeom {
    float minX, maxX, minY, maxY, wInv, hInv;
};

struct Window {
    int x0, x1, y0, y1;  // bucket ranges [x0,x1) x [y0,y1); empty when x1 <= x0
};

// Bucket range of GetFeaturesInArea(x, y, r), src/Frame.cc:91-112.
static __device__ __forceinline__ Window window_cells(float x, float y, float r, const GridGeom& G, bool excl) {
    Window w;
    w.x0 = w.x1 = w.y0 = w.y1 = 0;
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, G.minX), r), G.wInv)));
    if (nMinCellX >= ORB_GRID_COLS) return w;
    const int nMaxCellX = min(ORB_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, G.minX), r), G.wInv)));
    if (nMaxCellX < 0) return w;
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, G.minY), r), G.hInv)));
    if (excl ? nMinCellY > ORB_GRID_ROWS : nMinCellY >= ORB_GRID_ROWS) return w;
    const int nMaxCellY = min(ORB_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, G.minY), r), G.hInv)));
    if (nMaxCellY < 0) return w;
    w.x0 = nMinCellX;
    w.x1 = excl ? nMaxCellX : nMaxCellX + 1;
    w.y0 = nMinCellY;
    w.y1 = excl ? nMaxCellY : nMaxCellY + 1;
    if (w.y0 > ORB_GRID_ROWS) w.y0 = ORB_GRID_ROWS;
    if (w.y1 < w.y0) w.y1 = w.y0;
    return w;
}


#define M_LIST_SEG 256   // list entries every query owns (the resolve kernel stages the first 128 without a test)
#define M_ERR_LIMIT 1    // error word bits: keypoint counts beyond the kernels' limits
#define M_ERR_POOL 2     // the candidate pool ran out; the second error word holds the entries the call needed

// Device scratch of one call, per pair: the sorted grid of the searched frame and the per-query
// candidate lists.
struct MatchBuffers {
    uint32_t* gSorted;   // [npairs][stride2]  bucket << 16 | keypoint index, ascending
    float2* gXY;         // [npairs][stride2]  keypoint position by sorted position
    uint8_t* gOct;       // [npairs][stride2]  keypoint octave by sorted position
    int* cellStart;      // [npairs][M_CELLS + 1]
    uint32_t* lists;     // [npairs][stride1][capQ]  distance << 16 | sorted position: a query's first capQ candidates
    int* counts;         // [npairs][stride1]
    int* accepted;       // [npairs][stride1]  keypoint a query was matched to when it was resolved, -1 otherwise
    int* order;          // [npairs][stride1]  queries with a non-empty list, ascending
    int capQ;            // min(stride2, M_LIST_SEG)
    uint32_t* pool;      // [poolCap] candidates beyond a query's first capQ, handed out by an atomic counter
    int* ovfStart;       // [npairs][stride1]  where in the pool a query's candidate number capQ sits
    unsigned poolCap;
    unsigned* poolUsed;  // entries asked for so far in this call (may exceed poolCap: then M_ERR_POOL is set)
};

// A query's candidate list: its first capQ entries in a fixed segment, the rest (wide windows over dense
// frames) in the shared pool.
static __device__ __forceinline__ uint32_t list_entry(const MatchBuffers& B, const uint32_t* seg, int ovf, int c) {
    return c < B.capQ ? seg[c] : B.pool[(unsigned)ovf + (unsigned)(c - B.capQ)];
}

// Kernel 1.  Bucket assignment as the Frame constructor does it (PosInGrid): rounded position,
// keypoints that fall outside the grid are left out.  maxOctave >= 0 also leaves out the keypoints of coarser
// levels: a search whose every query asks for levels up to maxOctave only (SearchForInitialization: level 0,
// src/ORBmatcher.cc:41-46) never looks at them, and a 2000-feature frame has 434 keypoints on level 0.
// (1024 threads: a call on one pair of frames is a single workgroup, whose every phase is a few dependent steps
// per thread; with 256 threads the kernel took 18 us for 1000 keypoints.)
#define GB_THREADS 1024
__global__ __launch_bounds__(GB_THREADS) void k_grid_build(const orb_keypoint* __restrict__ k2all, const int* __restrict__ n2s,
                                                    int n2const, int stride2, GridGeom G, MatchBuffers B,
                                                    int* __restrict__ error, int maxOctave) {
    __shared__ uint32_t s_sorted[M_MAXKP];
    __shared__ uint32_t s_key[M_MAXKP];
    __shared__ int s_start[M_CELLS + 1];
    __shared__ int s_cursor[M_CELLS];
    __shared__ int s_part[GB_THREADS / 64];
    __shared__ int s_big[M_MAXKP / 32];
    __shared__ int s_nBig;
    const int pair = blockIdx.x, t = threadIdx.x;
    const orb_keypoint* k2 = k2all + (int64_t)pair * stride2;
    int* cellStart = B.cellStart + (int64_t)pair * (M_CELLS + 1);
    const int n2 = n2s ? n2s[pair] : n2const;
    if (pair == 0 && t == 0) *B.poolUsed = 0u;  // the call's pool counter starts at zero (the list kernel, next on the stream, counts)
    if (n2 > stride2 || n2 > M_MAXKP || n2 < 0) {
        if (t == 0) atomicOr(error, M_ERR_LIMIT);
        for (int c = t; c <= M_CELLS; c += GB_THREADS) cellStart[c] = 0;  // an empty grid: the later kernels find nothing
        return;
    }
    // Counting sort by bucket: histogram, prefix sum, scatter; a bucket's few entries are then put in
    // index order by one thread (the scatter's order depends on how the atomics were served).
    for (int c = t; c <= M_CELLS; c += GB_THREADS) s_start[c] = 0;
    __syncthreads();
    for (int i = t; i < n2; i += GB_THREADS) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(k2[i].x, G.minX), G.wInv));  // Frame::PosInGrid
        const int py = (int)roundf(__fmul_rn(__fsub_rn(k2[i].y, G.minY), G.hInv));
        uint32_t key = 0xFFFFFFFFu;
        if (px >= 0 && px < ORB_GRID_COLS && py >= 0 && py < ORB_GRID_ROWS && (maxOctave < 0 || k2[i].octave <= maxOctave)) {
            key = ((uint32_t)(px * ORB_GRID_ROWS + py) << 16) | (uint32_t)i;
            atomicAdd(&s_start[px * ORB_GRID_ROWS + py], 1);
        }
        s_key[i] = key;
    }
    __syncthreads();
    {   // exclusive prefix sum of the M_CELLS counts: three consecutive buckets per thread, then across threads
        constexpr int per = M_CELLS / GB_THREADS;
        static_assert(per * GB_THREADS == M_CELLS, "buckets divide evenly among the threads");
        int v[per], sum = 0;
#pragma unroll
        for (int j = 0; j < per; j++) {
            v[j] = s_start[t * per + j];
            sum += v[j];
        }
        const int lane = t & 63, wv = t >> 6;
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w2 = 0; w2 < wv; w2++) run += s_part[w2];
#pragma unroll
        for (int j = 0; j < per; j++) {
            s_start[t * per + j] = run;
            s_cursor[t * per + j] = run;
            run += v[j];
        }
        if (t == GB_THREADS - 1) s_start[M_CELLS] = run;
    }
    __syncthreads();
    for (int i = t; i < n2; i += GB_THREADS) {
        const uint32_t key = s_key[i];
        if (key != 0xFFFFFFFFu) s_sorted[atomicAdd(&s_cursor[key >> 16], 1)] = key;
    }
    __syncthreads();
    if (t == 0) s_nBig = 0;
    __syncthreads();
    for (int c = t; c < M_CELLS; c += GB_THREADS) {
        const int lo = s_start[c], m = s_start[c + 1] - lo;
        if (m > 32) {  // crowded bucket: left to the whole workgroup below (at most M_MAXKP / 32 of them)
            s_big[atomicAdd(&s_nBig, 1)] = c;
            continue;
        }
        for (int i = 1; i < m; i++) {
            const uint32_t x = s_sorted[lo + i];
            int j = i - 1;
            while (j >= 0 && s_sorted[lo + j] > x) {
                s_sorted[lo + j + 1] = s_sorted[lo + j];
                j--;
            }
            s_sorted[lo + j + 1] = x;
        }
    }
    __syncthreads();
    for (int b = 0; b < s_nBig; b++) {  // rank sort through s_key, which is free again
        const int lo = s_start[s_big[b]], m = s_start[s_big[b] + 1] - lo;
        for (int i = t; i < m; i += GB_THREADS) {
            const uint32_t x = s_sorted[lo + i];
            int r = 0;
            for (int j = 0; j < m; j++) r += s_sorted[lo + j] < x;
            s_key[r] = x;
        }
        __syncthreads();
        for (int i = t; i < m; i += GB_THREADS) s_sorted[lo + i] = s_key[i];
        __syncthreads();
    }
    for (int c = t; c <= M_CELLS; c += GB_THREADS) cellStart[c] = s_start[c];
    __syncthreads();
    const int nIn = s_start[M_CELLS];
    uint32_t* gS = B.gSorted + (int64_t)pair * stride2;
    float2* gXY = B.gXY + (int64_t)pair * stride2;
    uint8_t* gO = B.gOct + (int64_t)pair * stride2;
    for (int p = t; p < nIn; p += GB_THREADS) {
        const uint32_t key = s_sorted[p];
        const orb_keypoint kp = k2[key & 0xFFFFu];
        gS[p] = key;
        gXY[p] = make_float2(kp.x, kp.y);
        gO[p] = (uint8_t)(kp.octave < 0 ? 255 : (kp.octave > 254 ? 254 : kp.octave));
    }
}

// Kernel 2.  PROJECTED = false: queries are the first frame's keypoints at octave 0, window centre
// vbPrevMatched, fixed radius, level range [octave, octave] (src/ORBmatcher.cc:41-49).
// PROJECTED = true: queries are projected map points with their own radius and level range (:323-336).
template <bool PROJECTED>
__global__ __launch_bounds__(256) void k_match_lists(const orb_keypoint* __restrict__ k1all,
                                                     const uint8_t* __restrict__ d1all, const int* __restrict__ n1s,
                                                     int nqConst, int stride1, const float* __restrict__ prevAll,
                                                     float window, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ radius,
                                                     const int* __restrict__ minLevel, const int* __restrict__ maxLevel,
                                                     const uint8_t* __restrict__ d2all, int stride2, GridGeom G,
                                                     unsigned quirks, MatchBuffers B, int* __restrict__ error) {
    const int pair = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform per wavefront
    const int nq = PROJECTED ? nqConst : n1s[pair];
    if (q >= nq || q >= stride1) return;
    int* counts = B.counts + (int64_t)pair * stride1;
    float x, y, r;
    int lmin, lmax;
    bool active;
    if (PROJECTED) {
        x = u[q];
        y = v[q];
        r = radius[q];
        lmin = minLevel[q];
        lmax = maxLevel[q];
        active = !(x < G.minX || x > G.maxX) && !(y < G.minY || y > G.maxY);  // :323-326
    } else {
        const orb_keypoint kp1 = k1all[(int64_t)pair * stride1 + q];
        x = prevAll[((int64_t)pair * stride1 + q) * 2];
        y = prevAll[((int64_t)pair * stride1 + q) * 2 + 1];
        r = window;
        lmin = lmax = kp1.octave;
        active = !(kp1.octave > 0);  // :41-44
    }
    const bool excl = (quirks & ORB_Q_GRID_EXCLUSIVE) != 0, circ = (quirks & ORB_Q_WINDOW_CIRCULAR) != 0;
    Window w;
    w.x0 = w.x1 = w.y0 = w.y1 = 0;
    if (active) w = window_cells(x, y, r, G, excl);
    if (!active || w.x1 <= w.x0 || w.y1 <= w.y0) {
        if (lane == 0) counts[q] = 0;
        return;
    }
    const int* cellStart = B.cellStart + (int64_t)pair * (M_CELLS + 1);
    const uint32_t* gS = B.gSorted + (int64_t)pair * stride2;
    const float2* gXY = B.gXY + (int64_t)pair * stride2;
    const uint8_t* gO = B.gOct + (int64_t)pair * stride2;
    const uint8_t* d2 = d2all + (int64_t)pair * stride2 * 32;
    uint32_t* list = B.lists + ((int64_t)pair * stride1 + q) * B.capQ;
    const int lo = cellStart[w.x0 * ORB_GRID_ROWS + w.y0];
    const int hi = cellStart[(w.x1 - 1) * ORB_GRID_ROWS + w.y1];
    const uint8_t* qd = d1all + ((int64_t)pair * stride1 + q) * 32;
    const uint4 a0 = ((const uint4*)qd)[0], a1 = ((const uint4*)qd)[1];
    const bool checkLevels = lmin >= 0 || lmax >= 0;
    const float r2 = __fmul_rn(r, r);
    // pass 0 fills the query's segment; a query with more than capQ candidates (rare) then takes room for the
    // rest from the pool and walks its window once more for them
    int n = 0, ovf = 0;
    for (int pass = 0; pass < 2; pass++) {
        n = 0;
        for (int base = lo; base < hi; base += 64) {
            const int p = base + lane;
            bool ok = false;
            uint32_t key = 0;
            if (p < hi) {
                key = gS[p];
                const int iy = (int)(key >> 16) % ORB_GRID_ROWS;
                ok = iy >= w.y0 && iy < w.y1;
                if (ok && checkLevels) {
                    const int o = gO[p];
                    if (o < lmin) ok = false;
                    if (lmax >= 0 && o > lmax) ok = false;
                }
                if (ok) {
                    const float2 c = gXY[p];
                    const float dx = __fsub_rn(c.x, x), dy = __fsub_rn(c.y, y);
                    ok = circ ? (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < r2) : (fabsf(dx) < r && fabsf(dy) < r);
                }
            }
            const unsigned long long b = __ballot(ok);
            if (ok) {
                const int pos = n + __popcll(b & ((1ull << lane) - 1ull));
                if (pass == 0 ? pos < B.capQ : pos >= B.capQ) {
                    const uint32_t dist = (uint32_t)hamming_rows(a0, a1, d2 + (int64_t)(key & 0xFFFFu) * 32);
                    if (pass == 0) list[pos] = (dist << 16) | (uint32_t)p;
                    else B.pool[(unsigned)ovf + (unsigned)(pos - B.capQ)] = (dist << 16) | (uint32_t)p;
                }
            }
            n += __popcll(b);
        }
        if (pass == 1 || n <= B.capQ) break;
        unsigned start = 0;
        if (lane == 0) start = atomicAdd(B.poolUsed, (unsigned)(n - B.capQ));
        start = __shfl(start, 0, 64);
        if (start + (unsigned)(n - B.capQ) > B.poolCap || start + (unsigned)(n - B.capQ) < start) {
            // The list is cut at capQ entries, and never silently: the error word makes the call report
            // ORB_ERR_CAPACITY with what it would have needed, and the query's pool offset -1 makes the resolve
            // kernel give the pair a match count of -1.
            if (lane == 0) atomicOr(error, M_ERR_POOL);
            n = B.capQ;
            ovf = -1;
            break;
        }
        ovf = (int)start;
    }
    if (lane == 0) {
        counts[q] = n;
        B.ovfStart[(int64_t)pair * stride1 + q] = ovf;
    }
}

// Best candidate of every query on its own, for the searches in which one query's outcome does not
// change what the next may take: the candidate loops of ORBmatcher::Fuse (both overloads,
// include/ORBmatcher.h:163-184) and ORBmatcher::SearchBySim3 (:146-160).  One wavefront per query:
// window query, level range, optionally the reprojection-error gate of Fuse (squared pixel error
// times the candidate level's inverse sigma^2 above chi2Max rejects), smallest distance, first
// candidate in window order among equals; kept when the distance is at most `th`.
__global__ __launch_bounds__(256) void k_match_best(const float* __restrict__ u, const float* __restrict__ v,
                                                    const float* __restrict__ radius, const int* __restrict__ minLevel,
                                                    const int* __restrict__ maxLevel, const uint8_t* __restrict__ qdesc,
                                                    int nq, const uint8_t* __restrict__ d2,
                                                    const float* __restrict__ invSigma2, int nSigma, double chi2Max, int th,
                                                    GridGeom G, unsigned quirks, MatchBuffers B,
                                                    int* __restrict__ bestIdx, int* __restrict__ bestDist) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = u[q], y = v[q], r = radius[q];
    const int lmin = minLevel[q], lmax = maxLevel[q];
    const bool excl = (quirks & ORB_Q_GRID_EXCLUSIVE) != 0, circ = (quirks & ORB_Q_WINDOW_CIRCULAR) != 0;
    const Window w = window_cells(x, y, r, G, excl);
    uint32_t best = 0xFFFFFFFFu;
    if (w.x1 > w.x0 && w.y1 > w.y0) {
        const int lo = B.cellStart[w.x0 * ORB_GRID_ROWS + w.y0];
        const int hi = B.cellStart[(w.x1 - 1) * ORB_GRID_ROWS + w.y1];
        const uint4 a0 = ((const uint4*)(qdesc + (int64_t)q * 32))[0], a1 = ((const uint4*)(qdesc + (int64_t)q * 32))[1];
        const bool checkLevels = lmin >= 0 || lmax >= 0;
        const float r2 = __fmul_rn(r, r);
        for (int p = lo + lane; p < hi; p += 64) {
            const uint32_t key = B.gSorted[p];
            const int iy = (int)(key >> 16) % ORB_GRID_ROWS;
            if (iy < w.y0 || iy >= w.y1) continue;
            const int o = B.gOct[p];
            if (checkLevels && (o < lmin || (lmax >= 0 && o > lmax))) continue;
            const float2 c = B.gXY[p];
            const float dx = __fsub_rn(c.x, x), dy = __fsub_rn(c.y, y);
            const float e2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            if (circ ? !(e2 < r2) : !(fabsf(dx) < r && fabsf(dy) < r)) continue;
            if (invSigma2) {
                if (o >= nSigma) continue;  // no sigma for that level
                if ((double)__fmul_rn(e2, invSigma2[o]) > chi2Max) continue;
            }
            const uint32_t dist = (uint32_t)hamming_rows(a0, a1, d2 + (int64_t)(key & 0xFFFFu) * 32);
            best = min(best, (dist << 16) | (uint32_t)p);
        }
    }
    best = wave_min_u32(best);
    if (lane == 0) {
        const int dist = best == 0xFFFFFFFFu ? 256 : (int)(best >> 16);  // the searches start from 256 with a strict '<'
        const bool ok = dist < 256 && dist <= th;
        bestIdx[q] = ok ? (int)(B.gSorted[best & 0xFFFFu] & 0xFFFFu) : -1;
        bestDist[q] = dist;
    }
}

// Kernel 3.  MODE 0: window search (SearchForInitialization).  MODE 1: projected search, best
// candidate only (SearchByProjection(Frame&, const Frame&, ..)).  MODE 2: projected search with the
// best / second-best rule of the local-map search (SearchByProjection(Frame&, const vector<MapPoint*>&,
// th), include/ORBmatcher.h:24-39): the ratio test applies only when both candidates lie on the same
// pyramid level.
struct ResolveLds {
    uint32_t tile[2][64][128];      // first 128 list entries of 64 queries, double buffered
    int metaI1[2][64];              // the 64 queries of a chunk: index, list length, "map point has observations"
    int metaCnt[2][64];
    int metaObs[2][64];
    uint16_t idx[M_MAXKP];          // keypoint index by sorted position
    // State of the searched frame's keypoints, by sorted position (the lists name positions).  Low half:
    // vMatchedDistance (0xFFFF = INT_MAX) for the window search; occupancy flag in bits 0-7 and (MODE 2)
    // the keypoint's octave in bits 8-15 for the projected ones.  High half: owner (vnMatches21 / the
    // query whose map point the keypoint holds).
    uint32_t so[M_MAXKP];
    int hist[ORB_HISTO + 1];  // entry ORB_HISTO: rotations rot_bin could not place
    int wcount[4];
    int nmatches;
    int cut;  // some query's candidate list was cut short (pool overrun): the pair's results are void
};

template <int MODE>
__global__ __launch_bounds__(256) void k_match_resolve(const orb_keypoint* __restrict__ k1all,
                                                       const int* __restrict__ n1s, int nqConst, int stride1,
                                                       const orb_keypoint* __restrict__ k2all,
                                                       const int* __restrict__ n2s, int n2const, int stride2,
                                                       float* __restrict__ prevAll, int* __restrict__ m12All,
                                                       uint8_t* __restrict__ binOfAll, const float* __restrict__ qangle,
                                                       const uint8_t* __restrict__ qobs,
                                                       const uint8_t* __restrict__ occupied2, int* __restrict__ match2,
                                                       float nnratio, int thHigh, int checkOri, MatchBuffers B,
                                                       int* __restrict__ nmatchesOut, int* __restrict__ error) {
    __shared__ ResolveLds S;
    constexpr bool PROJECTED = MODE != 0;
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const orb_keypoint* k1 = PROJECTED ? nullptr : k1all + (int64_t)pair * stride1;
    const orb_keypoint* k2 = k2all + (int64_t)pair * stride2;
    float* prev = PROJECTED ? nullptr : prevAll + (int64_t)pair * stride1 * 2;
    int* m12 = PROJECTED ? nullptr : m12All + (int64_t)pair * stride1;
    uint8_t* binOf = binOfAll + (int64_t)pair * stride1;
    const int* counts = B.counts + (int64_t)pair * stride1;
    int* accepted = B.accepted + (int64_t)pair * stride1;
    const uint32_t* lists = B.lists + (int64_t)pair * stride1 * B.capQ;
    const int* cellStart = B.cellStart + (int64_t)pair * (M_CELLS + 1);
    const uint32_t* gS = B.gSorted + (int64_t)pair * stride2;
    const int nq = PROJECTED ? nqConst : n1s[pair];
    const int n2 = n2s ? n2s[pair] : n2const;
    if (nq > stride1 || nq > 32767 || n2 > stride2 || n2 > M_MAXKP) {
        if (t == 0) {
            atomicOr(error, M_ERR_LIMIT);
            nmatchesOut[pair] = 0;
        }
        return;
    }
    const int nIn = cellStart[M_CELLS];
    for (int i = t; i < nq; i += 256) {
        if (!PROJECTED) m12[i] = -1;
        binOf[i] = 0xFF;
        accepted[i] = -1;
    }
    if (PROJECTED)
        for (int i = t; i < n2; i += 256) match2[i] = -1;
    for (int p = t; p < nIn; p += 256) {
        const int i2 = (int)(gS[p] & 0xFFFFu);
        S.idx[p] = (uint16_t)i2;
        if (PROJECTED)  // owner -1
            S.so[p] = 0xFFFF0000u | (MODE == 2 ? (uint32_t)B.gOct[(int64_t)pair * stride2 + p] << 8 : 0u) |
                      (occupied2 && occupied2[i2] ? 1u : 0u);
        else S.so[p] = 0xFFFFFFFFu;                                            // distance INT_MAX, owner -1
    }
    if (t <= ORB_HISTO) S.hist[t] = 0;
    if (t == 0) {
        S.nmatches = 0;
        S.cut = 0;
    }
    __syncthreads();

    // Queries with a non-empty list, in index order (the others cannot match and change nothing).
    int* order = B.order + (int64_t)pair * stride1;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nActive = 0;
    for (int b0 = 0; b0 < nq; b0 += 256) {
        const int i = b0 + t;
        const bool act = i < nq && counts[i] > 0;
        if (act && B.ovfStart[(int64_t)pair * stride1 + i] < 0) S.cut = 1;  // (benign race: every writer stores 1)
        const unsigned long long m = __ballot(act);
        if (lane == 0) S.wcount[wave] = __popcll(m);
        __syncthreads();
        int off = nActive;
        for (int w2 = 0; w2 < wave; w2++) off += S.wcount[w2];
        if (act) order[off + __popcll(m & below)] = i;
        nActive += S.wcount[0] + S.wcount[1] + S.wcount[2] + S.wcount[3];
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();

    // Stages chunk `base` (64 consecutive active queries) into half `buf` of the tile: this wavefront
    // takes queries qFirst, qFirst + qStep, ...; all of its loads are in flight at once.
    auto stage = [&](int buf, int base, int qFirst, int qStep, bool meta) {
        const int slot = base + lane;
        const int i1L = slot < nActive ? order[slot] : -1;
        const int cntL = i1L >= 0 ? counts[i1L] : 0;
        if (meta) {
            S.metaI1[buf][lane] = i1L;
            S.metaCnt[buf][lane] = cntL;
            S.metaObs[buf][lane] = (PROJECTED && qobs && i1L >= 0) ? qobs[i1L] : 1;
        }
        uint32_t e[22], f[22];
#pragma unroll
        for (int j = 0; j < 22; j++) {
            const int qq = qFirst + j * qStep;
            e[j] = f[j] = 0xFFFFFFFFu;
            if (qq < 64) {
                const int c = __shfl(cntL, qq, 64), i1 = __shfl(i1L, qq, 64);
                if (lane < c) e[j] = lists[(int64_t)i1 * B.capQ + lane];
                if (lane + 64 < c) f[j] = lists[(int64_t)i1 * B.capQ + lane + 64];
            }
        }
#pragma unroll
        for (int j = 0; j < 22; j++) {
            const int qq = qFirst + j * qStep;
            if (qq < 64) {
                S.tile[buf][qq][lane] = e[j];
                S.tile[buf][qq][lane + 64] = f[j];
            }
        }
    };

    // Wavefront 0 resolves chunk k from one half of the tile while wavefronts 1-3 stage chunk k + 1
    // into the other half.
    int nmatches = 0;  // wave 0
    const int nChunks = (nActive + 63) >> 6;
    if (nChunks > 0) stage(0, 0, wave, 4, wave == 0);
    __syncthreads();
    for (int k = 0; k < nChunks; k++) {
        const int base = k * 64, buf = k & 1;
        if (wave != 0) {
            if (k + 1 < nChunks) stage(buf ^ 1, base + 64, wave - 1, 3, wave == 1);
        } else {
            const int i1L = S.metaI1[buf][lane], cntL = S.metaCnt[buf][lane], obsL = S.metaObs[buf][lane];
            const int nHere = nActive - base < 64 ? nActive - base : 64;
            uint32_t e0 = S.tile[buf][0][lane], e1 = S.tile[buf][0][lane + 64];
            int accL = -1;  // lane qq: keypoint the chunk's query qq was matched to
#pragma nounroll
            for (int qq = 0; qq < nHere; qq++) {
                const int i1 = __builtin_amdgcn_readlane(i1L, qq);
                const int cnt = __builtin_amdgcn_readlane(cntL, qq);
                const int obs = __builtin_amdgcn_readlane(obsL, qq);
                // the next query's entries do not depend on this one's outcome: fetch them now
                const int qn = qq + 1 < 64 ? qq + 1 : 63;
                const uint32_t n0 = S.tile[buf][qn][lane], n1 = S.tile[buf][qn][lane + 64];
                // this lane's two smallest surviving (distance << 16 | position) keys
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
                auto consider = [&](uint32_t e) {
                    const uint32_t st = S.so[e & 0xFFFFu] & 0xFFFFu;
                    const bool skip = PROJECTED ? ((st & 0xFFu) != 0) : (st <= (e >> 16));  // :345-347 / :63-64
                    if (!skip) {
                        k2 = min(k2, max(k1, e));
                        k1 = min(k1, e);
                    }
                };
                if (lane < cnt) consider(e0);
                if (lane + 64 < cnt) consider(e1);
                if (cnt > 128) {
                    const uint32_t* seg = lists + (int64_t)i1 * B.capQ;
                    const int ovf = B.ovfStart[(int64_t)pair * stride1 + i1];
                    for (int c = lane + 128; c < cnt; c += 64) consider(list_entry(B, seg, ovf, c));
                }
                if (MODE == 1) k1 = wave_min_u32(k1);
                else wave_top2_u32(k1, k2);
                e0 = n0;
                e1 = n1;
                if (k1 == 0xFFFFFFFFu) continue;  // nothing survived: bestDist stays at its initial value and fails the test
                const int bestDist = (int)(k1 >> 16);
                const int bestPos = (int)(k1 & 0xFFFFu);
                bool accept;
                if (MODE == 1) {
                    accept = bestDist < 256 && bestDist <= thHigh;  // bestDist starts at 256 with strict '<', :339,357,363
                } else if (MODE == 2) {
                    // both distances start at 256 and both levels at -1; a candidate at distance 256 replaces neither
                    accept = bestDist < 256 && bestDist <= thHigh;
                    const int dist2 = (k2 == 0xFFFFFFFFu) ? 256 : (int)(k2 >> 16);
                    if (accept && dist2 < 256) {
                        const uint32_t lv1 = (S.so[bestPos] >> 8) & 0xFFu, lv2 = (S.so[k2 & 0xFFFFu] >> 8) & 0xFFu;
                        if (lv1 == lv2 && (float)bestDist > __fmul_rn(nnratio, (float)dist2)) accept = false;
                    }
                } else {
                    // second best = smallest distance among everything but the winning entry
                    const float bestDist2 = (k2 == 0xFFFFFFFFu) ? (float)INT_MAX : (float)(int)(k2 >> 16);
                    accept = bestDist <= M_TH_LOW && (float)bestDist < __fmul_rn(bestDist2, nnratio);  // :78-80
                }
                if (accept) {
                    // No global memory is touched here: the owner of every keypoint lives in LDS (vnMatches12 /
                    // the map-point assignment follow from it after the loop), the acceptance in a register.
                    const int bestIdx2 = S.idx[bestPos];
                    if (lane == qq) accL = bestIdx2;
                    if (lane == 0)
                        S.so[bestPos] = (PROJECTED ? ((S.so[bestPos] & 0xFF00u) | (obs ? 1u : 0u)) : (uint32_t)bestDist) |
                                        ((uint32_t)i1 << 16);
                    nmatches++;
                    wave_lds_order();
                }
            }
            if (lane < nHere) accepted[i1L] = accL;
        }
        __syncthreads();
    }
    // vnMatches12 (window search): query i1 holds keypoint i2 exactly when i2's owner is i1 at the end, a
    // stolen match having lost its owner (:82-86).  Projected search: the keypoint holds the map point of
    // its last owner; every acceptance counts (:364-366).
    if (t == 0) S.nmatches = PROJECTED ? nmatches : 0;
    __threadfence_block();
    __syncthreads();
    {
        int mine = 0;
        for (int p = t; p < nIn; p += 256) {
            const int own = (int)(int16_t)(S.so[p] >> 16);
            if (own < 0) continue;
            if (PROJECTED) {
                match2[S.idx[p]] = own;
            } else {
                m12[own] = S.idx[p];
                mine++;
            }
        }
        if (!PROJECTED && mine) atomicAdd(&S.nmatches, mine);
    }
    __threadfence_block();
    __syncthreads();
    nmatches = S.nmatches;

    if (checkOri) {  // :88-99 / :371-383 after the fact: one histogram entry per acceptance
        for (int i = t; i < nq; i += 256) {
            const int i2 = accepted[i];
            if (i2 < 0) continue;
            const int bin = rot_bin(PROJECTED ? qangle[i] : k1[i].angle, k2[i2].angle);
            binOf[i] = (uint8_t)bin;
            atomicAdd(&S.hist[bin], 1);
        }
        __syncthreads();
        int ind1, ind2, ind3;
        three_maxima(S.hist, ind1, ind2, ind3);
        __syncthreads();
        if (t == 0) S.nmatches = 0;
        __syncthreads();
        int removed = 0;
        for (int i = t; i < nq; i += 256) {
            const int bin = binOf[i];
            if (bin == 0xFF || bin == ind1 || bin == ind2 || bin == ind3) continue;
            if (PROJECTED) {
                match2[accepted[i]] = -2;  // :400-401, once per list entry (-2: assigned, then cleared)
                removed++;
            } else if (m12[i] >= 0) {  // :122-126 (a stolen match is already -1)
                m12[i] = -1;
                removed++;
            }
        }
        if (removed) atomicAdd(&S.nmatches, removed);
        __syncthreads();
        nmatches -= S.nmatches;
    }
    if (!PROJECTED) {
        __threadfence_block();
        __syncthreads();
        for (int i = t; i < nq; i += 256) {  // :131-133
            const int m = m12[i];
            if (m >= 0) {
                prev[2 * i] = k2[m].x;
                prev[2 * i + 1] = k2[m].y;
            }
        }
    }
    if (t == 0) nmatchesOut[pair] = S.cut ? -1 : nmatches;
}

// Kernel 3, 64 queries at a time.  MODE as in k_match_resolve: 0 window search (SearchForInitialization),
// 1 projected search, best candidate only, 2 projected search with the local-map rule.
//
// Queries interact only through the keypoints they are matched to.  Window search: an acceptance lowers
// vMatchedDistance of one keypoint (src/ORBmatcher.cc:63-64, 89), which hides that keypoint from later queries whose
// distance to it is not smaller.  Projected searches: an acceptance by a map point with observations occupies the
// keypoint (:345-347) and hides it from every later query.  The next 64 queries with candidates (a "run") are
// resolved together, 16 lanes per query, by iteration: every query is evaluated against the state the runs before
// left AND the claims the earlier queries of its own run made in the previous round (a claim = "query a accepts
// keypoint p"; what a later query k sees of p is the claim of the last claimer before k, exactly what the state
// would hold when the walk in order reaches k); queries whose outcome changed post their new claim, and the rounds
// go on until nothing changes.  Query 0 of the run sees no claims and is final after round 1, query 1 after round 2
// at the latest, and so on: the iteration ends, in at most 65 rounds, at the outcome of the walk in order (its only
// fixed point).  On ordinary frames two or three rounds do.  Then the run is committed.  State per keypoint, by
// sorted position -- window search: distance << 16 | owner, so that an atomic minimum keeps the latest (smallest)
// accepted distance; projected: the owner (the last query that took the keypoint: an atomic maximum) and an
// occupancy flag next to the keypoint's octave.  The lists of the next run are fetched while the rounds of the
// current one go on.  Everything after the walk (vnMatches12 / the map-point assignment from the final owners,
// rotation histogram over all acceptances, three maxima, removal, vbPrevMatched) is as in k_match_resolve.
#define M_RUN 64            // queries resolved together
#define M_PAR_MAXKP 4096    // keypoints of the searched frame, and queries with candidates, this kernel's LDS holds
#define M_TILE_W 128        // list entries per query held in LDS (the rest is read from the list itself)
#define M_TILE_PITCH 129
#define M_PAR_THREADS 1024  // sixteen lanes per query of a run
struct ResolveParLds {
    unsigned long long claims[M_PAR_MAXKP];  // by sorted position: bit k = query k of the run claims the keypoint (projected: and occupies it)
    uint32_t so[M_PAR_MAXKP];                // window: vMatchedDistance << 16 | vnMatches21 (0xFFFFFFFF: INT_MAX, -1); projected: owner + 1 (0: none)
    uint16_t occ[M_PAR_MAXKP];               // projected: occupancy flag (bit 0), octave << 8
    uint16_t idx[M_PAR_MAXKP];               // keypoint index by sorted position
    uint16_t qI1[M_PAR_MAXKP], qCnt[M_PAR_MAXKP];  // the queries with candidates, in order: index, list length
    uint32_t tile[2][M_RUN * M_TILE_PITCH];  // first entries of the run's lists, double buffered
    uint32_t k1[M_RUN];                      // a query's outcome of the previous round: best key, 0xFFFFFFFF = no acceptance
    int hist[ORB_HISTO + 1];
    int wcount[M_PAR_THREADS / 64];
    int changed[2];
    int nmatches, cut;
};

// Workgroup barrier for code that communicates through LDS only: waits for this wavefront's LDS operations, not for
// its loads from memory (__syncthreads does: a run's rounds would each wait for the next run's lists, which are
// fetched meanwhile precisely so that nobody waits for them).
static __device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (compiler ordering; LDS operations of a wavefront execute in order)
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ __launch_bounds__(M_PAR_THREADS) void k_match_resolve_par(const orb_keypoint* __restrict__ k1all,
                                                              const int* __restrict__ n1s, int nqConst, int stride1,
                                                              const orb_keypoint* __restrict__ k2all,
                                                              const int* __restrict__ n2s, int n2const, int stride2,
                                                              float* __restrict__ prevAll, int* __restrict__ m12All,
                                                              uint8_t* __restrict__ binOfAll, const float* __restrict__ qangle,
                                                              const uint8_t* __restrict__ qobs,
                                                              const uint8_t* __restrict__ occupied2, int* __restrict__ match2,
                                                              float nnratio, int thHigh, int checkOri,
                                                              MatchBuffers B, int* __restrict__ nmatchesOut,
                                                              int* __restrict__ error) {
    __shared__ ResolveParLds S;
    constexpr bool PROJECTED = MODE != 0;
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const orb_keypoint* k1 = PROJECTED ? nullptr : k1all + (int64_t)pair * stride1;
    const orb_keypoint* k2 = k2all + (int64_t)pair * stride2;
    float* prev = PROJECTED ? nullptr : prevAll + (int64_t)pair * stride1 * 2;
    int* m12 = PROJECTED ? nullptr : m12All + (int64_t)pair * stride1;
    uint8_t* binOf = binOfAll + (int64_t)pair * stride1;
    const int* counts = B.counts + (int64_t)pair * stride1;
    int* accepted = B.accepted + (int64_t)pair * stride1;
    const uint32_t* lists = B.lists + (int64_t)pair * stride1 * B.capQ;
    const int* cellStart = B.cellStart + (int64_t)pair * (M_CELLS + 1);
    const uint32_t* gS = B.gSorted + (int64_t)pair * stride2;
    const int nq = PROJECTED ? nqConst : n1s[pair];
    const int n2 = n2s ? n2s[pair] : n2const;
    if (nq > stride1 || nq > M_PAR_MAXKP || n2 > stride2 || n2 > M_PAR_MAXKP) {  // (the host picks this kernel by the strides)
        if (t == 0) {
            atomicOr(error, M_ERR_LIMIT);
            nmatchesOut[pair] = 0;
        }
        return;
    }
    const int nIn = cellStart[M_CELLS];
    for (int i = t; i < nq; i += M_PAR_THREADS) {
        if (!PROJECTED) m12[i] = -1;
        binOf[i] = 0xFF;
        accepted[i] = -1;
    }
    if (PROJECTED)
        for (int i = t; i < n2; i += M_PAR_THREADS) match2[i] = -1;
    for (int p = t; p < nIn; p += M_PAR_THREADS) {
        const int i2 = (int)(gS[p] & 0xFFFFu);
        S.idx[p] = (uint16_t)i2;
        S.claims[p] = 0ull;
        if (PROJECTED) {
            S.so[p] = 0u;
            S.occ[p] = (uint16_t)((MODE == 2 ? (uint32_t)B.gOct[(int64_t)pair * stride2 + p] << 8 : 0u) | (occupied2 && occupied2[i2] ? 1u : 0u));
        } else {
            S.so[p] = 0xFFFFFFFFu;
        }
    }
    if (t <= ORB_HISTO) S.hist[t] = 0;
    if (t == 0) {
        S.nmatches = 0;
        S.cut = 0;
        S.changed[0] = S.changed[1] = 0;
    }
    __syncthreads();

    // queries with a non-empty list, in index order (the others cannot match and change nothing)
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nActive = 0;
    for (int b0 = 0; b0 < nq; b0 += M_PAR_THREADS) {
        const int i = b0 + t;
        const int cnt = i < nq ? counts[i] : 0;
        const bool act = cnt > 0;
        if (act && B.ovfStart[(int64_t)pair * stride1 + i] < 0) S.cut = 1;
        const unsigned long long m = __ballot(act);
        if (lane == 0) S.wcount[wave] = __popcll(m);
        __syncthreads();
        int off = nActive;
        int all = 0;
        for (int w2 = 0; w2 < M_PAR_THREADS / 64; w2++) {
            if (w2 < wave) off += S.wcount[w2];
            all += S.wcount[w2];
        }
        if (act) {
            const int slot = off + __popcll(m & below);
            S.qI1[slot] = (uint16_t)i;
            S.qCnt[slot] = (uint16_t)(cnt < 0xFFFF ? cnt : 0xFFFF);
        }
        nActive += all;
        __syncthreads();
    }

    // the lists of a run -> one half of the tile: thread = (query t / 16, every sixteenth entry); sixteen lanes, one
    // DPP row, per query, and four wavefronts per SIMD to cover each other's LDS round trips
    const int kq = t >> 4, sub = t & 15;
    uint32_t held[M_TILE_W / 16];
    auto fetch = [&](int base) {
        const int a = base + kq;
        const bool in = a < nActive;
        const int cnt = in ? (int)S.qCnt[a] : 0;
        const uint32_t* seg = lists + (int64_t)(in ? (int)S.qI1[a] : 0) * B.capQ;
#pragma unroll
        for (int j = 0; j < M_TILE_W / 16; j++) {
            const int c = sub + 16 * j;
            held[j] = (c < cnt && c < B.capQ) ? seg[c] : 0xFFFFFFFFu;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < M_TILE_W / 16; j++) S.tile[buf][kq * M_TILE_PITCH + sub + 16 * j] = held[j];
    };
    if (nActive > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();

    int flip = 0;  // which of the two `changed` words the current round uses
#ifdef MATCH_PROFILE
    int profRuns = 0, profRounds = 0;
    const long long profT0 = wall_clock64();
#endif
    for (int base = 0, buf = 0; base < nActive; base += M_RUN, buf ^= 1) {
        const int a = base + kq;
        const bool in = a < nActive;
        const int i1 = in ? (int)S.qI1[a] : 0;
        int cnt = in ? (int)S.qCnt[a] : 0;
        if (cnt == 0xFFFF) cnt = counts[i1];
        const int obs = (PROJECTED && qobs && in) ? (int)qobs[i1] : 1;  // projected: the map point has observations
        if (base + M_RUN < nActive) fetch(base + M_RUN);  // in flight during the rounds below
        const uint32_t* row = &S.tile[buf][kq * M_TILE_PITCH];
        const uint32_t* seg = lists + (int64_t)i1 * B.capQ;
        const int ovf = (in && cnt > M_TILE_W) ? B.ovfStart[(int64_t)pair * stride1 + i1] : 0;
        const unsigned long long earlier = kq == 0 ? 0ull : (~0ull >> (64 - kq));
        if (sub == 15) S.k1[kq] = 0xFFFFFFFFu;
        uint32_t mine = 0xFFFFFFFFu;  // this query's claim as posted (best key when accepted)
        lds_barrier();
        for (;;) {
            // evaluate against the committed state and the earlier queries' claims of the previous round
            uint32_t a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
            auto consider = [&](uint32_t e) {
                const uint32_t p = e & 0xFFFFu, d = e >> 16;
                const unsigned long long m = S.claims[p] & earlier;
                bool hidden;
                if (PROJECTED) {
                    hidden = (S.occ[p] & 1u) != 0 || m != 0ull;  // :345-347 (claims are only posted by map points with observations)
                } else {
                    hidden = (S.so[p] >> 16) <= d;  // :63-64
                    if (m) hidden = hidden || (S.k1[63 - __clzll(m)] >> 16) <= d;  // the last claimer before this query
                }
                if (!hidden) {
                    a2 = min(a2, max(a1, e));
                    a1 = min(a1, e);
                }
            };
            const int inTile = cnt < M_TILE_W ? cnt : M_TILE_W;
            for (int c = sub; c < inTile; c += 16) consider(row[c]);
            for (int c = M_TILE_W + sub; c < cnt; c += 16) consider(list_entry(B, seg, ovf, c));
            dpp_top2_step<0x111, 0xf>(a1, a2);
            dpp_top2_step<0x112, 0xf>(a1, a2);
            dpp_top2_step<0x114, 0xf>(a1, a2);
            dpp_top2_step<0x118, 0xf>(a1, a2);  // lane 15 of the row holds the query's two smallest keys
            uint32_t now = 0xFFFFFFFFu;
            if (a1 != 0xFFFFFFFFu) {
                const int bestDist = (int)(a1 >> 16);
                bool accept;
                if (MODE == 1) {
                    accept = bestDist < 256 && bestDist <= thHigh;  // bestDist starts at 256 with strict '<', :339,357,363
                } else if (MODE == 2) {
                    // both distances start at 256 and both levels at -1; a candidate at distance 256 replaces neither
                    accept = bestDist < 256 && bestDist <= thHigh;
                    const int dist2 = (a2 == 0xFFFFFFFFu) ? 256 : (int)(a2 >> 16);
                    if (accept && dist2 < 256) {
                        const uint32_t lv1 = S.occ[a1 & 0xFFFFu] >> 8, lv2 = S.occ[a2 & 0xFFFFu] >> 8;
                        if (lv1 == lv2 && (float)bestDist > __fmul_rn(nnratio, (float)dist2)) accept = false;
                    }
                } else {
                    const float bestDist2 = (a2 == 0xFFFFFFFFu) ? (float)INT_MAX : (float)(int)(a2 >> 16);
                    accept = bestDist <= M_TH_LOW && (float)bestDist < __fmul_rn(bestDist2, nnratio);  // :78-80
                }
                if (accept) now = a1;
            }
            lds_barrier();  // every query has read the claims of the previous round
            if (sub == 15 && in && now != mine) {
                // (projected: a map point without observations takes the keypoint but does not occupy it: no claim)
                const bool claims = !PROJECTED || obs != 0;
                if (claims && mine != 0xFFFFFFFFu) atomicAnd(&S.claims[mine & 0xFFFFu], ~(1ull << kq));
                if (claims && now != 0xFFFFFFFFu) atomicOr(&S.claims[now & 0xFFFFu], 1ull << kq);
                S.k1[kq] = now;
                S.changed[flip] = 1;
            }
            mine = now;
            if (t == 0) S.changed[flip ^ 1] = 0;
            lds_barrier();
            const int again = S.changed[flip];
            flip ^= 1;
#ifdef MATCH_PROFILE
            profRounds++;
#endif
            if (!again) break;
        }
#ifdef MATCH_PROFILE
        profRuns++;
#endif
        // commit the run; take its claims back
        if (sub == 15 && in && mine != 0xFFFFFFFFu) {
            const uint32_t pos = mine & 0xFFFFu;
            if (PROJECTED) {
                atomicMax(&S.so[pos], (uint32_t)i1 + 1u);  // the keypoint holds the map point of its last taker
                if (obs) S.occ[pos] |= 1u;                  // (several takers of one keypoint: at most the last has observations; all write 1)
                atomicAdd(&S.nmatches, 1);                  // every acceptance counts (:364-366)
            } else {
                atomicMin(&S.so[pos], ((mine >> 16) << 16) | (uint32_t)i1);
            }
            accepted[i1] = (int)S.idx[pos];
            S.claims[pos] = 0ull;
        }
        if (base + M_RUN < nActive) stash(buf ^ 1);
        lds_barrier();
    }
#ifdef MATCH_PROFILE
    if (t == 0 && pair == 0)
        printf("resolve mode %d: %d queries with candidates, %d runs, %d rounds, %.1f us in the runs\n", MODE, nActive, profRuns, profRounds,
               (double)(wall_clock64() - profT0) / 100.0);
#endif
    // Window search: query i1 holds keypoint i2 exactly when i2's owner is i1 at the end, a stolen match having lost
    // its owner (:82-86).  Projected search: the keypoint holds the map point of its last taker.
    __threadfence_block();
    __syncthreads();
    {
        int mine = 0;
        for (int p = t; p < nIn; p += M_PAR_THREADS) {
            const uint32_t v = S.so[p];
            if (PROJECTED) {
                if (v != 0u) match2[S.idx[p]] = (int)v - 1;
            } else if (v != 0xFFFFFFFFu) {
                m12[v & 0xFFFFu] = S.idx[p];
                mine++;
            }
        }
        if (!PROJECTED && mine) atomicAdd(&S.nmatches, mine);
    }
    __threadfence_block();
    __syncthreads();
    int nmatches = S.nmatches;
    if (checkOri) {  // :88-99 / :371-383 after the fact: one histogram entry per acceptance
        for (int i = t; i < nq; i += M_PAR_THREADS) {
            const int i2 = accepted[i];
            if (i2 < 0) continue;
            const int bin = rot_bin(PROJECTED ? qangle[i] : k1[i].angle, k2[i2].angle);
            binOf[i] = (uint8_t)bin;
            atomicAdd(&S.hist[bin], 1);
        }
        __syncthreads();
        int ind1, ind2, ind3;
        three_maxima(S.hist, ind1, ind2, ind3);
        __syncthreads();
        if (t == 0) S.nmatches = 0;
        __syncthreads();
        int removed = 0;
        for (int i = t; i < nq; i += M_PAR_THREADS) {
            const int bin = binOf[i];
            if (bin == 0xFF || bin == ind1 || bin == ind2 || bin == ind3) continue;
            if (PROJECTED) {
                match2[accepted[i]] = -2;  // :400-401, once per list entry (-2: assigned, then cleared)
                removed++;
            } else if (m12[i] >= 0) {  // :122-126 (a stolen match is already -1)
                m12[i] = -1;
                removed++;
            }
        }
        if (removed) atomicAdd(&S.nmatches, removed);
        __syncthreads();
        nmatches -= S.nmatches;
    }
    if (!PROJECTED) {
        __threadfence_block();
        __syncthreads();
        for (int i = t; i < nq; i += M_PAR_THREADS) {  // :131-133
            const int m = m12[i];
            if (m >= 0) {
                prev[2 * i] = k2[m].x;
                prev[2 * i + 1] = k2[m].y;
            }
        }
    }
    if (t == 0) nmatchesOut[pair] = S.cut ? -1 : nmatches;
}

// ---- host side ------------------------------------------------------------------------
struct MatchState {
    // Error words (flags, pool entries asked for) of the asynchronous entry point, read by orb_check_last_batch.
    // A host-pointer entry point keeps its own pair inside the block it uploads and downloads, so a pending
    // asynchronous report survives any synchronous call made before it is collected.
    void* err = nullptr;
    // scratch, grown on demand and kept
    void* gSorted = nullptr; void* gXY = nullptr; void* gOct = nullptr; void* cellStart = nullptr;
    void* lists = nullptr; void* counts = nullptr; void* accepted = nullptr; void* order = nullptr;
    void* binOf = nullptr; void* ovfStart = nullptr;
    void* pool = nullptr;   // shared candidate pool (see MatchBuffers)
    size_t poolCap = 0;     // entries
    size_t poolWanted = 0;  // entries the next call should have (set when a call ran out, or by orb_match_reserve_candidates)
    size_t capPairs2 = 0;   // npairs * stride2 the grid arrays hold
    size_t capPairs = 0;    // pairs cellStart holds
    size_t capPairs1 = 0;   // npairs * stride1 the per-query arrays hold
    size_t capLists = 0;    // list entries
    void* blockDev = nullptr;   // staging of the host-pointer entry points, see below
    void* blockHost = nullptr;
    void* blockHostDev = nullptr;  // blockHost as the device addresses it
    size_t blockBytes = 0;
};

void orb_match_state_free(void* p) {
    MatchState* s = (MatchState*)p;
    if (!s) return;
    void* bufs[] = {s->err, s->gSorted, s->gXY, s->gOct, s->cellStart, s->lists, s->counts, s->accepted, s->order, s->binOf,
                    s->ovfStart, s->pool};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (s->blockDev) (void)hipFree(s->blockDev);
    if (s->blockHost) (void)hipHostFree(s->blockHost);
    delete s;
}

// accessors implemented in orb_api.hip
int orb_ctx_device(const orb_ctx* c);
unsigned orb_ctx_quirks(const orb_ctx* c);
int orb_ctx_max_batch(const orb_ctx* c);
void** orb_ctx_match_state(orb_ctx* c);
void* orb_ctx_stream(orb_ctx* c);  // the context's own (non-blocking) stream: the host-pointer entry points run on it

static int grow(void** p, size_t bytes) {
    if (*p) {
        HIP_TRY(hipDeviceSynchronize());  // earlier launches may still use the old buffer
        (void)hipFree(*p);
        *p = nullptr;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        orb_set_error("matcher scratch: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    return ORB_OK;
}

#define M_POOL_DEFAULT ((size_t)4 << 20)   // pool entries (16 MiB) a context starts with

// Makes the scratch large enough for npairs pairs of (stride1 queries, stride2 searched keypoints).
static int match_buffers(orb_ctx* c, int npairs, int stride1, int stride2, MatchState** out, MatchBuffers* B,
                         int** errw = nullptr) {
    void** slot = orb_ctx_match_state(c);
    if (!*slot) {
        MatchState* s = new MatchState();
        hipError_t e = hipMalloc(&s->err, 8);
        if (e == hipSuccess) e = hipMemset(s->err, 0, 8);
        if (e != hipSuccess) {
            delete s;
            orb_set_error("matcher state allocation failed: %s", hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        *slot = s;
    }
    MatchState* s = (MatchState*)*slot;
    int rc;
    const size_t p2 = (size_t)npairs * stride2, p1 = (size_t)npairs * stride1;
    // Every query owns min(stride2, M_LIST_SEG) list entries; what a wide window over a dense frame holds beyond
    // that comes from a pool shared by the call's queries.  The worst case (every window holds every keypoint)
    // would take p1 * stride2 entries - 8.8 GB for 512 pairs of 2000-feature frames; the pool starts at what
    // the worst case needs or M_POOL_DEFAULT entries, whichever is smaller, and a call that runs out reports
    // ORB_ERR_CAPACITY (host-pointer entry points grow the pool and run again instead).
    const int capQ = stride2 < M_LIST_SEG ? stride2 : M_LIST_SEG;
    const size_t nl = p1 * (size_t)capQ;
    const size_t worst = p1 * (size_t)(stride2 - capQ);
    size_t poolNeed = worst < M_POOL_DEFAULT ? worst : M_POOL_DEFAULT;
    if (s->poolWanted > poolNeed) poolNeed = s->poolWanted < worst ? s->poolWanted : worst;
    if (poolNeed > 0xFFFFFFF0ull) poolNeed = 0xFFFFFFF0ull;
    if (poolNeed > s->poolCap) {
        s->poolCap = 0;
        if ((rc = grow(&s->pool, poolNeed * 4))) return rc;
        s->poolCap = poolNeed;
    }
    if (p2 > s->capPairs2) {
        s->capPairs2 = 0;
        if ((rc = grow(&s->gSorted, p2 * 4))) return rc;
        if ((rc = grow(&s->gXY, p2 * 8))) return rc;
        if ((rc = grow(&s->gOct, p2))) return rc;
        s->capPairs2 = p2;
    }
    if ((size_t)npairs > s->capPairs) {
        s->capPairs = 0;
        if ((rc = grow(&s->cellStart, (size_t)npairs * (M_CELLS + 1) * 4))) return rc;
        s->capPairs = npairs;
    }
    if (p1 > s->capPairs1) {
        s->capPairs1 = 0;
        if ((rc = grow(&s->counts, p1 * 4))) return rc;
        if ((rc = grow(&s->accepted, p1 * 4))) return rc;
        if ((rc = grow(&s->order, p1 * 4))) return rc;
        if ((rc = grow(&s->binOf, p1))) return rc;
        if ((rc = grow(&s->ovfStart, p1 * 4))) return rc;
        s->capPairs1 = p1;
    }
    if (nl > s->capLists) {
        s->capLists = 0;
        if ((rc = grow(&s->lists, nl * 4))) return rc;
        s->capLists = nl;
    }
    B->gSorted = (uint32_t*)s->gSorted;
    B->gXY = (float2*)s->gXY;
    B->gOct = (uint8_t*)s->gOct;
    B->cellStart = (int*)s->cellStart;
    B->lists = (uint32_t*)s->lists;
    B->counts = (int*)s->counts;
    B->accepted = (int*)s->accepted;
    B->order = (int*)s->order;
    B->capQ = capQ;
    B->pool = (uint32_t*)s->pool;
    B->ovfStart = (int*)s->ovfStart;
    B->poolCap = (unsigned)s->poolCap;
    // the asynchronous entry points' error words; a host-pointer entry point points these at its own block instead
    B->poolUsed = (unsigned*)s->err + 1;
    if (errw) *errw = (int*)s->err;
    *out = s;
    return ORB_OK;
}

int orb_ctx_get_bounds(const orb_ctx* c, float b[4]);

// Grid geometry of the frames being matched: the context's image bounds when set (lens distortion),
// else the image rectangle; inverses as the Frame constructor computes them.
static GridGeom grid_geom(const orb_ctx* c, int img_w, int img_h) {
    float b[4];
    if (!orb_ctx_get_bounds(c, b)) {
        b[0] = 0.f;
        b[1] = (float)img_w;
        b[2] = 0.f;
        b[3] = (float)img_h;
    }
    GridGeom G;
    G.minX = b[0];
    G.maxX = b[1];
    G.minY = b[2];
    G.maxY = b[3];
    G.wInv = static_cast<float>(ORB_GRID_COLS) / static_cast<float>(b[1] - b[0]);
    G.hInv = static_cast<float>(ORB_GRID_ROWS) / static_cast<float>(b[3] - b[2]);
    return G;
}

static int check_match_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("match launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

static int match_window_batch(orb_ctx* c, int npairs, const orb_keypoint* d_k1, const uint8_t* d_d1,
                              const int* d_n1, int stride1, const orb_keypoint* d_k2,
                              const uint8_t* d_d2, const int* d_n2, int stride2, int img_w, int img_h,
                              float* d_prev_xy, int* d_matches12, int window, float nnratio,
                              int check_ori, int* d_nmatches, void* stream, bool sync, int* blockErr = nullptr) {
    if (!c || npairs < 1 || !d_k1 || !d_d1 || !d_n1 || !d_k2 || !d_d2 || !d_n2 || !d_prev_xy || !d_matches12 ||
        !d_nmatches || stride1 < 1 || stride2 < 1 || img_w < 1 || img_h < 1) {
        orb_set_error("orb_match_window_batch_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (npairs > orb_ctx_max_batch(c)) {
        orb_set_error("%d pairs exceed the context's max_batch", npairs);
        return ORB_ERR_INVALID;
    }
    if (stride2 > M_MAXKP || stride1 > 32767) {
        orb_set_error("windowed matcher handles at most %d keypoints in the searched frame and 32767 in the query frame",
                      M_MAXKP);
        return ORB_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    MatchBuffers B;
    int* errw = nullptr;
    int rc = match_buffers(c, npairs, stride1, stride2, &s, &B, &errw);
    if (rc) return rc;
    if (blockErr) {  // error words inside the caller's block (both zero on arrival)
        errw = blockErr;
        B.poolUsed = (unsigned*)blockErr + 1;
    }
    hipStream_t st = (hipStream_t)stream;
    const GridGeom G = grid_geom(c, img_w, img_h);
    hipLaunchKernelGGL(k_grid_build, dim3(npairs), dim3(GB_THREADS), 0, st, d_k2, d_n2, 0, stride2, G, B,
                       errw, 0);
    if ((rc = check_match_launch())) return rc;
    hipLaunchKernelGGL(k_match_lists<false>, dim3((stride1 + 3) / 4, npairs), dim3(256), 0, st, d_k1, d_d1, d_n1, 0,
                       stride1, (const float*)d_prev_xy, (float)window, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const int*)nullptr, (const int*)nullptr, d_d2, stride2, G, orb_ctx_quirks(c), B,
                       errw);
    if ((rc = check_match_launch())) return rc;
    static const bool inOrder = getenv("ORBHIP_MATCH_IN_ORDER") != nullptr;  // the one-query-at-a-time walk (tests compare the two)
    if (inOrder || stride1 > M_PAR_MAXKP || stride2 > M_PAR_MAXKP)  // (frames beyond what the 64-at-a-time kernel's LDS holds)
        hipLaunchKernelGGL(k_match_resolve<0>, dim3(npairs), dim3(256), 0, st, d_k1, d_n1, 0, stride1, d_k2, d_n2, 0,
                           stride2, d_prev_xy, d_matches12, (uint8_t*)s->binOf, (const float*)nullptr,
                           (const uint8_t*)nullptr, (const uint8_t*)nullptr, (int*)nullptr, nnratio, 0, check_ori, B,
                           d_nmatches, errw);
    else
        hipLaunchKernelGGL(k_match_resolve_par<0>, dim3(npairs), dim3(M_PAR_THREADS), 0, st, d_k1, d_n1, 0, stride1, d_k2, d_n2, 0,
                           stride2, d_prev_xy, d_matches12, (uint8_t*)s->binOf, (const float*)nullptr,
                           (const uint8_t*)nullptr, (const uint8_t*)nullptr, (int*)nullptr, nnratio, 0, check_ori, B,
                           d_nmatches, errw);
    return check_match_launch();
}

extern "C" int orb_match_window_batch_device(orb_ctx* c, int npairs, const orb_keypoint* d_k1, const uint8_t* d_d1,
                                             const int* d_n1, int stride1, const orb_keypoint* d_k2,
                                             const uint8_t* d_d2, const int* d_n2, int stride2, int img_w, int img_h,
                                             float* d_prev_xy, int* d_matches12, int window, float nnratio,
                                             int check_ori, int* d_nmatches, void* stream) {
    return match_window_batch(c, npairs, d_k1, d_d1, d_n1, stride1, d_k2, d_d2, d_n2, stride2, img_w, img_h, d_prev_xy,
                              d_matches12, window, nnratio, check_ori, d_nmatches, stream, false);
}

// Host-pointer entry points: every input and output of a call is laid out in one block that exists
// twice, in page-locked host memory and on the device (both grown on demand and kept).  A call packs
// its inputs on the host, moves them with one DMA, runs the kernels, moves the outputs (and the error
// flag) back with one DMA and synchronises once.
namespace {
struct Block {
    size_t off = 0;
    size_t add(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};
}  // namespace

// Transfers between the page-locked block and its device copy by a kernel that reads or writes the host memory over
// the link: a call moves some tens of KB each way, for which the copy engine's start-up and the wait between it
// and the next kernel (7 + 33 us in front of a projected search) cost more than the transfer.
__global__ __launch_bounds__(256) void k_block_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned n16) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
// [off, off + bytes) of the block (offsets and sizes are multiples of 16, as Block::add lays them out)
static int block_copy(MatchState* s, size_t off, size_t bytes, bool toDevice, hipStream_t st) {
    if (!bytes) return ORB_OK;
    static const bool dma = getenv("ORBHIP_UPLOAD_DMA") != nullptr;
    uint8_t* const h = (uint8_t*)s->blockHostDev + off;
    uint8_t* const d = (uint8_t*)s->blockDev + off;
    if (dma || (bytes >> 4) > 0x7fffffffull) {
        HIP_TRY(hipMemcpyAsync(toDevice ? (void*)d : (void*)((uint8_t*)s->blockHost + off), toDevice ? (const void*)((uint8_t*)s->blockHost + off) : (const void*)d,
                               bytes, toDevice ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
        return ORB_OK;
    }
    const unsigned n16 = (unsigned)(bytes >> 4);
    hipLaunchKernelGGL(k_block_copy, dim3(std::min(1024u, (n16 + 255u) / 256u)), dim3(256), 0, st,
                       (const uint4*)(toDevice ? h : d), (uint4*)(toDevice ? d : h), n16);
    return check_match_launch();
}

static int ensure_block(MatchState* s, size_t bytes) {
    if (bytes <= s->blockBytes) return ORB_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (s->blockDev) (void)hipFree(s->blockDev);
    if (s->blockHost) (void)hipHostFree(s->blockHost);
    s->blockDev = s->blockHost = nullptr;
    s->blockBytes = 0;
    hipError_t e = hipMalloc(&s->blockDev, bytes);
    if (e == hipSuccess) e = hipHostMalloc(&s->blockHost, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&s->blockHostDev, s->blockHost, 0);
    if (e != hipSuccess) {
        orb_set_error("matcher staging (%zu bytes): %s", bytes, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    s->blockBytes = bytes;
    return ORB_OK;
}

// Error words of the last call as the host read them (flag[0] bits, flag[1] pool entries asked for).
// M_RETRY: the candidate pool ran out and has been marked for growth; the caller runs the call again.
#define M_RETRY 1
static int match_error(const int* flag, MatchState* s, const char* what, bool mayRetry, bool sync = true) {
    if (flag[0]) {
        // (the host-pointer entry points keep their two error words inside the call's block, uploaded as zeros with
        // the inputs and fetched with the outputs: nothing of theirs stays on the device)
        if (!sync) HIP_TRY(hipMemset(s->err, 0, 4));
        if (flag[0] & M_ERR_LIMIT) {
            orb_set_error("%s: keypoint count exceeds the kernel's limits", what);
            return ORB_ERR_CAPACITY;
        }
        const size_t need = (size_t)(unsigned)flag[1];
        s->poolWanted = need + need / 8;
        if (mayRetry) return M_RETRY;
        orb_set_error("%s: search windows hold more candidates than the context's candidate pool (%zu entries, %zu needed); "
                      "the pool grows with the next call, or reserve it with orb_match_reserve_candidates", what, s->poolCap,
                      need);
        return ORB_ERR_CAPACITY;
    }
    return ORB_OK;
}

// The context's transfer block for the other small host-pointer entry points of the library (kernels_frame.hip):
// *host and *dev receive `bytes` bytes of page-locked memory and their device copy; orb_stage_copy moves
// [off, off + bytes) one way or the other on the stream (multiples of 16).
int orb_stage_block(orb_ctx* c, size_t bytes, void** host, void** dev) {
    MatchState* s = nullptr;
    MatchBuffers B;
    int rc = match_buffers(c, 1, 1, 1, &s, &B);  // (creates the state on first use)
    if (rc) return rc;
    if ((rc = ensure_block(s, (bytes + 255) & ~(size_t)255))) return rc;
    *host = s->blockHost;
    *dev = s->blockDev;
    return ORB_OK;
}
int orb_stage_copy(orb_ctx* c, size_t off, size_t bytes, int toDevice, void* stream) {
    MatchState* s = (MatchState*)*orb_ctx_match_state(c);
    if (!s || off + bytes > s->blockBytes || ((off | bytes) & 15)) {
        orb_set_error("internal: transfer outside the context's block");
        return ORB_ERR_INVALID;
    }
    return block_copy(s, off, bytes, toDevice != 0, (hipStream_t)stream);
}

// For the asynchronous entry points: called by orb_check_last_batch once the device is idle.
int orb_match_check_error(orb_ctx* c) {
    MatchState* s = (MatchState*)*orb_ctx_match_state(c);
    if (!s) return ORB_OK;
    int flag[2] = {0, 0};
    HIP_TRY(hipMemcpy(flag, s->err, 8, hipMemcpyDeviceToHost));
    return match_error(flag, s, "windowed matcher", false, false);
}

extern "C" int orb_match_reserve_candidates(orb_ctx* c, long long entries) {
    if (!c || entries < 0) {
        orb_set_error("orb_match_reserve_candidates: bad arguments");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    MatchBuffers B;
    int rc = match_buffers(c, 1, 1, 1, &s, &B);  // (creates the state on first use)
    if (rc) return rc;
    if ((size_t)entries > s->poolWanted) s->poolWanted = (size_t)entries;
    return ORB_OK;
}

static int match_window_once(orb_ctx* c, const orb_keypoint* k1, const uint8_t* d1, int n1, const orb_keypoint* k2,
                             const uint8_t* d2, int n2, int img_w, int img_h, float* prev_xy, int* matches12, int window,
                             float nnratio, int check_ori, int* nmatches, bool mayRetry) {
    if (!c || n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && (!k1 || !d1 || !prev_xy || !matches12)) ||
        (n2 > 0 && (!k2 || !d2)) || img_w < 1 || img_h < 1) {
        orb_set_error("orb_match_window: bad arguments");
        return ORB_ERR_INVALID;
    }
    *nmatches = 0;
    if (n1 == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    const int s1 = n1, s2 = n2 > 0 ? n2 : 1;
    MatchState* s = nullptr;
    MatchBuffers B;
    int rc = match_buffers(c, 1, s1, s2, &s, &B);  // (creates the state on first use)
    if (rc) return rc;
    Block L;
    const size_t oK1 = L.add((size_t)n1 * sizeof(orb_keypoint)), oD1 = L.add((size_t)n1 * 32);
    const size_t oK2 = L.add((size_t)s2 * sizeof(orb_keypoint)), oD2 = L.add((size_t)s2 * 32), oN = L.add(8);
    const size_t oPrev = L.add((size_t)n1 * 8);  // in and out
    const size_t oErr = L.add(8);                // error words: zero in, fetched with the outputs
    const size_t oM = L.add((size_t)n1 * 4), oNm = L.add(4);
    if ((rc = ensure_block(s, L.off))) return rc;
    uint8_t* h = (uint8_t*)s->blockHost;
    uint8_t* d = (uint8_t*)s->blockDev;
    memcpy(h + oK1, k1, (size_t)n1 * sizeof(orb_keypoint));
    memcpy(h + oD1, d1, (size_t)n1 * 32);
    if (n2 > 0) {
        memcpy(h + oK2, k2, (size_t)n2 * sizeof(orb_keypoint));
        memcpy(h + oD2, d2, (size_t)n2 * 32);
    }
    const int ns[2] = {n1, n2};
    memcpy(h + oN, ns, 8);
    memcpy(h + oPrev, prev_xy, (size_t)n1 * 8);
    memset(h + oErr, 0, 8);
    hipStream_t st = (hipStream_t)orb_ctx_stream(c);
    if ((rc = block_copy(s, 0, oM, true, st))) return rc;
    rc = match_window_batch(c, 1, (const orb_keypoint*)(d + oK1), d + oD1, (const int*)(d + oN), s1,
                            (const orb_keypoint*)(d + oK2), d + oD2, (const int*)(d + oN) + 1, s2, img_w, img_h,
                            (float*)(d + oPrev), (int*)(d + oM), window, nnratio, check_ori, (int*)(d + oNm), st, true,
                            (int*)(d + oErr));
    if (rc) return rc;
    if ((rc = block_copy(s, oPrev, L.off - oPrev, false, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = match_error((const int*)(h + oErr), s, "windowed matcher", mayRetry))) return rc;
    memcpy(prev_xy, h + oPrev, (size_t)n1 * 8);
    memcpy(matches12, h + oM, (size_t)n1 * 4);
    memcpy(nmatches, h + oNm, 4);
    return ORB_OK;
}

extern "C" int orb_match_window(orb_ctx* c, const orb_keypoint* k1, const uint8_t* d1, int n1, const orb_keypoint* k2,
                                const uint8_t* d2, int n2, int img_w, int img_h, float* prev_xy, int* matches12,
                                int window, float nnratio, int check_ori, int* nmatches) {
    // inputs are only read and outputs only written on success, so a call whose windows overran the candidate
    // pool is simply made again once the pool has grown
    int rc = match_window_once(c, k1, d1, n1, k2, d2, n2, img_w, img_h, prev_xy, matches12, window, nnratio, check_ori,
                               nmatches, true);
    if (rc == M_RETRY)
        rc = match_window_once(c, k1, d1, n1, k2, d2, n2, img_w, img_h, prev_xy, matches12, window, nnratio, check_ori,
                               nmatches, false);
    return rc;
}

// mode 1: best candidate only, rotation check on request; mode 2: local-map rule (no rotation check).
static int match_projected_once(orb_ctx* c, int mode, const float* u, const float* v, const float* radius,
                                const int* min_level, const int* max_level, const float* qangle, const uint8_t* qdesc,
                                const uint8_t* qobs, int nq, const orb_keypoint* k2, const uint8_t* d2, int n2, int img_w,
                                int img_h, const uint8_t* occupied2, int* match2, int th_high, float nnratio,
                                int check_ori, int* nmatches, bool mayRetry) {
    if (!c || nq < 0 || n2 < 0 || !nmatches || (n2 > 0 && (!k2 || !d2 || !match2)) ||
        (nq > 0 && (!u || !v || !radius || !min_level || !max_level || (check_ori && !qangle) || !qdesc)) || img_w < 1 ||
        img_h < 1) {
        orb_set_error("projected matcher: bad arguments");
        return ORB_ERR_INVALID;
    }
    *nmatches = 0;
    if (n2 > M_MAXKP || nq > 32767) {
        orb_set_error("projected matcher handles at most %d keypoints in the searched frame and 32767 queries", M_MAXKP);
        return ORB_ERR_UNSUPPORTED;
    }
    if (n2 == 0) return ORB_OK;
    if (nq == 0) {
        for (int i = 0; i < n2; i++) match2[i] = -1;
        return ORB_OK;
    }
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    MatchBuffers B;
    int rc = match_buffers(c, 1, nq, n2, &s, &B);
    if (rc) return rc;
    Block L;
    const size_t oU = L.add((size_t)nq * 4), oV = L.add((size_t)nq * 4), oR = L.add((size_t)nq * 4);
    const size_t oL0 = L.add((size_t)nq * 4), oL1 = L.add((size_t)nq * 4), oA = L.add((size_t)nq * 4);
    const size_t oQ = L.add((size_t)nq * 32), oObs = L.add((size_t)nq);
    const size_t oK2 = L.add((size_t)n2 * sizeof(orb_keypoint)), oD2 = L.add((size_t)n2 * 32), oOcc = L.add((size_t)n2);
    const size_t oErr = L.add(8);                                             // error words: zero in, fetched with the outputs
    const size_t oM = L.add((size_t)n2 * 4), oNm = L.add(4);                  // outputs
    if ((rc = ensure_block(s, L.off))) return rc;
    uint8_t* h = (uint8_t*)s->blockHost;
    uint8_t* d = (uint8_t*)s->blockDev;
    memcpy(h + oU, u, (size_t)nq * 4);
    memcpy(h + oV, v, (size_t)nq * 4);
    memcpy(h + oR, radius, (size_t)nq * 4);
    memcpy(h + oL0, min_level, (size_t)nq * 4);
    memcpy(h + oL1, max_level, (size_t)nq * 4);
    if (qangle) memcpy(h + oA, qangle, (size_t)nq * 4);
    memcpy(h + oQ, qdesc, (size_t)nq * 32);
    if (qobs) memcpy(h + oObs, qobs, (size_t)nq);
    memcpy(h + oK2, k2, (size_t)n2 * sizeof(orb_keypoint));
    memcpy(h + oD2, d2, (size_t)n2 * 32);
    if (occupied2) memcpy(h + oOcc, occupied2, (size_t)n2);
    memset(h + oErr, 0, 8);
    hipStream_t st = (hipStream_t)orb_ctx_stream(c);
    int* const errw = (int*)(d + oErr);
    B.poolUsed = (unsigned*)errw + 1;
    if ((rc = block_copy(s, 0, oM, true, st))) return rc;
    const GridGeom G = grid_geom(c, img_w, img_h);
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_THREADS), 0, st, (const orb_keypoint*)(d + oK2), (const int*)nullptr, n2, n2,
                       G, B, errw, -1);
    if ((rc = check_match_launch())) return rc;
    hipLaunchKernelGGL(k_match_lists<true>, dim3((nq + 3) / 4, 1), dim3(256), 0, st, (const orb_keypoint*)nullptr,
                       (const uint8_t*)(d + oQ), (const int*)nullptr, nq, nq, (const float*)nullptr, 0.f,
                       (const float*)(d + oU), (const float*)(d + oV), (const float*)(d + oR), (const int*)(d + oL0),
                       (const int*)(d + oL1), (const uint8_t*)(d + oD2), n2, G, orb_ctx_quirks(c), B, errw);
    if ((rc = check_match_launch())) return rc;
    static const bool inOrder = getenv("ORBHIP_MATCH_IN_ORDER") != nullptr;
    const bool par = !inOrder && nq <= M_PAR_MAXKP && n2 <= M_PAR_MAXKP;
    hipLaunchKernelGGL(par ? (mode == 2 ? k_match_resolve_par<2> : k_match_resolve_par<1>) : (mode == 2 ? k_match_resolve<2> : k_match_resolve<1>),
                       dim3(1), dim3(par ? M_PAR_THREADS : 256), 0, st,
                       (const orb_keypoint*)nullptr, (const int*)nullptr, nq, nq, (const orb_keypoint*)(d + oK2),
                       (const int*)nullptr, n2, n2, (float*)nullptr, (int*)nullptr, (uint8_t*)s->binOf,
                       (const float*)(d + oA), qobs ? (const uint8_t*)(d + oObs) : (const uint8_t*)nullptr,
                       occupied2 ? (const uint8_t*)(d + oOcc) : (const uint8_t*)nullptr, (int*)(d + oM), nnratio, th_high,
                       check_ori, B, (int*)(d + oNm), errw);
    if ((rc = check_match_launch())) return rc;
    if ((rc = block_copy(s, oErr, L.off - oErr, false, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = match_error((const int*)(h + oErr), s, "projected matcher", mayRetry))) return rc;
    memcpy(match2, h + oM, (size_t)n2 * 4);
    memcpy(nmatches, h + oNm, 4);
    return ORB_OK;
}

static int match_projected(orb_ctx* c, int mode, const float* u, const float* v, const float* radius,
                           const int* min_level, const int* max_level, const float* qangle, const uint8_t* qdesc,
                           const uint8_t* qobs, int nq, const orb_keypoint* k2, const uint8_t* d2, int n2, int img_w,
                           int img_h, const uint8_t* occupied2, int* match2, int th_high, float nnratio, int check_ori,
                           int* nmatches) {
    int rc = match_projected_once(c, mode, u, v, radius, min_level, max_level, qangle, qdesc, qobs, nq, k2, d2, n2, img_w,
                                  img_h, occupied2, match2, th_high, nnratio, check_ori, nmatches, true);
    if (rc == M_RETRY)
        rc = match_projected_once(c, mode, u, v, radius, min_level, max_level, qangle, qdesc, qobs, nq, k2, d2, n2, img_w,
                                  img_h, occupied2, match2, th_high, nnratio, check_ori, nmatches, false);
    return rc;
}

extern "C" int orb_match_projected(orb_ctx* c, const float* u, const float* v, const float* radius,
                                   const int* min_level, const int* max_level, const float* qangle,
                                   const uint8_t* qdesc, const uint8_t* qobs, int nq, const orb_keypoint* k2,
                                   const uint8_t* d2, int n2, int img_w, int img_h, const uint8_t* occupied2,
                                   int* match2, int th_high, int check_ori, int* nmatches) {
    if (nq > 0 && !qangle) {
        orb_set_error("orb_match_projected: bad arguments");
        return ORB_ERR_INVALID;
    }
    return match_projected(c, 1, u, v, radius, min_level, max_level, qangle, qdesc, qobs, nq, k2, d2, n2, img_w, img_h,
                           occupied2, match2, th_high, 0.f, check_ori, nmatches);
}

extern "C" int orb_match_local_map(orb_ctx* c, const float* u, const float* v, const float* radius,
                                   const int* min_level, const int* max_level, const uint8_t* qdesc,
                                   const uint8_t* qobs, int nq, const orb_keypoint* k2, const uint8_t* d2, int n2,
                                   int img_w, int img_h, const uint8_t* occupied2, int* match2, int th_high,
                                   float nnratio, int* nmatches) {
    return match_projected(c, 2, u, v, radius, min_level, max_level, nullptr, qdesc, qobs, nq, k2, d2, n2, img_w, img_h,
                           occupied2, match2, th_high, nnratio, 0, nmatches);
}

extern "C" int orb_match_best(orb_ctx* c, const float* u, const float* v, const float* radius, const int* min_level,
                              const int* max_level, const uint8_t* qdesc, int nq, const orb_keypoint* k2,
                              const uint8_t* d2, int n2, int img_w, int img_h, const float* inv_level_sigma2,
                              int nlevels, double chi2_max, int th, int* best_idx, int* best_dist) {
    if (!c || nq < 0 || n2 < 0 || (n2 > 0 && (!k2 || !d2)) || img_w < 1 || img_h < 1 ||
        (nq > 0 && (!u || !v || !radius || !min_level || !max_level || !qdesc || !best_idx)) ||
        (inv_level_sigma2 && (nlevels < 1 || nlevels > 255))) {
        orb_set_error("orb_match_best: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (n2 > M_MAXKP) {
        orb_set_error("orb_match_best handles at most %d keypoints in the searched frame", M_MAXKP);
        return ORB_ERR_UNSUPPORTED;
    }
    if (nq == 0) return ORB_OK;
    if (n2 == 0) {
        for (int i = 0; i < nq; i++) {
            best_idx[i] = -1;
            if (best_dist) best_dist[i] = 256;
        }
        return ORB_OK;
    }
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    MatchBuffers B;
    int rc = match_buffers(c, 1, 1, n2, &s, &B);  // the grid arrays; no candidate lists are written
    if (rc) return rc;
    const int ns = inv_level_sigma2 ? nlevels : 0;
    Block L;
    const size_t oU = L.add((size_t)nq * 4), oV = L.add((size_t)nq * 4), oR = L.add((size_t)nq * 4);
    const size_t oL0 = L.add((size_t)nq * 4), oL1 = L.add((size_t)nq * 4), oQ = L.add((size_t)nq * 32);
    const size_t oK2 = L.add((size_t)n2 * sizeof(orb_keypoint)), oD2 = L.add((size_t)n2 * 32), oSig = L.add(256 * 4);
    const size_t oErr = L.add(8);  // error words: zero in, fetched with the outputs
    const size_t oIdx = L.add((size_t)nq * 4), oDist = L.add((size_t)nq * 4);  // outputs
    if ((rc = ensure_block(s, L.off))) return rc;
    uint8_t* h = (uint8_t*)s->blockHost;
    uint8_t* d = (uint8_t*)s->blockDev;
    memcpy(h + oU, u, (size_t)nq * 4);
    memcpy(h + oV, v, (size_t)nq * 4);
    memcpy(h + oR, radius, (size_t)nq * 4);
    memcpy(h + oL0, min_level, (size_t)nq * 4);
    memcpy(h + oL1, max_level, (size_t)nq * 4);
    memcpy(h + oQ, qdesc, (size_t)nq * 32);
    memcpy(h + oK2, k2, (size_t)n2 * sizeof(orb_keypoint));
    memcpy(h + oD2, d2, (size_t)n2 * 32);
    if (ns) memcpy(h + oSig, inv_level_sigma2, (size_t)ns * 4);
    memset(h + oErr, 0, 8);
    hipStream_t st = (hipStream_t)orb_ctx_stream(c);
    int* const errw = (int*)(d + oErr);
    B.poolUsed = (unsigned*)errw + 1;
    if ((rc = block_copy(s, 0, oIdx, true, st))) return rc;
    const GridGeom G = grid_geom(c, img_w, img_h);
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_THREADS), 0, st, (const orb_keypoint*)(d + oK2), (const int*)nullptr, n2, n2,
                       G, B, errw, -1);
    if ((rc = check_match_launch())) return rc;
    hipLaunchKernelGGL(k_match_best, dim3((nq + 3) / 4), dim3(256), 0, st, (const float*)(d + oU), (const float*)(d + oV),
                       (const float*)(d + oR), (const int*)(d + oL0), (const int*)(d + oL1), (const uint8_t*)(d + oQ), nq,
                       (const uint8_t*)(d + oD2), ns ? (const float*)(d + oSig) : (const float*)nullptr, ns, chi2_max, th, G,
                       orb_ctx_quirks(c), B, (int*)(d + oIdx), (int*)(d + oDist));
    if ((rc = check_match_launch())) return rc;
    if ((rc = block_copy(s, oErr, L.off - oErr, false, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = match_error((const int*)(h + oErr), s, "orb_match_best", false))) return rc;
    memcpy(best_idx, h + oIdx, (size_t)nq * 4);
    if (best_dist) memcpy(best_dist, h + oDist, (size_t)nq * 4);
    return ORB_OK;
}
