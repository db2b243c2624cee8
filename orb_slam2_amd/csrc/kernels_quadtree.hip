// DistributeOctTree for every (frame, level) of a batch: one workgroup each.
// The algorithm itself is in quadtree_core.h.
//
// Fast path (histogram form): the level's FAST keys are never sorted or even
// gathered.  One pass over the per-cell key slots counts them into buckets =
// quadtree cells of a fixed depth and keeps the best key of every bucket (LDS
// atomics); the distribution then works on the prefix sums of the counts, and
// the winner of a node is the best entry over its buckets.  That is exact as
// long as no node with several keys has to be divided below the bucket depth,
// which the depth is chosen to make rare (qt_hist_depth).
//
// Slow path (sorted form), taken by a level only when the fast path reports
// such a node: the keys are turned into sort entries numbered in the
// reference's order (src/ORBextractor.cc:825-870: cell rows, cells, rows inside
// the cell), sorted by quadtree path in global memory, and distributed by
// binary searches over the sorted array.  Same result by construction (both
// forms are checked against the oracle's std::list restatement, tests/emu).
#include <atomic>
#include <hip/hip_runtime.h>

#include <algorithm>

#include "quadtree_core.h"

#ifdef QT_PROFILE
__device__ unsigned long long g_qtProfile[32];
extern "C" int orb_debug_quadtree_profile(unsigned long long* out, int clear) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qtProfile), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qtProfile), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#define QT_MAX_THREADS 1024
#define QT_LDS_KEYS_MAX 4096
#define QT_STATIC_LDS 1024     // bound on sizeof(QtShared)

struct QtLaunch {
    int level0, nlevels;  // levels [level0, level0 + nlevels) of every frame
    int ldsKeys;          // sort entries the LDS block holds (sorted form; 0: the sorted keys stay in global memory)
    int binBits;          // key-sort buckets = 2^binBits (sorted form)
    int regionInts;       // ints behind the keys: histogram or bucket arrays, then node scratch
    unsigned nframesMagic;  // orb_div_magic(frames of the launch)
    // 0: every (frame, level) is finished by its workgroup.  A batch runs in two passes instead: pass 1 with one
    // path digit fewer buckets (a quarter of the LDS, twice the workgroups per CU), whose workgroups hand the
    // levels that need the finer buckets to a list; pass 2 takes the list with the full depth.
    int pass;
};

// Path of a key from its rank (= cell * cellCap + position inside the cell): the histogram form needs it for
// single keys below the bucket depth.
struct QtDevRankCode {
    const uint32_t* lck;
    const uint32_t* codeX;
    const uint32_t* codeY;
    int cellCap, nCols, wCell, hCell;
    unsigned cellCapMagic, nColsMagic;
    __device__ __forceinline__ uint32_t operator()(uint32_t rank) const {
        const uint32_t key = lck[rank];
        const int cell = (int)orb_div(rank, (unsigned)cellCap, cellCapMagic);
        const int ci = (int)orb_div((unsigned)cell, (unsigned)nCols, nColsMagic), cj = cell - ci * nCols;
        return codeX[3 + cj * wCell + (int)(key & 0xff)] | codeY[3 + ci * hCell + (int)((key >> 8) & 0xff)];
    }
};

// Winners of the final list -> selXY / selScore in list order.
template <class Space>
static __device__ __forceinline__ int qt_finish(const OrbGeom* g, const OrbLevel& L, const Space& sp, const QtWork& w,
                                                OrbNode* list, int count, const uint32_t* __restrict__ levelCellKeys,
                                                uint32_t* __restrict__ selXY, int* __restrict__ selScore,
                                                int* __restrict__ error, QtShared* sh) {
    const int t = threadIdx.x, T = (int)blockDim.x;
    if (sh->err && t == 0) atomicOr(error, sh->err == 2 ? 8 : 16);
    if (count > L.outCap) {  // cannot happen (quota + 2 bound); fail loudly, never truncate silently
        if (t == 0) atomicOr(error, 32);
        count = L.outCap;
    }
    qt_select_all(sp, list, count, w.scanA, sh, L.nIni);
    QT_STAMP(6);
    for (int i = t; i < count; i += T) {
        const int rank = w.scanA[i];
        const uint32_t key = levelCellKeys[rank];  // rank = cell * cellCap + position inside the cell
        const int cell = (int)orb_div((unsigned)rank, (unsigned)g->cellCap, g->cellCapMagic);
        const int ci = (int)orb_div((unsigned)cell, (unsigned)L.nCols, L.nColsMagic), cj = cell - ci * L.nCols;
        const int x = 3 + cj * L.wCell + (int)(key & 0xff);  // relative to the 16-pixel border, src/ORBextractor.cc:864-865
        const int y = 3 + ci * L.hCell + (int)((key >> 8) & 0xff);
        selXY[i] = (uint32_t)x | ((uint32_t)y << 16);
        selScore[i] = (int)(key >> 16);
    }
    return count;
}

// Sorted form: sort, distribute, select and emit one level.  Inlined once per memory placement so that the
// compiler sees which address space `sorted` and the node scratch live in.
static __device__ __forceinline__ int qt_level(const OrbGeom* g, const OrbLevel& L, const unsigned long long* tmp,
                                               unsigned long long* sorted, const QtWork& w0, int n,
                                               const uint32_t* __restrict__ levelCellKeys, uint32_t* __restrict__ selXY,
                                               int* __restrict__ selScore, int* __restrict__ error, const QtBins bins,
                                               QtShared* sh) {
    qt_sort_keys(tmp, sorted, n, L.nIni, bins, sh);
    __syncthreads();  // the bucket arrays may now be reused as node scratch
    QT_STAMP(4);
    QtWork w = w0;
    w.sorted = sorted;
    const QtSortedSpace sp = {sorted, n};
    OrbNode* list = nullptr;
    int count = 0;
    qt_distribute(sp, w, L.nIni, L.quota, (g->quirks & ORB_Q_DIVIDE_N3) ? 1 : 0, sh, &list, &count);
    QT_STAMP(5);
    return qt_finish(g, L, sp, w, list, count, levelCellKeys, selXY, selScore, error, sh);
}

// Node scratch of a (frame, level) in the global buffers.
static __device__ __forceinline__ void qt_work_global(QtWork* w, const OrbGeom* g, const OrbLevel& L, int frame, int level,
                                                      OrbNode* nodes, int* ints) {
    OrbNode* nb = nodes + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 2;
    int* p = ints + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 9 + (int64_t)(frame * g->nlevels + level) * 4;
    w->nodeCap = L.nodeCap;
    w->listA = nb;
    w->listB = nb + L.nodeCap;
    w->rec = p;
    p += L.nodeCap;
    w->recSorted = p;
    p += L.nodeCap;
    w->bounds = p;
    p += 4 * L.nodeCap;
    w->scanA = p;
    p += L.nodeCap + 1;
    w->scanB = p;
    p += L.nodeCap + 1;
    w->scanC = p;
}

// One (frame, level): `wg` is its number in the level-major order of the launch.  F: cells whose loads a lane group
// of the counting pass has under way at once (twice that with the round asked for ahead).
// (The node scratch lies in LDS or in global memory, so the list accesses are flat instructions.  An instance that
// knew it to be in LDS ran slower: 0.119 against 0.108 ms at 512 frames.)
template <int F>
static __device__ __forceinline__ void qt_one(const OrbGeom* __restrict__ g, const QtLaunch& q, int wg, int frame0, int nframes,
                                              const uint32_t* __restrict__ cellKeys, const int* __restrict__ cellCount,
                                              const uint32_t* __restrict__ codeTab, unsigned long long* __restrict__ sortBuf,
                                              OrbNode* __restrict__ nodes, int* __restrict__ ints,
                                              uint32_t* __restrict__ selXY, int* __restrict__ selScore,
                                              int* __restrict__ levelCount, int* __restrict__ levelKeys,
                                              int* __restrict__ retry, int* __restrict__ error, QtShared& sh,
                                              unsigned long long* s_keys, int* s_region) {
    const QtBins bins = {s_region, s_region + (1 << q.binBits) + 1, q.binBits};
    const int nlevels = g->nlevels;
    // level-major launch order: the levels with the most keys (the longest-running workgroups) start first
    const int levelLocal = (int)orb_div((unsigned)wg, (unsigned)nframes, q.nframesMagic);
    const int level = q.level0 + levelLocal, frame = frame0 + wg - levelLocal * nframes;
    const OrbLevel& L = g->lv[level];
    const int t = threadIdx.x, T = (int)blockDim.x;
    unsigned long long* sortedG = sortBuf + ((int64_t)frame * g->keysPerFrame + L.keyBase) * 2;
    unsigned long long* tmp = sortedG + L.keyCap;
    const uint32_t* lck = cellKeys + ((int64_t)frame * g->cellsPerFrame + L.cellBase) * g->cellCap;

    const int* cc = cellCount + (int64_t)frame * g->cellsPerFrame + L.cellBase;
    const int nCells = L.nCols * L.nRows, cellCap = g->cellCap;
    const uint32_t* codeX = codeTab + L.codeTabX;  // quadtree path words of the level's columns and rows
    const uint32_t* codeY = codeTab + L.codeTabY;
    const int64_t ob = (int64_t)frame * g->outCapFrame + L.outBase;
    const int n3quirk = (g->quirks & ORB_Q_DIVIDE_N3) ? 1 : 0;
    QT_STAMP(0);

    // ---- histogram form ------------------------------------------------------------------------------
    const int Db = q.pass == 1 ? L.histDepthFirst : L.histDepth, nb = L.nIni << (2 * Db);
    if (2 * nb + 1 <= q.regionInts) {
        int* start = s_region;
        uint32_t* best = (uint32_t*)(s_region + nb + 1);
        // Four lanes per cell, four key slots per lane: a cell's first 16 slots are read whether or not they hold
        // keys (slots are 16-byte aligned, at least 16 long, and share a cache line or two), so that the load does
        // not wait for the cell's count.  F cells per lane group are under way at once, the first round's loads are
        // issued before the histogram is cleared, and the next round's counts and slots are asked for before this
        // round's keys are looked at: a level costs about three dependent memory round trips (counts and slots;
        // path words; the next round hides behind them) instead of three per round.  Keys beyond a cell's first
        // 16 follow one by one (rare).
        const int shiftB = 2 * (ORB_CODE_DEPTH - Db);
        const int sub = t & 3, ngrp = T >> 2;
        auto fetch = [&](int c0, int (&cnt)[F], uint4 (&slot)[F]) {
#pragma unroll
            for (int u = 0; u < F; u++) {
                const int c = c0 + u * ngrp;
                const bool in = c < nCells;
                cnt[u] = in ? cc[c] : 0;
                slot[u] = in ? *(const uint4*)(lck + (int64_t)c * cellCap + 4 * sub) : make_uint4(0u, 0u, 0u, 0u);
            }
        };
        auto add_key = [&](uint32_t rank, uint32_t key, uint32_t code) {
            const uint32_t prio = ((key >> 16) << 24) | (~rank & 0xffffffu);  // = qt_entry_prio of the key's sort entry
            atomicAdd(&start[code >> shiftB], 1);
            atomicMax(&best[code >> shiftB], prio);
        };
        int cnt[F];
        uint4 slot[F];
        int c0 = t >> 2;
        fetch(c0, cnt, slot);
        for (int b = t; b < nb; b += T) {
            start[b] = 0;
            best[b] = 0u;
        }
        if (t == 0) start[nb] = 0;
        __syncthreads();
        while (c0 < nCells) {
            int cntN[F];
            uint4 slotN[F];
            fetch(c0 + F * ngrp, cntN, slotN);
            uint32_t key[F][4], code[F][4];
#pragma unroll
            for (int u = 0; u < F; u++) {
                const int c = c0 + u * ngrp;
                const int ci = (int)orb_div((unsigned)c, (unsigned)L.nCols, L.nColsMagic), cj = c - ci * L.nCols;
                const int bx = 3 + cj * L.wCell, by = 3 + ci * L.hCell;  // relative to the 16-pixel border, as :864-865
                const uint32_t k4[4] = {slot[u].x, slot[u].y, slot[u].z, slot[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (a slot without a key holds anything, and cells past the last lie outside the tables: such
                    // slots read the tables' first words)
                    const bool has = 4 * sub + j < cnt[u];
                    key[u][j] = k4[j];
                    code[u][j] = codeX[has ? bx + (int)(k4[j] & 0xff) : 0] | codeY[has ? by + (int)((k4[j] >> 8) & 0xff) : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < F; u++) {
                const uint32_t rank0 = (uint32_t)((c0 + u * ngrp) * cellCap + 4 * sub);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * sub + j < cnt[u]) add_key(rank0 + (uint32_t)j, key[u][j], code[u][j]);
            }
#pragma unroll
            for (int u = 0; u < F; u++) {
                if (cnt[u] <= 16) continue;
                const int c = c0 + u * ngrp;
                const int ci = (int)orb_div((unsigned)c, (unsigned)L.nCols, L.nColsMagic), cj = c - ci * L.nCols;
                const int bx = 3 + cj * L.wCell, by = 3 + ci * L.hCell;
                for (int k = 16 + sub; k < cnt[u]; k += 4) {
                    const uint32_t kk = lck[(int64_t)c * cellCap + k];
                    add_key((uint32_t)(c * cellCap + k), kk, codeX[bx + (int)(kk & 0xff)] | codeY[by + (int)((kk >> 8) & 0xff)]);
                }
            }
#pragma unroll
            for (int u = 0; u < F; u++) {
                cnt[u] = cntN[u];
                slot[u] = slotN[u];
            }
            c0 += F * ngrp;
        }
        __syncthreads();
        QT_STAMP(1);
        qt_scan_buckets(start, nb, &sh);  // (start = the dynamic LDS block's first word: 16-byte aligned)
        const int n = start[nb];
        if (t == 0) levelKeys[frame * nlevels + level] = n;
        if (n > L.keyCap) {  // cannot happen: keyCap is the 2x2-block bound summed over the cells
            if (t == 0) {
                atomicOr(error, 4);
                levelCount[frame * nlevels + level] = 0;
            }
            return;
        }
        if (n == 0) {
            if (t == 0) levelCount[frame * nlevels + level] = 0;
            return;
        }
        QtWork w;
        if (QT_NODE_INTS(L.nodeCap) <= q.regionInts - (2 * nb + 1)) qt_work_from_pool(&w, s_region + 2 * nb + 1, L.nodeCap);
        else qt_work_global(&w, g, L, frame, level, nodes, ints);
        w.sorted = nullptr;
        const QtDevRankCode rc = {lck, codeX, codeY, cellCap, L.nCols, L.wCell, L.hCell, g->cellCapMagic, L.nColsMagic};
        const QtHistSpace<QtDevRankCode> sp = {start, best, Db, rc};
        OrbNode* list = nullptr;
        int count = 0;
        QT_STAMP(4);
        qt_distribute(sp, w, L.nIni, L.quota, n3quirk, &sh, &list, &count);
        QT_STAMP(5);
        if (!sh.deeper) {
            count = qt_finish(g, L, sp, w, list, count, lck, selXY + ob, selScore + ob, error, &sh);
            if (t == 0) levelCount[frame * nlevels + level] = count;
            QT_STAMP(7);
#ifdef QT_PROFILE
            if (blockIdx.x == 0 && t == 0) {
                g_qtProfile[8] = (unsigned long long)n;
                g_qtProfile[9] = (unsigned long long)count;
                g_qtProfile[10] = 2ull;
            }
#endif
            return;
        }
        if (q.pass == 1) {  // a node below the bucket depth holds several keys: the second pass takes the level
            if (t == 0) {
                const int k = atomicAdd(&retry[0], 1);
                if (k < nframes * q.nlevels) retry[3 + k] = wg;  // (always: a (frame, level) lists itself once and the count starts at zero)
                else atomicOr(error, 4);
            }
            return;
        }
        __syncthreads();  // ... or, without a second pass (or in it), the sorted form does
    }

    // ---- sorted form ---------------------------------------------------------------------------------
    // Gather: cells are taken in chunks whose key offsets fit the LDS bucket array; every thread then
    // owns keys (not cells), finds each key's cell by binary search over the chunk's offsets and
    // writes its sort entry.  All loads of a thread are independent of each other.
    int n = 0;
    const int chunkCells = 1 << q.binBits;
    for (int c0 = 0; c0 < nCells; c0 += chunkCells) {
        const int m = nCells - c0 < chunkCells ? nCells - c0 : chunkCells;
        int* off = bins.start;
        for (int c = t; c < m; c += T) off[c] = cc[c0 + c];
        __syncthreads();
        qt_scan(off, m, &sh);
        const int chunk = off[m];
        for (int r = t; r < chunk; r += T) {
            int lo = 0, hi = m;  // largest c with off[c] <= r
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= r) lo = mid; else hi = mid;
            }
            const int k = r - off[lo], cell = c0 + lo;
            const uint32_t key = lck[(int64_t)cell * cellCap + k];
            const int ci = (int)orb_div((unsigned)cell, (unsigned)L.nCols, L.nColsMagic), cj = cell - ci * L.nCols;
            const int x = 3 + cj * L.wCell + (int)(key & 0xff);  // relative to the 16-pixel border, as :864-865
            const int y = 3 + ci * L.hCell + (int)((key >> 8) & 0xff);
            if (n + r < L.keyCap)
                tmp[n + r] = qt_entry(codeX[x] | codeY[y], (uint32_t)(cell * cellCap + k), key >> 16);
        }
        n += chunk;
        __syncthreads();
    }
    if (t == 0) levelKeys[frame * nlevels + level] = n;
    if (n > L.keyCap) {  // cannot happen: keyCap is the 2x2-block bound summed over the cells
        if (t == 0) {
            atomicOr(error, 4);
            levelCount[frame * nlevels + level] = 0;
        }
        return;
    }
    int count = 0;
    if (n > 0) {
        QtWork w;
        if (n <= q.ldsKeys && QT_NODE_INTS(L.nodeCap) <= q.regionInts) {
            qt_work_from_pool(&w, s_region, L.nodeCap);
            count = qt_level(g, L, tmp, s_keys, w, n, lck, selXY + ob, selScore + ob, error, bins, &sh);
        } else {
            if (QT_NODE_INTS(L.nodeCap) <= q.regionInts) qt_work_from_pool(&w, s_region, L.nodeCap);  // (behind the sort's buckets in time)
            else qt_work_global(&w, g, L, frame, level, nodes, ints);
            count = qt_level(g, L, tmp, sortedG, w, n, lck, selXY + ob, selScore + ob, error, bins, &sh);
        }
    }
    if (t == 0) levelCount[frame * nlevels + level] = count;
}

// MAXT = 512: up to 168 registers per lane (three wavefronts per SIMD, as many as the LDS blocks allow anyway);
// 1024: the full workgroups of large frames.
template <int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_quadtree(const OrbGeom* __restrict__ g, const QtLaunch q, int frame0, int nframes,
                                                        const uint32_t* __restrict__ cellKeys,
                                                        const int* __restrict__ cellCount,
                                                        const uint32_t* __restrict__ codeTab,
                                                        unsigned long long* __restrict__ sortBuf,
                                                        OrbNode* __restrict__ nodes, int* __restrict__ ints,
                                                        uint32_t* __restrict__ selXY, int* __restrict__ selScore,
                                                        int* __restrict__ levelCount, int* __restrict__ levelKeys,
                                                        int* __restrict__ retry, int* __restrict__ error) {
    static_assert(sizeof(QtShared) <= QT_STATIC_LDS, "QtShared outgrew its bound");
    __shared__ QtShared sh;
    extern __shared__ __align__(16) unsigned long long s_dyn[];  // q.ldsKeys entries, then q.regionInts ints
    unsigned long long* const s_keys = s_dyn;
    int* const s_region = (int*)(s_dyn + q.ldsKeys);
    if (q.pass != 2) {
        qt_one<(MAXT <= 512 ? 4 : 2)>(g, q, (int)blockIdx.x, frame0, nframes, cellKeys, cellCount, codeTab, sortBuf, nodes, ints, selXY, selScore, levelCount,
               levelKeys, retry, error, sh, s_keys, s_region);
        return;
    }
    // second pass: the levels the first pass listed (its launch is complete: same stream), a few per workgroup.
    // retry = count, workgroups of this pass that are done, count of the last launch, entries.  The last workgroup
    // to finish leaves count and done at zero for the next launch (no clearing launch in between).
    const int listed = min(retry[0], nframes * q.nlevels);
    for (int i = (int)blockIdx.x; i < listed; i += (int)gridDim.x) {
        qt_one<(MAXT <= 512 ? 4 : 2)>(g, q, retry[3 + i], frame0, nframes, cellKeys, cellCount, codeTab, sortBuf, nodes, ints, selXY, selScore, levelCount,
               levelKeys, retry, error, sh, s_keys, s_region);
        __syncthreads();
    }
    // Only the workgroups that had levels to take count themselves done (hundreds of atomics on one word cost more
    // than the clearing launch they replace); the others found nothing to do whether they read the count before or
    // after it was cleared.
    const int takers = min(listed, (int)gridDim.x);
    if ((int)blockIdx.x >= takers) {
        if (listed == 0 && blockIdx.x == 0 && threadIdx.x == 0) retry[2] = 0;
        return;
    }
    if (threadIdx.x == 0) {  // (behind the loop's last barrier: every wavefront of the workgroup has read the count)
        __threadfence();
        if (atomicAdd(&retry[1], 1) == takers - 1) {
            retry[2] = listed;
            retry[0] = 0;
            retry[1] = 0;
        }
    }
}

// LDS block of a launch over all levels.  The histogram arrays (2 nb + 1 ints for nb buckets) and the node
// scratch behind them decide it; the sorted form, which only levels the histogram form gives up on take, puts
// its bucket arrays over the same words, keeps its sorted keys in global memory and its node scratch wherever
// the histogram form had it.
static QtLaunch qt_plan(const OrbGeom& g, bool bigFrames, int pass) {
    QtLaunch q;
    q.pass = pass;
    q.level0 = 0;
    q.nlevels = g.nlevels;
    q.ldsKeys = 0;
    q.nframesMagic = 0;
    int hist = 0, both = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int h = 2 * (g.lv[l].nIni << (2 * (pass == 1 ? g.lv[l].histDepthFirst : g.lv[l].histDepth))) + 1;
        hist = std::max(hist, h);
        both = std::max(both, h + QT_NODE_INTS(g.lv[l].nodeCap));
    }
    // large frames: a workgroup of 1024 threads has its CU to itself, the node scratch of levels with quotas in
    // the hundreds stays in global memory; otherwise every level's scratch sits behind its histogram
    const size_t limit = (bigFrames ? 160 : 64) * 1024 - QT_STATIC_LDS;
    q.regionInts = ((size_t)both * 4 <= limit) ? both : hist;
    q.binBits = 8;
    while (q.binBits < (bigFrames ? QT_MAX_BIN_BITS_BIG : QT_MAX_BIN_BITS) && QT_BIN_INTS(q.binBits + 1) <= q.regionInts) q.binBits++;
    return q;
}

// More (frame, level) pairs than the chip holds workgroups of at once (768 with the full-depth histograms)?
// ORBHIP_QT_TWO_PASS_MIN moves the threshold (tests).
static bool qt_large_launch(int pairs) {
    const char* e = getenv("ORBHIP_QT_TWO_PASS_MIN");
    return pairs >= (e ? atoi(e) : 768);
}

static int qt_launch(const OrbBatch& b, bool bigFrames, int pass) {
    const OrbGeom& g = *b.h_geom;
    // (a few frames: twice the threads shorten the counting pass of the one workgroup everything waits for,
    // level 0's, 39 -> 35 us; four times gain nothing more.  A launch with more workgroups than the chip holds is
    // the opposite case: most of a level's distribution runs on a few dozen threads, so half the threads per
    // workgroup and twice the workgroups per CU do more: 0.148 -> 0.128 ms at 512 frames, 64 threads no better.)
    const int threads = bigFrames ? 1024 : b.nframes < ORB_FEW_FRAMES ? 512 : qt_large_launch(b.nframes * g.nlevels) ? 128 : 256;
    QtLaunch q = qt_plan(g, bigFrames, pass);
    q.nframesMagic = orb_div_magic((uint32_t)b.nframes, (uint64_t)b.nframes * q.nlevels);
    size_t lds = (size_t)q.ldsKeys * 8 + (size_t)q.regionInts * 4;
#ifdef QT_PROFILE
    if (const char* pad = getenv("ORBHIP_QT_LDS_PAD")) lds += (size_t)atoi(pad);  // fewer workgroups per CU, to see what they cost each other
#endif
    if (lds + QT_STATIC_LDS > 160 * 1024) {
        orb_set_error("internal: quadtree LDS block of %zu bytes", lds);
        return ORB_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {
        // More dynamic LDS than the default per-workgroup limit: ask for it, once per device (the attribute belongs
        // to the device's copy of the kernel; contexts of several devices run on several host threads).
        static std::atomic<bool> asked[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !asked[dev].load(std::memory_order_acquire)) {
            hipError_t ea = hipFuncSetAttribute((const void*)k_quadtree<QT_MAX_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - QT_STATIC_LDS);
            if (ea != hipSuccess) {
                orb_set_error("quadtree: %zu bytes of LDS per workgroup refused: %s", lds, hipGetErrorString(ea));
                return ORB_ERR_HIP;
            }
            if (dev >= 0 && dev < 64) asked[dev].store(true, std::memory_order_release);
        }
    }
    // every frame group of a batch has its own list (a count, then at most one entry per (frame, level))
    int* retry = b.d_qtRetry + (int64_t)b.frame0 * (g.nlevels + 1);
    const int all = b.nframes * q.nlevels;
    hipLaunchKernelGGL(bigFrames ? k_quadtree<QT_MAX_THREADS> : k_quadtree<512>, dim3(pass == 2 ? std::min(all, 1024) : all), dim3(threads), lds, (hipStream_t)b.stream, b.d_geom, q,
                       b.frame0, b.nframes, b.d_cellKeys, b.d_cellCount, b.d_codeTab, b.d_sort, b.d_nodes, b.d_ints, b.d_selXY, b.d_selScore, b.d_levelCount,
                       b.d_levelKeys, retry, b.d_error);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("quadtree launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

// A large batch is bound by how many workgroups a CU holds, a smaller one (all of whose workgroups are resident at
// once anyway) by the latency of one workgroup: two passes when the first one's smaller histograms make a
// difference to the former.
bool orb_quadtree_two_passes(const OrbGeom& g, int nframes) {
    // (three frames at least: a frame group's list has room for three words in front of its nframes x nlevels entries)
    if (g.lv[0].keyCap > ORB_QT_BIG_KEYCAP || !qt_large_launch(nframes * g.nlevels) || nframes < 3) return false;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].histDepthFirst < g.lv[l].histDepth) return true;
    return false;
}

int orb_launch_quadtree(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    // large levels (4K frames hand ~1e5 keys to the quadtree) get a full 1024-thread workgroup
    const bool bigFrames = g.lv[0].keyCap > ORB_QT_BIG_KEYCAP;
    if (!orb_quadtree_two_passes(g, b.nframes)) return qt_launch(b, bigFrames, 0);
    const int rc = qt_launch(b, bigFrames, 1);
    return rc ? rc : qt_launch(b, bigFrames, 2);
}
