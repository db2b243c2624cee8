// DistributeOctTree for every (frame, level) of a batch: one workgroup each.
// The algorithm itself is in quadtree_core.h.  This kernel first turns the
// level's FAST keys (per-cell slots) into sort entries numbered in the
// reference's order (src/ORBextractor.cc:825-870: cell rows, cells, rows inside
// the cell), then sorts them by quadtree path, runs the distribution and
// writes the winners in list order.
//
// Levels of everyday frame sizes keep the sorted keys and all node scratch in
// LDS; larger ones (a 4K level holds ~1e5 keys) use the global-memory buffers
// of the same layout.  The kernel is bound by latency, not by issue rate, so
// its speed is proportional to the workgroups a CU can hold: the LDS block is
// sized per launch (QtLaunch) to what the levels of that launch need.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "quadtree_core.h"

#define QT_MAX_THREADS 1024
#define QT_LDS_KEYS_MAX 4096
#define QT_STATIC_LDS 1024     // bound on sizeof(QtShared)

struct QtLaunch {
    int level0, nlevels;  // levels [level0, level0 + nlevels) of every frame
    int ldsKeys;          // sort entries the LDS block holds
    int binBits;          // key-sort buckets = 2^binBits
    int regionInts;       // ints behind the keys: bucket arrays, then node scratch
    unsigned nframesMagic;  // orb_div_magic(frames of the launch)
};

// Sort, distribute, select and emit one level.  Inlined once per memory placement so that the
// compiler sees which address space `sorted` and the node scratch live in.
static __device__ __forceinline__ int qt_level(const OrbGeom* g, const OrbLevel& L, const unsigned long long* tmp,
                                               unsigned long long* sorted, const QtWork& w0, int n,
                                               const uint32_t* __restrict__ levelCellKeys, uint32_t* __restrict__ selXY,
                                               int* __restrict__ selScore, int* __restrict__ error, const QtBins bins,
                                               QtShared* sh) {
    const int t = threadIdx.x, T = (int)blockDim.x;
    qt_sort_keys(tmp, sorted, n, L.nIni, bins, sh);
    __syncthreads();  // the bucket arrays may now be reused as node scratch
    QtWork w = w0;
    w.sorted = sorted;
    OrbNode* list = nullptr;
    int count = 0;
    qt_distribute(w, n, L.nIni, L.quota, (g->quirks & ORB_Q_DIVIDE_N3) ? 1 : 0, sh, &list, &count);
    if (sh->err && t == 0) atomicOr(error, sh->err == 2 ? 8 : 16);
    if (count > L.outCap) {  // cannot happen (quota + 2 bound); fail loudly, never truncate silently
        if (t == 0) atomicOr(error, 32);
        count = L.outCap;
    }
    qt_select_all(sorted, list, count, w.scanA, sh);
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

__global__ __launch_bounds__(QT_MAX_THREADS) void k_quadtree(const OrbGeom* __restrict__ g, const QtLaunch q, int frame0, int nframes,
                                                        const uint32_t* __restrict__ cellKeys,
                                                        const int* __restrict__ cellCount,
                                                        unsigned long long* __restrict__ sortBuf,
                                                        OrbNode* __restrict__ nodes, int* __restrict__ ints,
                                                        uint32_t* __restrict__ selXY, int* __restrict__ selScore,
                                                        int* __restrict__ levelCount, int* __restrict__ levelKeys,
                                                        int* __restrict__ error) {
    static_assert(sizeof(QtShared) <= QT_STATIC_LDS, "QtShared outgrew its bound");
    __shared__ QtShared sh;
    extern __shared__ unsigned long long s_dyn[];  // q.ldsKeys entries, then q.regionInts ints
    unsigned long long* const s_keys = s_dyn;
    int* const s_region = (int*)(s_dyn + q.ldsKeys);
    const QtBins bins = {s_region, s_region + (1 << q.binBits) + 1, q.binBits};
    const int nlevels = g->nlevels;
    // level-major launch order: the levels with the most keys (the longest-running workgroups) start first
    const int levelLocal = (int)orb_div(blockIdx.x, (unsigned)nframes, q.nframesMagic);
    const int level = q.level0 + levelLocal, frame = frame0 + (int)blockIdx.x - levelLocal * nframes;
    const OrbLevel& L = g->lv[level];
    const int t = threadIdx.x, T = (int)blockDim.x;
    unsigned long long* sortedG = sortBuf + ((int64_t)frame * g->keysPerFrame + L.keyBase) * 2;
    unsigned long long* tmp = sortedG + L.keyCap;
    const uint32_t* lck = cellKeys + ((int64_t)frame * g->cellsPerFrame + L.cellBase) * g->cellCap;

    // Gather: cells are taken in chunks whose key offsets fit the LDS bucket array; every thread then
    // owns keys (not cells), finds each key's cell by binary search over the chunk's offsets and
    // writes its sort entry.  All loads of a thread are independent of each other.
    const int* cc = cellCount + (int64_t)frame * g->cellsPerFrame + L.cellBase;
    const int nCells = L.nCols * L.nRows, cellCap = g->cellCap;
    const int height = L.maxBorderY - ORB_BORDER;
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
                tmp[n + r] = qt_entry(qt_code(x, y, L.nIni, L.hX, height), (uint32_t)(cell * cellCap + k), key >> 16);
        }
        n += chunk;
        __syncthreads();
    }
    if (t == 0) {
        levelKeys[frame * nlevels + level] = n;
        if (n > error[1]) atomicMax(&error[1], n);  // sizes the LDS block of later launches
    }
    if (n > L.keyCap) {  // cannot happen: keyCap is the 2x2-block bound summed over the cells
        if (t == 0) {
            atomicOr(error, 4);
            levelCount[frame * nlevels + level] = 0;
        }
        return;
    }
    int count = 0;
    if (n > 0) {
        const int64_t ob = (int64_t)frame * g->outCapFrame + L.outBase;
        QtWork w;
        if (n <= q.ldsKeys && QT_NODE_INTS(L.nodeCap) <= q.regionInts) {
            qt_work_from_pool(&w, s_region, L.nodeCap);
            count = qt_level(g, L, tmp, s_keys, w, n, lck, selXY + ob, selScore + ob, error, bins, &sh);
        } else {
            OrbNode* nb = nodes + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 2;
            int* p = ints + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 9 + (int64_t)(frame * nlevels + level) * 4;
            w.nodeCap = L.nodeCap;
            w.listA = nb;
            w.listB = nb + L.nodeCap;
            w.rec = p;
            p += L.nodeCap;
            w.recSorted = p;
            p += L.nodeCap;
            w.bounds = p;
            p += 4 * L.nodeCap;
            w.scanA = p;
            p += L.nodeCap + 1;
            w.scanB = p;
            p += L.nodeCap + 1;
            w.scanC = p;
            count = qt_level(g, L, tmp, sortedG, w, n, lck, selXY + ob, selScore + ob, error, bins, &sh);
        }
    }
    if (t == 0) levelCount[frame * nlevels + level] = count;
}

// LDS block for levels [level0, level0 + nlevels): the bucket arrays and the largest node pool of these
// levels share the region behind the keys; levels whose node pool would push the block past 64 KB
// (quotas in the thousands) run from global memory anyway.
static QtLaunch qt_plan(const OrbGeom& g, int level0, int nlevels, int ldsKeys) {
    QtLaunch q;
    q.level0 = level0;
    q.nlevels = nlevels;
    q.ldsKeys = ldsKeys;
    int pool = 0;
    for (int l = level0; l < level0 + nlevels; l++) pool = std::max(pool, QT_NODE_INTS(g.lv[l].nodeCap));
    // about four keys per bucket when the block is full; more buckets when the node pool leaves the room
    for (int attempt = 0; attempt < 2; attempt++) {
        int bits = 8;
        while ((4 << bits) < ldsKeys && bits < QT_MAX_BIN_BITS) bits++;
        while (bits < QT_MAX_BIN_BITS && QT_BIN_INTS(bits + 1) <= pool) bits++;
        q.binBits = bits;
        q.regionInts = std::max(pool, QT_BIN_INTS(bits));
        if ((size_t)ldsKeys * 8 + (size_t)q.regionInts * 4 + QT_STATIC_LDS <= 65536) break;
        // such levels use the global buffers: no keys in LDS, and all the buckets the sort can use
        pool = 0;
        ldsKeys = 0;
        q.ldsKeys = 0;
        q.binBits = QT_MAX_BIN_BITS;
        q.regionInts = QT_BIN_INTS(QT_MAX_BIN_BITS);
        break;
    }
    return q;
}

int orb_launch_quadtree(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    // large levels (4K frames hand ~1e5 keys to the quadtree) get a full 1024-thread workgroup
    const int threads = g.lv[0].keyCap > 400000 ? 1024 : 256;
    // keys in LDS: the largest level met so far plus an eighth; the full block until a count has arrived
    int ldsKeys = QT_LDS_KEYS_MAX;
    if (b.qtKeysSeen > 0) ldsKeys = std::min(QT_LDS_KEYS_MAX, std::max(512, (b.qtKeysSeen + b.qtKeysSeen / 8 + 63) & ~63));
    QtLaunch q = qt_plan(g, 0, g.nlevels, ldsKeys);
    if (q.ldsKeys > 0 && q.ldsKeys < QT_LDS_KEYS_MAX) {
        // The block decides how many workgroups a CU holds; between two such steps a larger block costs
        // nothing, so the keys get all the room up to the next step (measured: four blocks of 40 908 bytes
        // fit the 160 KB of a CU, four of 41 036 do not; sizes are rounded up to 512 bytes here to stay clear
        // of the allocation granularity).
        const size_t unit = 512, perCU = 160 * 1024;
        auto block_bytes = [&](const QtLaunch& p) {
            return (sizeof(QtShared) + (size_t)p.ldsKeys * 8 + (size_t)p.regionInts * 4 + unit - 1) / unit * unit;
        };
        const size_t wgs = perCU / block_bytes(q);
        if (wgs >= 1) {
            const size_t room = perCU / wgs / unit * unit;
            const int more = (int)((room - block_bytes(q)) / 8);
            const QtLaunch q2 = qt_plan(g, 0, g.nlevels, std::min(QT_LDS_KEYS_MAX, q.ldsKeys + more));
            if (q2.ldsKeys > 0 && block_bytes(q2) <= room) q = q2;
        }
    }
    q.nframesMagic = orb_div_magic((uint32_t)b.nframes, (uint64_t)b.nframes * q.nlevels);
    const size_t lds = (size_t)q.ldsKeys * 8 + (size_t)q.regionInts * 4;
    hipLaunchKernelGGL(k_quadtree, dim3(b.nframes * q.nlevels), dim3(threads), lds, (hipStream_t)b.stream, b.d_geom, q,
                       b.frame0, b.nframes, b.d_cellKeys, b.d_cellCount, b.d_sort, b.d_nodes, b.d_ints, b.d_selXY, b.d_selScore, b.d_levelCount,
                       b.d_levelKeys, b.d_error);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("quadtree launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}
