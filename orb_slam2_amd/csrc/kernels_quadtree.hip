// DistributeOctTree for every (frame, level) of a batch: one workgroup each.
// The algorithm itself is in quadtree_core.h; this file gathers the level's
// FAST keys from the per-cell slots in the reference's order
// (src/ORBextractor.cc:825-870: cell rows, cells, rows inside the cell), sorts
// them by quadtree path and writes the winners in list order.
#include <hip/hip_runtime.h>

#include "quadtree_core.h"

#define QT_MAX_THREADS 1024

__global__ __launch_bounds__(QT_MAX_THREADS) void k_quadtree(const OrbGeom* __restrict__ g, int nframes,
                                                        const uint32_t* __restrict__ cellKeys,
                                                        const int* __restrict__ cellCount, int* __restrict__ cellOff,
                                                        unsigned long long* __restrict__ sortBuf,
                                                        unsigned long long* __restrict__ keyInfo,
                                                        OrbNode* __restrict__ nodes, int* __restrict__ ints,
                                                        uint32_t* __restrict__ selXY, int* __restrict__ selScore,
                                                        int* __restrict__ levelCount, int* __restrict__ levelKeys,
                                                        int* __restrict__ error) {
    __shared__ QtShared sh;
    const int nlevels = g->nlevels;
    const int frame = blockIdx.x / nlevels, level = blockIdx.x - frame * nlevels;
    const OrbLevel& L = g->lv[level];
    const int t = threadIdx.x, T = (int)blockDim.x;
    const int nCells = L.nCols * L.nRows;
    const int64_t cell0 = (int64_t)frame * g->cellsPerFrame + L.cellBase;
    const int* cc = cellCount + cell0;
    int* off = cellOff + (int64_t)frame * (g->cellsPerFrame + nlevels) + L.cellBase + level;  // nCells + 1 entries
    unsigned long long* sorted = sortBuf + ((int64_t)frame * g->keysPerFrame + L.keyBase) * 2;
    unsigned long long* tmp = sorted + L.keyCap;
    unsigned long long* info = keyInfo + (int64_t)frame * g->keysPerFrame + L.keyBase;

    for (int c = t; c < nCells; c += T) off[c] = cc[c];
    __syncthreads();
    qt_scan(off, nCells, &sh);
    const int n = off[nCells];
    if (n > L.keyCap) {  // cannot happen: keyCap is the 2x2-block bound summed over the cells
        if (t == 0) {
            atomicOr(error, 4);
            levelCount[frame * nlevels + level] = 0;
            levelKeys[frame * nlevels + level] = n;
        }
        return;
    }
    const int cellCap = g->cellCap;
    const int height = L.maxBorderY - ORB_BORDER;
    for (int c = t; c < nCells; c += T) {
        const int cnt = cc[c], base = off[c];
        const int ci = c / L.nCols, cj = c - ci * L.nCols;
        const uint32_t* slot = cellKeys + (cell0 + c) * cellCap;
        for (int k = 0; k < cnt; k++) {
            const uint32_t key = slot[k];
            const int x = 3 + cj * L.wCell + (int)(key & 0xff);  // relative to the 16-pixel border, as :864-865
            const int y = 3 + ci * L.hCell + (int)((key >> 8) & 0xff);
            const int r = base + k;
            tmp[r] = ((unsigned long long)qt_code(x, y, L.nIni, L.hX, height) << 32) | (unsigned)r;
            info[r] = (unsigned long long)(unsigned)x | ((unsigned long long)(unsigned)y << 16) |
                      ((unsigned long long)(key >> 16) << 32);
        }
    }
    __syncthreads();
    int count = 0;
    if (n > 0) {
        qt_sort_keys(tmp, sorted, n, L.nIni, &sh);
        QtWork w;
        w.sorted = sorted;
        w.nodeCap = L.nodeCap;
        OrbNode* nb = nodes + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 2;
        w.listA = nb;
        w.listB = nb + L.nodeCap;
        int* p = ints + ((int64_t)frame * g->nodesPerFrame + L.nodeBase) * 9 + (int64_t)(frame * nlevels + level) * 4;
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
        OrbNode* list = nullptr;
        qt_distribute(w, n, L.nIni, L.quota, (g->quirks & ORB_Q_DIVIDE_N3) ? 1 : 0, &sh, &list, &count);
        if (sh.err && t == 0) atomicOr(error, sh.err == 2 ? 8 : 16);
        if (count > L.outCap) {  // cannot happen (quota + 2 bound); fail loudly, never truncate silently
            if (t == 0) atomicOr(error, 32);
            count = L.outCap;
        }
        const int64_t ob = (int64_t)frame * g->outCapFrame + L.outBase;
        qt_select_all(sorted, info, list, count, w.scanA, &sh);
        for (int i = t; i < count; i += T) {
            const unsigned long long ki = info[w.scanA[i]];
            selXY[ob + i] = (uint32_t)ki;
            selScore[ob + i] = (int)(ki >> 32);
        }
    }
    if (t == 0) {
        levelCount[frame * nlevels + level] = count;
        levelKeys[frame * nlevels + level] = n;
    }
}

int orb_launch_quadtree(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    // large levels (4K frames hand ~1e5 keys to the quadtree) get a full 1024-thread workgroup
    const int threads = g.lv[0].keyCap > 400000 ? 1024 : 256;
    hipLaunchKernelGGL(k_quadtree, dim3(b.nframes * g.nlevels), dim3(threads), 0, (hipStream_t)b.stream, b.d_geom,
                       b.nframes, b.d_cellKeys, b.d_cellCount, b.d_cellOff, b.d_sort, b.d_keyInfo, b.d_nodes, b.d_ints,
                       b.d_selXY, b.d_selScore, b.d_levelCount, b.d_levelKeys, b.d_error);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("quadtree launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}
