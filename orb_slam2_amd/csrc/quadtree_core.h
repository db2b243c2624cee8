// Quadtree keypoint distribution for one (frame, level), written once for the
// device (one workgroup, QT_NT threads, phases separated by workgroup
// barriers) and for a single-threaded host build used by the unit tests
// (ORB_HOST_EMU: one "thread", barriers are no-ops).
//
// Follows ORBextractor::DistributeOctTree / ExtractorNode::DivideNode of the
// reference (src/ORBextractor.cc:511-796) with these observations:
//  * a node's rectangle is fixed by its initial node and its path of quadrant
//    choices, and a key's path depends only on its own coordinates; keys are
//    therefore sorted once by (initial node, path) and every node owns a
//    contiguous range of that order — dividing a node is three binary searches;
//  * a coarse pass divides every divisible node, so the new list is
//    [children of the last divisible node (n4..n1), ..., children of the first]
//    followed by the untouched nodes in their old order — prefix sums;
//  * the fine phase divides recorded nodes by decreasing (size, creation
//    number) and stops at the first node that brings the list to the quota —
//    a prefix sum over that order finds the stopping point;
//  * the per-node winner is the largest response, earliest key on ties.
#pragma once
#include <stdint.h>

#include "orb_internal.h"

#ifdef ORB_HOST_EMU
#define QT_FN static inline
#define QT_BARRIER() ((void)0)
#define QT_TID 0
#define QT_NT 1
#define QT_NT_SHIFT 0
#define QT_FDIV(a, b) ((a) / (b))
#define QT_ATOMIC_ADD(p, v) qt_host_fetch_add((p), (v))
#define QT_ATOMIC_MAX32(p, v) (*(p) = (*(p) > (v)) ? *(p) : (v))
static inline int qt_host_fetch_add(int* p, int v) {
    const int o = *p;
    *p = o + v;
    return o;
}
#else
#define QT_FN static __device__ __forceinline__
#define QT_BARRIER() __syncthreads()
#define QT_TID ((int)threadIdx.x)
#define QT_NT ((int)blockDim.x)
#define QT_NT_SHIFT (31 - __clz((int)blockDim.x))  // the kernel runs with a power of two of threads: a shift, not a division
#define QT_FDIV(a, b) __fdiv_rn((a), (b))
#define QT_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define QT_ATOMIC_MAX32(p, v) atomicMax((p), (v))
#endif

// Development aid: with -DQT_PROFILE the first workgroup of a launch stamps the 100 MHz wall clock at phase
// boundaries into g_qtProfile (kernels_quadtree.hip, read back with orb_debug_quadtree_profile).
#if defined(QT_PROFILE) && !defined(ORB_HOST_EMU)
extern __device__ unsigned long long g_qtProfile[32];
#ifdef QT_PROFILE_SUM
// ... or, with -DQT_PROFILE_SUM as well, every workgroup adds the engine cycles since its previous stamp to
// g_qtProfile[k] (and counts itself in g_qtProfile[k + 16] for k < 16): where the workgroups of a full launch
// spend their time
static __shared__ unsigned long long g_qtPrev;
#define QT_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0) {                                                                       \
            const unsigned long long now_ = clock64();                                                \
            if ((k) != 0) atomicAdd(&g_qtProfile[k], now_ - g_qtPrev);                                \
            g_qtPrev = now_;                                                                          \
        }                                                                                             \
    } while (0)
#else
#define QT_STAMP(k)                                                                   \
    do {                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_qtProfile[k] = wall_clock64();     \
    } while (0)
#endif
#define QT_COUNT(k)                                                       \
    do {                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_qtProfile[k] += 1ull;  \
    } while (0)
#else
#define QT_STAMP(k) ((void)0)
#define QT_COUNT(k) ((void)0)
#endif

#define QT_NOMORE 0x100
#define QT_DEPTH(n) ((n).depthFlags & 0xff)
// depthFlags also carries, from bit 9 up, the first bucket of the node when the keys are held as a histogram
#define QT_B0_SHIFT 9
#define QT_B0(n) ((int)((unsigned)(n).depthFlags >> QT_B0_SHIFT))
#ifdef ORB_HOST_EMU
#define QT_MFN inline
#else
#define QT_MFN __device__ __forceinline__
#endif

#define QT_MAX_BIN_BITS 12      // the key sort uses 2^bits buckets, bits chosen per launch (QtBins)
#define QT_MAX_BIN_BITS_BIG 14  // ... for the levels of large frames, whose workgroups take a CU's LDS for themselves
#define QT_SMALL_BIN 40   // buckets up to this size are ordered by counting, larger ones by the bitonic network
#ifdef ORB_HOST_EMU
#define QT_RANK_K 64      // keys one thread holds per round of that counting (the emulation has one thread)
#else
#define QT_RANK_K 8
#endif

#define QT_BIG_LIST 128   // nodes handed to the whole workgroup by the winner selection
#define QT_BIN_INTS(bits) ((2 << (bits)) + 1)  // ints of bucket scratch: 2^bits + 1 starts, 2^bits cursors
#define QT_NODE_INTS(cap) (17 * (cap) + 3)  // ints of node scratch one (frame, level) needs, see qt_work_from_pool

// Sort entry: quadtree path << 32 | rank << 8 | response.  The rank (24 bits) numbers the level's keys
// in the reference's order (cell rows, cells, rows inside the cell, with gaps); entries are unique, and
// ascending order of the whole word is (path, rank) order.
#define QT_RANK_BITS 24
QT_FN unsigned long long qt_entry(uint32_t code, uint32_t rank, uint32_t score) {
    return ((unsigned long long)code << 32) | ((unsigned long long)rank << 8) | (unsigned long long)(score & 255u);
}
QT_FN uint32_t qt_entry_rank(unsigned long long e) { return (uint32_t)(e >> 8) & 0xffffffu; }
// Winner order (src/ORBextractor.cc:777-793): largest response, earliest key on ties.
QT_FN uint32_t qt_entry_prio(unsigned long long e) {
    return (((uint32_t)e & 255u) << 24) | (~(uint32_t)(e >> 8) & 0xffffffu);
}

struct QtShared {
    // per-wavefront partial sums of the block scans: ints (qt_scan) or three counts packed into one word (qt_scan3)
    union {
        int part[64];
        unsigned long long part3[16];
    };
    int bigList[QT_BIG_LIST];
    int nBig;
    unsigned best32;
    int total;
    int nL, prevSize, nToExpand, seqCounter, nRec, finish, fine, stop, err;
    int deeper;  // histogram form only: a node with several keys would have to be divided below the bucket depth
};

// Key-sort buckets.  Once the keys are sorted the same words serve as node scratch when a level's
// node pool fits (QT_NODE_INTS ints from `start` on).
struct QtBins {
    int* start;   // 2^bits + 1
    int* cursor;  // 2^bits
    int bits;
};

struct QtWork {  // scratch of one (frame, level): LDS when it fits, global memory otherwise
    const unsigned long long* sorted;  // n entries, ascending
    OrbNode* listA;              // nodeCap
    OrbNode* listB;              // nodeCap
    int* rec;                    // nodeCap: positions (in the current list) of recorded nodes
    int* recSorted;              // nodeCap
    int* bounds;                 // nodeCap * 4: child boundaries b1,b2,b3 and flags per node
    int* scanA;                  // nodeCap + 1
    int* scanB;                  // nodeCap + 1
    int* scanC;                  // nodeCap + 1
    int nodeCap;
};

// Lays the node scratch out in `pool` (QT_NODE_INTS(nodeCap) ints).
QT_FN void qt_work_from_pool(QtWork* w, int* pool, int nodeCap) {
    w->nodeCap = nodeCap;
    w->listA = (OrbNode*)pool;
    pool += 4 * nodeCap;
    w->listB = (OrbNode*)pool;
    pool += 4 * nodeCap;
    w->rec = pool;
    pool += nodeCap;
    w->recSorted = pool;
    pool += nodeCap;
    w->bounds = pool;
    pool += 4 * nodeCap;
    w->scanA = pool;
    pool += nodeCap + 1;
    w->scanB = pool;
    pool += nodeCap + 1;
    w->scanC = pool;
}

// Quadtree path of a key: 4 bits initial node, then ORB_CODE_DEPTH two-bit
// quadrant digits (0 = n1 left/top, 1 = n2 right/top, 2 = n3 left/bottom,
// 3 = n4 right/bottom), src/ORBextractor.cc:516-559, 583-602.  Left/right
// choices depend on x alone (the initial node and the halving of its x range),
// top/bottom choices on y alone, so the path is the OR of a word made from x
// and a word made from y; the host tabulates both per level (qt_code_x over the
// level's columns, qt_code_y over its rows) and the kernel looks them up.
#ifdef __HIPCC__
#define QT_HD static __host__ __device__ __forceinline__
#else
#define QT_HD static inline
#endif
#ifdef __HIP_DEVICE_COMPILE__
#define QT_FDIV_HD(a, b) __fdiv_rn((a), (b))
#else
#define QT_FDIV_HD(a, b) ((a) / (b))  // IEEE division: the host is built without fast-math
#endif
QT_HD uint32_t qt_code_x(int x, int nIni, float hX) {
    int ini = (int)QT_FDIV_HD((float)x, hX);
    if (ini > nIni - 1) ini = nIni - 1;  // (never taken for in-range keys; keeps the index safe)
    int x0 = (int)(hX * (float)ini), x1 = (int)(hX * (float)(ini + 1));
    uint32_t code = (uint32_t)ini;
    for (int d = 0; d < ORB_CODE_DEPTH; d++) {
        const int halfX = (x1 - x0 + 1) >> 1;  // ceil((UR.x-UL.x)/2)
        const int mx = x0 + halfX;
        const int right = !(x < mx);
        code = (code << 2) | (uint32_t)right;
        if (right) x0 = mx; else x1 = mx;
    }
    return code;
}
QT_HD uint32_t qt_code_y(int y, int height) {
    int y0 = 0, y1 = height;
    uint32_t code = 0u;
    for (int d = 0; d < ORB_CODE_DEPTH; d++) {
        const int halfY = (y1 - y0 + 1) >> 1;
        const int my = y0 + halfY;
        const int bottom = !(y < my);
        code = (code << 2) | (uint32_t)(bottom * 2);
        if (bottom) y0 = my; else y1 = my;
    }
    return code;
}
QT_HD uint32_t qt_code(int x, int y, int nIni, float hX, int height) {
    return qt_code_x(x, nIni, hX) | qt_code_y(y, height);
}

// First index in [lo, hi) whose code is >= key (codes are the high 32 bits).
QT_FN int qt_lower_bound(const unsigned long long* s, int lo, int hi, uint32_t key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((uint32_t)(s[mid] >> 32) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Exclusive prefix sum of a[0..n) in place; a[n] and sh->total receive the sum.
// Each thread sums a contiguous chunk; the per-thread sums are scanned inside
// each wavefront with shuffles and across wavefronts through LDS.
QT_FN void qt_scan(int* a, int n, QtShared* sh) {
    const int T = QT_NT, t = QT_TID;
    const int chunk = (n + T - 1) >> QT_NT_SHIFT;
    const int b = t * chunk, e = (b + chunk < n) ? b + chunk : n;
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
#ifdef ORB_HOST_EMU
    int run = 0;
    sh->total = s;
    a[n] = s;
#else
    const int lane = t & 63, wv = t >> 6, nw = (T + 63) >> 6;
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) sh->part[wv] = incl;
    QT_BARRIER();
    int wprefix = 0, total = 0;
    for (int i = 0; i < nw; i++) {
        const int v = sh->part[i];
        if (i < wv) wprefix += v;
        total += v;
    }
    int run = wprefix + incl - s;
    if (t == 0) {
        sh->total = total;
        a[n] = total;
    }
#endif
    for (int i = b; i < e; i++) {
        const int v = a[i];
        a[i] = run;
        run += v;
    }
    QT_BARRIER();
}

// The same for an array that starts 16-byte aligned and whose length is a multiple of four (the bucket counts):
// a thread's chunk is read and written as four-word accesses and held in registers in between, a quarter of the
// LDS instructions and hardly any bank conflicts (a chunk of eight words read word by word is an eight-way one).
#define QT_SCAN_REG 16  // words of a chunk the registers hold
QT_FN void qt_scan_buckets(int* a, int n, QtShared* sh) {
#ifdef ORB_HOST_EMU
    qt_scan(a, n, sh);
#else
    const int T = QT_NT, t = QT_TID;
    const int chunk = (((n + T - 1) >> QT_NT_SHIFT) + 3) & ~3;
    if (chunk > QT_SCAN_REG || (n & 3)) {
        qt_scan(a, n, sh);
        return;
    }
    const int b = t * chunk;
    int4 v[QT_SCAN_REG / 4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < QT_SCAN_REG / 4; k++) {
        v[k] = (4 * k < chunk && b + 4 * k < n) ? *(const int4*)(a + b + 4 * k) : make_int4(0, 0, 0, 0);
        s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    const int lane = t & 63, wv = t >> 6, nw = (T + 63) >> 6;
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) sh->part[wv] = incl;
    QT_BARRIER();
    int wprefix = 0, total = 0;
    for (int i = 0; i < nw; i++) {
        const int u = sh->part[i];
        if (i < wv) wprefix += u;
        total += u;
    }
    int run = wprefix + incl - s;
    if (t == 0) {
        sh->total = total;
        a[n] = total;
    }
#pragma unroll
    for (int k = 0; k < QT_SCAN_REG / 4; k++) {
        if (4 * k < chunk && b + 4 * k < n) {
            int4 o;
            o.x = run;
            o.y = o.x + v[k].x;
            o.z = o.y + v[k].y;
            o.w = o.z + v[k].z;
            run = o.w + v[k].w;
            *(int4*)(a + b + 4 * k) = o;
        }
    }
    QT_BARRIER();
#endif
}

// Three exclusive prefix sums at once: a, b (and c unless null) hold small non-negative counts whose
// totals stay below 2^21; they travel packed in one 64-bit word, so the three scans cost one pass over
// the data, one shuffle ladder and two barriers.  a[n], b[n], c[n] receive the totals.
QT_FN void qt_scan3(int* a, int* b, int* c, int n, QtShared* sh) {
    const int T = QT_NT, t = QT_TID;
    const int chunk = (n + T - 1) >> QT_NT_SHIFT;
    const int lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    unsigned long long s = 0ull;
    for (int i = lo; i < hi; i++)
        s += (unsigned long long)a[i] | ((unsigned long long)b[i] << 21) | ((unsigned long long)(c ? c[i] : 0) << 42);
#ifdef ORB_HOST_EMU
    unsigned long long run = 0ull;
    const unsigned long long total = s;
#else
    const int lane = t & 63, wv = t >> 6, nw = (T + 63) >> 6;
    unsigned long long incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned vl = (unsigned)__shfl_up((int)(unsigned)incl, o, 64);
        const unsigned vh = (unsigned)__shfl_up((int)(unsigned)(incl >> 32), o, 64);
        if (lane >= o) incl += ((unsigned long long)vh << 32) | vl;
    }
    if (lane == 63) sh->part3[wv] = incl;
    QT_BARRIER();
    unsigned long long wprefix = 0ull, total = 0ull;
    for (int i = 0; i < nw; i++) {
        const unsigned long long v = sh->part3[i];
        if (i < wv) wprefix += v;
        total += v;
    }
    unsigned long long run = wprefix + incl - s;
#endif
    if (t == 0) {
        a[n] = (int)(total & 0x1fffffull);
        b[n] = (int)((total >> 21) & 0x1fffffull);
        if (c) c[n] = (int)(total >> 42);
    }
    for (int i = lo; i < hi; i++) {
        const unsigned long long v = (unsigned long long)a[i] | ((unsigned long long)b[i] << 21) | ((unsigned long long)(c ? c[i] : 0) << 42);
        a[i] = (int)(run & 0x1fffffull);
        b[i] = (int)((run >> 21) & 0x1fffffull);
        if (c) c[i] = (int)(run >> 42);
        run += v;
    }
    QT_BARRIER();
}

// Child boundaries of a node: b[0..2] = first key of n2, n3, n4 inside [lo,hi).
// Returns 0 and sets *err when the node is deeper than the stored path.
QT_FN void qt_children(const unsigned long long* s, const OrbNode& nd, int* b, int* err) {
    const int d = QT_DEPTH(nd);
    if (d >= ORB_CODE_DEPTH) {  // cannot happen for frames within the size limit; fail loudly
        *err = 1;
        b[0] = b[1] = b[2] = nd.hi;
        return;
    }
    const int shift = 2 * (ORB_CODE_DEPTH - 1 - d);
    const uint32_t prefix = (uint32_t)(s[nd.lo] >> 32) >> shift >> 2 << 2;  // parent's digits, child digit cleared
    if (nd.hi - nd.lo == 1) {
        const int q = (int)(((uint32_t)(s[nd.lo] >> 32) >> shift) & 3);
        b[0] = q >= 1 ? nd.lo : nd.hi;
        b[1] = q >= 2 ? nd.lo : nd.hi;
        b[2] = q >= 3 ? nd.lo : nd.hi;
        return;
    }
    // three lower bounds over the same range, advanced together: their loads are independent of each other,
    // so the three searches take the time of one
    const uint32_t k1 = (prefix | 1u) << shift, k2 = (prefix | 2u) << shift, k3 = (prefix | 3u) << shift;
    int lo1 = nd.lo, hi1 = nd.hi, lo2 = nd.lo, hi2 = nd.hi, lo3 = nd.lo, hi3 = nd.hi;
    while (lo1 < hi1 || lo2 < hi2 || lo3 < hi3) {
        // (a finished search probes index lo == hi, which may be nd.hi: clamped into the range, result ignored)
        const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1, m3 = (lo3 + hi3) >> 1;
        const int last = nd.hi - 1;
        const uint32_t c1 = (uint32_t)(s[m1 < last ? m1 : last] >> 32);
        const uint32_t c2 = (uint32_t)(s[m2 < last ? m2 : last] >> 32);
        const uint32_t c3 = (uint32_t)(s[m3 < last ? m3 : last] >> 32);
        if (lo1 < hi1) { if (c1 < k1) lo1 = m1 + 1; else hi1 = m1; }
        if (lo2 < hi2) { if (c2 < k2) lo2 = m2 + 1; else hi2 = m2; }
        if (lo3 < hi3) { if (c3 < k3) lo3 = m3 + 1; else hi3 = m3; }
    }
    b[0] = lo1;
    b[1] = lo2;
    b[2] = lo3;
}

// Bucket depth of the histogram form for a level with nIni initial nodes and quota N: the coarse passes run at
// least until 4^d nIni >= N (depth dmin) and the fine phase goes one deeper, so dmin + 2 digits cover ordinary
// images with a level to spare; fewer when that many buckets exceed `bucketCap` (at least one digit).
static inline int qt_hist_depth(int nIni, int N, int bucketCap, int margin = 2) {
    int dmin = 0;
    while (((long long)nIni << (2 * dmin)) < N && dmin < ORB_CODE_DEPTH) dmin++;
    int d = dmin + margin;
    if (d > ORB_CODE_DEPTH) d = ORB_CODE_DEPTH;
    while (d > 1 && ((long long)nIni << (2 * d)) > bucketCap) d--;
    return d;
}

// ---- the two forms the keys of a level are held in ------------------------------------------------------
// A "key space" answers the three questions the distribution asks about keys: which index range an initial
// node owns, where the children of a node begin, and which key of a node wins.  Node ranges [lo, hi) are key
// index ranges in both forms, so everything else (list bookkeeping, sizes, orders) is shared.
//
// QtSortedSpace: the keys as an array sorted by (path, rank); boundaries by binary search, winner by a scan.
struct QtSortedSpace {
    const unsigned long long* s;
    int n;
    QT_MFN void ini_range(int i, int nIni, int* lo, int* hi, int* b0) const {
        *lo = qt_lower_bound(s, 0, n, (uint32_t)i << (2 * ORB_CODE_DEPTH));
        *hi = (i + 1 == nIni) ? n : qt_lower_bound(s, *lo, n, (uint32_t)(i + 1) << (2 * ORB_CODE_DEPTH));
        *b0 = 0;
    }
    QT_MFN void children(const OrbNode& nd, int* b, int* err, int* deeper) const {
        (void)deeper;
        qt_children(s, nd, b, err);
    }
    QT_MFN int child_b0(const OrbNode&, int) const { return 0; }
    QT_MFN int jump_depth() const { return 0; }  // (no shortcut over the first coarse passes in this form)
    QT_MFN int count_at(int, int) const { return 0; }
    QT_MFN int cell_lo(int, int) const { return 0; }
    QT_MFN int items(const OrbNode& nd) const { return nd.hi - nd.lo; }
    QT_MFN uint32_t prio_at(const OrbNode& nd, int i) const { return qt_entry_prio(s[nd.lo + i]); }
    static constexpr bool kWinnerTable = false;
};

// QtHistSpace: no key array at all.  The keys are counted into buckets = quadtree cells of depth Db (bucket number
// = the first 4 + 2 Db bits of the path), start[] is the prefix sum of the counts and best[] the largest priority
// word (response << 24 | ~rank) per bucket.  A node of depth d <= Db is a run of 4^(Db - d) buckets: its children
// begin at three entries of start[], its winner is the largest entry of best[] over the run.  Below the bucket
// depth only nodes with a single key can be followed (the key's own path says which child it falls into; the
// fork's rule divides such nodes over and over); a node with several keys there sets *deeper and the caller
// runs the level again in the sorted form.  RankCode maps a key's rank to its path.
template <class RankCode>
struct QtHistSpace {
    const int* start;      // nb + 1
    const uint32_t* best;  // nb
    int Db;
    RankCode codeOf;
    QT_MFN void ini_range(int i, int nIni, int* lo, int* hi, int* b0) const {
        (void)nIni;
        const int span = 1 << (2 * Db);
        *b0 = i * span;
        *lo = start[i * span];
        *hi = start[(i + 1) * span];
    }
    QT_MFN void children(const OrbNode& nd, int* b, int* err, int* deeper) const {
        const int d = QT_DEPTH(nd), b0 = QT_B0(nd);
        if (d >= ORB_CODE_DEPTH) {
            *err = 1;
            b[0] = b[1] = b[2] = nd.hi;
            return;
        }
        if (d < Db) {
            const int span = 1 << (2 * (Db - d - 1));
            b[0] = start[b0 + span];
            b[1] = start[b0 + 2 * span];
            b[2] = start[b0 + 3 * span];
            return;
        }
        if (nd.hi - nd.lo != 1) {
            *deeper = 1;
            b[0] = b[1] = b[2] = nd.hi;
            return;
        }
        const uint32_t code = codeOf(~best[b0] & 0xffffffu);
        const int q = (int)((code >> (2 * (ORB_CODE_DEPTH - 1 - d))) & 3u);
        b[0] = q >= 1 ? nd.lo : nd.hi;
        b[1] = q >= 2 ? nd.lo : nd.hi;
        b[2] = q >= 3 ? nd.lo : nd.hi;
    }
    QT_MFN int child_b0(const OrbNode& nd, int q) const {
        const int d = QT_DEPTH(nd);
        return d < Db ? QT_B0(nd) + q * (1 << (2 * (Db - d - 1))) : QT_B0(nd);
    }
    // Every quadtree cell down to the bucket depth is a difference of two entries of start[]: the first coarse
    // passes can be evaluated from the counts alone (qt_jump).
    QT_MFN int jump_depth() const { return Db; }
    QT_MFN int cell_lo(int d, int c) const { return start[c << (2 * (Db - d))]; }
    QT_MFN int count_at(int d, int c) const { return start[(c + 1) << (2 * (Db - d))] - start[c << (2 * (Db - d))]; }
    QT_MFN int items(const OrbNode& nd) const {
        const int d = QT_DEPTH(nd);
        return d < Db ? 1 << (2 * (Db - d)) : 1;
    }
    QT_MFN uint32_t prio_at(const OrbNode& nd, int i) const { return best[QT_B0(nd) + i]; }  // 0: empty bucket
    // A node of depth d < Db is one quadtree cell = an aligned run of 4^(Db - d) buckets, so the winners of all
    // possible nodes form a pyramid over best[]: level k holds the largest entry of every aligned run of 4^k
    // buckets.  Built once the distribution is over (start[] is free then and holds the levels one after the
    // other: fewer than nb / 3 + Db words for nb = nIni 4^Db buckets), it turns a node's winner into one read.
    static constexpr bool kWinnerTable = true;
    QT_MFN int winner_levels() const { return Db; }
    QT_MFN void winner_table_build(int nIni) const {
        const int T = QT_NT, t = QT_TID;
        uint32_t* const tab = (uint32_t*)start;
        const uint32_t* src = best;
        int o = 0;
        for (int k = 1; k <= Db; k++) {
            const int m = nIni << (2 * (Db - k));
            for (int i = t; i < m; i += T) {
                const uint32_t a = src[4 * i], b = src[4 * i + 1], c = src[4 * i + 2], d = src[4 * i + 3];
                const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
                tab[o + i] = ab > cd ? ab : cd;
            }
            QT_BARRIER();
            src = tab + o;
            o += m;
        }
    }
    QT_MFN uint32_t winner_of(const OrbNode& nd, int nIni) const {
        const int d = QT_DEPTH(nd), b0 = QT_B0(nd);
        if (d >= Db) return best[b0];
        const int k = Db - d;
        // level k starts behind levels 1 .. k - 1: nIni (4^(Db-1) + .. + 4^(Db-k+1)) words
        const int o = (int)((((long long)nIni << (2 * Db)) - ((long long)nIni << (2 * (Db - k + 1)))) / 3);
        return ((const uint32_t*)start)[o + (b0 >> (2 * k))];
    }
};

QT_FN int qt_child_flags(int q, int cnt, int n3quirk) {
    // src/ORBextractor.cc:561-568: one-key children stop; with the fork's stray
    // ';' n3 always stops and n4 never does.
    if (n3quirk) {
        if (q == 2) return QT_NOMORE;
        if (q == 3) return 0;
    }
    return cnt == 1 ? QT_NOMORE : 0;
}

// Writes the children of `nd` (boundaries b) in list order n4,n3,n2,n1 starting
// at dst[pos]; creation numbers run n1..n4 from seq0.  Children with more than
// one key are recorded at rec[recPos..].
template <class Space>
QT_FN void qt_emit_children(const Space& sp, const OrbNode& nd, const int* b, OrbNode* dst, int pos, int seq0, int* rec,
                            int recPos, int n3quirk) {
    const int lo[4] = {nd.lo, b[0], b[1], b[2]};
    const int hi[4] = {b[0], b[1], b[2], nd.hi};
    int seqs[4], s = seq0;
    for (int q = 0; q < 4; q++) seqs[q] = (hi[q] > lo[q]) ? s++ : -1;
    const int d = QT_DEPTH(nd) + 1;
    for (int q = 3; q >= 0; q--) {
        const int cnt = hi[q] - lo[q];
        if (cnt <= 0) continue;
        OrbNode c;
        c.lo = lo[q];
        c.hi = hi[q];
        c.depthFlags = d | qt_child_flags(q, cnt, n3quirk) | (sp.child_b0(nd, q) << QT_B0_SHIFT);
        c.seq = seqs[q];
        dst[pos] = c;
        pos++;
    }
    // record in creation order (n1..n4); the position of child q in dst is pos_end - 1 - (index among non-empty)
    int idx = 0, nonEmpty = 0;
    for (int q = 0; q < 4; q++) nonEmpty += (hi[q] > lo[q]);
    const int base = pos - nonEmpty;
    for (int q = 0; q < 4; q++) {
        const int cnt = hi[q] - lo[q];
        if (cnt <= 0) continue;
        if (cnt > 1) rec[recPos++] = base + (nonEmpty - 1 - idx);
        idx++;
    }
}

// ---- the first coarse passes in one step ------------------------------------------------------------------
// While a coarse pass neither ends the distribution nor hands over to the fine phase, the next pass divides
// every divisible node again (src/ORBextractor.cc:625-701), so what the list looks like after K such passes does
// not depend on the lists in between:
//  * it holds the non-empty quadtree cells of depth K all of whose ancestors were divisible, followed by the
//    cells of depth K - 1, K - 2, .. 0 that stopped being divisible there (one key; with the fork's rule every
//    n3), ancestors divisible likewise;
//  * a pass puts the children of the LAST divisible node first and n4 before n1, and moves what stays behind them
//    in the old order; so the cells of one depth d stand in the order of their paths with digit j (1 .. d)
//    descending when d - j is even, ascending when it is odd, and the initial node descending when d is odd;
//  * whether pass p is the last one follows from three counts per depth: cells of the list's front block T[p],
//    those of them with several keys E[p] (what the fine phase would divide), and those that stop D[p]:
//    the list holds T[p] + D[0] + .. + D[p - 1] nodes after pass p.
// qt_jump classifies every cell of depths 0 .. Dj at once (a cell's count is a difference of two prefix sums;
// its ancestors are a few more), finds K from the counts, and writes the list after pass K with one prefix sum.
// Creation numbers: the front block's, which the fine phase orders by, are exact; cells of the tail (which are
// never divided again) get -1.
#define QT_JUMP_MAX 12  // deepest pass the shortcut evaluates (bounded by the scratch below and the bucket depth)
#ifdef ORB_HOST_EMU
static int g_qtJumpStats[QT_JUMP_MAX + 4];  // tests: how often the shortcut ended in a finish [0], the fine phase [1], another coarse pass [2]; by K from [3]
#endif
// The three counts of a depth travel as one 64-bit word (21 bits each), so that a cell adds itself with one atomic.
#define QT_AGG_T(v) ((int)((v) & 0x1fffffull))
#define QT_AGG_E(v) ((int)(((v) >> 21) & 0x1fffffull))
#define QT_AGG_D(v) ((int)((v) >> 42))
// bits of a cell's class
#define QT_C_EXISTS 1
#define QT_C_LIVE 2
#define QT_C_MULTI 4

template <class Space>
QT_FN int qt_cell_class(const Space& sp, int d, int c, int n3quirk) {
    const int cnt = sp.count_at(d, c);
    // every ancestor divisible?  (depth 0: more than one key; below: the child's flag, qt_child_flags.)  No early
    // exit: the loads of all ancestors are independent of each other and go out together.
    bool ok = cnt > 0;
    for (int a = d - 1; a >= 0; a--) {
        const int ca = c >> (2 * (d - a));
        const int q = ca & 3;
        const bool several = sp.count_at(a, ca) > 1;
        const bool byQuadrant = a > 0 && n3quirk && q >= 2;  // n4 never stops, n3 always does
        ok = ok && (byQuadrant ? q == 3 : several);
    }
    const int q = c & 3;
    const bool live = (d > 0 && n3quirk && q >= 2) ? q == 3 : cnt > 1;
    return ok ? (QT_C_EXISTS | (cnt > 1 ? QT_C_MULTI : 0) | (live ? QT_C_LIVE : 0)) : 0;
}

// natural cell number of entry t of depth d's block (the order the list keeps a depth's cells in, see above)
QT_FN int qt_jump_cell(int d, int t, int nIni) {
    const int low = (1 << (2 * d)) - 1;
    const int ini = t >> (2 * d);
    return (((d & 1) ? nIni - 1 - ini : ini) << (2 * d)) | ((t & low) ^ (0x33333333 & low));
}

// Returns 0 when the shortcut does not apply (the caller starts with an ordinary coarse pass); otherwise the list
// after pass K is in `cur`, the recorded nodes in w.rec and sh->nL / nRec / seqCounter / fine / finish are as that
// pass leaves them.
template <class Space>
QT_FN int qt_jump(const Space& sp, const QtWork& w, int nIni, int N, int n3quirk, QtShared* sh, OrbNode* cur) {
    const int T = QT_NT, t = QT_TID;
    // depths 0 .. Dj: as far as the passes can go before the list reaches the quota by cell count alone, the
    // bucket depth, and what the scratch (bounds + scanA + scanB + scanC: 7 nodeCap + 3 ints) holds
    int Dj = 0;
    while (((long long)nIni << (2 * Dj)) < N && Dj < QT_JUMP_MAX) Dj++;
    if (Dj > sp.jump_depth()) Dj = sp.jump_depth();
    int* const F = w.bounds;
    const long long room = 7ll * w.nodeCap + 3;
    while (Dj >= 1 && ((((long long)nIni << (2 * (Dj + 1))) - nIni) / 3 + 1) > room) Dj--;  // nIni (4^(Dj+1) - 1) / 3 cells + 1
    if (Dj < 1) return 0;
    // the cells of depth d are a block of nIni 4^d entries, deepest depth first, as the list has them
    const int M = (int)((((long long)nIni << (2 * (Dj + 1))) - nIni) / 3);
    auto locate = [&](int i, int* depth, int* entry) {  // flat index -> (depth, entry of the depth's block)
        int d = Dj, o = 0, sz = nIni << (2 * Dj);
        while (d > 0 && i >= o + sz) {
            o += sz;
            sz >>= 2;
            d--;
        }
        *depth = d;
        *entry = i - o;
    };
    unsigned long long* const agg = (unsigned long long*)sh->bigList;  // per depth: T | E << 21 | D << 42
    static_assert(QT_BIG_LIST * sizeof(int) >= (QT_JUMP_MAX + 1) * sizeof(unsigned long long), "per-depth counters live in bigList");
    for (int i = t; i <= Dj; i += T) agg[i] = 0ull;
    QT_BARRIER();
    for (int base = 0; base < M; base += T) {
        const int i = base + t;
        int d, e;
        locate(i < M ? i : M - 1, &d, &e);
        const bool in = i < M;
        const int cls = in ? qt_cell_class(sp, d, qt_jump_cell(d, e, nIni), n3quirk) : 0;
        if (in) F[i] = cls | (d << 8);
        const unsigned long long add = (cls & QT_C_EXISTS) ? 1ull | ((cls & QT_C_MULTI) ? 1ull << 21 : 0ull) | ((cls & QT_C_LIVE) ? 0ull : 1ull << 42) : 0ull;
#ifdef ORB_HOST_EMU
        agg[d] += add;
#else
        // a wavefront whose cells all lie at one depth (most do) adds its three counts with one atomic
        const int d0 = __builtin_amdgcn_readfirstlane(d);
        if (__all(d == d0)) {
            const unsigned long long mT = __ballot((cls & QT_C_EXISTS) != 0), mE = __ballot((cls & QT_C_EXISTS) && (cls & QT_C_MULTI)),
                                     mD = __ballot((cls & QT_C_EXISTS) && !(cls & QT_C_LIVE));
            if ((t & 63) == 0 && mT)
                atomicAdd(&agg[d0], (unsigned long long)__popcll(mT) | ((unsigned long long)__popcll(mE) << 21) | ((unsigned long long)__popcll(mD) << 42));
        } else if (add) {
            atomicAdd(&agg[d], add);
        }
#endif
    }
    QT_BARRIER();
    QT_STAMP(13);
    // which pass is the last of the run (every thread finds the same answer)
    int K = 1, size = 0, prev = QT_AGG_T(agg[0]), tail = QT_AGG_D(agg[0]), fine = 0, finish = 0, seqNext = nIni, nRec = 0;
    for (int p = 1;; p++) {
        const unsigned long long ap = agg[p];
        size = QT_AGG_T(ap) + tail;
        seqNext += QT_AGG_T(ap);
        nRec = QT_AGG_E(ap);
        finish = size >= N || size == prev;
        fine = !finish && size + 3 * nRec > N;
        K = p;
        if (finish || fine || p == Dj) break;
        tail += QT_AGG_D(ap);
        prev = size;
    }
#ifdef ORB_HOST_EMU
    g_qtJumpStats[finish ? 0 : fine ? 1 : 2]++;
    g_qtJumpStats[3 + K]++;
#endif
    // list positions: front block, then the stopped cells of the shallower depths; recorded nodes counted alongside
    for (int i = t; i < M; i += T) {
        const int v = F[i], d = v >> 8, cls = v & 255;
        int f = 0;
        if (cls & QT_C_EXISTS) {
            if (d == K) f = 1 | ((cls & QT_C_MULTI) ? 1 << 16 : 0);
            else if (d < K && !(cls & QT_C_LIVE)) f = 1;
        }
        F[i] = f;
    }
    QT_BARRIER();
    qt_scan(F, M, sh);
    QT_STAMP(14);
    if (size > w.nodeCap) {  // cannot happen by the size bound; fail loudly
        if (t == 0) { sh->err = 2; sh->finish = 1; sh->nL = 0; }
        QT_BARRIER();
        return 1;
    }
    for (int i = t; i < M; i += T) {
        const int a = F[i], f = F[i + 1] - a;
        if (!(f & 0xffff)) continue;
        int d, e;
        locate(i, &d, &e);
        const int pos = a & 0xffff, c = qt_jump_cell(d, e, nIni);
        const int cnt = sp.count_at(d, c), q = c & 3;
        OrbNode nd;
        nd.lo = sp.cell_lo(d, c);
        nd.hi = nd.lo + cnt;
        const int flags = d == 0 ? (cnt == 1 ? QT_NOMORE : 0) : qt_child_flags(q, cnt, n3quirk);
        nd.depthFlags = d | flags | ((c << (2 * (sp.jump_depth() - d))) << QT_B0_SHIFT);
        nd.seq = d == K ? seqNext - 1 - pos : -1;  // created in the reverse of the list order, numbers ending at seqNext - 1
        cur[pos] = nd;
        if (f >> 16) w.rec[a >> 16] = pos;
    }
    if (t == 0) {
        sh->prevSize = prev;
        sh->nL = size;
        sh->nRec = nRec;
        sh->seqCounter = seqNext;
        sh->fine = fine;
        sh->finish = finish;
    }
    QT_BARRIER();
    QT_STAMP(15);
#if defined(QT_PROFILE) && !defined(ORB_HOST_EMU)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_qtProfile[30] = (unsigned long long)K, g_qtProfile[31] = (unsigned long long)M;
#endif
    return 1;
}

// Runs the distribution over the keys of `sp`.  On return *outCount nodes are in `*outList` in list order;
// sh->deeper != 0 means the histogram form could not follow the reference any further (the list is then void).
template <class Space>
QT_FN void qt_distribute(const Space& sp, const QtWork& w, int nIni, int N, int n3quirk, QtShared* sh, OrbNode** outList,
                         int* outCount) {
    const int T = QT_NT, t = QT_TID;
    OrbNode* cur = w.listA;
    OrbNode* nxt = w.listB;

    // ---- initial nodes, src/ORBextractor.cc:583-616 -------------------------------
    if (t == 0) {
        int nL = 0;
        for (int i = 0; i < nIni; i++) {
            int lo, hi, b0;
            sp.ini_range(i, nIni, &lo, &hi, &b0);
            if (hi == lo) continue;
            OrbNode nd;
            nd.lo = lo;
            nd.hi = hi;
            nd.depthFlags = ((hi - lo == 1) ? QT_NOMORE : 0) | (b0 << QT_B0_SHIFT);
            nd.seq = i;
            cur[nL++] = nd;
        }
        sh->nL = nL;
        sh->seqCounter = nIni;
        sh->finish = 0;
        sh->err = 0;
        sh->deeper = 0;
    }
    QT_BARRIER();

    // the run of coarse passes that neither finish nor hand over to the fine phase, in one step where the key
    // space allows it; the loop below then starts behind the coarse pass
    int jumped = 0;
#ifndef QT_NO_JUMP
    if (sp.jump_depth() > 0) jumped = qt_jump(sp, w, nIni, N, n3quirk, sh, cur);
#endif

    while (!sh->finish) {
      if (!jumped) {
        // ---- coarse pass, src/ORBextractor.cc:625-697 ------------------------------
        QT_COUNT(16);
#if defined(QT_PROFILE) && !defined(ORB_HOST_EMU)
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_qtProfile[16] < 6) g_qtProfile[19 + g_qtProfile[16]] = wall_clock64();
#endif
        const int nL = sh->nL;
        for (int i = t; i < nL; i += T) {
            const OrbNode nd = cur[i];
            int k = 0, e = 0, nm = 1;
            if (!(nd.depthFlags & QT_NOMORE)) {
                int b[3];
                int err = 0, deeper = 0;
                sp.children(nd, b, &err, &deeper);
                if (err) sh->err = 1;
                if (deeper) sh->deeper = 1;
                w.bounds[4 * i + 0] = b[0];
                w.bounds[4 * i + 1] = b[1];
                w.bounds[4 * i + 2] = b[2];
                const int c0 = b[0] - nd.lo, c1 = b[1] - b[0], c2 = b[2] - b[1], c3 = nd.hi - b[2];
                k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                e = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                nm = 0;
            }
            w.scanA[i] = k;
            w.scanB[i] = e;
            w.scanC[i] = nm;
        }
        QT_BARRIER();
        if (sh->deeper) break;  // (same for every thread: written before the barrier)
        qt_scan3(w.scanA, w.scanB, w.scanC, nL, sh);
        const int totalChildren = w.scanA[nL], nToExpand = w.scanB[nL], nStay = w.scanC[nL];
        if (totalChildren + nStay > w.nodeCap) {  // cannot happen by the size bound; fail loudly
            if (t == 0) { sh->err = 2; sh->finish = 1; }
            QT_BARRIER();
            break;
        }
        const int seq0 = sh->seqCounter;
        for (int i = t; i < nL; i += T) {
            const OrbNode nd = cur[i];
            if (nd.depthFlags & QT_NOMORE) {
                nxt[totalChildren + w.scanC[i]] = nd;
            } else {
                const int k = w.scanA[i + 1] - w.scanA[i];
                const int pos = totalChildren - w.scanA[i] - k;  // blocks of later nodes come first
                qt_emit_children(sp, nd, &w.bounds[4 * i], nxt, pos, seq0 + w.scanA[i], w.rec, w.scanB[i], n3quirk);
            }
        }
        QT_BARRIER();
        if (t == 0) {
            const int newL = totalChildren + nStay;
            sh->prevSize = nL;
            sh->nL = newL;
            sh->nRec = nToExpand;
            sh->seqCounter = seq0 + totalChildren;
            sh->fine = 0;
            if (newL >= N || newL == nL) sh->finish = 1;
            else if (newL + nToExpand * 3 > N) sh->fine = 1;
        }
        { OrbNode* tmp = cur; cur = nxt; nxt = tmp; }
        QT_BARRIER();
      }
        jumped = 0;

        // ---- fine phase, src/ORBextractor.cc:702-770 ------------------------------
        while (sh->fine && !sh->finish) {
            QT_COUNT(17);
#if defined(QT_PROFILE) && !defined(ORB_HOST_EMU)
            if (blockIdx.x == 0 && threadIdx.x == 0 && g_qtProfile[17] < 4) {
                g_qtProfile[25 + g_qtProfile[17]] = wall_clock64();
                g_qtProfile[11] = (unsigned long long)sh->nRec;
                g_qtProfile[12] = (unsigned long long)sh->nL;
            }
#endif
            const int nLf = sh->nL, nR = sh->nRec;
            // order recorded nodes by (size, creation number) descending: both are first laid out densely as one
            // 64-bit word per node (in `bounds`, which is free until the children are computed), so that the
            // counting loop reads one word per step and nothing else; words are unique (creation numbers are)
            unsigned long long* const ordKey = (unsigned long long*)w.bounds;  // nodeCap * 4 ints = 2 nodeCap words
            for (int j = t; j < nR; j += T) {
                const OrbNode a = cur[w.rec[j]];
                ordKey[j] = ((unsigned long long)(unsigned)(a.hi - a.lo) << 32) | (unsigned)a.seq;
            }
            QT_BARRIER();
            for (int j = t; j < nR; j += T) {
                const unsigned long long mine = ordKey[j];
                int r = 0;
                int m = 0;
#ifndef ORB_HOST_EMU
                for (; m + 8 <= nR; m += 8) {  // eight independent loads per step
                    unsigned long long o[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) o[u] = ordKey[m + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) r += o[u] > mine;
                }
#endif
                for (; m < nR; m++) r += ordKey[m] > mine;
                w.recSorted[r] = w.rec[j];
            }
            QT_BARRIER();
            for (int j = t; j < nR; j += T) {
                const OrbNode nd = cur[w.recSorted[j]];
                int b[3];
                int err = 0, deeper = 0;
                sp.children(nd, b, &err, &deeper);
                if (err) sh->err = 1;
                if (deeper) sh->deeper = 1;
                w.bounds[4 * j + 0] = b[0];
                w.bounds[4 * j + 1] = b[1];
                w.bounds[4 * j + 2] = b[2];
                const int c0 = b[0] - nd.lo, c1 = b[1] - b[0], c2 = b[2] - b[1], c3 = nd.hi - b[2];
                w.scanA[j] = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                w.scanB[j] = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
            }
            if (t == 0) sh->stop = nR - 1;
            QT_BARRIER();
            if (sh->deeper) break;
            qt_scan3(w.scanA, w.scanB, nullptr, nR, sh);
            // the list reaches the quota after processing node j when nLf + (children - parents) >= N
            for (int j = t; j < nR; j += T) {
                const int after = nLf + w.scanA[j + 1] - (j + 1);
                const int before = nLf + w.scanA[j] - j;
                if (after >= N && before < N) sh->stop = j;  // unique: the running size never decreases
            }
            QT_BARRIER();
            const int nProc = (nR > 0) ? sh->stop + 1 : 0;
            const int childrenProc = w.scanA[nProc];
            // survivors of the old list keep their order behind the new children
            for (int i = t; i < nLf; i += T) w.scanC[i] = 1;
            QT_BARRIER();
            for (int j = t; j < nProc; j += T) w.scanC[w.recSorted[j]] = 0;
            QT_BARRIER();
            qt_scan(w.scanC, nLf, sh);
            const int nSurv = w.scanC[nLf];
            if (childrenProc + nSurv > w.nodeCap) {
                if (t == 0) { sh->err = 2; sh->finish = 1; }
                QT_BARRIER();
                break;
            }
            const int seqF = sh->seqCounter;
            for (int i = t; i < nLf; i += T)
                if (w.scanC[i + 1] - w.scanC[i]) nxt[childrenProc + w.scanC[i]] = cur[i];
            for (int j = t; j < nProc; j += T) {
                const OrbNode nd = cur[w.recSorted[j]];
                const int k = w.scanA[j + 1] - w.scanA[j];
                const int pos = childrenProc - w.scanA[j] - k;
                qt_emit_children(sp, nd, &w.bounds[4 * j], nxt, pos, seqF + w.scanA[j], w.rec, w.scanB[j], n3quirk);
            }
            QT_BARRIER();
            if (t == 0) {
                const int newL = childrenProc + nSurv;
                sh->nRec = w.scanB[nProc];
                sh->seqCounter = seqF + childrenProc;
                sh->nL = newL;
                if (newL >= N || newL == nLf) sh->finish = 1;
            }
            { OrbNode* tmp = cur; cur = nxt; nxt = tmp; }
            QT_BARRIER();
        }
        if (sh->deeper) break;
    }
    *outList = cur;
    *outCount = sh->deeper ? 0 : sh->nL;
}

// In-place ascending sort of a[0..n) by the whole 64-bit value: bitonic network
// in its all-ascending form (first step of every merge mirrors the upper half),
// so indices >= n act as +infinity without being stored.
QT_FN void qt_sort(unsigned long long* a, int n) {
    const int T = QT_NT, t = QT_TID;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    const int pairs = np2 >> 1;
    // (k, hk and jj are powers of two: block and offset by shift and mask)
    for (int k = 2, lk = 1; k <= np2; k <<= 1, lk++) {
        const int hk = k >> 1;
        for (int p = t; p < pairs; p += T) {
            const int blk = p >> (lk - 1), o = p & (hk - 1);
            const int i = blk * k + o, j = blk * k + k - 1 - o;
            if (j < n) {
                const unsigned long long x = a[i], y = a[j];
                if (x > y) { a[i] = y; a[j] = x; }
            }
        }
        QT_BARRIER();
        for (int jj = k >> 2, lj = lk - 2; jj > 0; jj >>= 1, lj--) {
            for (int p = t; p < pairs; p += T) {
                const int blk = p >> lj, o = p & (jj - 1);
                const int i = blk * 2 * jj + o, j = i + jj;
                if (j < n) {
                    const unsigned long long x = a[i], y = a[j];
                    if (x > y) { a[i] = y; a[j] = x; }
                }
            }
            QT_BARRIER();
        }
    }
}

// Winner of a node: rank of the key with the largest priority word (priorities are unique: they hold the rank).
template <class Space>
QT_FN int qt_select(const Space& sp, const OrbNode& nd) {
    const int m = sp.items(nd);
    uint32_t best = sp.prio_at(nd, 0);
    for (int i = 1; i < m; i++) {
        const uint32_t pr = sp.prio_at(nd, i);
        if (pr > best) best = pr;
    }
    return (int)(~best & 0xffffffu);
}

// Sorts the n entries of `tmp` into `sorted`, ascending.
// Entries are first bucketed by the leading bits of the quadtree path (one
// histogram + one scatter over all keys, LDS counters), which already puts
// every bucket into its final slice; buckets are then ordered internally, small
// ones by counting (one key per thread), large ones by the whole workgroup.  Entries are
// unique, so the result does not depend on the order atomics were served in.
QT_FN void qt_sort_keys(const unsigned long long* tmp, unsigned long long* sorted, int n, int nIni, const QtBins bins,
                        QtShared* sh) {
    const int T = QT_NT, t = QT_TID;
    const uint32_t maxCode = ((uint32_t)nIni << (2 * ORB_CODE_DEPTH)) - 1u;  // nIni <= 16
    int bits = 0;
    while ((maxCode >> bits) != 0u && bits < 32) bits++;
    // about four keys per bucket, as many buckets as the scratch holds at most: the passes over the bucket
    // arrays are a fixed cost that a level with few keys should not pay in full
    int useBits = bins.bits < 4 ? bins.bits : 4;
    while ((4 << useBits) < n && useBits < bins.bits) useBits++;
    const int shift = bits > useBits ? bits - useBits : 0;
    const int nb = (int)(maxCode >> shift) + 1;  // <= 2^bins.bits
    int* const binStart = bins.start;
    int* const binCursor = bins.cursor;
    for (int b = t; b <= nb; b += T) binStart[b] = 0;
    if (t == 0) sh->nBig = 0;
    QT_BARRIER();
    for (int r = t; r < n; r += T) QT_ATOMIC_ADD(&binStart[(uint32_t)(tmp[r] >> 32) >> shift], 1);
    QT_BARRIER();
    qt_scan(binStart, nb, sh);
    for (int b = t; b < nb; b += T) binCursor[b] = binStart[b];
    QT_BARRIER();
    for (int r = t; r < n; r += T) {
        const unsigned long long e = tmp[r];
        const int pos = QT_ATOMIC_ADD(&binCursor[(uint32_t)(e >> 32) >> shift], 1);
        sorted[pos] = e;
    }
    QT_BARRIER();
    QT_STAMP(2);
    // Order inside the buckets.  Buckets of up to QT_SMALL_BIN entries: every key counts the entries of its
    // bucket that are smaller (entries are unique, so that count is its place), all keys at once, holds
    // itself in a register across a barrier and moves.  The keys are taken in rounds of QT_RANK_K per
    // thread; a round ends on a bucket boundary, so no key moves onto one that has not been read yet.
    // binCursor is free again: it collects the larger buckets, which the whole workgroup sorts below.
    for (int base = 0; base < n;) {
        int end = base + QT_RANK_K * T;
        if (end >= n) {
            end = n;
        } else {
            const int b = (int)((uint32_t)(sorted[end] >> 32) >> shift);
            end = binStart[b];
            if (end <= base) end = binStart[b + 1];  // one bucket fills the round: a large one, nothing to do here
        }
        unsigned long long held[QT_RANK_K];
        int dst[QT_RANK_K];
#ifndef ORB_HOST_EMU
#pragma unroll
#endif
        for (int k = 0; k < QT_RANK_K; k++) {
            const int r = base + t + k * T;
            dst[k] = -1;
            if (r < end) {
                const unsigned long long e = sorted[r];
                const int b = (int)((uint32_t)(e >> 32) >> shift);
                const int lo = binStart[b], m = binStart[b + 1] - lo;
                if (m > QT_SMALL_BIN) {
                    if (r == lo) binCursor[QT_ATOMIC_ADD(&sh->nBig, 1)] = b;
                } else if (m > 1) {
                    int rank = 0;
                    for (int j = 0; j < m; j++) rank += sorted[lo + j] < e;
                    held[k] = e;
                    dst[k] = lo + rank;
                }
            }
        }
        QT_BARRIER();
#ifndef ORB_HOST_EMU
#pragma unroll
#endif
        for (int k = 0; k < QT_RANK_K; k++)
            if (dst[k] >= 0) sorted[dst[k]] = held[k];
        QT_BARRIER();
        base = end;
    }
    const int nBig = sh->nBig;
    QT_STAMP(3);
    for (int k = 0; k < nBig; k++) {
        QT_COUNT(18);
        const int b = binCursor[k];
        const int lo = binStart[b];
        qt_sort(sorted + lo, binStart[b + 1] - lo);
    }
}

// Winners of all `count` nodes of the final list: outRank[i] = rank of node i's
// winner.  Nodes with few items (keys, or buckets in the histogram form) are handled by one thread each; a
// node that still holds many (with the fork's n3 rule whole quadrants are never divided) is scanned by a whole
// wavefront.
#define QT_SELECT_SERIAL 64
template <class Space>
QT_FN void qt_select_all(const Space& sp, const OrbNode* list, int count, int* outRank, QtShared* sh, int nIni = 0) {
    const int T = QT_NT, t = QT_TID;
    if constexpr (Space::kWinnerTable) {
        if (nIni > 0) {
            sp.winner_table_build(nIni);
            for (int i = t; i < count; i += T) outRank[i] = (int)(~sp.winner_of(list[i], nIni) & 0xffffffu);
            QT_BARRIER();
            return;
        }
    }
    if (t == 0) sh->nBig = 0;
    QT_BARRIER();
    for (int i = t; i < count; i += T) {
        const OrbNode nd = list[i];
        if (sp.items(nd) > QT_SELECT_SERIAL) {
            const int k = QT_ATOMIC_ADD(&sh->nBig, 1);
            if (k < QT_BIG_LIST) {  // (more large nodes than list slots: the rest stay with their thread)
                sh->bigList[k] = i;
                continue;
            }
        }
        outRank[i] = qt_select(sp, nd);
    }
    QT_BARRIER();
    const int nBig = sh->nBig < QT_BIG_LIST ? sh->nBig : QT_BIG_LIST;
#ifdef ORB_HOST_EMU
    for (int k = 0; k < nBig; k++) {
        const int i = sh->bigList[k];
        outRank[i] = qt_select(sp, list[i]);
    }
#else
    // one wavefront per large node, the wavefronts of the workgroup working on different nodes at once
    const int lane = t & 63, wv = t >> 6, nw = (T + 63) >> 6;
    for (int k = wv; k < nBig; k += nw) {
        const int i = sh->bigList[k];
        const OrbNode nd = list[i];
        const int m = sp.items(nd);
        uint32_t best = 0u;
        for (int p = lane; p < m; p += 64) {
            const uint32_t pr = sp.prio_at(nd, p);
            if (pr > best) best = pr;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t v = (uint32_t)__shfl_xor((int)best, o, 64);
            if (v > best) best = v;
        }
        if (lane == 0) outRank[i] = (int)(~best & 0xffffffu);
    }
    QT_BARRIER();
#endif
}
