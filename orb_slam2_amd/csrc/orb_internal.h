// Internal declarations shared by the host side of liborbhip and its kernels.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orb_hip.h"

#define ORB_MAX_LEVELS 12
#define ORB_FEW_FRAMES 8        // launches of fewer frames are bound by launch gaps and latency, not by throughput
#define ORB_CHAIN_G 16          // most tiles per axis of the few-frames pyramid kernel
#define ORB_MAX_TAPS 4          // source cells one destination pixel of the area resize may cover per axis
#define ORB_EDGE 19             // EDGE_THRESHOLD, src/ORBextractor.cc:21
#define ORB_BORDER 16           // EDGE_THRESHOLD - 3, src/ORBextractor.cc:808
#define ORB_FAST_SMALL_W 37      // largest cell interior the FAST kernel's small-tile instance holds
#define ORB_FAST_SMALL_H 35
#define ORB_CELL_MAX 64         // a FAST cell interior is at most 59 pixels wide / high
#define ORB_CODE_DEPTH 14       // quadtree path digits kept per key (2 bits each) + 4 bits initial node
#define ORB_GRID_COLS 64        // include/Frame.h:17
#define ORB_GRID_ROWS 48        // include/Frame.h:16
#define ORB_HISTO 30            // src/ORBmatcher.cc:18
#define ORB_QT_BIG_KEYCAP 400000 // level-0 key capacity from which a frame counts as large for the quadtree kernel (4K: 2 million)

// Geometry of one pyramid level for the current frame size.
struct OrbLevel {
    int w, h, pitch;            // level image, row pitch in the pyramid buffer
    int64_t off;                // byte offset of frame 0 inside the pyramid buffer (levels >= 1)
    int64_t frameStride;        // bytes between consecutive frames of this level
    int nCols, nRows, wCell, hCell;  // FAST cell grid, src/ORBextractor.cc:816-823
    unsigned nColsMagic;             // orb_div_magic(nCols)
    int maxBorderX, maxBorderY;
    int cellBase;               // index of the level's first cell within a frame
    int fastClass, classBase;   // FAST kernel instance the level's cells fit (0: small tile, 1: large), index of
                                // the level's first cell among the frame's cells of that class
    int nIni;                   // quadtree initial nodes, src/ORBextractor.cc:574
    float hX;                   // :576
    int quota;                  // mnFeaturesPerLevel[level]
    int outCap;                 // most keypoints the level can emit
    int outBase;                // sum of outCap of the lower levels
    int64_t keyBase;            // sum of keyCap of the lower levels
    int keyCap;                 // most keys FAST can hand to the quadtree on this level
    int nodeCap;                // node-pool entries per (frame, level)
    int64_t nodeBase;           // sum of nodeCap of the lower levels
    float scale;                // mvScaleFactor[level]
    float patchSize;            // (float)(int)(31 * scale)
    int tabX, tabY;             // first entry of the level's resize taps in the tap table
    int rsTpExact, rsRows8;     // resize into this level in 64 x 8 tiles whose source spans are read off the tap tables: LDS row
                                // pitch (whole dwords from an aligned start) and source rows of the largest tile (0: taps not ordered)
    int codeTabX, codeTabY;     // first entry of the level's quadtree path words per column / per row (border-relative)
    int histDepth;              // bucket depth (path digits) of the quadtree's histogram form, qt_hist_depth
    int histDepthFirst;         // ... of the first pass when a batch runs the quadtree in two passes (one digit less)
};

struct OrbGeom {
    int nlevels, w, h;
    int cellsPerFrame, cellCap;   // cellCap: key slots per cell
    unsigned cellsPerFrameMagic, outCapFrameMagic, cellCapMagic;  // orb_div_magic of cellsPerFrame, outCapFrame, cellCap
    int maxCellW, maxCellH;       // largest FAST cell interior over all levels
    int fastCells[2], fastFirst[2];  // per FAST kernel instance: cells per frame, first level of the class (-1: none)
    unsigned fastCellsMagic[2];      // orb_div_magic of fastCells
    int areaTapsMax;              // most source cells any area-resize tap entry of the pyramid uses (per axis)
    int outCapFrame;              // sum of outCap
    int64_t keysPerFrame;         // sum of keyCap
    int64_t nodesPerFrame;        // sum of nodeCap
    int iniTh, minTh;
    unsigned quirks;
    OrbLevel lv[ORB_MAX_LEVELS];
    // IC_Angle byte weights (src/ORBextractor.cc:23-63) for the describe kernel, which reads the 31 patch rows as
    // aligned dwords of its staged window: for window alignment off (0..3), patch row r (v = r - 15) and dword
    // group qi (bytes u0 .. u0 + 3, u0 = 4 * (qi + 1) - off - 21), word 0 holds 1 in every byte with |u| <= umax[|v|]
    // and word 1 holds u + 32 in those bytes (0 elsewhere).
    uint32_t icMask[4][31][9][2];
    // Few-frames pyramid (one launch builds every level): the frame is cut into chainGX x chainGY tiles at every
    // level alike (tile i of an axis of n pixels owns [i * n / G, (i + 1) * n / G)); chainX[level][i] = {r0, r1, o0,
    // o1}: the columns [r0, r1) of that level the workgroup of tile column i computes (what it owns at [o0, o1) and
    // whatever its part of the next level is made from), chainY alike for rows.  chainGX = 0: not available for
    // this frame size.  chainLds: bytes of LDS the kernel needs.
    int chainGX, chainGY, chainLds;
    short chainX[ORB_MAX_LEVELS][ORB_CHAIN_G][4];
    short chainY[ORB_MAX_LEVELS][ORB_CHAIN_G][4];
};

struct OrbAreaTap {  // one destination index of the area resize: up to 4 source cells and their weights
    int si[ORB_MAX_TAPS];
    float a[ORB_MAX_TAPS];
};
struct OrbLinTap {   // one destination index of the linear resize
    int s0, s1, a0, a1;
};

// Resize kernel tiling: a workgroup makes ORB_RS_TW x ORB_RS_TH destination pixels from a source
// rectangle held in LDS.  orb_resize_span gives, per axis, a source index range that contains every
// tap of destination indices [d0, d1): area taps lie in [floor(d0 * scale), floor(d1 * scale)],
// linear taps in [floor(d0 * scale) - 1, floor(d1 * scale) + 2] once float rounding of the sample
// position is allowed for; the span is at least that wide.  The host checks the tap tables against it
// before the first launch.
#define ORB_RS_TW 64
#define ORB_RS_TH 32
#define ORB_RS_TAP_BYTES (ORB_RS_TH * 32)   // the tile's vertical taps, staged in LDS
#ifdef __HIPCC__
#define ORB_HD __host__ __device__
#else
#define ORB_HD
#endif
// The source-to-destination ratio as cv::resize forms it, floored to 32.32 fixed point: (d * fx) >> 32 is
// floor(d * scale) or one less, (d * (fx + 1)) >> 32 is floor(d * scale) or one more.  Whole-number arithmetic
// on values that are the same for a whole workgroup, which the compiler keeps on the scalar unit (the
// double-precision products it replaces ran on the vector unit in every lane).
static inline uint64_t orb_resize_scale_fx(int ssize, int dsize) {
    const double scale = 1. / ((double)dsize / ssize);
    return (uint64_t)(scale * 4294967296.0);
}
static inline ORB_HD void orb_resize_span(int d0, int d1, uint64_t scaleFx, int ssize, int* s0, int* s1) {
    int lo = (int)(((uint64_t)(unsigned)d0 * scaleFx) >> 32) - 1;
    int hi = (int)(((uint64_t)(unsigned)d1 * (scaleFx + 1)) >> 32) + 2;
    if (lo < 0) lo = 0;
    if (hi > ssize - 1) hi = ssize - 1;
    *s0 = lo;
    *s1 = hi;
}
// Largest LDS extent any tile needs along one axis: bytes per row (dword-aligned start, whole dwords)
// for the horizontal axis, rows for the vertical one.
static inline void orb_resize_tile_extent(int ssize, int dsize, uint64_t scale, int tileSize, bool dwordAligned,
                                          int* extent) {
    int best = 0;
    for (int d0 = 0; d0 < dsize; d0 += tileSize) {
        const int d1 = d0 + tileSize < dsize ? d0 + tileSize : dsize;
        int s0, s1;
        orb_resize_span(d0, d1, scale, ssize, &s0, &s1);
        const int e = dwordAligned ? 4 * ((s1 - (s0 & ~3) + 4) >> 2) : s1 - s0 + 1;
        if (e > best) best = e;
    }
    *extent = best;
}

// Workgroups are dealt round robin over the 8 XCDs (workgroups b and b + 8 share one), each with an L2 of
// its own.  Kernels whose workgroups read overlapping pieces of one frame (cell halos, keypoint windows,
// source rows that straddle tiles) number their work so that a frame stays on one XCD: with X = 8 lanes
// of frames, workgroup b takes item (b / X) % perFrame of frame (b % X) + X * ((b / X) / perFrame).  The
// grid covers X * ceil(nframes / X) frames; workgroups of frames beyond nframes leave at once.  Fewer than
// 8 frames use X = nframes: nothing is launched in vain, and a single frame spreads over all XCDs.
#define ORB_XCDS 8
static inline ORB_HD unsigned orb_xcd_lanes(int nframes) { return nframes < ORB_XCDS ? (unsigned)nframes : ORB_XCDS; }

// Division by a number that is the same for the whole workgroup.  The scalar unit multiplies but does not
// divide, and the general sequence runs on the vector unit in every lane; with magic = ceil(2^32 / d) from the
// host, n / d = high word of n * magic, exact as long as n * d < 2^32.  The host knows the largest dividend
// (nMax) of each use; where the bound does not hold, and for d == 1, it passes magic 0 and the kernel divides
// the ordinary way.
static inline uint32_t orb_div_magic(uint32_t d, uint64_t nMax) {
    if (d <= 1 || nMax * d >= 0x100000000ull) return 0u;
    if (getenv("ORBHIP_PLAIN_DIVISION")) return 0u;  // lets the tests run the ordinary-division path
    return (uint32_t)((0x100000000ull + d - 1) / d);
}
#ifdef __HIPCC__
static __device__ __forceinline__ unsigned orb_div(unsigned n, unsigned d, unsigned magic) {
    return magic ? __umulhi(n, magic) : n / d;
}
// t / d for t < 256 and a divisor 1..256 that is only known on the device: multiplier 65536 / d + 1 from a table,
// exact because t * d < 65536.
struct OrbDiv16Table {
    uint32_t m[257];
    constexpr OrbDiv16Table() : m() {
        for (int d = 1; d <= 256; d++) m[d] = 65536u / (uint32_t)d + 1u;
    }
};
static __constant__ OrbDiv16Table c_orbDiv16 = OrbDiv16Table();
static __device__ __forceinline__ int orb_div_small(int t, int d) { return (int)(__umul24((unsigned)t, c_orbDiv16.m[d]) >> 16); }

static __device__ __forceinline__ bool orb_xcd_map(unsigned b, unsigned perFrame, unsigned perFrameMagic, int nframes,
                                                   int* frameLocal, unsigned* item) {
    const unsigned X = orb_xcd_lanes(nframes);
    const unsigned j = X == ORB_XCDS ? b >> 3 : b / X;  // (fewer than eight frames: a single-frame call, not a batch)
    const unsigned k = orb_div(j, perFrame, perFrameMagic);
    *item = j - k * perFrame;
    *frameLocal = (int)(b - j * X + X * k);
    return *frameLocal < nframes;
}
#endif
static inline int64_t orb_xcd_grid(int nframes, int64_t perFrame) {
    const int64_t X = orb_xcd_lanes(nframes);
    return X * ((nframes + X - 1) / X) * perFrame;
}

// Quadtree node as the kernel keeps it (keys of a node are a contiguous range of the code-sorted key array).
struct OrbNode {
    int lo, hi;      // key range [lo, hi)
    int depthFlags;  // depth | noMore << 8
    int seq;         // creation number
};

// ---- kernel launchers (defined in the .hip files) --------------------------------
struct OrbBatch {  // everything one extraction launch sequence needs
    const OrbGeom* d_geom;
    const OrbGeom* h_geom;
    const uint8_t* d_img0;      // level 0 = the caller's frames
    int stride0;
    int64_t frameStride0;
    int frame0, nframes;        // this launch sequence covers frames [frame0, frame0 + nframes) of the batch
    uint8_t* d_pyr;             // levels >= 1, level-major
    const OrbAreaTap* d_areaTaps;
    const OrbLinTap* d_linTaps;
    const uint32_t* d_codeTab;  // quadtree path words per column and per row of every level (OrbLevel::codeTabX / codeTabY)
    uint32_t* d_cellKeys;       // [frame][cell][cellCap]: x | y << 8 | score << 16, cell-interior coordinates
    int* d_cellCount;           // [frame][cell]
    unsigned long long* d_sort; // [frame][2 * keysPerFrame]: per level keyCap sorted entries, then keyCap unsorted ones
    OrbNode* d_nodes;           // [frame][nodesPerFrame * 2]
    int* d_ints;                // [frame][nodesPerFrame * 9 + nlevels * 4]
    uint32_t* d_selXY;          // [frame][outCapFrame]: x | y << 16 (relative to the 16-pixel border)
    int* d_selScore;            // [frame][outCapFrame]
    int* d_levelCount;          // [frame][nlevels] keypoints kept per level
    int* d_levelKeys;           // [frame][nlevels] keys handed to the quadtree (counted by the FAST kernel)
    int* d_qtRetry;             // quadtree levels handed to the second pass: per frame group count, workgroups done, last count, then workgroup numbers
    int* d_error;               // flag word, OR of limit bits
    int* d_status;              // optional: per frame, the flag word as the last stage saw it (may be host memory)
    orb_keypoint* d_kps;
    uint8_t* d_desc;
    int capPerFrame;
    int* d_nOut;
    void* stream;
};

int orb_launch_pyramid(const OrbBatch& b);
int orb_launch_realign(const uint8_t* src, int w, int h, int sstride, int64_t sFrame, uint8_t* dst, int dpitch, int n,
                       void* stream);
bool orb_quadtree_two_passes(const OrbGeom& g, int nframes);  // whether orb_launch_quadtree runs such a launch in two passes
int orb_launch_fast(const OrbBatch& b);
int orb_launch_quadtree(const OrbBatch& b);
int orb_launch_describe(const OrbBatch& b);

int orb_launch_resize_single(const uint8_t* d_src, int sw, int sh, int sstride, uint8_t* d_dst, int dw, int dh,
                             int dstride, const OrbAreaTap* d_ax, const OrbAreaTap* d_ay, const OrbLinTap* d_lx,
                             const OrbLinTap* d_ly, int area, void* stream);
int orb_launch_blur_single(const uint8_t* d_src, int w, int h, int sstride, uint8_t* d_dst, int dstride, int legacy,
                           void* stream);

int orb_launch_distinctive(const uint8_t* d_desc, const int* d_offsets, int npoints, int selfDistance, int* d_best,
                           int* d_median, void* stream);

void orb_set_error(const char* fmt, ...);

// For host functions that return an ORB_* code: a failing HIP call records its message and returns ORB_ERR_HIP.
#define HIP_TRY(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            orb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ORB_ERR_HIP;                                                                       \
        }                                                                                             \
    } while (0)
