// Per-cell FAST-9 detection (the cell loop of
// ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:816-870, with
// cv::FAST(.., nonmaxSuppression=true) inside) for gfx950.
//
// One wavefront owns one 30-pixel-class cell.  The reference runs cv::FAST on
// the cell's ROI with iniThFAST and, if that finds nothing, again with
// minThFAST; the kernel does the same, at most two attempts per cell.  cv::FAST's
// corner score is the largest threshold at which the pixel is still a corner,
// independent of the threshold it was called with, and its non-maximum
// suppression only sees scores inside the ROI's interior, non-corners counting
// as 0.  So, per attempt with threshold th, with t* the score:
//   corner  <=>  t* >= th
//   kept    <=>  t* >= th  and  t* > t*(q) for the 8 neighbours q that are corners
// and the interiors of the cells tile the level without overlap.
//
// Steps per cell: (1) stage the interior + 3-pixel halo in LDS; per attempt
// (2) a cheap necessary test on every interior pixel queues candidates, which
// get their score t* 64 at a time into an LDS score tile with a zero rim and
// are listed when they are corners, (3) local-maximum test over the listed
// corners, ballot + prefix popcount for row-major ordered compaction (the order
// cv::FAST emits).
#include <hip/hip_runtime.h>

#include "orb_internal.h"

// Development aid: with -DFAST_PROFILE every cell adds what it did to a table per level (cells, interior pixels,
// pixels that passed the candidate test, score_chunk calls, corners, keys kept, cells that went to the second
// attempt, candidates of second attempts), read back with orb_debug_fast_profile (tools/fast_chunk_fill.py).
#ifdef FAST_PROFILE
__device__ unsigned long long g_fastProfile[ORB_MAX_LEVELS][8];
extern "C" int orb_debug_fast_profile(unsigned long long* out, int clear) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastProfile), sizeof(unsigned long long) * ORB_MAX_LEVELS * 8) != hipSuccess) return -1;
    if (clear) {
        static unsigned long long z[ORB_MAX_LEVELS * 8];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fastProfile), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define FAST_PROF(x) x
#else
#define FAST_PROF(x)
#endif

#define FAST_WAVES 1  // wavefronts per workgroup; with one, a cell's LDS is free again as soon as its wavefront ends
// LDS geometry per wavefront, a template parameter of the kernel: the pixel tile holds the cell interior
// plus a 3-pixel halo (plus up to 3 bytes of alignment slack per row and one spare row that the
// two-rows-per-step walk may touch), the score tile the interior plus a zero rim.  Two instances: cells up
// to 37 x 35 (every level of the usual frame sizes) and the general one up to 59 x 59.
// A level belongs to the instance its largest cell fits (OrbLevel::fastClass, set by the host); a frame size whose
// coarse levels have tall cells still runs its many fine-level cells on the small instance.
struct TileSmall {
    static constexpr int TILE_PITCH = 48, TILE_ROWS = 42, SCORE_PITCH = 40, SCORE_ROWS = 37, MAX_W = ORB_FAST_SMALL_W, MAX_H = ORB_FAST_SMALL_H,
                         WAVES_PER_SIMD = 8, CLASS = 0;
};
struct TileLarge {
    static constexpr int TILE_PITCH = 72, TILE_ROWS = 66, SCORE_PITCH = 64, SCORE_ROWS = 62, MAX_W = 59, MAX_H = 59, WAVES_PER_SIMD = 4, CLASS = 1;
};
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
#define CORN_CAP 256                                    // corners listed per cell for the suppression pass

static __device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wavefront execute in issue order; this only stops the
    // compiler from moving LDS accesses across the point and drains the counter.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// base + the number of lanes below this one whose bit is set in a wavefront mask (two instructions: the count
// instructions add their third operand).
static __device__ __forceinline__ int lane_rank(unsigned long long m, int base = 0) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

// Three-input minimum / maximum as single instructions.  Written as min(a, min(b, c)) the compiler shares
// the inner two-input results between neighbouring arcs and ends up with twice as many instructions.
static __device__ __forceinline__ int imin3(int a, int b, int c) {
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
static __device__ __forceinline__ int imax3(int a, int b, int c) {
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// a * b + c on 24-bit factors in one instruction (left to itself the compiler forms both signs of the
// difference and selects).
static __device__ __forceinline__ int imad24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Maximum over the 16 arcs of 9 contiguous ring pixels of the arc's smallest entry: 16 -> 6 -> 2 -> 1.
static __device__ __forceinline__ int arc_max_min(const int* e) {
    int lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = imin3(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
    int a9[16];  // minimum over the arc of 9 starting at k
#pragma unroll
    for (int k = 0; k < 16; k++) a9[k] = imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    return max(imax3(imax3(a9[0], a9[1], a9[2]), imax3(a9[3], a9[4], a9[5]), imax3(a9[6], a9[7], a9[8])),
               imax3(imax3(a9[9], a9[10], a9[11]), imax3(a9[12], a9[13], a9[14]), a9[15]));
}

// Minimum over the 16 arcs of the arc's largest entry.
static __device__ __forceinline__ int arc_min_max(const int* e) {
    int hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) hi3[k] = imax3(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
    int b9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b9[k] = imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    return min(imin3(imin3(b9[0], b9[1], b9[2]), imin3(b9[3], b9[4], b9[5]), imin3(b9[6], b9[7], b9[8])),
               imin3(imin3(b9[9], b9[10], b9[11]), imin3(b9[12], b9[13], b9[14]), b9[15]));
}

// t* = max(A, B) - 1 with A = max over the arcs of min(v - ring), B likewise for (ring - v)
// (cv::cornerScore<16>), exact whenever t* >= th and below th otherwise.  A side can only reach th if its
// adjacent compass pair clears th (every arc of 9 holds one), so the side with the larger compass value is
// evaluated -- the differences are formed with that side's sign -- and the other one only for pixels whose
// compass points clear th on both sides (one wavefront-wide branch).
template <int TILE_PITCH>
static __device__ __forceinline__ int fast_score(const uint8_t* c, int th) {
    // c points at the centre inside the LDS tile
    const int v = c[0];
    int r[16];
    r[0] = c[3 * TILE_PITCH + 0];
    r[1] = c[3 * TILE_PITCH + 1];
    r[2] = c[2 * TILE_PITCH + 2];
    r[3] = c[1 * TILE_PITCH + 3];
    r[4] = c[0 * TILE_PITCH + 3];
    r[5] = c[-1 * TILE_PITCH + 3];
    r[6] = c[-2 * TILE_PITCH + 2];
    r[7] = c[-3 * TILE_PITCH + 1];
    r[8] = c[-3 * TILE_PITCH + 0];
    r[9] = c[-3 * TILE_PITCH - 1];
    r[10] = c[-2 * TILE_PITCH - 2];
    r[11] = c[-1 * TILE_PITCH - 3];
    r[12] = c[0 * TILE_PITCH - 3];
    r[13] = c[1 * TILE_PITCH - 3];
    r[14] = c[2 * TILE_PITCH - 2];
    r[15] = c[3 * TILE_PITCH - 1];
    const int m1 = max(min(r[0], r[8]), min(r[4], r[12])), m2 = min(max(r[0], r[8]), max(r[4], r[12]));
    const int darkv = v - m1, brightv = m2 - v;  // the prefilter's two quantities
    const bool bright = brightv > darkv;
    const int sgn = bright ? 1 : -1, base = bright ? -v : v;
    int e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = imad24(r[k], sgn, base);  // ring - v, or v - ring
    int A = arc_max_min(e);
    const bool both = min(darkv, brightv) > th;
    if (__ballot(both) != 0ull) {  // (diagonal edges do this: common enough to keep it short)
        const int A2 = -arc_min_max(e);  // the other side's differences are -e
        if (both) A = max(A, A2);
    }
    return A - 1;
}

// The kernel waits on LDS round trips far more than it computes: the register budget is set so that the
// wavefronts the LDS footprint admits are also resident (8 per SIMD for the small instance).
template <class TG>
__global__ __launch_bounds__(64 * FAST_WAVES)
__attribute__((amdgpu_waves_per_eu(TG::WAVES_PER_SIMD, TG::WAVES_PER_SIMD))) void k_fast_cells(const OrbGeom* __restrict__ g,
                                                               const uint8_t* __restrict__ img0, int stride0,
                                                               int64_t frameStride0, const uint8_t* __restrict__ pyr,
                                                               int frame0, int nframes,
                                                               uint32_t* __restrict__ cellKeys,
                                                               int* __restrict__ cellCount) {
    constexpr int TILE_PITCH = TG::TILE_PITCH, TILE_ROWS = TG::TILE_ROWS, SCORE_PITCH = TG::SCORE_PITCH, SCORE_ROWS = TG::SCORE_ROWS;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[FAST_WAVES][TILE_ROWS * TILE_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t s_score[FAST_WAVES][(SCORE_ROWS * SCORE_PITCH + 15) / 16 * 16];
    __shared__ uint16_t s_cand[FAST_WAVES][128];
    __shared__ uint16_t s_corn[FAST_WAVES][CORN_CAP];
    // the wavefront index is the same in all 64 lanes: telling the compiler so keeps everything derived from
    // it (frame, level, cell geometry, loop bounds) in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // one cell per workgroup, a frame's cells on one XCD (neighbouring cells share halo pixels and cache lines)
    static_assert(FAST_WAVES == 1, "the cell numbering assumes one wavefront per workgroup");
    const int cellsPerFrame = g->cellsPerFrame;
    int frameLocal;
    unsigned cellU;  // the cell's number among the frame's cells of this instance's class
    if (!orb_xcd_map(blockIdx.x, (unsigned)g->fastCells[TG::CLASS], g->fastCellsMagic[TG::CLASS], nframes, &frameLocal, &cellU)) return;
    const int frame = frame0 + frameLocal;
    int level = g->fastFirst[TG::CLASS];
    for (int l = level + 1; l < g->nlevels; l++)
        if (g->lv[l].fastClass == TG::CLASS && (int)cellU >= g->lv[l].classBase) level = l;
    const OrbLevel& L = g->lv[level];
    const int ci = (int)cellU - L.classBase;
    const int c = L.cellBase + ci;
    const int64_t cellGlobal = (int64_t)frame * cellsPerFrame + c;
    const int i = (int)orb_div((unsigned)ci, (unsigned)L.nCols, L.nColsMagic), j = ci - i * L.nCols;

    // Cell ROI and interior, src/ORBextractor.cc:825-844 (note the 3 / 6 asymmetry of the skip tests).
    const int iniY = ORB_BORDER + i * L.hCell, iniX = ORB_BORDER + j * L.wCell;
    int maxY = iniY + L.hCell + 6, maxX = iniX + L.wCell + 6;
    if (maxY > L.maxBorderY) maxY = L.maxBorderY;
    if (maxX > L.maxBorderX) maxX = L.maxBorderX;
    const int X0 = iniX + 3, Y0 = iniY + 3;
    int cw = maxX - 3 - X0, ch = maxY - 3 - Y0;
    if (iniY >= L.maxBorderY - 3 || iniX >= L.maxBorderX - 6) cw = 0;
    if (cw <= 0 || ch <= 0) {
        if (lane == 0) cellCount[cellGlobal] = 0;
        return;
    }

    const uint8_t* img;
    int pitch;
    if (level == 0) {
        img = img0 + (int64_t)frame * frameStride0;
        pitch = stride0;
    } else {
        img = pyr + L.off + (int64_t)frame * L.frameStride;
        pitch = L.pitch;
    }
    uint8_t* tile = s_tile[wave];
    uint8_t* score = s_score[wave];

    // (1) interior + halo -> LDS; zero the score tile.  On 4-byte-aligned rows the tile is fetched as
    // whole dwords (the tile then starts `off` bytes into each LDS row), otherwise byte by byte.
    const int tw = cw + 6, th = ch + 6;
    const int tx0 = X0 - 3;
    const int xa = tx0 & ~3;
    const int nd = ((tx0 - xa) + tw + 3) >> 2;  // dwords per tile row, <= 18
    const bool fastLoad = (pitch & 3) == 0 && (((unsigned long long)img) & 3ull) == 0 &&
                          xa + 4 * nd <= (level == 0 ? L.w : pitch);
    const int off = fastLoad ? tx0 - xa : 0;
    if (fastLoad) {
        // lane = (row within a pass, dword): COLS dword columns (lanes beyond the tile's width idle), 64 / COLS
        // rows per pass.  All of a lane's loads are issued before the first one is awaited: one after the other
        // they would make up most of the wavefront's life.  Straight-line code with wavefront-uniform row
        // numbers: a pass starts at row min(k RP, th - RP), so the last ones fetch and store rows the earlier
        // ones already did (same bytes to the same place) instead of being cut short or skipped; the row's
        // address is scalar arithmetic, a lane adds its constant part (one vector instruction per store, none
        // per load).
        constexpr int COLS = TILE_PITCH / 4 <= 16 ? 16 : 32, RP = 64 / COLS;
        constexpr int MAX_PASSES = (TG::MAX_H + 6 + RP - 1) / RP;
        static_assert(TILE_PITCH / 4 <= COLS && RP <= 7, "a pass holds whole rows; a tile is at least 7 rows high (one pixel and its halo)");
        const int rl = lane / COLS, d = lane & (COLS - 1);
        if (d < nd) {
            const uint8_t* base = img + (int64_t)(Y0 - 3) * pitch + xa;
            const unsigned lo = __umul24((unsigned)rl, (unsigned)pitch) + 4u * (unsigned)d;
            // the lane's part of the LDS address, opaque to the compiler (left to itself it adds the row numbers
            // first and multiplies per lane, in 64 bits)
            unsigned ldsLane = (unsigned)(uintptr_t)(lds_u8*)tile + (unsigned)(rl * TILE_PITCH + 4 * d);
            asm volatile("" : "+v"(ldsLane));
            uint32_t v[MAX_PASSES];
#pragma unroll
            for (int k = 0; k < MAX_PASSES; k++) {
                const unsigned r0 = (unsigned)min(k * RP, th - RP);
                v[k] = *(const uint32_t*)(base + (r0 * (unsigned)pitch + lo));  // (a tile's bytes lie within 2^32 of its first)
            }
#pragma unroll
            for (int k = 0; k < MAX_PASSES; k++) {
                const unsigned r0 = (unsigned)min(k * RP, th - RP);
                unsigned ro = r0 * (unsigned)TILE_PITCH;
                asm volatile("" : "+s"(ro));  // a scalar product and one 32-bit vector add, not a 64-bit multiply-add per lane
                *(lds_u32*)(uintptr_t)(ldsLane + ro) = v[k];
            }
        }
    } else {
        const uint8_t* src = img + (int64_t)(Y0 - 3) * pitch + tx0;
        for (int r = 0; r < th; r++) {
            for (int x = lane; x < tw; x += 64) tile[r * TILE_PITCH + x] = src[(int64_t)r * pitch + x];
        }
    }
    tile += off;  // tile pixel (r, c) is tile[r * TILE_PITCH + c] from here on
    {
        // the whole score tile, sixteen bytes per lane and step (rows 0 .. ch+1 are read)
        constexpr int N16 = (SCORE_ROWS * SCORE_PITCH + 15) / 16;
        uint4* s128 = (uint4*)score;
#pragma unroll
        for (int k = 0; k < (N16 + 63) / 64; k++)
            if (k * 64 + 64 <= N16 || lane < N16 - k * 64) s128[k * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
    }
    wave_lds_sync();

    // (2)+(3) as the reference does it (src/ORBextractor.cc:846-858): first with iniThFAST, and only when
    // that keeps nothing again with minThFAST.  Cells up to 32 wide are walked two rows per step.  A pixel
    // can only reach t* >= th if two adjacent compass points of its ring (any arc of 9 holds such a pair)
    // both differ from the centre by more than th with the same sign; pixels passing that test are queued
    // in a small LDS ring, in row-major order, and scored 64 at a time; the others keep the score 0, which
    // is what cv::FAST's suppression sees for non-corners.  Corners go to a list in row-major order; local
    // maxima are taken over that list (bit k of a lane's mask remembers chunk k), or, in a cell with more
    // corners than the list holds, over all its pixels (bit p remembers step p).
    const int minTh = g->minTh, iniTh = g->iniTh;
    const int two = cw <= 32;
    const int ly = two ? (lane >> 5) : 0, lx = two ? (lane & 31) : lane;
    const int rowsPerStep = two ? 2 : 1;
    const int steps = (ch + rowsPerStep - 1) / rowsPerStep;
    uint16_t* cand = s_cand[wave];
    uint16_t* corn = s_corn[wave];
    const int cellCap = g->cellCap;
    uint32_t* out = cellKeys + cellGlobal * cellCap;
    const bool xin = lx < cw;
    int total = 0;
    FAST_PROF(int profCand = 0; int profChunks = 0; int profCorn = 0; int profSecond = 0; int profCand2 = 0;)
    for (int attempt = 0; attempt < 2; attempt++) {
        const int th = attempt == 0 ? iniTh : minTh;
        int head = 0, tail = 0;  // wave-uniform ring positions, tail - head <= 127
        int nCorn = 0;           // corners found (t* >= th); the first CORN_CAP are listed
        // scores 64 queued candidates (lanes below `count`), marks corners in the score tile and the list
        auto score_chunk = [&](int count) {
            wave_lds_sync();
            bool corner = false;
            unsigned q = 0;
            if (lane < count) {
                q = cand[(head + lane) & 127];
                const int qx = q & 255, qy = q >> 8;
                const int t = fast_score<TILE_PITCH>(&tile[(qy + 3) * TILE_PITCH + qx + 3], th);
                corner = t >= th;
                if (corner) score[(qy + 1) * SCORE_PITCH + qx + 1] = (uint8_t)t;
            }
            const unsigned long long m = __ballot(corner);
            if (corner) {
                const int pos = lane_rank(m, nCorn);
                if (pos < CORN_CAP) corn[pos] = (uint16_t)q;
            }
            nCorn += __popcll(m);
            head += count;
            FAST_PROF(profChunks++; profCorn += __popcll(m); if (attempt) profCand2 += count; else profCand += count;)
        };
        {
            // An adjacent compass pair darker than v - th:  max(min(S,N), min(E,W)) < v - th; one brighter than
            // v + th:  min(max(S,N), max(E,W)) > v + th; either one:  max(v - m1, m2 - v) > th.  Lanes outside the
            // cell (they read tile bytes that belong to no pixel of it) compare against a threshold no difference
            // of two bytes reaches, so that the test is one compare whose result is the wavefront mask.
            const uint8_t* c = &tile[(ly + 3) * TILE_PITCH + lx + 3];
            auto prefilter = [&](unsigned q, int thLane) {  // q = lx | y << 8, the queue entry of this lane's pixel
                const int v = c[0];
                const int pS = c[3 * TILE_PITCH], pE = c[3], pN = c[-3 * TILE_PITCH], pW = c[-3];
                const int m1 = max(min(pS, pN), min(pE, pW)), m2 = min(max(pS, pN), max(pE, pW));
                const bool pass = max(v - m1, m2 - v) > thLane;
                const unsigned long long m = __ballot(pass);
                if (pass) cand[lane_rank(m, tail) & 127] = (uint16_t)q;
                tail += __popcll(m);
                if (tail - head >= 64) score_chunk(64);
            };
            const int thIn = xin ? th : 0x7fff;
            unsigned q = (unsigned)lx | ((unsigned)ly << 8);
            for (int p = 0; p + 1 < steps; p++, q += (unsigned)rowsPerStep << 8, c += rowsPerStep * TILE_PITCH) prefilter(q, thIn);
            prefilter(q, ly + (steps - 1) * rowsPerStep < ch ? thIn : 0x7fff);  // only the last step can reach below the cell
        }
        if (tail > head) score_chunk(tail - head);
        wave_lds_sync();

        unsigned long long keep = 0ull;
        const bool listed = nCorn <= CORN_CAP;
        const int rounds = listed ? (nCorn + 63) >> 6 : steps;
        for (int p = 0; p < rounds; p++) {
            bool lm = false;
            int px = lx, py = p * rowsPerStep + ly;
            bool in = xin && py < ch;
            if (listed) {
                in = p * 64 + lane < nCorn;
                const unsigned q = in ? corn[p * 64 + lane] : 0u;
                px = q & 255;
                py = q >> 8;
            }
            if (in) {
                const uint8_t* sp = &score[(py + 1) * SCORE_PITCH + px + 1];
                const int sc = sp[0];
                if (sc >= th) {  // (a second attempt also sees the first one's scores: same values, all >= iniTh)
                    const int nb = max(max(imax3(sp[-SCORE_PITCH - 1], sp[-SCORE_PITCH], sp[-SCORE_PITCH + 1]),
                                           imax3(sp[SCORE_PITCH - 1], sp[SCORE_PITCH], sp[SCORE_PITCH + 1])),
                                       max((int)sp[-1], (int)sp[1]));
                    lm = sc > nb;
                }
            }
            keep |= (unsigned long long)lm << p;
            total += __popcll(__ballot(lm));
        }
        FAST_PROF(if (total == 0 && attempt == 0 && minTh < iniTh) profSecond = 1;)
        if (total == 0 && attempt == 0 && minTh < iniTh) continue;  // nothing kept: lower threshold, :852-858
        int run = 0;
        for (int p = 0; p < rounds; p++) {
            const bool f = (keep >> p) & 1ull;
            const unsigned long long b = __ballot(f);
            if (f) {
                int px = lx, py = p * rowsPerStep + ly;
                if (listed) {
                    const unsigned q = corn[p * 64 + lane];
                    px = q & 255;
                    py = q >> 8;
                }
                const int pos = lane_rank(b, run);
                const int sc = score[(py + 1) * SCORE_PITCH + px + 1];
                if (pos < cellCap) out[pos] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)sc << 16);
            }
            run += __popcll(b);
        }
        break;
    }
    if (lane == 0) cellCount[cellGlobal] = total;  // total <= cellCap by the 2x2-block bound on strict local maxima
#ifdef FAST_PROFILE
    if (lane == 0) {
        unsigned long long* pr = g_fastProfile[level];
        atomicAdd(&pr[0], 1ull);
        atomicAdd(&pr[1], (unsigned long long)(cw * ch));
        atomicAdd(&pr[2], (unsigned long long)profCand);
        atomicAdd(&pr[3], (unsigned long long)profChunks);
        atomicAdd(&pr[4], (unsigned long long)profCorn);
        atomicAdd(&pr[5], (unsigned long long)total);
        atomicAdd(&pr[6], (unsigned long long)profSecond);
        atomicAdd(&pr[7], (unsigned long long)profCand2);
    }
#endif
}

int orb_launch_fast(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    // (orb_api.hip refuses geometries whose cells exceed 59 x 59 before anything is launched)
    if (g.fastCells[0] > 0)
        hipLaunchKernelGGL(k_fast_cells<TileSmall>, dim3((unsigned)orb_xcd_grid(b.nframes, g.fastCells[0])), dim3(64 * FAST_WAVES), 0,
                           (hipStream_t)b.stream, b.d_geom, b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, b.nframes,
                           b.d_cellKeys, b.d_cellCount);
    if (g.fastCells[1] > 0)
        hipLaunchKernelGGL(k_fast_cells<TileLarge>, dim3((unsigned)orb_xcd_grid(b.nframes, g.fastCells[1])), dim3(64 * FAST_WAVES), 0,
                           (hipStream_t)b.stream, b.d_geom, b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, b.nframes,
                           b.d_cellKeys, b.d_cellCount);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("fast launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}
