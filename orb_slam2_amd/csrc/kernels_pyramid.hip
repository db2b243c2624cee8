// Scale pyramid (ORBextractor::ComputePyramid, src/ORBextractor.cc:913-949) and
// the stand-alone 7x7 Gaussian blur (cv::GaussianBlur call at :493) for gfx950.
//
// Both are exact-arithmetic streaming kernels: the area resize accumulates
// float products in the order cv::resize(INTER_AREA) does (row taps inside
// column taps, every product and sum rounded on its own — this file is built
// with -ffp-contract=off), the linear resize is 11-bit fixed point, the blur
// is 8.8 fixed point.  Results therefore do not depend on the tiling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "orb_internal.h"

// Resize, both interpolation modes, separably.  A workgroup produces destination tiles 64 pixels wide -- one
// wavefront tiles of 8 rows, or four wavefronts tiles of 32 rows (see k_resize_tiled): the source rectangle
// a tile's taps touch (read off the tap tables, or orb_resize_span's superset of it, which the host checks
// against the tables) is fetched into LDS as whole dwords; a horizontal pass evaluates, for every
// staged source row and every destination column, the row's contribution
//   area:   buf = sum_k src(sx_k, row) * alpha_k        (float, each product and sum rounded)
//   linear: H   = src(s0, row) * a0 + src(s1, row) * a1 (11-bit weights)
// into LDS, and a vertical pass combines the rows of each destination pixel,
//   area:   saturate(round_half_even(sum_j beta_j * buf_j))
//   linear: (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2
// four adjacent pixels per thread, stored as one dword.  This is the order cv::resize itself works in,
// so sharing the horizontal results between destination rows changes nothing.
typedef float float2v __attribute__((ext_vector_type(2)));  // element-wise products and sums: v_pk_mul_f32 / v_pk_add_f32
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Vertically adjacent tiles a four-wavefront workgroup makes in turn (PAIR instances): two, or four in launches over
// many frames (fewer, longer workgroups: -3 % at 512 frames; at 8 and 64 frames four were 15 and 10 % slower than
// two).  The one-wavefront shape takes 2 .. 8 (launch_resize).
#define RS_TILES_PER_WG 2
#define RS_TILES_PER_WG_MANY 4
#define RS_MANY_FRAMES 256
#define RS_STAGE_REGS 6  // dwords of a source tile one thread can hold while the previous tile is worked on
#define RS_WAVE_TH 8     // tile height of the one-wavefront shape

// Everything about one workgroup's column of tiles that does not change from tile to tile.
struct RsTileCtx {
    uint8_t* tile;   // staged source rows, tp bytes each
    float* bufF;     // horizontal results, [rows][ORB_RS_TW]
    int tp, sxa;     // LDS row pitch, first staged source column (multiple of 4)
    int dx, q;       // column of this thread in the horizontal pass, wavefront index
    bool colValid;
};

// Horizontal pass: column dx of every staged row.
template <bool AREA, int NT, int NW>
static __device__ __forceinline__ void rs_horizontal(const RsTileCtx& c, const OrbAreaTap& txa, const OrbLinTap& txl,
                                                     int nrows) {
    if (!c.colValid) return;
    if (AREA) {
        // The taps of a column are consecutive source pixels starting at si[0]: fetch the two aligned dwords
        // that hold them, shift the first tap to byte 0, and always take NT (unused taps have weight 0,
        // and buf + pixel * 0 == buf exactly; what lies under them is some byte of the LDS allocation).
        const int b0 = txa.si[0] - c.sxa;
        const int sh = b0 & 3;
        const int rowStep = c.tp >> 2;
        // two staged rows at a time (the LDS regions hold one row more than any tile stages, so the
        // partner of a last odd row exists); the first product needs no sum: 0 + x == x for x >= +0.
        // The two rows are walked with an LDS address each, kept apart from the compiler's view: as one address plus
        // the row pitch it forms both addresses anew in every step.
        unsigned a0 = (unsigned)(uintptr_t)(lds_u8*)c.tile + 4u * (unsigned)((b0 >> 2) + 2 * c.q * rowStep);
        unsigned a1 = a0 + 4u * (unsigned)rowStep;
        const unsigned aStep = 8u * NW * (unsigned)rowStep;
        float* out = c.bufF + 2 * c.q * ORB_RS_TW + c.dx;
        for (int r = 2 * c.q; r < nrows; r += 2 * NW, a0 += aStep, a1 += aStep, out += 2 * NW * ORB_RS_TW) {
            const lds_u32* p0 = (const lds_u32*)(uintptr_t)a0;
            const lds_u32* p1 = (const lds_u32*)(uintptr_t)a1;
            const uint32_t w0 = __builtin_amdgcn_alignbyte(p0[1], p0[0], sh);
            const uint32_t w1 = __builtin_amdgcn_alignbyte(p1[1], p1[0], sh);
            float2v px = {(float)(w0 & 255u), (float)(w1 & 255u)};
            float2v buf = px * txa.a[0];
            px = float2v{(float)((w0 >> 8) & 255u), (float)((w1 >> 8) & 255u)};
            buf = buf + px * txa.a[1];
            px = float2v{(float)((w0 >> 16) & 255u), (float)((w1 >> 16) & 255u)};
            buf = buf + px * txa.a[2];
            if (NT > 3) {
                px = float2v{(float)(w0 >> 24), (float)(w1 >> 24)};
                buf = buf + px * txa.a[3];
            }
            out[0] = buf.x;
            out[ORB_RS_TW] = buf.y;
            asm volatile("" : "+v"(a0), "+v"(a1));  // (the two addresses stay what the loop carries)
        }
    } else {
        int* bufI = (int*)c.bufF;
        for (int r = c.q; r < nrows; r += NW) {
            const uint8_t* rowp = c.tile + r * c.tp - c.sxa;  // rowp[sx] is source pixel (sx, sy0 + r)
            bufI[r * ORB_RS_TW + c.dx] = rowp[txl.s0] * txl.a0 + rowp[txl.s1] * txl.a1;
        }
    }
}

// Vertical pass: four adjacent columns of one destination row per thread and round; the tile's vertical taps
// are in LDS at `taps`.
// TAPREG (one wavefront): the taps of the lane's rows come in registers (regTaps[k] / regLin[k] for the k-th row the
// lane makes, fetched from the table while the horizontal pass ran, source rows as in the table) instead of from LDS.
template <bool AREA, int NT, int NW, int TH, bool TAPREG>
static __device__ __forceinline__ void rs_vertical(const RsTileCtx& c, const void* taps, const OrbAreaTap* regTaps, const OrbLinTap* regLin, int t, int x0, int y0, int sy0,
                                                   uint8_t* __restrict__ dstFrame, int dw, int dh, int dpitch) {
    const int cg = t & 15;
    const int x = x0 + 4 * cg;
    if (x >= dw) return;
    const bool dwordRows = ((((unsigned long long)dstFrame) | (unsigned long long)(unsigned)dpitch) & 3ull) == 0;  // uniform
    const int* bufI = (const int*)c.bufF;
#pragma unroll
    for (int row = t >> 4; row < TH; row += 4 * NW) {
        const int y = y0 + row;
        if (y >= dh) break;
        uint32_t outw = 0u;
        if (AREA) {
            // all NT taps, no early exit: an unused tap has weight 0 and names the row of the last used one
            // (the staged taps name rows by their byte offset in the tile's horizontal results)
            OrbAreaTap ty;
            if (TAPREG) {
                ty = regTaps[(row - (t >> 4)) / (4 * NW)];  // (a constant index once the loop is unrolled)
#pragma unroll
                for (int j = 0; j < NT; j++) ty.si[j] = (ty.si[j] - sy0) * (ORB_RS_TW * 4);
            } else {
                ty = ((const OrbAreaTap*)taps)[row];
            }
            float2v s01, s23;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float4 b = *(const float4*)((const uint8_t*)c.bufF + ty.si[j] + 16 * cg);
                const float2v t01 = float2v{b.x, b.y} * ty.a[j], t23 = float2v{b.z, b.w} * ty.a[j];
                s01 = (j == 0) ? t01 : s01 + t01;
                s23 = (j == 0) ? t23 : s23 + t23;
            }
            // saturate_cast<uchar>: round half to even, clamp, one instruction per pixel
            outw = __builtin_amdgcn_cvt_pk_u8_f32(s01.x, 0u, outw);
            outw = __builtin_amdgcn_cvt_pk_u8_f32(s01.y, 1u, outw);
            outw = __builtin_amdgcn_cvt_pk_u8_f32(s23.x, 2u, outw);
            outw = __builtin_amdgcn_cvt_pk_u8_f32(s23.y, 3u, outw);
        } else {
            const OrbLinTap ty = TAPREG ? regLin[(row - (t >> 4)) / (4 * NW)] : ((const OrbLinTap*)taps)[row];
            const int4 h0 = *(const int4*)&bufI[(ty.s0 - sy0) * ORB_RS_TW + 4 * cg];
            const int4 h1 = *(const int4*)&bufI[(ty.s1 - sy0) * ORB_RS_TW + 4 * cg];
            const int H0[4] = {h0.x, h0.y, h0.z, h0.w}, H1[4] = {h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int v = (((ty.a0 * (H0[i] >> 4)) >> 16) + ((ty.a1 * (H1[i] >> 4)) >> 16) + 2) >> 2;
                outw |= (uint32_t)(v & 255) << (8 * i);
            }
        }
        // (a frame's bytes lie within 2^31 of its first: one 24-bit multiply-add on top of the uniform frame address;
        // x is a multiple of four, so whether the four pixels sit on a dword is the same for every thread)
        uint8_t* D = dstFrame + (__umul24((unsigned)y, (unsigned)dpitch) + (unsigned)x);
        if (x + 3 < dw && dwordRows) {
            *(uint32_t*)D = outw;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x + i < dw) D[i] = (uint8_t)(outw >> (8 * i));
        }
    }
}

// NT = taps per axis the area tables use at most (3 for any level ratio below 2, else ORB_MAX_TAPS).
// PAIR: the workgroup makes several vertically adjacent tiles and fetches the source rows of the next into
// registers while it works on the current one (a wavefront otherwise spends most of its life waiting for its
// tile); needs a tile to be at most RS_STAGE_REGS dwords per thread, which the launcher checks.
// NW wavefronts make tiles of ORB_RS_TW x TH pixels.  Two shapes are launched: four wavefronts on 64 x 32 tiles,
// and one wavefront on 64 x 8 tiles (EXACT: the staged source span is read off the tap tables -- first tap of the
// tile's first index to last tap of its last -- instead of the wider span orb_resize_span bounds it with; a
// one-wavefront workgroup waits for nobody at its barriers, and its LDS is free again the moment it ends).
template <bool AREA, int NT, bool PAIR, int NW, int TH, bool EXACT>
__global__ __launch_bounds__(64 * NW) void k_resize_tiled(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                                      int64_t sFrame, uint8_t* __restrict__ dst, int dw, int dh,
                                                      int dpitch, int64_t dFrame, const void* __restrict__ xtab,
                                                      const void* __restrict__ ytab, uint64_t scaleX, uint64_t scaleY,
                                                      int tp, int rowsMax, int tilesX, int tilesY, int nframes,
                                                      unsigned tilesXMagic, unsigned tilesMagic, int srcPadded, int tilesPerWg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    constexpr int NTHREADS = 64 * NW;
    constexpr int RPP = 2 * NW;  // rows per staging pass of the uniform-row mapping (32 dword columns per row)
    const int t = threadIdx.x;
    RsTileCtx c;
    c.tile = s_mem + TH * 32;
    c.bufF = (float*)(c.tile + (((size_t)tp * rowsMax + 15) & ~(size_t)15));  // [rowsMax][ORB_RS_TW]
    c.tp = tp;
    c.dx = t & 63;
    c.q = __builtin_amdgcn_readfirstlane(t >> 6);  // wavefront index: uniform
    // tiles of one frame on one XCD: neighbouring tiles share source rows and cache lines
    int frameLocal;
    unsigned tileId;
    if (!orb_xcd_map(blockIdx.x, (unsigned)(tilesX * tilesY), tilesMagic, nframes, &frameLocal, &tileId)) return;
    const int by = (int)orb_div(tileId, (unsigned)tilesX, tilesXMagic), bx = (int)tileId - by * tilesX;
    const int x0 = bx * ORB_RS_TW, x1 = min(x0 + ORB_RS_TW, dw);
    c.colValid = x0 + c.dx < dw;
    OrbAreaTap txa = {};
    OrbLinTap txl = {};
    if (c.colValid) {
        if (AREA) txa = ((const OrbAreaTap*)xtab)[x0 + c.dx];
        else txl = ((const OrbLinTap*)xtab)[x0 + c.dx];
    }
    int sx0, sx1;
    if (EXACT) {  // (uniform addresses: scalar loads)
        if (AREA) {
            sx0 = ((const OrbAreaTap*)xtab)[x0].si[0];
            sx1 = ((const OrbAreaTap*)xtab)[x1 - 1].si[ORB_MAX_TAPS - 1];
        } else {
            sx0 = ((const OrbLinTap*)xtab)[x0].s0;
            sx1 = ((const OrbLinTap*)xtab)[x1 - 1].s1;
        }
    } else {
        orb_resize_span(x0, x1, scaleX, sw, &sx0, &sx1);
    }
    c.sxa = sx0 & ~3;
    const int nd = (sx1 - c.sxa + 4) >> 2;
    const uint8_t* S = src + (int64_t)frameLocal * sFrame;
    uint8_t* D = dst + (int64_t)frameLocal * dFrame;
    const bool aligned = (sstride & 3) == 0 && (((unsigned long long)S) & 3ull) == 0;
    // staging: thread = (row within a pass, dword); a pass covers NTHREADS / nd rows (nd <= NTHREADS: checked by the launcher)
    const int rpp = orb_div_small(NTHREADS, nd);
    const int rr = orb_div_small(t, nd), d = t - rr * nd;
    const int col = c.sxa + 4 * d;
    const bool stager = rr < rpp;
    // A thread's dwords of all rows share their column: whether it can be fetched whole is settled once (on the
    // library's own levels, srcPadded, also a dword that straddles the image width: the row pitch holds it, and the
    // bytes beyond the width are no tap of any pixel), and a row's byte offset inside the frame fits 32 bits
    // (checked by the launcher): one 24-bit multiply.
    const bool dwordOk = aligned && (col + 3 < sw || (srcPadded && col + 4 <= sstride));
    auto fetch = [&](int sy0, int r) -> uint32_t {
        const uint8_t* p = S + (__umul24((unsigned)(sy0 + r), (unsigned)sstride) + (unsigned)col);
        if (dwordOk) return *(const uint32_t*)p;
        uint32_t v = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (col + i < sw) v |= (uint32_t)p[i] << (8 * i);
        return v;
    };
    // the tile's vertical taps; the area taps' source rows as byte offsets into the horizontal results of this tile
    constexpr bool TAPREG = NW == 1 && EXACT;
    constexpr int NREG = TAPREG ? TH / 4 : 1;
    OrbAreaTap regTaps[AREA ? NREG : 1];
    OrbLinTap regLin[AREA ? 1 : NREG];
    auto fetch_taps = [&](int y0) {  // TAPREG: the taps of the rows this lane makes in the tile at y0
        if (TAPREG) {
#pragma unroll
            for (int k = 0; k < NREG; k++) {
                const int y = min(y0 + (t >> 4) + 4 * k, dh - 1);
                if (AREA) regTaps[k] = ((const OrbAreaTap*)ytab)[y];
                else regLin[k] = ((const OrbLinTap*)ytab)[y];
            }
        }
    };
    auto stage_taps = [&](int y0, int sy0) {
        if (TAPREG) return;
        if (t < TH && y0 + t < dh) {
            if (AREA) {
                OrbAreaTap ty = ((const OrbAreaTap*)ytab)[y0 + t];
#pragma unroll
                for (int j = 0; j < ORB_MAX_TAPS; j++) ty.si[j] = (ty.si[j] - sy0) * (ORB_RS_TW * 4);
                ((OrbAreaTap*)s_mem)[t] = ty;
            } else {
                ((OrbLinTap*)s_mem)[t] = ((const OrbLinTap*)ytab)[y0 + t];
            }
        }
    };
    auto rows_of = [&](int y0, int* sy0, int* nrows) {
        int s0, s1;
        const int y1 = min(y0 + TH, dh);
        if (EXACT) {
            if (AREA) {
                s0 = ((const OrbAreaTap*)ytab)[y0].si[0];
                s1 = ((const OrbAreaTap*)ytab)[y1 - 1].si[ORB_MAX_TAPS - 1];
            } else {
                s0 = ((const OrbLinTap*)ytab)[y0].s0;
                s1 = ((const OrbLinTap*)ytab)[y1 - 1].s1;
            }
        } else {
            orb_resize_span(y0, y1, scaleY, sh, &s0, &s1);
        }
        *sy0 = s0;
        *nrows = s1 - s0 + 1;
    };
    // (a one-wavefront workgroup needs no hardware barrier: its LDS operations execute in order)
    auto wg_sync = [&]() {
        if (NW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };

    if (!PAIR) {
        const int y0 = by * TH;
        int sy0, nrows;
        rows_of(y0, &sy0, &nrows);
        stage_taps(y0, sy0);
        if (stager)
            for (int r = rr; r < nrows; r += rpp) *(uint32_t*)&c.tile[r * tp + 4 * d] = fetch(sy0, r);
        wg_sync();
        rs_horizontal<AREA, NT, NW>(c, txa, txl, nrows);
        wg_sync();
        fetch_taps(y0);
        rs_vertical<AREA, NT, NW, TH, TAPREG>(c, s_mem, regTaps, regLin, t, x0, y0, sy0, D, dw, dh, dpitch);
        return;
    }

    // Staging of a tile, PAIR instances.  On 4-byte-aligned sources whose tile rows are at most 32 dwords a thread
    // is (row within a pass of RPP, dword) and a pass starts at row min(RPP k, nrows - RPP): the row numbers are the
    // same for all lanes of a wavefront, so a row's address is scalar arithmetic plus the thread's constant part --
    // one vector instruction per load and per store, no branches; the last passes fetch and store rows the earlier
    // ones already did (same bytes to the same place) instead of being cut short.  Threads whose dword straddles
    // the image width (caller's level 0 only) fetch theirs byte by byte afterwards.  Other sources keep the general
    // mapping (rr, d) with its per-row tests.
    const bool rowsUniform = aligned && nd <= 32;
    const int rl = t >> 5, dd = t & 31;
    const int colU = c.sxa + 4 * dd;
    const bool activeU = dd < nd;
    const bool dwordOkU = colU + 3 < sw || (srcPadded && colU + 4 <= sstride);
    // (the two mappings run as two instances of the tile loop: were they merged after every fetch, the merge would
    // wait for the loads it has to copy, and the point of fetching ahead would be lost)
    auto tile_loop = [&](auto uniformTag) {
        constexpr bool UNI = decltype(uniformTag)::value;
        auto fetch_tile = [&](int fy0, int fn, uint32_t (&held)[RS_STAGE_REGS]) {
            if (UNI) {
                if (activeU && dwordOkU) {
                    const unsigned lo = __umul24((unsigned)min(rl, fn - 1), (unsigned)sstride) + (unsigned)colU;
#pragma unroll
                    for (int k = 0; k < RS_STAGE_REGS; k++) {
                        const unsigned r0 = (unsigned)max(0, min(RPP * k, fn - RPP));
                        held[k] = *(const uint32_t*)(S + (((unsigned)fy0 + r0) * (unsigned)sstride + lo));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < RS_STAGE_REGS; k++) {
                    const int r = rr + k * rpp;
                    held[k] = (stager && r < fn) ? fetch(fy0, r) : 0u;
                }
            }
        };
        auto store_tile = [&](int fy0, int fn, const uint32_t (&held)[RS_STAGE_REGS]) {
            if (UNI) {
                if (!activeU) return;
                const int rlc = min(rl, fn - 1);
                // the thread's part of the LDS address, opaque to the compiler (left to itself it adds the row
                // numbers first and multiplies per lane)
                unsigned ldsLane = (unsigned)(uintptr_t)(lds_u8*)c.tile + (unsigned)(rlc * tp + 4 * dd);
                asm volatile("" : "+v"(ldsLane));
                if (dwordOkU) {
#pragma unroll
                    for (int k = 0; k < RS_STAGE_REGS; k++) {
                        unsigned ro = (unsigned)max(0, min(RPP * k, fn - RPP)) * (unsigned)tp;
                        asm volatile("" : "+s"(ro));
                        *(lds_u32*)(uintptr_t)(ldsLane + ro) = held[k];
                    }
                } else {  // a dword that straddles the image width: fetched now, byte by byte
                    for (int k = 0; k < RS_STAGE_REGS; k++) {
                        const int r0 = max(0, min(RPP * k, fn - RPP));
                        const uint8_t* p = S + (((unsigned)(fy0 + r0 + rlc)) * (unsigned)sstride + (unsigned)colU);
                        uint32_t v = 0u;
                        for (int i = 0; i < 4; i++)
                            if (colU + i < sw) v |= (uint32_t)p[i] << (8 * i);
                        *(lds_u32*)(uintptr_t)(ldsLane + (unsigned)(r0 * tp)) = v;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < RS_STAGE_REGS; k++) {
                    const int r = rr + k * rpp;
                    if (stager && r < fn) *(uint32_t*)&c.tile[r * tp + 4 * d] = held[k];
                }
            }
        };

        int y0 = by * (tilesPerWg * TH);
        int sy0, nrows;
        rows_of(y0, &sy0, &nrows);
        uint32_t held[RS_STAGE_REGS];
        fetch_tile(sy0, nrows, held);
        for (int s = 0; s < tilesPerWg; s++) {
            store_tile(sy0, nrows, held);
            stage_taps(y0, sy0);
            fetch_taps(y0);
            wg_sync();
            // the next tile's rows are requested now and awaited after this tile is done
            const int ny0 = y0 + TH;
            const bool more = s + 1 < tilesPerWg && ny0 < dh;
            int nsy0 = 0, nnrows = 0;
            if (more) {
                rows_of(ny0, &nsy0, &nnrows);
                fetch_tile(nsy0, nnrows, held);
            }
            rs_horizontal<AREA, NT, NW>(c, txa, txl, nrows);
            wg_sync();
            rs_vertical<AREA, NT, NW, TH, TAPREG>(c, s_mem, regTaps, regLin, t, x0, y0, sy0, D, dw, dh, dpitch);
            if (!more) break;
            wg_sync();  // tile, horizontal results and taps are free again
            y0 = ny0;
            sy0 = nsy0;
            nrows = nnrows;
        }
    };
    // (the one-wavefront shape is only launched on aligned sources with tile rows of at most 32 dwords)
    if (EXACT || rowsUniform) tile_loop(std::true_type{});
    else tile_loop(std::false_type{});
}

static __device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// 7x7 separable blur, 8.8 fixed point taps, BORDER_REFLECT_101.  One workgroup
// blurs a 64x16 tile: the (64+6)x(16+6) source window goes to LDS, the
// horizontal pass writes 16-bit sums for 22 rows to LDS, the vertical pass
// rounds once: (sum + 2^15) >> 16.
#define BLUR_TW 64
#define BLUR_TH 16
__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ src, int w, int h, int sstride,
                                               uint8_t* __restrict__ dst, int dstride, int k0, int k1, int k2,
                                               int k3) {
    __shared__ uint8_t s_in[(BLUR_TH + 6) * (BLUR_TW + 8)];
    __shared__ uint16_t s_h[(BLUR_TH + 6) * BLUR_TW];
    const int x0 = blockIdx.x * BLUR_TW, y0 = blockIdx.y * BLUR_TH;
    for (int i = threadIdx.x; i < (BLUR_TH + 6) * (BLUR_TW + 6); i += 256) {
        const int r = i / (BLUR_TW + 6), c = i - r * (BLUR_TW + 6);
        const int sx = reflect101(x0 + c - 3, w), sy = reflect101(y0 + r - 3, h);
        s_in[r * (BLUR_TW + 8) + c] = src[(int64_t)sy * sstride + sx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (BLUR_TH + 6) * BLUR_TW; i += 256) {
        const int r = i / BLUR_TW, c = i - r * BLUR_TW;
        const uint8_t* p = &s_in[r * (BLUR_TW + 8) + c];
        s_h[i] = (uint16_t)(k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BLUR_TH * BLUR_TW; i += 256) {
        const int r = i / BLUR_TW, c = i - r * BLUR_TW;
        const int x = x0 + c, y = y0 + r;
        if (x >= w || y >= h) continue;
        const uint16_t* p = &s_h[r * BLUR_TW + c];
        uint32_t v = (uint32_t)k0 * (p[0] + p[6 * BLUR_TW]) + (uint32_t)k1 * (p[BLUR_TW] + p[5 * BLUR_TW]) +
                     (uint32_t)k2 * (p[2 * BLUR_TW] + p[4 * BLUR_TW]) + (uint32_t)k3 * p[3 * BLUR_TW];
        v = (v + 32768u) >> 16;
        dst[(int64_t)y * dstride + x] = (uint8_t)(v > 255u ? 255u : v);
    }
}

// Frames whose rows do not start on 4-byte boundaries (a 1241-pixel-wide frame stored without padding) would send
// every kernel that stages level 0 down its byte-by-byte path; they are copied once into rows of an aligned
// pitch instead.  Thread = sixteen destination bytes of one row, made from the five aligned source dwords that hold
// them (v_alignbyte_b32); a source dword is only read when one of its bytes is a pixel of the row, so no read
// leaves the pages the frames lie in.
__global__ __launch_bounds__(256) void k_realign_rows(const uint8_t* __restrict__ src, int w, int sstride, int64_t sFrame,
                                                      uint8_t* __restrict__ dst, int dpitch, int64_t dFrame, int rowsPerFrame,
                                                      unsigned perRow, long long total) {
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int row = (int)(id / perRow);  // over all frames
    const int x = 16 * (int)(id - (long long)row * perRow);
    const int frame = row / rowsPerFrame, y = row - frame * rowsPerFrame;
    const uint8_t* rowp = src + frame * sFrame + (int64_t)y * sstride;
    const unsigned long long first = (unsigned long long)(rowp + x), last = (unsigned long long)(rowp + w - 1);
    const uint32_t* a0 = (const uint32_t*)(first & ~3ull);
    const unsigned sh = (unsigned)(first & 3ull);
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = ((unsigned long long)(a0 + k) <= last && (k < 4 || sh)) ? a0[k] : 0u;
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
    o.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
    o.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
    o.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
    *(uint4*)(dst + frame * dFrame + (int64_t)y * dpitch + x) = o;  // (x + 16 <= dpitch: the pitch is a multiple of 64)
}

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

// exactTp / exactRows8: LDS row pitch and source rows of the largest 64 x 8 tile when its span is read off the tap
// tables (OrbLevel::rsTpExact / rsRows8; 0 = not known, as for a stand-alone resize).
static int launch_resize(bool area, int ntaps, const uint8_t* src, int sw, int sh, int sstride, int64_t sFrame, uint8_t* dst, int dw,
                         int dh, int dpitch, int64_t dFrame, const void* xt, const void* yt, int nframes,
                         hipStream_t st, bool srcPadded, int exactTp = 0, int exactRows8 = 0) {
    if ((int64_t)sh * sstride > 0x7fffffffll || sh >= (1 << 24) || sstride >= (1 << 24)) {
        orb_set_error("resize: a source frame of %d rows of %d bytes is beyond the kernel's 32-bit row offsets", sh, sstride);
        return ORB_ERR_UNSUPPORTED;
    }
    if ((int64_t)dh * dpitch > 0x7fffffffll || dh >= (1 << 24) || dpitch >= (1 << 24)) {
        orb_set_error("resize: a destination frame of %d rows of %d bytes is beyond the kernel's 32-bit row offsets", dh, dpitch);
        return ORB_ERR_UNSUPPORTED;
    }
    const uint64_t scaleX = orb_resize_scale_fx(sw, dw), scaleY = orb_resize_scale_fx(sh, dh);
    const int tilesX = (dw + ORB_RS_TW - 1) / ORB_RS_TW;
    // One wavefront per workgroup on 64 x 8 tiles: sources in 4-byte-aligned rows whose tiles are at most 32 dwords
    // wide and 2 * RS_STAGE_REGS rows high (any level ratio up to about 1.3); ORBHIP_RS_WAVE=0 keeps the four-wavefront shape.
    static const int waveMode = getenv("ORBHIP_RS_WAVE") ? atoi(getenv("ORBHIP_RS_WAVE")) : 1;
    static const int waveTiles = getenv("ORBHIP_RS_WAVE_TILES") ? atoi(getenv("ORBHIP_RS_WAVE_TILES")) : 0;
    const int tilesY8 = (dh + RS_WAVE_TH - 1) / RS_WAVE_TH;
    if (waveMode && exactTp > 0 && exactRows8 > 0 && exactTp / 4 <= 32 && exactRows8 <= 2 * RS_STAGE_REGS && tilesY8 > 1 &&
        ((((unsigned long long)src) | (unsigned long long)sstride | (unsigned long long)sFrame) & 3ull) == 0) {
        const int rowsW = exactRows8 + 1;  // the horizontal pass works on pairs of rows
        const size_t ldsW = RS_WAVE_TH * 32 + (((size_t)exactTp * rowsW + 15) & ~(size_t)15) + (size_t)rowsW * ORB_RS_TW * 4;
        // tiles per wavefront: as many as leave the launch about three wavefronts per slot of the chip, 2 .. 8
        const int64_t tilesAll = (int64_t)tilesX * tilesY8 * nframes;
        int tpw = waveTiles > 0 ? waveTiles : (int)std::max<int64_t>(2, std::min<int64_t>(8, tilesAll / (3 * 8192)));
        tpw = std::min(tpw, tilesY8);
        const int tilesYwgW = (tilesY8 + tpw - 1) / tpw;
        const dim3 gridW((unsigned)orb_xcd_grid(nframes, (int64_t)tilesX * tilesYwgW));
#define RS_LAUNCH_WAVE(AREA_, NT_)                                                                                          \
    hipLaunchKernelGGL((k_resize_tiled<AREA_, NT_, true, 1, RS_WAVE_TH, true>), gridW, dim3(64), ldsW, st, src, sw, sh, sstride, \
                       sFrame, dst, dw, dh, dpitch, dFrame, xt, yt, scaleX, scaleY, exactTp, rowsW, tilesX, tilesYwgW, nframes, \
                       orb_div_magic((uint32_t)tilesX, (uint64_t)tilesX * tilesYwgW),                                      \
                       orb_div_magic((uint32_t)(tilesX * tilesYwgW), (uint64_t)gridW.x), srcPadded ? 1 : 0, tpw)
        if (area && ntaps <= 3) RS_LAUNCH_WAVE(true, 3);
        else if (area) RS_LAUNCH_WAVE(true, ORB_MAX_TAPS);
        else RS_LAUNCH_WAVE(false, 2);
#undef RS_LAUNCH_WAVE
        return check_launch("resize");
    }
    int tp = 0, rows = 0;
    orb_resize_tile_extent(sw, dw, scaleX, ORB_RS_TW, true, &tp);
    orb_resize_tile_extent(sh, dh, scaleY, ORB_RS_TH, false, &rows);
    rows += 1;  // the horizontal pass works on pairs of rows
    const size_t lds = ORB_RS_TAP_BYTES + (((size_t)tp * rows + 15) & ~(size_t)15) + (size_t)rows * ORB_RS_TW * 4;
    if (lds > 64 * 1024 || tp / 4 > 256) {  // (the kernel stages one tile row with at most 256 threads)
        orb_set_error("resize %dx%d -> %dx%d: source tile of %zu bytes does not fit the kernel's LDS budget", sw, sh, dw,
                      dh, lds);
        return ORB_ERR_UNSUPPORTED;
    }
    const int tilesY = (dh + ORB_RS_TH - 1) / ORB_RS_TH;
    // two tiles per workgroup when one tile is at most RS_STAGE_REGS dwords per thread (any level ratio up to ~1.5)
    // (tile rows of at most 32 dwords may be staged eight rows per pass, see the kernel)
    const int rowsPerPass = tp / 4 <= 32 ? 8 : 256 / (tp / 4);
    const bool pair = (rows + rowsPerPass - 1) / rowsPerPass <= RS_STAGE_REGS && tilesY > 1;
    const int tilesPerWg = !pair ? 1 : nframes >= RS_MANY_FRAMES ? RS_TILES_PER_WG_MANY : RS_TILES_PER_WG;
    const int tilesYwg = (tilesY + tilesPerWg - 1) / tilesPerWg;
    const dim3 grid((unsigned)orb_xcd_grid(nframes, (int64_t)tilesX * tilesYwg));
#define RS_LAUNCH(AREA_, NT_, PAIR_)                                                                                  \
    hipLaunchKernelGGL((k_resize_tiled<AREA_, NT_, PAIR_, 4, ORB_RS_TH, false>), grid, dim3(256), lds, st, src, sw, sh, sstride, sFrame, dst, \
                       dw, dh, dpitch, dFrame, xt, yt, scaleX, scaleY, tp, rows, tilesX, tilesYwg, nframes,       \
                       orb_div_magic((uint32_t)tilesX, (uint64_t)tilesX * tilesYwg),                               \
                       orb_div_magic((uint32_t)(tilesX * tilesYwg), (uint64_t)grid.x), srcPadded ? 1 : 0, tilesPerWg)
    if (area && ntaps <= 3) {
        if (pair) RS_LAUNCH(true, 3, true); else RS_LAUNCH(true, 3, false);
    } else if (area) {
        if (pair) RS_LAUNCH(true, ORB_MAX_TAPS, true); else RS_LAUNCH(true, ORB_MAX_TAPS, false);
    } else {
        if (pair) RS_LAUNCH(false, 2, true); else RS_LAUNCH(false, 2, false);
    }
#undef RS_LAUNCH
    return check_launch("resize");
}

// Few frames (the reference's one frame per call): seven dependent launches of a few microseconds each cost more in
// launch gaps than in work.  Here one launch builds every level.  The frame is cut into a grid of tiles, the same
// grid at every level; a workgroup reads the part of level 0 its tile depends on into LDS and resizes it level
// by level from one LDS buffer into the other, recomputing the pixels near its borders that its neighbours also
// compute (identical arithmetic, identical values) instead of waiting for them, and stores the pixels it owns.
// The regions come from the tap tables (OrbGeom::chainX / chainY, built on the host).  Per pixel the arithmetic is
// that of k_resize_tiled: same products, same sums, same order.
#define PF_THREADS 1024
#define PF_COLS 128  // threads along x; the eight wavefront pairs of a workgroup take rows in turn
template <bool AREA, int NT>
__global__ __launch_bounds__(PF_THREADS) void k_pyramid_few(const OrbGeom* __restrict__ g, const uint8_t* __restrict__ img0,
                                                            int stride0, int64_t frameStride0, uint8_t* __restrict__ pyr,
                                                            int frame0, const void* __restrict__ taps, int bufBytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const int GX = g->chainGX, tiles = GX * g->chainGY;
    const int frameLocal = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - frameLocal * tiles;
    const int tj = tile / GX, ti = tile - tj * GX;
    const int frame = frame0 + frameLocal;
    const int t = threadIdx.x, lx = t & (PF_COLS - 1);
    const int ly = __builtin_amdgcn_readfirstlane(t / PF_COLS);  // the same in all lanes of a wavefront
    constexpr int LY = PF_THREADS / PF_COLS;
    uint8_t* cur = s_mem;
    uint8_t* nxt = s_mem + bufBytes;

    // level 0: the region's rows as dwords (its first column is a multiple of 4)
    int rx0 = g->chainX[0][ti][0], rx1 = g->chainX[0][ti][1], ry0 = g->chainY[0][tj][0], ry1 = g->chainY[0][tj][1];
    int pitch = ((rx1 - rx0) + 3) & ~3;
    {
        const uint8_t* S = img0 + (int64_t)frame * frameStride0;
        const int w0 = g->lv[0].w;
        const bool aligned = (stride0 & 3) == 0 && (((unsigned long long)S) & 3ull) == 0;
        const int nd = pitch >> 2, col = rx0 + 4 * lx;
        if (lx < nd)
            for (int r = ly; r < ry1 - ry0; r += LY) {
                const uint8_t* p = S + (int64_t)(ry0 + r) * stride0 + col;
                uint32_t v = 0u;
                if (aligned && col + 3 < w0) {
                    v = *(const uint32_t*)p;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (col + i < w0) v |= (uint32_t)p[i] << (8 * i);
                }
                *(uint32_t*)&cur[r * pitch + 4 * lx] = v;
            }
    }
    __syncthreads();

    for (int k = 1; k < g->nlevels; k++) {
        const OrbLevel& L = g->lv[k];
        const int dx0 = g->chainX[k][ti][0], dx1 = g->chainX[k][ti][1], ox0 = g->chainX[k][ti][2], ox1 = g->chainX[k][ti][3];
        const int dy0 = g->chainY[k][tj][0], dy1 = g->chainY[k][tj][1], oy0 = g->chainY[k][tj][2], oy1 = g->chainY[k][tj][3];
        const int dpitch = ((dx1 - dx0) + 3) & ~3;
        uint8_t* D = pyr + L.off + (int64_t)frame * L.frameStride;
        for (int x = dx0 + lx; x < dx1; x += PF_COLS) {
            const bool ownX = x >= ox0 && x < ox1;
            if (AREA) {
                const OrbAreaTap tx = ((const OrbAreaTap*)taps)[L.tabX + x];
                // the taps of a column are consecutive source pixels from si[0] on: the two aligned dwords that
                // hold them, shifted; unused taps have weight 0 (buf + pixel * 0 == buf), whatever lies under them
                const int b0 = tx.si[0] - rx0, sh = b0 & 3;
                const uint32_t* rowp = (const uint32_t*)cur + (b0 >> 2);
                const int rowStep = pitch >> 2;
                for (int y = dy0 + ly; y < dy1; y += LY) {
                    const OrbAreaTap ty = ((const OrbAreaTap*)taps)[L.tabY + y];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; j++) {  // (taps beyond NT have weight 0 in every table of this geometry)
                        const uint32_t* p = rowp + (ty.si[j] - ry0) * rowStep;
                        const uint32_t w = __builtin_amdgcn_alignbyte(p[1], p[0], sh);
                        float buf = (float)(w & 255u) * tx.a[0];
                        buf = buf + (float)((w >> 8) & 255u) * tx.a[1];
                        buf = buf + (float)((w >> 16) & 255u) * tx.a[2];
                        if (NT > 3) buf = buf + (float)(w >> 24) * tx.a[3];
                        s = (j == 0) ? buf * ty.a[0] : s + buf * ty.a[j];
                    }
                    const uint8_t v = (uint8_t)(__builtin_amdgcn_cvt_pk_u8_f32(s, 0u, 0u) & 255u);  // saturate_cast<uchar>
                    nxt[(y - dy0) * dpitch + (x - dx0)] = v;
                    if (ownX && y >= oy0 && y < oy1) D[(int64_t)y * L.pitch + x] = v;
                }
            } else {
                const OrbLinTap tx = ((const OrbLinTap*)taps)[L.tabX + x];
                const uint8_t* c0 = cur + (tx.s0 - rx0);
                const uint8_t* c1 = cur + (tx.s1 - rx0);
                for (int y = dy0 + ly; y < dy1; y += LY) {
                    const OrbLinTap ty = ((const OrbLinTap*)taps)[L.tabY + y];
                    const int r0 = (ty.s0 - ry0) * pitch, r1 = (ty.s1 - ry0) * pitch;
                    const int H0 = c0[r0] * tx.a0 + c1[r0] * tx.a1, H1 = c0[r1] * tx.a0 + c1[r1] * tx.a1;
                    const uint8_t v = (uint8_t)(((((ty.a0 * (H0 >> 4)) >> 16) + ((ty.a1 * (H1 >> 4)) >> 16) + 2) >> 2) & 255);
                    nxt[(y - dy0) * dpitch + (x - dx0)] = v;
                    if (ownX && y >= oy0 && y < oy1) D[(int64_t)y * L.pitch + x] = v;
                }
            }
        }
        __syncthreads();
        uint8_t* sw = cur;
        cur = nxt;
        nxt = sw;
        rx0 = dx0;
        ry0 = dy0;
        pitch = dpitch;
    }
}

static int launch_pyramid_few(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    const bool area = (g.quirks & ORB_Q_RESIZE_AREA) != 0;
    const int bufBytes = g.chainLds / 2;
    const dim3 grid((unsigned)(b.nframes * g.chainGX * g.chainGY));
    if (area && g.areaTapsMax <= 3)
        hipLaunchKernelGGL((k_pyramid_few<true, 3>), grid, dim3(PF_THREADS), (size_t)g.chainLds, (hipStream_t)b.stream, b.d_geom,
                           b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, (const void*)b.d_areaTaps, bufBytes);
    else if (area)
        hipLaunchKernelGGL((k_pyramid_few<true, ORB_MAX_TAPS>), grid, dim3(PF_THREADS), (size_t)g.chainLds, (hipStream_t)b.stream, b.d_geom,
                           b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, (const void*)b.d_areaTaps, bufBytes);
    else
        hipLaunchKernelGGL((k_pyramid_few<false, 2>), grid, dim3(PF_THREADS), (size_t)g.chainLds, (hipStream_t)b.stream, b.d_geom,
                           b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, (const void*)b.d_linTaps, bufBytes);
    return check_launch("pyramid");
}

int orb_launch_realign(const uint8_t* src, int w, int h, int sstride, int64_t sFrame, uint8_t* dst, int dpitch, int n,
                       void* stream) {
    // (dpitch is a multiple of 64 and at least w: the last dword of a row lies inside it)
    const unsigned perRow = (unsigned)((w + 15) / 16);
    const long long total = (long long)n * h * perRow;
    hipLaunchKernelGGL(k_realign_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, w, sstride, sFrame,
                       dst, dpitch, (int64_t)dpitch * h, h, perRow, total);
    return check_launch("row realignment");
}

int orb_launch_pyramid(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    hipStream_t st = (hipStream_t)b.stream;
    const bool area = (g.quirks & ORB_Q_RESIZE_AREA) != 0;
    if (b.nframes < ORB_FEW_FRAMES && g.chainGX > 0 && g.nlevels > 1) return launch_pyramid_few(b);
    for (int l = 1; l < g.nlevels; l++) {
        const OrbLevel& D = g.lv[l];
        const OrbLevel& P = g.lv[l - 1];
        const uint8_t* src = (l == 1) ? b.d_img0 : b.d_pyr + P.off;
        const int sstride = (l == 1) ? b.stride0 : P.pitch;
        const int64_t sFrame = (l == 1) ? b.frameStride0 : P.frameStride;
        const void* xt = area ? (const void*)(b.d_areaTaps + D.tabX) : (const void*)(b.d_linTaps + D.tabX);
        const void* yt = area ? (const void*)(b.d_areaTaps + D.tabY) : (const void*)(b.d_linTaps + D.tabY);
        int rc = launch_resize(area, g.areaTapsMax, src + b.frame0 * sFrame, P.w, P.h, sstride, sFrame,
                               b.d_pyr + D.off + b.frame0 * D.frameStride, D.w, D.h, D.pitch, D.frameStride, xt, yt,
                               b.nframes, st, l > 1, D.rsTpExact, D.rsRows8);  // (levels >= 1 lie in the library's padded rows)
        if (rc) return rc;
    }
    return ORB_OK;
}

int orb_launch_resize_single(const uint8_t* d_src, int sw, int sh, int sstride, uint8_t* d_dst, int dw, int dh,
                             int dstride, const OrbAreaTap* d_ax, const OrbAreaTap* d_ay, const OrbLinTap* d_lx,
                             const OrbLinTap* d_ly, int area, void* stream) {
    return launch_resize(area != 0, ORB_MAX_TAPS, d_src, sw, sh, sstride, 0, d_dst, dw, dh, dstride, 0,
                         area ? (const void*)d_ax : (const void*)d_lx, area ? (const void*)d_ay : (const void*)d_ly, 1,
                         (hipStream_t)stream, false);
}

int orb_launch_blur_single(const uint8_t* d_src, int w, int h, int sstride, uint8_t* d_dst, int dstride, int legacy,
                           void* stream) {
    dim3 grid((w + BLUR_TW - 1) / BLUR_TW, (h + BLUR_TH - 1) / BLUR_TH, 1);
    const int k2 = legacy ? 49 : 48, k3 = legacy ? 55 : 56;
    hipLaunchKernelGGL(k_blur7, grid, dim3(256), 0, (hipStream_t)stream, d_src, w, h, sstride, d_dst, dstride, 18, 34,
                       k2, k3);
    return check_launch("blur");
}
