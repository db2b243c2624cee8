// Scale pyramid (ORBextractor::ComputePyramid, src/ORBextractor.cc:913-949) and
// the stand-alone 7x7 Gaussian blur (cv::GaussianBlur call at :493) for gfx950.
//
// Both are exact-arithmetic streaming kernels: the area resize accumulates
// float products in the order cv::resize(INTER_AREA) does (row taps inside
// column taps, every product and sum rounded on its own — this file is built
// with -ffp-contract=off), the linear resize is 11-bit fixed point, the blur
// is 8.8 fixed point.  Results therefore do not depend on the tiling.
#include <hip/hip_runtime.h>

#include "orb_internal.h"

// Resize, both interpolation modes.  One workgroup produces a 64x16 destination tile: the source
// rectangle its taps can touch (orb_resize_span, a superset checked against the tap tables on the
// host) is fetched into LDS as whole dwords, the tile's tap entries likewise, and every thread then
// computes four horizontally adjacent destination pixels from LDS and stores them as one dword.
//
// area:   dst(dx,dy) = saturate(round_half_even( sum_j beta_j * ( sum_k src(sx_k, sy_j) * alpha_k ) ))
// linear: 11-bit weights, (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2
template <bool AREA>
__global__ __launch_bounds__(256) void k_resize_tiled(const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                                      int64_t sFrame, uint8_t* __restrict__ dst, int dw, int dh,
                                                      int dpitch, int64_t dFrame, const void* __restrict__ xtab,
                                                      const void* __restrict__ ytab, double scaleX, double scaleY,
                                                      int tp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    OrbAreaTap* s_xa = (OrbAreaTap*)s_mem;
    OrbAreaTap* s_ya = s_xa + ORB_RS_TW;
    OrbLinTap* s_xl = (OrbLinTap*)s_mem;
    OrbLinTap* s_yl = s_xl + ORB_RS_TW;
    uint8_t* tile = s_mem + ORB_RS_TAP_BYTES;
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * ORB_RS_TW, y0 = blockIdx.y * ORB_RS_TH;
    const int x1 = min(x0 + ORB_RS_TW, dw), y1 = min(y0 + ORB_RS_TH, dh);
    if (t < ORB_RS_TW) {
        if (x0 + t < dw) {
            if (AREA) s_xa[t] = ((const OrbAreaTap*)xtab)[x0 + t];
            else s_xl[t] = ((const OrbLinTap*)xtab)[x0 + t];
        }
    } else if (t < ORB_RS_TW + ORB_RS_TH) {
        const int r = t - ORB_RS_TW;
        if (y0 + r < dh) {
            if (AREA) s_ya[r] = ((const OrbAreaTap*)ytab)[y0 + r];
            else s_yl[r] = ((const OrbLinTap*)ytab)[y0 + r];
        }
    }
    int sx0, sx1, sy0, sy1;
    orb_resize_span(x0, x1, scaleX, sw, &sx0, &sx1);
    orb_resize_span(y0, y1, scaleY, sh, &sy0, &sy1);
    const int sxa = sx0 & ~3;
    const int nd = (sx1 - sxa + 4) >> 2, nrows = sy1 - sy0 + 1;
    const uint8_t* S = src + (int64_t)blockIdx.z * sFrame;
    const bool aligned = (sstride & 3) == 0 && (((unsigned long long)S) & 3ull) == 0;
    for (int k = t; k < nrows * nd; k += 256) {
        const int r = k / nd, d = k - r * nd;
        const int col = sxa + 4 * d;
        const uint8_t* p = S + (int64_t)(sy0 + r) * sstride + col;
        uint32_t v;
        if (aligned && col + 3 < sw) {
            v = *(const uint32_t*)p;
        } else {
            v = 0u;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (col + i < sw) v |= (uint32_t)p[i] << (8 * i);
        }
        *(uint32_t*)&tile[r * tp + 4 * d] = v;
    }
    __syncthreads();
    const int cg = t & 15, row = t >> 4;
    const int x = x0 + 4 * cg, y = y0 + row;
    if (x >= dw || y >= dh) return;
    const uint8_t* T = tile - (int64_t)sy0 * tp - sxa;  // T[sy * tp + sx] is source pixel (sx, sy)
    uint32_t outw = 0u;
    if (AREA) {
        const OrbAreaTap ty = s_ya[row];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x + i >= dw) break;
            const OrbAreaTap tx = s_xa[4 * cg + i];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < ORB_MAX_TAPS; j++) {
                if (ty.a[j] == 0.f) break;
                const uint8_t* rowp = T + ty.si[j] * tp;
                float buf = 0.f;
#pragma unroll
                for (int k = 0; k < ORB_MAX_TAPS; k++) {
                    if (tx.a[k] == 0.f) break;
                    buf = __fadd_rn(buf, __fmul_rn((float)rowp[tx.si[k]], tx.a[k]));
                }
                const float term = __fmul_rn(ty.a[j], buf);
                sum = (j == 0) ? term : __fadd_rn(sum, term);
            }
            int v = (int)rintf(sum);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            outw |= (uint32_t)v << (8 * i);
        }
    } else {
        const OrbLinTap ty = s_yl[row];
        const uint8_t* S0 = T + ty.s0 * tp;
        const uint8_t* S1 = T + ty.s1 * tp;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x + i >= dw) break;
            const OrbLinTap tx = s_xl[4 * cg + i];
            const int H0 = S0[tx.s0] * tx.a0 + S0[tx.s1] * tx.a1;
            const int H1 = S1[tx.s0] * tx.a0 + S1[tx.s1] * tx.a1;
            const int v = (((ty.a0 * (H0 >> 4)) >> 16) + ((ty.a1 * (H1 >> 4)) >> 16) + 2) >> 2;
            outw |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    uint8_t* D = dst + (int64_t)blockIdx.z * dFrame + (int64_t)y * dpitch + x;
    if (x + 3 < dw && (((unsigned long long)D) & 3ull) == 0) {
        *(uint32_t*)D = outw;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (x + i < dw) D[i] = (uint8_t)(outw >> (8 * i));
    }
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

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

static int launch_resize(bool area, const uint8_t* src, int sw, int sh, int sstride, int64_t sFrame, uint8_t* dst, int dw,
                         int dh, int dpitch, int64_t dFrame, const void* xt, const void* yt, int nframes,
                         hipStream_t st) {
    const double scaleX = orb_resize_scale(sw, dw), scaleY = orb_resize_scale(sh, dh);
    int tp = 0, rows = 0;
    orb_resize_tile_extent(sw, dw, scaleX, ORB_RS_TW, true, &tp);
    orb_resize_tile_extent(sh, dh, scaleY, ORB_RS_TH, false, &rows);
    const size_t lds = ORB_RS_TAP_BYTES + (size_t)tp * rows;
    if (lds > 64 * 1024) {
        orb_set_error("resize %dx%d -> %dx%d: source tile of %zu bytes does not fit the kernel's LDS budget", sw, sh, dw,
                      dh, lds);
        return ORB_ERR_UNSUPPORTED;
    }
    dim3 grid((dw + ORB_RS_TW - 1) / ORB_RS_TW, (dh + ORB_RS_TH - 1) / ORB_RS_TH, nframes);
    if (area)
        hipLaunchKernelGGL(k_resize_tiled<true>, grid, dim3(256), lds, st, src, sw, sh, sstride, sFrame, dst, dw, dh,
                           dpitch, dFrame, xt, yt, scaleX, scaleY, tp);
    else
        hipLaunchKernelGGL(k_resize_tiled<false>, grid, dim3(256), lds, st, src, sw, sh, sstride, sFrame, dst, dw, dh,
                           dpitch, dFrame, xt, yt, scaleX, scaleY, tp);
    return check_launch("resize");
}

int orb_launch_pyramid(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    hipStream_t st = (hipStream_t)b.stream;
    const bool area = (g.quirks & ORB_Q_RESIZE_AREA) != 0;
    for (int l = 1; l < g.nlevels; l++) {
        const OrbLevel& D = g.lv[l];
        const OrbLevel& P = g.lv[l - 1];
        const uint8_t* src = (l == 1) ? b.d_img0 : b.d_pyr + P.off;
        const int sstride = (l == 1) ? b.stride0 : P.pitch;
        const int64_t sFrame = (l == 1) ? b.frameStride0 : P.frameStride;
        const void* xt = area ? (const void*)(b.d_areaTaps + D.tabX) : (const void*)(b.d_linTaps + D.tabX);
        const void* yt = area ? (const void*)(b.d_areaTaps + D.tabY) : (const void*)(b.d_linTaps + D.tabY);
        int rc = launch_resize(area, src, P.w, P.h, sstride, sFrame, b.d_pyr + D.off, D.w, D.h, D.pitch, D.frameStride,
                               xt, yt, b.nframes, st);
        if (rc) return rc;
    }
    return ORB_OK;
}

int orb_launch_resize_single(const uint8_t* d_src, int sw, int sh, int sstride, uint8_t* d_dst, int dw, int dh,
                             int dstride, const OrbAreaTap* d_ax, const OrbAreaTap* d_ay, const OrbLinTap* d_lx,
                             const OrbLinTap* d_ly, int area, void* stream) {
    return launch_resize(area != 0, d_src, sw, sh, sstride, 0, d_dst, dw, dh, dstride, 0,
                         area ? (const void*)d_ax : (const void*)d_lx, area ? (const void*)d_ay : (const void*)d_ly, 1,
                         (hipStream_t)stream);
}

int orb_launch_blur_single(const uint8_t* d_src, int w, int h, int sstride, uint8_t* d_dst, int dstride, int legacy,
                           void* stream) {
    dim3 grid((w + BLUR_TW - 1) / BLUR_TW, (h + BLUR_TH - 1) / BLUR_TH, 1);
    const int k2 = legacy ? 49 : 48, k3 = legacy ? 55 : 56;
    hipLaunchKernelGGL(k_blur7, grid, dim3(256), 0, (hipStream_t)stream, d_src, w, h, sstride, d_dst, dstride, 18, 34,
                       k2, k3);
    return check_launch("blur");
}
