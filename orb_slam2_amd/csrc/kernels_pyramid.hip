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

// dst(dx,dy) = saturate(round_half_even( sum_j beta_j * ( sum_k src(sx_k, sy_j) * alpha_k ) ))
__global__ __launch_bounds__(256) void k_resize_area(const uint8_t* __restrict__ src, int sstride, int64_t sFrame,
                                                     uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                     int64_t dFrame, const OrbAreaTap* __restrict__ xt,
                                                     const OrbAreaTap* __restrict__ yt) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const uint8_t* S = src + (int64_t)blockIdx.z * sFrame;
    const OrbAreaTap tx = xt[dx];
    const OrbAreaTap ty = yt[dy];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < ORB_MAX_TAPS; j++) {
        if (ty.a[j] == 0.f) break;
        const uint8_t* row = S + (int64_t)ty.si[j] * sstride;
        float buf = 0.f;
#pragma unroll
        for (int k = 0; k < ORB_MAX_TAPS; k++) {
            if (tx.a[k] == 0.f) break;
            buf = __fadd_rn(buf, __fmul_rn((float)row[tx.si[k]], tx.a[k]));
        }
        const float term = __fmul_rn(ty.a[j], buf);
        sum = (j == 0) ? term : __fadd_rn(sum, term);
    }
    int v = (int)rintf(sum);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(int64_t)blockIdx.z * dFrame + (int64_t)dy * dpitch + dx] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* __restrict__ src, int sstride, int64_t sFrame,
                                                       uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                       int64_t dFrame, const OrbLinTap* __restrict__ xt,
                                                       const OrbLinTap* __restrict__ yt) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const uint8_t* S = src + (int64_t)blockIdx.z * sFrame;
    const OrbLinTap tx = xt[dx];
    const OrbLinTap ty = yt[dy];
    const uint8_t* S0 = S + (int64_t)ty.s0 * sstride;
    const uint8_t* S1 = S + (int64_t)ty.s1 * sstride;
    const int H0 = S0[tx.s0] * tx.a0 + S0[tx.s1] * tx.a1;
    const int H1 = S1[tx.s0] * tx.a0 + S1[tx.s1] * tx.a1;
    const int v = (((ty.a0 * (H0 >> 4)) >> 16) + ((ty.a1 * (H1 >> 4)) >> 16) + 2) >> 2;
    dst[(int64_t)blockIdx.z * dFrame + (int64_t)dy * dpitch + dx] = (uint8_t)v;
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

int orb_launch_pyramid(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    hipStream_t st = (hipStream_t)b.stream;
    for (int l = 1; l < g.nlevels; l++) {
        const OrbLevel& D = g.lv[l];
        const OrbLevel& P = g.lv[l - 1];
        const uint8_t* src = (l == 1) ? b.d_img0 : b.d_pyr + P.off;
        const int sstride = (l == 1) ? b.stride0 : P.pitch;
        const int64_t sFrame = (l == 1) ? b.frameStride0 : P.frameStride;
        dim3 grid((D.w + 63) / 64, (D.h + 3) / 4, b.nframes);
        if (g.quirks & ORB_Q_RESIZE_AREA)
            hipLaunchKernelGGL(k_resize_area, grid, dim3(256), 0, st, src, sstride, sFrame, b.d_pyr + D.off, D.w, D.h,
                               D.pitch, D.frameStride, b.d_areaTaps + D.tabX, b.d_areaTaps + D.tabY);
        else
            hipLaunchKernelGGL(k_resize_linear, grid, dim3(256), 0, st, src, sstride, sFrame, b.d_pyr + D.off, D.w,
                               D.h, D.pitch, D.frameStride, b.d_linTaps + D.tabX, b.d_linTaps + D.tabY);
        int rc = check_launch("resize");
        if (rc) return rc;
    }
    return ORB_OK;
}

int orb_launch_resize_single(const uint8_t* d_src, int sw, int sh, int sstride, uint8_t* d_dst, int dw, int dh,
                             int dstride, const OrbAreaTap* d_ax, const OrbAreaTap* d_ay, const OrbLinTap* d_lx,
                             const OrbLinTap* d_ly, int area, void* stream) {
    (void)sw;
    (void)sh;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, 1);
    if (area)
        hipLaunchKernelGGL(k_resize_area, grid, dim3(256), 0, (hipStream_t)stream, d_src, sstride, (int64_t)0, d_dst,
                           dw, dh, dstride, (int64_t)0, d_ax, d_ay);
    else
        hipLaunchKernelGGL(k_resize_linear, grid, dim3(256), 0, (hipStream_t)stream, d_src, sstride, (int64_t)0, d_dst,
                           dw, dh, dstride, (int64_t)0, d_lx, d_ly);
    return check_launch("resize");
}

int orb_launch_blur_single(const uint8_t* d_src, int w, int h, int sstride, uint8_t* d_dst, int dstride, int legacy,
                           void* stream) {
    dim3 grid((w + BLUR_TW - 1) / BLUR_TW, (h + BLUR_TH - 1) / BLUR_TH, 1);
    const int k2 = legacy ? 49 : 48, k3 = legacy ? 55 : 56;
    hipLaunchKernelGGL(k_blur7, grid, dim3(256), 0, (hipStream_t)stream, d_src, w, h, sstride, d_dst, dstride, 18, 34,
                       k2, k3);
    return check_launch("blur");
}
