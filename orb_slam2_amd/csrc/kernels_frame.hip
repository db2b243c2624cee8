// The part of the Frame constructor that follows extraction (upstream ORB-SLAM2
// Frame.cc; the fork declares the functions at include/Frame.h:121-125 and lost
// the bodies): UndistortKeyPoints and ComputeImageBounds, both through
// cv::undistortPoints(src, dst, K, distCoef, noArray(), K).
//
// cvUndistortPointsInternal for that call: five fixed iterations of the inverse
// radial-tangential model per point, in double, then re-projection with K and
// one rounding to float.  One thread per keypoint; this file is built with
// -ffp-contract=off, so every product and sum is rounded as on the host.
#include <hip/hip_runtime.h>

#include <cstring>

#include "orb_internal.h"

struct UndistortParams {
    double fx, fy, cx, cy, ifx, ify;
    double k[5];  // k1, k2, p1, p2, k3
};

static __host__ __device__ inline void undistort_one(const UndistortParams& P, float xin, float yin, float* xo, float* yo) {
    double x = xin, y = yin;
    const double u = x, v = y;
    x = (x - P.cx) * P.ifx;
    y = (y - P.cy) * P.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        // the rational terms k[5..7] and the thin-prism terms k[8..11] are zero for a five-coefficient model
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((P.k[4] * r2 + P.k[1]) * r2 + P.k[0]) * r2);
        if (icdist < 0) {
            x = (u - P.cx) * P.ifx;
            y = (v - P.cy) * P.ify;
            break;
        }
        const double deltaX = 2 * P.k[2] * x * y + P.k[3] * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
        const double deltaY = P.k[2] * (r2 + 2 * y * y) + 2 * P.k[3] * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = P.fx * x + 0. * y + P.cx;
    const double yy = 0. * x + P.fy * y + P.cy;
    const double ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(const orb_keypoint* __restrict__ in, orb_keypoint* __restrict__ out,
                                                   int64_t n, UndistortParams P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    orb_keypoint kp = in[i];
    float x, y;
    undistort_one(P, kp.x, kp.y, &x, &y);
    kp.x = x;
    kp.y = y;
    out[i] = kp;
}

// The same over the result block of a frame batch: frame f's first count[f] of `cap` keypoint slots.
__global__ __launch_bounds__(256) void k_undistort_frames(const orb_keypoint* __restrict__ in, orb_keypoint* __restrict__ out,
                                                          const int* __restrict__ count, int cap, UndistortParams P) {
    const int f = blockIdx.y, i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= count[f] || i >= cap) return;
    orb_keypoint kp = in[(int64_t)f * cap + i];
    float x, y;
    undistort_one(P, kp.x, kp.y, &x, &y);
    kp.x = x;
    kp.y = y;
    out[(int64_t)f * cap + i] = kp;
}

static UndistortParams make_params(const float K[4], const float dist[5]) {
    UndistortParams P;
    P.fx = K[0];
    P.fy = K[1];
    P.cx = K[2];
    P.cy = K[3];
    P.ifx = 1. / P.fx;
    P.ify = 1. / P.fy;
    for (int i = 0; i < 5; i++) P.k[i] = dist[i];
    return P;
}

int orb_ctx_device(const orb_ctx* c);
void* orb_ctx_stream(orb_ctx* c);
void orb_ctx_store_bounds(orb_ctx* c, const float* b);
int orb_stage_block(orb_ctx* c, size_t bytes, void** host, void** dev);                       // kernels_match.hip
int orb_stage_copy(orb_ctx* c, size_t off, size_t bytes, int toDevice, void* stream);

static int bad_camera(const float* K, const float* dist, const char* who) {
    if (!K || !dist || !(K[0] != 0.f) || !(K[1] != 0.f)) {
        orb_set_error("%s: camera matrix (fx, fy, cx, cy) and five distortion coefficients are required, fx, fy != 0", who);
        return 1;
    }
    return 0;
}

// For orb_extract_undistorted (orb_api.hip): behind the extraction on its stream (dist[0] != 0: the caller copies
// otherwise, as Frame::UndistortKeyPoints does).
int orb_launch_undistort_frames(const float K[4], const float dist[5], const orb_keypoint* d_in, orb_keypoint* d_out,
                                const int* d_count, int cap, int nframes, void* stream) {
    if (bad_camera(K, dist, "orb_extract_undistorted")) return ORB_ERR_INVALID;
    const UndistortParams P = make_params(K, dist);
    hipLaunchKernelGGL(k_undistort_frames, dim3((unsigned)((cap + 255) / 256), (unsigned)nframes), dim3(256), 0, (hipStream_t)stream,
                       d_in, d_out, d_count, cap, P);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("orb_extract_undistorted: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_undistort_keypoints_device(orb_ctx* c, const float K[4], const float dist[5], const orb_keypoint* d_in,
                                              orb_keypoint* d_out, long long n, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out)) || bad_camera(K, dist, "orb_undistort_keypoints_device")) {
        if (!c || n < 0 || (n > 0 && (!d_in || !d_out))) orb_set_error("orb_undistort_keypoints_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (n == 0) return ORB_OK;
    hipError_t e = hipSetDevice(orb_ctx_device(c));
    if (e == hipSuccess) {
        if (dist[0] == 0.0f) {  // if(mDistCoef.at<float>(0)==0.0) mvKeysUn = mvKeys
            if (d_in != d_out)
                e = hipMemcpyAsync(d_out, d_in, (size_t)n * sizeof(orb_keypoint), hipMemcpyDeviceToDevice, (hipStream_t)stream);
        } else {
            hipLaunchKernelGGL(k_undistort, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out,
                               (int64_t)n, make_params(K, dist));
            e = hipGetLastError();
        }
    }
    if (e != hipSuccess) {
        orb_set_error("orb_undistort_keypoints_device: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_undistort_keypoints(orb_ctx* c, const float K[4], const float dist[5], const orb_keypoint* in,
                                       orb_keypoint* out, int n) {
    if (!c || n < 0 || (n > 0 && (!in || !out))) {
        orb_set_error("orb_undistort_keypoints: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (bad_camera(K, dist, "orb_undistort_keypoints")) return ORB_ERR_INVALID;
    if (n == 0) return ORB_OK;
    if (dist[0] == 0.0f) {
        if (in != out) memcpy(out, in, (size_t)n * sizeof(orb_keypoint));
        return ORB_OK;
    }
    // through the context's page-locked block and its device copy (no allocation per call), moved by kernels
    hipError_t e = hipSetDevice(orb_ctx_device(c));
    if (e != hipSuccess) {
        orb_set_error("orb_undistort_keypoints: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    const size_t bytes = ((size_t)n * sizeof(orb_keypoint) + 255) & ~(size_t)255;
    void *hb = nullptr, *db = nullptr;
    int rc = orb_stage_block(c, 2 * bytes, &hb, &db);
    if (rc) return rc;
    void* const st = orb_ctx_stream(c);
    memcpy(hb, in, (size_t)n * sizeof(orb_keypoint));
    if ((rc = orb_stage_copy(c, 0, bytes, 1, st))) return rc;
    if ((rc = orb_undistort_keypoints_device(c, K, dist, (const orb_keypoint*)db, (orb_keypoint*)((uint8_t*)db + bytes), n, st))) return rc;
    if ((rc = orb_stage_copy(c, bytes, bytes, 0, st))) return rc;
    e = hipStreamSynchronize((hipStream_t)st);
    if (e != hipSuccess) {
        orb_set_error("orb_undistort_keypoints: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    memcpy(out, (const uint8_t*)hb + bytes, (size_t)n * sizeof(orb_keypoint));
    return ORB_OK;
}

// Frame::ComputeImageBounds.  Four points: evaluated on the host with the same routine the kernel runs.
extern "C" int orb_compute_image_bounds(orb_ctx* c, int cols, int rows, const float K[4], const float dist[5],
                                        float bounds[4]) {
    if (!c || cols < 1 || rows < 1 || !bounds) {
        orb_set_error("orb_compute_image_bounds: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (bad_camera(K, dist, "orb_compute_image_bounds")) return ORB_ERR_INVALID;
    if (dist[0] != 0.0f) {
        const UndistortParams P = make_params(K, dist);
        const float src[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        float m[8];
        for (int i = 0; i < 4; i++) undistort_one(P, src[2 * i], src[2 * i + 1], &m[2 * i], &m[2 * i + 1]);
        bounds[0] = m[0] < m[4] ? m[0] : m[4];  // mnMinX = min(mat(0,0), mat(2,0))
        bounds[1] = m[2] > m[6] ? m[2] : m[6];  // mnMaxX = max(mat(1,0), mat(3,0))
        bounds[2] = m[1] < m[3] ? m[1] : m[3];  // mnMinY = min(mat(0,1), mat(1,1))
        bounds[3] = m[5] > m[7] ? m[5] : m[7];  // mnMaxY = max(mat(2,1), mat(3,1))
    } else {
        bounds[0] = 0.0f;
        bounds[1] = (float)cols;
        bounds[2] = 0.0f;
        bounds[3] = (float)rows;
    }
    orb_ctx_store_bounds(c, bounds);
    return ORB_OK;
}

extern "C" int orb_set_image_bounds(orb_ctx* c, const float bounds[4]) {
    if (!c) {
        orb_set_error("null context");
        return ORB_ERR_INVALID;
    }
    if (bounds && !(bounds[1] > bounds[0] && bounds[3] > bounds[2])) {
        orb_set_error("orb_set_image_bounds: need minX < maxX and minY < maxY");
        return ORB_ERR_INVALID;
    }
    orb_ctx_store_bounds(c, bounds);
    return ORB_OK;
}
