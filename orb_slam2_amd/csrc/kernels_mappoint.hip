// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:158-222) for many map
// points at once: among a point's observation descriptors pick the one with the
// least median Hamming distance to all of them (first on ties).
//
// One wavefront per map point.  Lane i owns descriptor i (i + 64, i + 128, ... for
// longer lists) in registers; the list itself sits in LDS and every lane reads
// descriptor j from there (a broadcast read).  The median of row i is the
// smallest v with #{j : d(i,j) <= v} > k, k = (size_t)(0.5 * (N - 1)), found by a
// binary search over v in [0, 256] with the distances recomputed in every step:
// 9 x N x 16 lane-ops per row, nothing stored.  The winner is a wave minimum
// over (median << 16 | i).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstring>

#include "orb_internal.h"

#define MP_WAVES 4
#define MP_LDS_DESC 512  // descriptors of one point kept in LDS; longer lists are read from global memory

static __device__ __forceinline__ int hamming8(const uint32_t a[8], const uint32_t* __restrict__ b) {
    const uint4 b0 = ((const uint4*)b)[0], b1 = ((const uint4*)b)[1];
    return __builtin_popcount(a[0] ^ b0.x) + __builtin_popcount(a[1] ^ b0.y) + __builtin_popcount(a[2] ^ b0.z) +
           __builtin_popcount(a[3] ^ b0.w) + __builtin_popcount(a[4] ^ b1.x) + __builtin_popcount(a[5] ^ b1.y) +
           __builtin_popcount(a[6] ^ b1.z) + __builtin_popcount(a[7] ^ b1.w);
}

__global__ __launch_bounds__(64 * MP_WAVES) void k_distinctive(const uint8_t* __restrict__ desc,
                                                              const int* __restrict__ offsets, int npoints,
                                                              int selfDistance, int* __restrict__ bestIndex,
                                                              int* __restrict__ bestMedian) {
    __shared__ __attribute__((aligned(16))) uint32_t s_desc[MP_WAVES][MP_LDS_DESC * 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform per wavefront
    const int pt = blockIdx.x * MP_WAVES + wave;
    if (pt >= npoints) return;
    const int lo = offsets[pt], N = offsets[pt + 1] - lo;
    if (N <= 0) {  // no usable observation: the reference returns without touching mDescriptor (:181-182)
        if (lane == 0) {
            bestIndex[pt] = -1;
            if (bestMedian) bestMedian[pt] = INT_MAX;
        }
        return;
    }
    if (N > 65535) {  // the winner key holds 16 index bits; reported, never silently wrong
        if (lane == 0) {
            bestIndex[pt] = -2;
            if (bestMedian) bestMedian[pt] = INT_MAX;
        }
        return;
    }
    if (selfDistance) {  // the fork compares every descriptor with itself (:195): all medians 0, the first row wins
        if (lane == 0) {
            bestIndex[pt] = 0;
            if (bestMedian) bestMedian[pt] = 0;
        }
        return;
    }
    const uint32_t* g = (const uint32_t*)(desc + (int64_t)lo * 32);  // 4-byte aligned: checked by the host
    const bool inLds = N <= MP_LDS_DESC;
    if (inLds) {
        for (int k = lane; k < N * 8; k += 64) s_desc[wave][k] = g[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const uint32_t* all = inLds ? s_desc[wave] : g;
    const int kth = (int)(0.5 * (double)(N - 1));
    uint32_t best = 0xFFFFFFFFu;
    for (int i = lane; i < N; i += 64) {
        uint32_t a[8];
#pragma unroll
        for (int w = 0; w < 8; w++) a[w] = all[i * 8 + w];
        int vlo = 0, vhi = 256;  // the median lies in [vlo, vhi]
        while (vlo < vhi) {
            const int mid = (vlo + vhi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) cnt += hamming8(a, all + j * 8) <= mid;
            if (cnt > kth) vhi = mid; else vlo = mid + 1;
        }
        const uint32_t key = ((uint32_t)vlo << 16) | (uint32_t)i;
        best = min(best, key);  // (lists longer than 65535 are rejected by the host: the key holds 16 index bits)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
    if (lane == 0) {
        bestIndex[pt] = (int)(best & 0xFFFFu);
        if (bestMedian) bestMedian[pt] = (int)(best >> 16);
    }
}

int orb_launch_distinctive(const uint8_t* d_desc, const int* d_offsets, int npoints, int selfDistance, int* d_best,
                           int* d_median, void* stream) {
    const int blocks = (npoints + MP_WAVES - 1) / MP_WAVES;
    hipLaunchKernelGGL(k_distinctive, dim3(blocks), dim3(64 * MP_WAVES), 0, (hipStream_t)stream, d_desc, d_offsets, npoints,
                       selfDistance, d_best, d_median);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("distinctive-descriptor launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

int orb_ctx_device(const orb_ctx* c);
void* orb_ctx_stream(orb_ctx* c);
int orb_stage_block(orb_ctx* c, size_t bytes, void** host, void** dev);                       // kernels_match.hip
int orb_stage_copy(orb_ctx* c, size_t off, size_t bytes, int toDevice, void* stream);

extern "C" int orb_distinctive_descriptors_device(orb_ctx* c, const uint8_t* d_desc, const int* d_offsets, int npoints,
                                                  int self_distance, int* d_best_index, int* d_best_median,
                                                  void* stream) {
    if (!c || npoints < 0 || (npoints > 0 && (!d_desc || !d_offsets || !d_best_index))) {
        orb_set_error("orb_distinctive_descriptors_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if ((((unsigned long long)d_desc) & 3ull) != 0) {
        orb_set_error("orb_distinctive_descriptors_device: descriptor array must be 4-byte aligned");
        return ORB_ERR_INVALID;
    }
    if (npoints == 0) return ORB_OK;
    hipError_t e = hipSetDevice(orb_ctx_device(c));
    if (e != hipSuccess) {
        orb_set_error("hipSetDevice failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return orb_launch_distinctive(d_desc, d_offsets, npoints, self_distance, d_best_index, d_best_median, stream);
}

extern "C" int orb_distinctive_descriptors(orb_ctx* c, const uint8_t* desc, const int* offsets, int npoints,
                                           int self_distance, int* best_index, int* best_median) {
    if (!c || npoints < 0 || (npoints > 0 && (!offsets || !best_index))) {
        orb_set_error("orb_distinctive_descriptors: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (npoints == 0) return ORB_OK;
    for (int p = 0; p < npoints; p++) {
        if (offsets[p] < 0 || offsets[p + 1] < offsets[p]) {
            orb_set_error("orb_distinctive_descriptors: offsets must be non-negative and ascending");
            return ORB_ERR_INVALID;
        }
        if (offsets[p + 1] - offsets[p] > 65535) {
            orb_set_error("orb_distinctive_descriptors: map point %d has more than 65535 observations", p);
            return ORB_ERR_UNSUPPORTED;
        }
    }
    const int total = offsets[npoints];
    if (total > 0 && !desc) {
        orb_set_error("orb_distinctive_descriptors: null descriptor array");
        return ORB_ERR_INVALID;
    }
    // through the context's page-locked block and its device copy (no allocation per call), on the context's stream
    hipError_t e = hipSetDevice(orb_ctx_device(c));
    if (e != hipSuccess) {
        orb_set_error("orb_distinctive_descriptors: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oD = 0, oO = oD + up((size_t)total * 32), oB = oO + up((size_t)(npoints + 1) * 4), oM = oB + up((size_t)npoints * 4);
    const size_t end = oM + up((size_t)npoints * 4);
    void *hb = nullptr, *db = nullptr;
    int rc = orb_stage_block(c, end, &hb, &db);
    if (rc) return rc;
    uint8_t* const h = (uint8_t*)hb;
    uint8_t* const d = (uint8_t*)db;
    void* const st = orb_ctx_stream(c);
    if (total > 0) memcpy(h + oD, desc, (size_t)total * 32);
    memcpy(h + oO, offsets, (size_t)(npoints + 1) * 4);
    if ((rc = orb_stage_copy(c, 0, oB, 1, st))) return rc;
    if ((rc = orb_launch_distinctive(d + oD, (const int*)(d + oO), npoints, self_distance, (int*)(d + oB), (int*)(d + oM), st))) return rc;
    if ((rc = orb_stage_copy(c, oB, end - oB, 0, st))) return rc;
    e = hipStreamSynchronize((hipStream_t)st);
    if (e != hipSuccess) {
        orb_set_error("orb_distinctive_descriptors: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    memcpy(best_index, h + oB, (size_t)npoints * 4);
    if (best_median) memcpy(best_median, h + oM, (size_t)npoints * 4);
    return ORB_OK;
}
