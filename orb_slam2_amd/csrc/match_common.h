// Device helpers shared by the windowed matchers and the bag-of-words matcher: wavefront
// reductions through the DPP path, the 256-bit Hamming distance, the rotation-histogram bin and
// ORBmatcher::ComputeThreeMaxima.
#pragma once
#include <hip/hip_runtime.h>

#include "orb_internal.h"

#define M_TH_LOW 50           // src/ORBmatcher.cc:17

static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Minimum over the 64 lanes, returned to all of them: row shifts and row broadcasts through the DPP
// path (no LDS round trips), then a read of lane 63.  Lanes a step does not write keep their value.
template <int CTRL>
static __device__ __forceinline__ uint32_t dpp_min_step(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
    return min(v, o);
}
static __device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_step<0x111>(v);  // row_shr:1
    v = dpp_min_step<0x112>(v);  // row_shr:2
    v = dpp_min_step<0x114>(v);  // row_shr:4
    v = dpp_min_step<0x118>(v);  // row_shr:8   -> lane 15 of every row holds the row's minimum
    v = dpp_min_step<0x142>(v);  // row_bcast:15 -> lane 31 / 63 hold the minimum of rows 0-1 / 2-3
    v = dpp_min_step<0x143>(v);  // row_bcast:31 -> lane 63 holds the minimum of the wavefront
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The two smallest of the wavefront's keys, given each lane's own two smallest (k1 <= k2), returned
// to all lanes.  Keys are unique or 0xFFFFFFFF ("none").  Same DPP ladder; lanes a step does not feed
// merge with "none", and the row broadcasts write only the rows that are meant to receive them.
template <int CTRL, int ROWS>
static __device__ __forceinline__ void dpp_top2_step(uint32_t& k1, uint32_t& k2) {
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k1, CTRL, ROWS, 0xf, false);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, CTRL, ROWS, 0xf, false);
    const uint32_t lo = min(k1, o1), hi = min(max(k1, o1), min(k2, o2));
    k1 = lo;
    k2 = hi;
}
static __device__ __forceinline__ void wave_top2_u32(uint32_t& k1, uint32_t& k2) {
    dpp_top2_step<0x111, 0xf>(k1, k2);
    dpp_top2_step<0x112, 0xf>(k1, k2);
    dpp_top2_step<0x114, 0xf>(k1, k2);
    dpp_top2_step<0x118, 0xf>(k1, k2);
    dpp_top2_step<0x142, 0xa>(k1, k2);  // row_bcast:15 into rows 1 and 3
    dpp_top2_step<0x143, 0xc>(k1, k2);  // row_bcast:31 into rows 2 and 3
    k1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63);
    k2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63);
}

// Orders LDS accesses of one wavefront for the compiler only: the hardware executes a wavefront's LDS
// instructions in issue order, so a store by one lane is seen by later loads of the other lanes.
static __device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static __device__ __forceinline__ int hamming_rows(const uint4 a0, const uint4 a1, const uint8_t* __restrict__ b) {
    const uint4 b0 = ((const uint4*)b)[0], b1 = ((const uint4*)b)[1];
    return __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) + __builtin_popcount(a0.z ^ b0.z) +
           __builtin_popcount(a0.w ^ b0.w) + __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
           __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
}

// Rotation histogram bin of a match, src/ORBmatcher.cc:94-102.  The reference asserts 0 <= bin < HISTO_LENGTH,
// which holds for keypoint angles in [0, 360); for angles a caller made up outside that range the result here
// is the extra bin ORB_HISTO, which the histograms hold as an entry three_maxima never looks at: such a match
// counts for no orientation and is removed by the rotation check instead of indexing outside the histogram.
static __device__ __forceinline__ int rot_bin(float a1, float a2) {
    const float factor = ORB_HISTO / 360.0f;  // src/ORBmatcher.cc:35
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == ORB_HISTO) bin = 0;
    if (bin < 0 || bin > ORB_HISTO) bin = ORB_HISTO;
    return bin;
}

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:413-454.
static __device__ void three_maxima(const int* histo, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < ORB_HISTO; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s; ind3 = i;
        }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        ind2 = -1; ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        ind3 = -1;
    }
}
