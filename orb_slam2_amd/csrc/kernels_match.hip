// Windowed Hamming matching for gfx950: ORBmatcher::SearchForInitialization
// (src/ORBmatcher.cc:26-136) and the search loop of
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (:302-405),
// both on top of Frame::GetFeaturesInArea (src/Frame.cc:86-144) and the 64x48
// bucket grid of include/Frame.h:16-17,96.
//
// One wavefront per frame pair.  The second frame's keypoints are sorted by
// (grid column * 48 + grid row, keypoint index): GetFeaturesInArea visits
// columns outermost, rows inside, indices inside a bucket in increasing order,
// which is exactly ascending position in that sorted array.  A window query
// therefore is, per grid column, one contiguous slice of the array, and "first
// candidate wins on equal distance" is "smallest array position wins".  Queries
// are resolved in index order because each accepted match changes what later
// queries may take (vMatchedDistance / map-point occupancy); inside a query the
// 64 lanes scan the candidates and reduce (distance, position) with shuffles.
#include <hip/hip_runtime.h>

#include <climits>
#include <vector>

#include "orb_internal.h"

#define M_MAXKP 8192          // keypoints per frame the LDS state is sized for
#define M_CELLS (ORB_GRID_COLS * ORB_GRID_ROWS)
#define M_TH_LOW 50           // src/ORBmatcher.cc:17

#define HIP_TRY(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            orb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
            return ORB_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

struct MatchLds {
    uint32_t sorted[M_MAXKP];      // bucket << 16 | keypoint index, ascending (index < 65536)
    float2 xy[M_MAXKP];            // keypoint position, by sorted position
    uint8_t oct[M_MAXKP];          // keypoint octave, by sorted position
    uint16_t cellStart[M_CELLS + 2];
    uint16_t matchedDist[M_MAXKP]; // vMatchedDistance (0xFFFF = INT_MAX) / occupancy flag for the projected search
    int16_t owner[M_MAXKP];        // vnMatches21 / query that owns the keypoint
    uint16_t cand[M_MAXKP];        // sorted positions that passed the window test of the current query
    int hist[ORB_HISTO];
};

static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static __device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

struct Window {
    int x0, x1, y0, y1;  // bucket ranges [x0,x1) x [y0,y1); empty when x1 <= x0
};

// Sort the frame's keypoints into bucket order and build bucket start offsets.
// Bucket assignment as the Frame constructor does it (PosInGrid): rounded
// position, keypoints that fall outside the grid are left out.
static __device__ void grid_build(MatchLds& S, const orb_keypoint* __restrict__ k2, int n2, float wInv, float hInv,
                                  int lane, int* nIn) {
    int np2 = 1;
    while (np2 < n2) np2 <<= 1;
    for (int i = lane; i < np2; i += 64) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < n2) {
            const int px = (int)roundf(__fmul_rn(k2[i].x, wInv));  // mnMinX = mnMinY = 0: no lens distortion
            const int py = (int)roundf(__fmul_rn(k2[i].y, hInv));
            if (px >= 0 && px < ORB_GRID_COLS && py >= 0 && py < ORB_GRID_ROWS)
                key = ((uint32_t)(px * ORB_GRID_ROWS + py) << 16) | (uint32_t)i;
        }
        S.sorted[i] = key;
    }
    wave_lds_sync();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < np2; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = S.sorted[i], b = S.sorted[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        S.sorted[i] = b;
                        S.sorted[l] = a;
                    }
                }
            }
            wave_lds_sync();
        }
    }
    // bucket c starts at the first entry whose bucket is >= c
    for (int c = lane; c <= M_CELLS; c += 64) {
        int lo = 0, hi = n2;
        const uint32_t key = (uint32_t)c << 16;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (S.sorted[mid] < key) lo = mid + 1; else hi = mid;
        }
        S.cellStart[c] = (uint16_t)lo;
    }
    wave_lds_sync();
    *nIn = S.cellStart[M_CELLS];
    for (int p = lane; p < *nIn; p += 64) {
        const orb_keypoint kp = k2[S.sorted[p] & 0xFFFFu];
        S.xy[p] = make_float2(kp.x, kp.y);
        S.oct[p] = (uint8_t)(kp.octave < 0 ? 255 : (kp.octave > 254 ? 254 : kp.octave));
    }
    wave_lds_sync();
}

// Window query: collects into S.cand the sorted positions of the keypoints GetFeaturesInArea would
// return (src/Frame.cc:110-136), in its order.  The buckets of the window's columns form one
// contiguous slice of the sorted array; rows outside [y0, y1) are skipped.  Returns the count.
static __device__ int window_candidates(MatchLds& S, const Window& w, float x, float y, float r, int minLevel,
                                        int maxLevel, bool circ, int lane) {
    if (w.x1 <= w.x0 || w.y1 <= w.y0) return 0;
    const int lo = S.cellStart[w.x0 * ORB_GRID_ROWS + w.y0];
    const int hi = S.cellStart[(w.x1 - 1) * ORB_GRID_ROWS + w.y1];
    const bool checkLevels = minLevel >= 0 || maxLevel >= 0;
    const float r2 = __fmul_rn(r, r);
    int n = 0;
    for (int base = lo; base < hi; base += 64) {
        const int p = base + lane;
        bool ok = false;
        if (p < hi) {
            const int iy = (int)(S.sorted[p] >> 16) % ORB_GRID_ROWS;
            ok = iy >= w.y0 && iy < w.y1;
            if (ok && checkLevels) {
                const int o = S.oct[p];
                if (o < minLevel) ok = false;
                if (maxLevel >= 0 && o > maxLevel) ok = false;
            }
            if (ok) {
                const float2 q = S.xy[p];
                const float dx = __fsub_rn(q.x, x), dy = __fsub_rn(q.y, y);
                ok = circ ? (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < r2) : (fabsf(dx) < r && fabsf(dy) < r);
            }
        }
        const unsigned long long b = __ballot(ok);
        if (ok) S.cand[n + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)p;
        n += __popcll(b);
    }
    wave_lds_sync();
    return n;
}

// Bucket range of GetFeaturesInArea(x, y, r), src/Frame.cc:91-112.
static __device__ __forceinline__ Window window_cells(float x, float y, float r, float wInv, float hInv, bool excl) {
    Window w;
    w.x0 = w.x1 = w.y0 = w.y1 = 0;
    const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(x, r), wInv)));
    if (nMinCellX >= ORB_GRID_COLS) return w;
    const int nMaxCellX = min(ORB_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(x, r), wInv)));
    if (nMaxCellX < 0) return w;
    const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(y, r), hInv)));
    if (excl ? nMinCellY > ORB_GRID_ROWS : nMinCellY >= ORB_GRID_ROWS) return w;
    const int nMaxCellY = min(ORB_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(y, r), hInv)));
    if (nMaxCellY < 0) return w;
    w.x0 = nMinCellX;
    w.x1 = excl ? nMaxCellX : nMaxCellX + 1;
    w.y0 = nMinCellY;
    w.y1 = excl ? nMaxCellY : nMaxCellY + 1;
    if (w.y0 > ORB_GRID_ROWS) w.y0 = ORB_GRID_ROWS;
    if (w.y1 < w.y0) w.y1 = w.y0;
    return w;
}

static __device__ __forceinline__ int hamming_rows(const uint4 a0, const uint4 a1, const uint8_t* __restrict__ b) {
    const uint4 b0 = ((const uint4*)b)[0], b1 = ((const uint4*)b)[1];
    return __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) + __builtin_popcount(a0.z ^ b0.z) +
           __builtin_popcount(a0.w ^ b0.w) + __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
           __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
}

static __device__ __forceinline__ int rot_bin(float a1, float a2) {
    const float factor = ORB_HISTO / 360.0f;  // src/ORBmatcher.cc:35
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == ORB_HISTO) bin = 0;
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

// SearchForInitialization for pair blockIdx.x.  binOf[i1] (global scratch, one
// byte per first-frame keypoint) remembers the rotation bin a match was filed under.
__global__ __launch_bounds__(64) void k_match_window(const orb_keypoint* __restrict__ k1all,
                                                    const uint8_t* __restrict__ d1all, const int* __restrict__ n1s,
                                                    int stride1, const orb_keypoint* __restrict__ k2all,
                                                    const uint8_t* __restrict__ d2all, const int* __restrict__ n2s,
                                                    int stride2, float imgW, float imgH, float* __restrict__ prevAll,
                                                    int* __restrict__ m12All, uint8_t* __restrict__ binOfAll,
                                                    float window, float nnratio, int checkOri, unsigned quirks,
                                                    int* __restrict__ nmatchesOut, int* __restrict__ error) {
    __shared__ MatchLds S;
    const int pair = blockIdx.x, lane = threadIdx.x;
    const orb_keypoint* k1 = k1all + (int64_t)pair * stride1;
    const uint8_t* d1 = d1all + (int64_t)pair * stride1 * 32;
    const orb_keypoint* k2 = k2all + (int64_t)pair * stride2;
    const uint8_t* d2 = d2all + (int64_t)pair * stride2 * 32;
    float* prev = prevAll + (int64_t)pair * stride1 * 2;
    int* m12 = m12All + (int64_t)pair * stride1;
    uint8_t* binOf = binOfAll + (int64_t)pair * stride1;
    const int n1 = n1s[pair], n2 = n2s[pair];
    if (n1 > stride1 || n2 > stride2 || n2 > M_MAXKP || n1 > 32767) {
        if (lane == 0) {
            atomicOr(error, 1);
            nmatchesOut[pair] = 0;
        }
        return;
    }
    const bool excl = (quirks & ORB_Q_GRID_EXCLUSIVE) != 0, circ = (quirks & ORB_Q_WINDOW_CIRCULAR) != 0;
    const float wInv = __fdiv_rn((float)ORB_GRID_COLS, imgW), hInv = __fdiv_rn((float)ORB_GRID_ROWS, imgH);
    for (int i = lane; i < n1; i += 64) {
        m12[i] = -1;
        binOf[i] = 0xFF;
    }
    for (int i = lane; i < n2; i += 64) {
        S.matchedDist[i] = 0xFFFF;
        S.owner[i] = -1;
    }
    if (lane < ORB_HISTO) S.hist[lane] = 0;
    int nIn = 0;
    grid_build(S, k2, n2, wInv, hInv, lane, &nIn);
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        const orb_keypoint kp1 = k1[i1];
        if (kp1.octave > 0) continue;  // :41-44
        const float x = prev[2 * i1], y = prev[2 * i1 + 1];
        const Window w = window_cells(x, y, window, wInv, hInv, excl);
        const uint4 a0 = ((const uint4*)(d1 + (int64_t)i1 * 32))[0], a1 = ((const uint4*)(d1 + (int64_t)i1 * 32))[1];
        // per-lane running (distance << 16 | array position) minimum and second-smallest distance
        uint32_t best = 0xFFFFFFFFu;
        uint32_t second = 0xFFFFu;
        const int nc = window_candidates(S, w, x, y, window, kp1.octave, kp1.octave, circ, lane);
        for (int c = lane; c < nc; c += 64) {
            const int p = S.cand[c];
            const int i2 = (int)(S.sorted[p] & 0xFFFFu);
            const uint32_t dist = (uint32_t)hamming_rows(a0, a1, d2 + (int64_t)i2 * 32);
            if ((uint32_t)S.matchedDist[i2] <= dist) continue;  // :63-64
            const uint32_t key = (dist << 16) | (uint32_t)p;
            if (key < best) {
                second = min(second, best >> 16);
                best = key;
            } else {
                second = min(second, dist);
            }
        }
        // wave reduction: smallest key; second = smallest distance among everything but that one entry
        const uint32_t wbest = wave_min_u32(best);
        const uint32_t cand2 = (best == wbest) ? second : min(second, best >> 16);
        const uint32_t wsecond = wave_min_u32(cand2);
        if (wbest == 0xFFFFFFFFu) continue;  // no candidate survived: bestDist = INT_MAX fails the TH_LOW test
        const int bestDist = (int)(wbest >> 16);
        const int bestIdx2 = (int)(S.sorted[wbest & 0xFFFFu] & 0xFFFFu);
        const float bestDist2 = (wsecond == 0xFFFFu) ? (float)INT_MAX : (float)(int)wsecond;
        if (bestDist <= M_TH_LOW && (float)bestDist < __fmul_rn(bestDist2, nnratio)) {  // :78-80
            if (lane == 0) {
                const int old = S.owner[bestIdx2];
                if (old >= 0) {
                    m12[old] = -1;
                    nmatches--;
                }
                m12[i1] = bestIdx2;
                S.owner[bestIdx2] = (int16_t)i1;
                S.matchedDist[bestIdx2] = (uint16_t)bestDist;
                nmatches++;
                if (checkOri) {
                    const int bin = rot_bin(kp1.angle, k2[bestIdx2].angle);
                    binOf[i1] = (uint8_t)bin;
                    S.hist[bin]++;
                }
            }
            wave_lds_sync();
        }
    }
    __threadfence_block();
    wave_lds_sync();
    nmatches = __shfl(nmatches, 0, 64);
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(S.hist, ind1, ind2, ind3);
        int removed = 0;
        for (int i = lane; i < n1; i += 64) {
            const int bin = binOf[i];
            if (bin == 0xFF || bin == ind1 || bin == ind2 || bin == ind3) continue;
            if (m12[i] >= 0) {  // :122-126 (a stolen match is already -1)
                m12[i] = -1;
                removed++;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o, 64);
        nmatches -= removed;
    }
    __threadfence_block();
    wave_lds_sync();
    for (int i = lane; i < n1; i += 64) {  // :131-133
        const int m = m12[i];
        if (m >= 0) {
            prev[2 * i] = k2[m].x;
            prev[2 * i + 1] = k2[m].y;
        }
    }
    if (lane == 0) nmatchesOut[pair] = nmatches;
}

// Search loop of SearchByProjection(Frame&, const Frame&, th, bMono) for one frame.
__global__ __launch_bounds__(64) void k_match_projected(const float* __restrict__ u, const float* __restrict__ v,
                                                       const float* __restrict__ radius,
                                                       const int* __restrict__ minLevel,
                                                       const int* __restrict__ maxLevel,
                                                       const float* __restrict__ qangle,
                                                       const uint8_t* __restrict__ qdesc,
                                                       const uint8_t* __restrict__ qobs, int nq,
                                                       const orb_keypoint* __restrict__ k2,
                                                       const uint8_t* __restrict__ d2, int n2, float imgW, float imgH,
                                                       const uint8_t* __restrict__ occupied2, int* __restrict__ match2,
                                                       int* __restrict__ qBin, int* __restrict__ qIdx2, int thHigh,
                                                       int checkOri, unsigned quirks, int* __restrict__ nmatchesOut,
                                                       int* __restrict__ error) {
    __shared__ MatchLds S;
    const int lane = threadIdx.x;
    if (n2 > M_MAXKP) {
        if (lane == 0) {
            atomicOr(error, 1);
            nmatchesOut[0] = 0;
        }
        return;
    }
    const bool excl = (quirks & ORB_Q_GRID_EXCLUSIVE) != 0, circ = (quirks & ORB_Q_WINDOW_CIRCULAR) != 0;
    const float wInv = __fdiv_rn((float)ORB_GRID_COLS, imgW), hInv = __fdiv_rn((float)ORB_GRID_ROWS, imgH);
    for (int i = lane; i < n2; i += 64) {
        match2[i] = -1;
        S.matchedDist[i] = occupied2 ? occupied2[i] : 0;  // here: occupancy flag
    }
    for (int i = lane; i < nq; i += 64) qBin[i] = -1;
    if (lane < ORB_HISTO) S.hist[lane] = 0;
    int nIn = 0;
    grid_build(S, k2, n2, wInv, hInv, lane, &nIn);
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        const float x = u[i], y = v[i];
        if (x < 0.f || x > imgW) continue;  // :323-326
        if (y < 0.f || y > imgH) continue;
        const float r = radius[i];
        const int lmin = minLevel[i], lmax = maxLevel[i];
        const Window w = window_cells(x, y, r, wInv, hInv, excl);
        const uint4 a0 = ((const uint4*)(qdesc + (int64_t)i * 32))[0], a1 = ((const uint4*)(qdesc + (int64_t)i * 32))[1];
        uint32_t best = 0xFFFFFFFFu;
        const int nc = window_candidates(S, w, x, y, r, lmin, lmax, circ, lane);
        for (int c = lane; c < nc; c += 64) {
            const int p = S.cand[c];
            const int i2 = (int)(S.sorted[p] & 0xFFFFu);
            if (S.matchedDist[i2]) continue;  // :345-347
            const uint32_t dist = (uint32_t)hamming_rows(a0, a1, d2 + (int64_t)i2 * 32);
            best = min(best, (dist << 16) | (uint32_t)p);
        }
        const uint32_t wbest = wave_min_u32(best);
        if (wbest == 0xFFFFFFFFu) continue;
        const int bestDist = (int)(wbest >> 16);
        if (bestDist >= 256) continue;  // bestDist starts at 256 with strict '<', :339,357
        if (bestDist <= thHigh) {
            const int bestIdx2 = (int)(S.sorted[wbest & 0xFFFFu] & 0xFFFFu);
            if (lane == 0) {
                match2[bestIdx2] = i;
                S.matchedDist[bestIdx2] = qobs ? qobs[i] : 1;
                nmatches++;
                if (checkOri) {
                    const int bin = rot_bin(qangle[i], k2[bestIdx2].angle);
                    qBin[i] = bin;
                    qIdx2[i] = bestIdx2;
                    S.hist[bin]++;
                }
            }
            wave_lds_sync();
        }
    }
    __threadfence_block();
    wave_lds_sync();
    nmatches = __shfl(nmatches, 0, 64);
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima(S.hist, ind1, ind2, ind3);
        int removed = 0;
        for (int i = lane; i < nq; i += 64) {
            const int bin = qBin[i];
            if (bin < 0 || bin == ind1 || bin == ind2 || bin == ind3) continue;
            match2[qIdx2[i]] = -2;  // :400-401, once per list entry (-2: assigned, then cleared)
            removed++;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o, 64);
        nmatches -= removed;
    }
    if (lane == 0) nmatchesOut[0] = nmatches;
}

// ---- host side ------------------------------------------------------------------------
struct MatchState {
    void* binOf = nullptr;
    size_t binBytes = 0;
    void* err = nullptr;
};

void orb_match_state_free(void* p) {
    MatchState* s = (MatchState*)p;
    if (!s) return;
    if (s->binOf) (void)hipFree(s->binOf);
    if (s->err) (void)hipFree(s->err);
    delete s;
}

// accessors implemented in orb_api.hip
int orb_ctx_device(const orb_ctx* c);
unsigned orb_ctx_quirks(const orb_ctx* c);
int orb_ctx_max_batch(const orb_ctx* c);
void** orb_ctx_match_state(orb_ctx* c);
void* orb_ctx_stream(orb_ctx* c);

static int match_state(orb_ctx* c, size_t binBytes, MatchState** out) {
    void** slot = orb_ctx_match_state(c);
    if (!*slot) {
        MatchState* s = new MatchState();
        hipError_t e = hipMalloc(&s->err, 4);
        if (e == hipSuccess) e = hipMemset(s->err, 0, 4);
        if (e != hipSuccess) {
            delete s;
            orb_set_error("matcher state allocation failed: %s", hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        *slot = s;
    }
    MatchState* s = (MatchState*)*slot;
    if (binBytes > s->binBytes) {
        if (s->binOf) (void)hipFree(s->binOf);
        s->binOf = nullptr;
        s->binBytes = 0;
        hipError_t e = hipMalloc(&s->binOf, binBytes);
        if (e != hipSuccess) {
            orb_set_error("hipMalloc(%zu) failed: %s", binBytes, hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        s->binBytes = binBytes;
    }
    *out = s;
    return ORB_OK;
}

extern "C" int orb_match_window_batch_device(orb_ctx* c, int npairs, const orb_keypoint* d_k1, const uint8_t* d_d1,
                                             const int* d_n1, int stride1, const orb_keypoint* d_k2,
                                             const uint8_t* d_d2, const int* d_n2, int stride2, int img_w, int img_h,
                                             float* d_prev_xy, int* d_matches12, int window, float nnratio,
                                             int check_ori, int* d_nmatches, void* stream) {
    if (!c || npairs < 1 || !d_k1 || !d_d1 || !d_n1 || !d_k2 || !d_d2 || !d_n2 || !d_prev_xy || !d_matches12 ||
        !d_nmatches || stride1 < 1 || stride2 < 1 || img_w < 1 || img_h < 1) {
        orb_set_error("orb_match_window_batch_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (npairs > orb_ctx_max_batch(c)) {
        orb_set_error("%d pairs exceed the context's max_batch", npairs);
        return ORB_ERR_INVALID;
    }
    if (stride2 > M_MAXKP || stride1 > 32767) {
        orb_set_error("windowed matcher handles at most %d keypoints in the searched frame and 32767 in the query frame",
                      M_MAXKP);
        return ORB_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    int rc = match_state(c, (size_t)orb_ctx_max_batch(c) * 32768, &s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_match_window, dim3(npairs), dim3(64), 0, (hipStream_t)stream, d_k1, d_d1, d_n1, stride1, d_k2,
                       d_d2, d_n2, stride2, (float)img_w, (float)img_h, d_prev_xy, d_matches12, (uint8_t*)s->binOf,
                       (float)window, nnratio, check_ori, orb_ctx_quirks(c), d_nmatches, (int*)s->err);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("match launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

namespace {
struct Tmp {  // scoped device allocation for the host-pointer entry points
    void* p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t put(const void* src, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 4);
        if (e == hipSuccess && src && bytes) e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        return e;
    }
};
}  // namespace

extern "C" int orb_match_window(orb_ctx* c, const orb_keypoint* k1, const uint8_t* d1, int n1, const orb_keypoint* k2,
                                const uint8_t* d2, int n2, int img_w, int img_h, float* prev_xy, int* matches12,
                                int window, float nnratio, int check_ori, int* nmatches) {
    if (!c || n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && (!k1 || !d1 || !prev_xy || !matches12)) ||
        (n2 > 0 && (!k2 || !d2)) || img_w < 1 || img_h < 1) {
        orb_set_error("orb_match_window: bad arguments");
        return ORB_ERR_INVALID;
    }
    *nmatches = 0;
    if (n1 == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    Tmp tk1, td1, tk2, td2, tn, tprev, tm, tnm;
    const int ns[2] = {n1, n2};
    const int s1 = n1, s2 = n2 > 0 ? n2 : 1;
    HIP_TRY(tk1.put(k1, (size_t)n1 * sizeof(orb_keypoint)));
    HIP_TRY(td1.put(d1, (size_t)n1 * 32));
    HIP_TRY(tk2.put(k2, (size_t)n2 * sizeof(orb_keypoint)));
    HIP_TRY(td2.put(d2, (size_t)n2 * 32));
    HIP_TRY(tn.put(ns, sizeof(ns)));
    HIP_TRY(tprev.put(prev_xy, (size_t)n1 * 8));
    HIP_TRY(tm.put(nullptr, (size_t)n1 * 4));
    HIP_TRY(tnm.put(nullptr, 4));
    int rc = orb_match_window_batch_device(c, 1, (const orb_keypoint*)tk1.p, (const uint8_t*)td1.p, (const int*)tn.p, s1,
                                           (const orb_keypoint*)tk2.p, (const uint8_t*)td2.p, (const int*)tn.p + 1, s2,
                                           img_w, img_h, (float*)tprev.p, (int*)tm.p, window, nnratio, check_ori,
                                           (int*)tnm.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    MatchState* s = (MatchState*)*orb_ctx_match_state(c);
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, s->err, 4, hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(s->err, 0, 4));
        orb_set_error("windowed matcher: keypoint count exceeds the kernel's limits");
        return ORB_ERR_CAPACITY;
    }
    HIP_TRY(hipMemcpy(prev_xy, tprev.p, (size_t)n1 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(matches12, tm.p, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nmatches, tnm.p, 4, hipMemcpyDeviceToHost));
    return ORB_OK;
}

extern "C" int orb_match_projected(orb_ctx* c, const float* u, const float* v, const float* radius,
                                   const int* min_level, const int* max_level, const float* qangle,
                                   const uint8_t* qdesc, const uint8_t* qobs, int nq, const orb_keypoint* k2,
                                   const uint8_t* d2, int n2, int img_w, int img_h, const uint8_t* occupied2,
                                   int* match2, int th_high, int check_ori, int* nmatches) {
    if (!c || nq < 0 || n2 < 0 || !nmatches || (n2 > 0 && (!k2 || !d2 || !match2)) ||
        (nq > 0 && (!u || !v || !radius || !min_level || !max_level || !qangle || !qdesc)) || img_w < 1 || img_h < 1) {
        orb_set_error("orb_match_projected: bad arguments");
        return ORB_ERR_INVALID;
    }
    *nmatches = 0;
    if (n2 > M_MAXKP) {
        orb_set_error("projected matcher handles at most %d keypoints in the searched frame", M_MAXKP);
        return ORB_ERR_UNSUPPORTED;
    }
    if (n2 == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    MatchState* s = nullptr;
    int rc = match_state(c, 4, &s);
    if (rc) return rc;
    Tmp tu, tv, tr, tl0, tl1, ta, tq, to, tk2, td2, tocc, tm, tb, ti, tnm;
    HIP_TRY(tu.put(u, (size_t)nq * 4));
    HIP_TRY(tv.put(v, (size_t)nq * 4));
    HIP_TRY(tr.put(radius, (size_t)nq * 4));
    HIP_TRY(tl0.put(min_level, (size_t)nq * 4));
    HIP_TRY(tl1.put(max_level, (size_t)nq * 4));
    HIP_TRY(ta.put(qangle, (size_t)nq * 4));
    HIP_TRY(tq.put(qdesc, (size_t)nq * 32));
    if (qobs) HIP_TRY(to.put(qobs, (size_t)nq));
    HIP_TRY(tk2.put(k2, (size_t)n2 * sizeof(orb_keypoint)));
    HIP_TRY(td2.put(d2, (size_t)n2 * 32));
    if (occupied2) HIP_TRY(tocc.put(occupied2, (size_t)n2));
    HIP_TRY(tm.put(nullptr, (size_t)n2 * 4));
    HIP_TRY(tb.put(nullptr, (size_t)nq * 4));
    HIP_TRY(ti.put(nullptr, (size_t)nq * 4));
    HIP_TRY(tnm.put(nullptr, 4));
    hipLaunchKernelGGL(k_match_projected, dim3(1), dim3(64), 0, (hipStream_t) nullptr, (const float*)tu.p,
                       (const float*)tv.p, (const float*)tr.p, (const int*)tl0.p, (const int*)tl1.p,
                       (const float*)ta.p, (const uint8_t*)tq.p, (const uint8_t*)to.p, nq, (const orb_keypoint*)tk2.p,
                       (const uint8_t*)td2.p, n2, (float)img_w, (float)img_h, (const uint8_t*)tocc.p, (int*)tm.p,
                       (int*)tb.p, (int*)ti.p, th_high, check_ori, orb_ctx_quirks(c), (int*)tnm.p, (int*)s->err);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("match launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(match2, tm.p, (size_t)n2 * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nmatches, tnm.p, 4, hipMemcpyDeviceToHost));
    return ORB_OK;
}
