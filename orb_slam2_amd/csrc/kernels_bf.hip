// Brute-force Hamming top-2 over a descriptor database shard for gfx950.
//
// The unit of work is ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:460-476), the popcount of a 256-bit XOR, and
// the scan applies the best / second-best rule of src/ORBmatcher.cc:53-76 to every database row in index order.
// All pairs of (query, row) with a 256-term sum each is a matrix product, so the distances are made on the matrix
// cores (k_bf_mfma below: bits as the FP4 values +-1 / 0-1, v_mfma_scale_f32_32x32x64_f8f6f4), and only the top-2 bookkeeping runs on
// the vector ALU.  The database is cut into chunks along blockIdx.y so that a shard fills the chip; per-chunk
// results are packed as best << 48 | second << 32 | global row and combined by the same merge kernel that combines
// the shards of several GPUs.  (The vector-ALU form of the scan, 8 x (v_xor_b32 + v_bcnt_u32_b32) per comparison at
// 0.88 of that pair's measured issue ceiling, is kept under tools/ for reference: the matrix form is 3.4 times as fast.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "orb_internal.h"

#define BF_NONE 0xFFFFu

struct orb_bf {
    int device = 0;
    const uint8_t* d_db = nullptr;
    bool ownsDb = false;
    int64_t n = 0, globalBase = 0;
    void* d_partial = nullptr;
    size_t partialBytes = 0;
    void* d_q = nullptr;
    size_t qBytes = 0;
    void* d_packed = nullptr;
    size_t packedBytes = 0;
    void* d_out = nullptr;
    size_t outBytes = 0;
    hipStream_t stream = nullptr;
    int residentBlocks = 0;  // workgroups of the scan kernel the device holds at once
    unsigned long long* d_static = nullptr;  // this device's copy of the library-held partials block (see below)
    unsigned long long* d_clock = nullptr;   // engine-clock probe (orb_bf_clock_probe): two sums the scan kernel adds to
};

// Per-chunk partial results for orb_bf_query_packed_noalloc, the entry point for callers that must not
// allocate: a fixed block in the library's device image instead of a buffer grown on demand.  Every device
// the library is loaded on holds its own copy at its own address; host code never passes the array's
// name to a kernel but the address hipGetSymbolAddress returned for that device (bf_static_block).
// All calls on one device share the block, so calls that overlap in time on one device race.
#define BF_STATIC_WORDS (4u << 20)  // 64-bit words: 32 MiB
static __device__ __align__(256) unsigned long long g_bfStaticPartials[BF_STATIC_WORDS];

#define BF_MAX_DEVICES 64
static unsigned long long* g_bfStaticAddr[BF_MAX_DEVICES];
static std::mutex g_bfStaticMutex;

// Address of g_bfStaticPartials on `device`, which must be the current device.
static int bf_static_block(int device, unsigned long long** out) {
    if (device < 0 || device >= BF_MAX_DEVICES) {
        orb_set_error("device ordinal %d beyond the %d the library keeps a partials block for", device, BF_MAX_DEVICES);
        return ORB_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lock(g_bfStaticMutex);
    if (!g_bfStaticAddr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_bfStaticPartials)));
        g_bfStaticAddr[device] = (unsigned long long*)p;
    }
    *out = g_bfStaticAddr[device];
    return ORB_OK;
}


// ---- the scan: all-pairs Hamming distances on the matrix cores ------------------------------------------------
// Query bits become the values +1 (set) / -1 (clear), database bits 1 / 0.  The dot product of a row and a query is
// then (bits set in both) - (bits set in the row only) = popcount(query) - distance, so per query the largest
// product is the nearest row.  The values are held as FP4 (E2M1: 0x0 = 0, 0x2 = 1, 0xA = -1; block scales 2^0), the
// narrowest operand type of the matrix cores, and v_mfma_scale_f32_32x32x64_f8f6f4 makes 32 rows x 32 queries x 64
// bits per instruction into f32 accumulators — sums of at most 256 terms of magnitude 1, exact in any order.
// (As int8 with v_mfma_i32_32x32x32_i8 the scan ran at 6.2 k Gcmp/s, as FP4 at 9.2 k; on the vector ALU at 2.7 k.)
//
// A wavefront keeps BFM_QB blocks of 32 queries as B operands in registers for the whole scan (lane (c, h) holds word
// 2 ks + h of query c for the four k-steps ks; the A operand below uses the same bit order, which is all the
// products need).  The workgroup's four wavefronts share the database rows: 256 threads fetch a tile of BFM_TR rows
// as bits (8 bytes per thread), expand them to FP4 in LDS (five instructions per byte, see bfm_expand32), two tiles in
// flight, and every wavefront reads its A operands from there with ds_read_b128.  The 32 x 32 products of a (row
// block, query block) arrive with the query on the lane and 16 rows in 16 registers (row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5)): a lane keeps the top-2 of one query over its rows, and the two lanes of a query report separately
// (the merge kernel combines them like two chunks).
//
// What the vector ALU does beside the matrix instructions decides the rate, in three ways:
//  * A matrix instruction holds the SIMD's issue port for part of its 32 cycles only; about four independent vector
//    instructions fit beside it, but a wavefront issues in order, so they have to stand between the matrix
//    instructions in the instruction stream.  The tile's two row blocks therefore use two sets of accumulators:
//    while the 16 matrix instructions of one block run, the filter of the other block's finished products (the
//    maximum of a lane's 16 against the query's threshold: eight v_max3_f32 and a compare per query block) and the
//    expansion of the next tile's rows are issued between them (bfm_half; sched_group_barrier gives the order).
//  * Looking at the 16 registers one by one happens only for query blocks in which some lane passed the filter.
//    That costs a few hundred cycles with no matrix instruction in flight, so how often it happens matters: the
//    threshold is the larger second best of the query's two lanes, raised by what the other chunks of the same query,
//    which run side by side, have published in `shared` (bfm_publish / the refresh in the tile loop).
//  * The file is built with -fno-honor-nans (or every accumulator is canonicalised by a v_max_f32 x, x before a
//    maximum) and with machine-code sinking disabled (or the matrix instructions, whose results are first used a
//    branch later, are moved out of the block they were interleaved in).
typedef int bfm_v4i __attribute__((ext_vector_type(4)));
typedef int bfm_v8i __attribute__((ext_vector_type(8)));
typedef float bfm_v16f __attribute__((ext_vector_type(16)));
#define BFM_QB 4      // blocks of 32 queries per wavefront: 512 queries per workgroup
#define BFM_TR 64     // database rows per tile
#define BFM_ROWP 144  // bytes per expanded row in LDS (128 + 16: consecutive rows start 16 bytes apart in the banks)
#define BFM_LDS (2 * BFM_TR * BFM_ROWP)
#define BFM_QUERIES_PER_WG (4 * BFM_QB * 32)
#define BFM_NONE (-100000.f)  // below any product
#define BFM_SCALE_ONE 0x7f7f7f7f  // E8M0 block scales 2^0

// 32 descriptor bits -> 32 FP4 values (16 bytes)
template <bool QUERY>
static __device__ __forceinline__ bfm_v4i bfm_expand32(uint32_t bits) {
    bfm_v4i o;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        // the byte's bit i -> bit 1 of nibble i (the FP4 value 1), five instructions: the byte's four bit pairs are
        // brought to the low bits of four bytes (shifted copies ORed together: no kept bit is hit twice), and each
        // pair selects a byte of the table {0x00, 0x02, 0x20, 0x22} through v_perm_b32
        uint32_t m = (bits >> (8 * n)) & 255u;
        m |= m << 6;
        m |= m << 12;
        m = __builtin_amdgcn_perm(0u, 0x22200200u, m & 0x03030303u);
        if (QUERY) m = 0x22222222u | ((m ^ 0x22222222u) << 2);  // clear bit: the sign on top of the 1
        o[n] = (int)m;
    }
    return o;
}

#define BFM_REFRESH 32  // tiles between two looks at the thresholds the other chunks have published (a power of two)
#define BFM_BIAS 1024.f // added to a published product: positive, so that the bit patterns order like the values

__global__ __launch_bounds__(256, 2) void k_bf_mfma(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ db,
                                                    long long n, long long rowsPerChunk, long long globalBase,
                                                    unsigned long long* __restrict__ partial,
                                                    uint32_t* __restrict__ shared,  // (null: a single chunk)
                                                    unsigned long long* __restrict__ clk) {  // (null: no clock probe)
    static_assert(BFM_QB == 4 && BFM_TR == 64, "the interleaving is written for four query blocks and two row blocks");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rows[];  // two buffers of BFM_TR expanded rows
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, c = lane & 31;
    const long long q0 = (long long)blockIdx.x * BFM_QUERIES_PER_WG + w * (BFM_QB * 32);
    const long long r0 = (long long)blockIdx.y * rowsPerChunk;
    const long long rEnd = r0 + rowsPerChunk < n ? r0 + rowsPerChunk : n;
    const int rows = (int)(rEnd - r0);
    if (rows <= 0) return;  // (the whole workgroup; the launcher makes no such chunk)
    // engine clock while the launch runs: shader-clock ticks per tick of the constant 100 MHz counter, summed over
    // the workgroups (each weighs in with its own duration)
    const unsigned long long clk0 = clk ? clock64() : 0ull, wall0 = clk ? wall_clock64() : 0ull;

    bfm_v4i bq[BFM_QB][4];
    int pcq[BFM_QB];  // popcount of the lane's queries: distance = popcount - product
#pragma unroll
    for (int b = 0; b < BFM_QB; b++) {
        const long long qi = q0 + b * 32 + c;
        pcq[b] = 0;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const uint32_t w0 = qi < nq ? q[qi * 8 + 2 * ks] : 0u, w1 = qi < nq ? q[qi * 8 + 2 * ks + 1] : 0u;
            pcq[b] += __popc(w0) + __popc(w1);
            bq[b][ks] = bfm_expand32<true>(h ? w1 : w0);
        }
    }
    // best / sec: the lane's own two largest products; thr: what a product has to beat to matter at all -- the
    // larger second best of the query's two lanes (they see rows 0-3, 8-11, .. and 4-7, 12-15, .. of every block; a
    // product that does not beat it has two earlier rows at least as good in one list or the other), or another
    // chunk's published second best less one half (a product equal to that one may still be the earlier row)
    float best[BFM_QB], sec[BFM_QB], thr[BFM_QB];
    int bidx[BFM_QB];
    bool flagged[BFM_QB];  // query blocks in which the last filter pass saw a lane above its threshold
#pragma unroll
    for (int b = 0; b < BFM_QB; b++) best[b] = sec[b] = thr[b] = BFM_NONE, bidx[b] = 0, flagged[b] = false;

    // staging: thread = (row t / 4, word pair t % 4) of a tile
    const int srow = t >> 2, swp = t & 3;
    auto fetch = [&](int tile) -> uint2 {
        const long long r = r0 + (long long)tile * BFM_TR + srow;
        return r < rEnd ? *(const uint2*)(db + r * 8 + 2 * swp) : uint2{0u, 0u};  // (rows past the end: no bit set,
    };                                                                              // and rejected by number below)

    // The registers of the flagged query blocks one by one, rows in ascending order: on equal products the earlier
    // row stays the best, as in a scan.
    auto update = [&](const bfm_v16f* acc, int rowBase) {
#pragma unroll
        for (int b = 0; b < BFM_QB; b++) {
            if (__ballot(flagged[b]) == 0ull) continue;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = rowBase + (i & 3) + 8 * (i >> 2);
                const float v = acc[b][i];
                if (r < rows && v > sec[b]) {
                    if (v > best[b]) {
                        sec[b] = best[b];
                        best[b] = v;
                        bidx[b] = r;
                    } else {
                        sec[b] = v;
                    }
                }
            }
            const float own = fmaxf(sec[b], __shfl_xor(sec[b], 32));
            thr[b] = fmaxf(thr[b], own);
            const long long qi = q0 + b * 32 + c;
            if (shared && h == 0 && qi < nq && own > -BFM_BIAS) atomicMax(&shared[qi], __float_as_uint(own + BFM_BIAS));
        }
    };

    // One row block: its 16 matrix instructions (operands at aBase -> accW), and between them the filter of accR,
    // the products of the block before, and the expansion of one word of the next tile's rows (stored at dst).
    // Returns whether any lane of the wavefront has a product above its threshold in accR.
    auto half = [&](const uint8_t* aBase, bfm_v16f* accW, const bfm_v16f* accR, uint32_t word, uint8_t* dst) -> bool {
        bfm_v4i aK[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) aK[ks] = *(const bfm_v4i*)(aBase + ks * 32);
        float mm[5], mA = 0.f;
        bool hit = false;
        uint32_t ex = 0u;
        bfm_v4i o = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int ks = j >> 2, b = j & 3;
            const bfm_v8i a = {aK[ks][0], aK[ks][1], aK[ks][2], aK[ks][3], 0, 0, 0, 0};  // (FP4 operands are the low four registers)
            const bfm_v8i bb = {bq[b][ks][0], bq[b][ks][1], bq[b][ks][2], bq[b][ks][3], 0, 0, 0, 0};
            if (ks == 0) {
                const bfm_v16f zero = {};
                accW[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, zero, 4, 4, 0, BFM_SCALE_ONE, 0, BFM_SCALE_ONE);
            } else {
                accW[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, accW[b], 4, 4, 0, BFM_SCALE_ONE, 0, BFM_SCALE_ONE);
            }
            // step j of the filter and the expansion: query block / byte j / 4, quarter j % 4
            const int eb = j >> 2, p = j & 3;
            const bfm_v16f s = accR[eb];
#define BFM_MAX3(a, b, c) fmaxf(fmaxf(a, b), c)
            if (p == 0) {
                mm[0] = BFM_MAX3(s[0], s[1], s[2]);
                mm[1] = BFM_MAX3(s[3], s[4], s[5]);
                ex = (word >> (8 * eb)) & 255u;
                ex |= ex << 6;
            } else if (p == 1) {
                mm[2] = BFM_MAX3(s[6], s[7], s[8]);
                mm[3] = BFM_MAX3(s[9], s[10], s[11]);
                ex |= ex << 12;
            } else if (p == 2) {
                mm[4] = BFM_MAX3(s[12], s[13], s[14]);
                mA = BFM_MAX3(mm[0], mm[1], mm[2]);
                ex &= 0x03030303u;
            } else {
                const float m = fmaxf(mA, BFM_MAX3(mm[3], mm[4], s[15]));
                flagged[eb] = m > thr[eb];
                hit |= flagged[eb];
                o[eb] = (int)__builtin_amdgcn_perm(0u, 0x22200200u, ex);  // (as in bfm_expand32)
            }
#undef BFM_MAX3
        }
        // the order the scheduler is to build from this block: the four operand reads, then one matrix instruction
        // and up to four vector instructions in turn
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        *(bfm_v4i*)dst = o;  // (past the last tile: rows that are never read)
        return __ballot(hit) != 0ull;
    };

    const int ntiles = (rows + BFM_TR - 1) / BFM_TR;
    {
        const uint2 w0 = fetch(0);
        uint8_t* d = s_rows + srow * BFM_ROWP + swp * 32;
        *(bfm_v4i*)(d + 0) = bfm_expand32<false>(w0.x);
        *(bfm_v4i*)(d + 16) = bfm_expand32<false>(w0.y);
    }
    uint2 ahead = fetch(1);
    __syncthreads();
    bfm_v16f acc0[BFM_QB], acc1[BFM_QB];
#pragma unroll
    for (int b = 0; b < BFM_QB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[b][i] = BFM_NONE;  // (what the first filter pass looks at)
    for (int tile = 0; tile < ntiles; tile++) {
        if (shared && (tile & (BFM_REFRESH - 1)) == BFM_REFRESH - 1) {
#pragma unroll
            for (int b = 0; b < BFM_QB; b++) {
                const long long qi = q0 + b * 32 + c;
                if (qi < nq) {
                    const uint32_t g = __hip_atomic_load(&shared[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g != 0u) thr[b] = fmaxf(thr[b], __uint_as_float(g) - (BFM_BIAS + 0.5f));
                }
            }
        }
        const int buf = tile & 1;
        const uint2 nextw = ahead;  // tile + 1, requested a tile ago
        ahead = fetch(tile + 2);
        const uint8_t* base = s_rows + buf * (BFM_TR * BFM_ROWP) + c * BFM_ROWP + h * 16;
        uint8_t* d = s_rows + (buf ^ 1) * (BFM_TR * BFM_ROWP) + srow * BFM_ROWP + swp * 32;
        __builtin_amdgcn_sched_barrier(0);
        if (half(base, acc0, acc1, nextw.x, d)) update(acc1, (tile - 1) * BFM_TR + 32 + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
        if (half(base + 32 * BFM_ROWP, acc1, acc0, nextw.y, d + 16)) update(acc0, tile * BFM_TR + 4 * h);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < BFM_QB; b++) flagged[b] = true;
    update(acc1, (ntiles - 1) * BFM_TR + 32 + 4 * h);
#pragma unroll
    for (int b = 0; b < BFM_QB; b++) {
        const long long qi = q0 + b * 32 + c;
        if (qi < nq) {
            const unsigned long long bd = best[b] == BFM_NONE ? (unsigned long long)BF_NONE : (unsigned long long)(pcq[b] - (int)best[b]);
            const unsigned long long sd = sec[b] == BFM_NONE ? (unsigned long long)BF_NONE : (unsigned long long)(pcq[b] - (int)sec[b]);
            const unsigned long long idx =
                best[b] == BFM_NONE ? 0xFFFFFFFFull : (unsigned long long)(globalBase + r0 + bidx[b]) & 0xFFFFFFFFull;
            partial[((long long)blockIdx.y * 2 + h) * nq + qi] = (bd << 48) | (sd << 32) | idx;
        }
    }
    if (clk && t == 0) {
        atomicAdd(&clk[0], clock64() - clk0);
        atomicAdd(&clk[1], wall_clock64() - wall0);
    }
}
#define BF_PARTS_PER_CHUNK 2  // partial words per (chunk, query): one per half of the wavefront


// Combine nshards packed candidates per query.  Inserting every shard's best
// (distance, row) and second distance into one running top-2 equals scanning
// the concatenated rows: the best is the smallest (distance, row), the second
// the second-smallest distance with multiplicity.
__global__ __launch_bounds__(256) void k_bf_merge(const unsigned long long* __restrict__ gathered, int nshards, int nq,
                                                  unsigned long long* __restrict__ packedOut,
                                                  int32_t* __restrict__ bestIdx, uint16_t* __restrict__ bestDist,
                                                  uint16_t* __restrict__ secondDist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    uint32_t bd = BF_NONE, sd = BF_NONE, bi = 0xFFFFFFFFu;
    for (int s = 0; s < nshards; s++) {
        const unsigned long long p = gathered[(long long)s * nq + i];
        const uint32_t d = (uint32_t)(p >> 48), d2 = (uint32_t)(p >> 32) & 0xFFFFu, idx = (uint32_t)p;
        if (d == BF_NONE) continue;
        if (d < bd || (d == bd && idx < bi)) {
            sd = bd;  // bd <= sd always, so the displaced best is the new second unless d2 beats it below
            bd = d;
            bi = idx;
        } else if (d < sd) {
            sd = d;
        }
        if (d2 < sd) sd = d2;
    }
    if (packedOut) packedOut[i] = ((unsigned long long)bd << 48) | ((unsigned long long)sd << 32) | bi;
    if (bestIdx) bestIdx[i] = (bd == BF_NONE) ? -1 : (int32_t)bi;
    if (bestDist) bestDist[i] = (uint16_t)bd;
    if (secondDist) secondDist[i] = (uint16_t)sd;
}

extern "C" int orb_bf_create(int device, orb_bf** out) {
    if (!out) {
        orb_set_error("orb_bf_create: null argument");
        return ORB_ERR_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        orb_set_error("device %d not present (%d devices)", device, ndev);
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    // everything a query needs besides buffers is settled here, so that the query entry points make no
    // runtime call other than launches (and, for the growing variants, hipMalloc)
    unsigned long long* block = nullptr;
    int rc = bf_static_block(device, &block);
    if (rc) return rc;
    int perCu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_bf_mfma, 256, BFM_LDS));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    orb_bf* bf = new orb_bf();
    bf->device = device;
    bf->stream = st;
    bf->d_static = block;
    bf->residentBlocks = std::max(1, perCu) * std::max(1, cus);
    *out = bf;
    return ORB_OK;
}

static void bf_free(void*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

extern "C" void orb_bf_destroy(orb_bf* bf) {
    if (!bf) return;
    (void)hipSetDevice(bf->device);
    (void)hipDeviceSynchronize();
    if (bf->ownsDb) {
        void* p = (void*)bf->d_db;
        bf_free(p);
    }
    bf_free(bf->d_partial);
    bf_free(bf->d_q);
    bf_free(bf->d_packed);
    bf_free(bf->d_out);
    if (bf->d_clock) (void)hipFree(bf->d_clock);
    if (bf->stream) (void)hipStreamDestroy(bf->stream);
    delete bf;
}

static int bf_ensure(void*& p, size_t& have, size_t need) {
    if (need <= have) return ORB_OK;
    bf_free(p);
    have = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
        orb_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    have = need;
    return ORB_OK;
}

extern "C" int orb_bf_set_database_device(orb_bf* bf, const uint8_t* d_db, int64_t n, int64_t global_base) {
    if (!bf || n < 0 || (n > 0 && !d_db) || global_base < 0 || global_base + n > 0xFFFFFFFFll) {
        orb_set_error("orb_bf_set_database: bad arguments (rows are indexed with 32 bits)");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(bf->device));
    if (bf->ownsDb) {
        HIP_TRY(hipDeviceSynchronize());
        void* p = (void*)bf->d_db;
        bf_free(p);
        bf->ownsDb = false;
    }
    bf->d_db = d_db;
    bf->n = n;
    bf->globalBase = global_base;
    return ORB_OK;
}

extern "C" int orb_bf_set_database(orb_bf* bf, const uint8_t* db, int64_t n, int64_t global_base) {
    if (!bf || n < 0 || (n > 0 && !db)) {
        orb_set_error("orb_bf_set_database: bad arguments");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(bf->device));
    void* d = nullptr;
    if (n > 0) {
        hipError_t e = hipMalloc(&d, (size_t)n * 32);
        if (e != hipSuccess) {
            orb_set_error("hipMalloc(%zu) failed: %s", (size_t)n * 32, hipGetErrorString(e));
            return ORB_ERR_NOMEM;
        }
        HIP_TRY(hipMemcpy(d, db, (size_t)n * 32, hipMemcpyHostToDevice));
    }
    int rc = orb_bf_set_database_device(bf, (const uint8_t*)d, n, global_base);
    if (rc) {
        if (d) (void)hipFree(d);
        return rc;
    }
    bf->ownsDb = (d != nullptr);
    return ORB_OK;
}

// Chunks along the database.  Every workgroup does the same amount of work, so
// the launch runs in rounds of `resident` workgroups; the chunk count is chosen
// so that the last round is (almost) full instead of leaving most of the chip
// idle behind a few stragglers, while chunks stay long enough that the first
// rows of a chunk (which update the top-2 state often) are a small share.
static int bf_chunks(int nq, int64_t nd, int resident, int queriesPerWg, int group, long long chunkLimit,
                     long long* rowsPerChunk) {
    const long long qBlocks = (nq + queriesPerWg - 1) / queriesPerWg;
    const long long maxChunks = std::max<long long>(1, std::min<long long>(std::min<long long>(nd / 8192, 4096), chunkLimit));
    long long best = 1;
    double bestEff = -1.0;
    if (resident < 1) resident = 1;
    for (int k = 1; k <= 8; k++) {
        long long chunks = ((long long)k * resident) / qBlocks;
        chunks = std::max<long long>(1, std::min(chunks, maxChunks));
        const long long wgs = chunks * qBlocks;
        const long long rounds = (wgs + resident - 1) / resident;
        const double eff = (double)wgs / ((double)rounds * resident);
        if (eff > bestEff + 0.02) {
            bestEff = eff;
            best = chunks;
        }
        if (eff >= 0.97) break;
    }
    long long rpc = (nd + best - 1) / best;
    rpc = (rpc + group - 1) / group * group;
    if (rpc < group) rpc = group;
    *rowsPerChunk = rpc;
    return (int)std::max<long long>(1, (nd + rpc - 1) / rpc);
}

// 64-bit words of scratch a scan of nq queries in `chunks` chunks uses: the thresholds the chunks of a query
// publish to each other (one 32-bit word per query, at the head) and the per-chunk partial results.
static size_t bf_threshold_words(int nq) { return ((size_t)nq + 1) / 2; }
static size_t bf_scratch_words(int nq, int chunks) {
    return bf_threshold_words(nq) + (size_t)BF_PARTS_PER_CHUNK * chunks * nq;
}

// Scan + per-chunk merge into d_packed; `scratch` has room for bf_scratch_words(nq, chunks).
static int bf_scan_into(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, unsigned long long* scratch,
                        int chunks, long long rpc, hipStream_t st) {
    const int qBlocks = (nq + BFM_QUERIES_PER_WG - 1) / BFM_QUERIES_PER_WG;
    int parts = BF_PARTS_PER_CHUNK * chunks;
    unsigned long long* partial = scratch + bf_threshold_words(nq);
    if (bf->n > 0) {
        // the published thresholds start at "nothing yet" (a single chunk has nobody to tell: no words, no clearing);
        // every partial word is written before the merge reads it
        uint32_t* shared = chunks > 1 ? (uint32_t*)scratch : nullptr;
        if (shared) HIP_TRY(hipMemsetAsync(shared, 0, bf_threshold_words(nq) * 8, st));
        hipLaunchKernelGGL(k_bf_mfma, dim3(qBlocks, chunks), dim3(256), BFM_LDS, st, (const uint32_t*)d_q, nq,
                           (const uint32_t*)bf->d_db, (long long)bf->n, rpc, (long long)bf->globalBase, partial, shared, bf->d_clock);
    } else {
        HIP_TRY(hipMemsetAsync(partial, 0xFF, (size_t)nq * 8, st));
        parts = 1;
    }
    hipLaunchKernelGGL(k_bf_merge, dim3((nq + 255) / 256), dim3(256), 0, st, (const unsigned long long*)partial, parts,
                       nq, (unsigned long long*)d_packed, (int32_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("brute-force launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_bf_clock_probe(orb_bf* bf, int enable, double* mhz) {
    if (!bf) {
        orb_set_error("orb_bf_clock_probe: bad arguments");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(bf->device));
    HIP_TRY(hipDeviceSynchronize());
    if (mhz) {
        *mhz = 0.0;
        if (bf->d_clock) {
            unsigned long long v[2] = {0, 0};
            HIP_TRY(hipMemcpy(v, bf->d_clock, 16, hipMemcpyDeviceToHost));
            if (v[1]) *mhz = (double)v[0] / (double)v[1] * 100.0;  // the constant counter runs at 100 MHz
        }
    }
    if (enable && !bf->d_clock) HIP_TRY(hipMalloc((void**)&bf->d_clock, 16));
    if (!enable && bf->d_clock) {
        (void)hipFree(bf->d_clock);
        bf->d_clock = nullptr;
    }
    if (bf->d_clock) HIP_TRY(hipMemset(bf->d_clock, 0, 16));
    return ORB_OK;
}

extern "C" int orb_bf_query_packed_device(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, void* stream) {
    if (!bf || nq < 0 || (nq > 0 && (!d_q || !d_packed))) {
        orb_set_error("orb_bf_query_packed_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(bf->device));
    long long rpc = BFM_TR;
    const int chunks = bf_chunks(nq, bf->n, bf->residentBlocks, BFM_QUERIES_PER_WG, BFM_TR, 4096, &rpc);
    int rc = bf_ensure(bf->d_partial, bf->partialBytes, bf_scratch_words(nq, chunks) * 8);
    if (rc) return rc;
    return bf_scan_into(bf, d_q, nq, d_packed, (unsigned long long*)bf->d_partial, chunks, rpc, (hipStream_t)stream);
}

extern "C" int orb_bf_query_packed_noalloc(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, void* stream) {
    if (!bf || nq < 0 || (nq > 0 && (!d_q || !d_packed))) {
        orb_set_error("orb_bf_query_packed_noalloc: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq > ORB_BF_NOALLOC_MAX_QUERIES) {
        orb_set_error("orb_bf_query_packed_noalloc: %d queries exceed the limit of %d per call (split the query set, or use "
                      "orb_bf_query_packed_device, which allocates)", nq, ORB_BF_NOALLOC_MAX_QUERIES);
        return ORB_ERR_CAPACITY;
    }
    if (nq == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(bf->device));
    // as many chunks as the fixed block has room for beside the threshold words: ORB_BF_NOALLOC_MAX_QUERIES
    // queries still get 7
    long long rpc = BFM_TR;
    const int chunks = bf_chunks(nq, bf->n, bf->residentBlocks, BFM_QUERIES_PER_WG, BFM_TR,
                                 (long long)((BF_STATIC_WORDS - bf_threshold_words(nq)) / BF_PARTS_PER_CHUNK / (unsigned)nq), &rpc);
    // the threshold words are cleared on the stream before the scan (no allocation); every (chunk, query) word is
    // written by the scan before the merge reads it
    return bf_scan_into(bf, d_q, nq, d_packed, bf->d_static, chunks, rpc, (hipStream_t)stream);
}

extern "C" int orb_bf_merge_packed_device(int device, const uint64_t* d_gathered, int nshards, int nq,
                                          int32_t* d_best_idx, uint16_t* d_best_dist, uint16_t* d_second_dist,
                                          void* stream) {
    if (nshards < 1 || nq < 0 || (nq > 0 && !d_gathered)) {
        orb_set_error("orb_bf_merge_packed_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_bf_merge, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)d_gathered, nshards, nq, (unsigned long long*)nullptr, d_best_idx,
                       d_best_dist, d_second_dist);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("merge launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_bf_query_top2(orb_bf* bf, const uint8_t* q, int nq, int32_t* best_idx, uint16_t* best_dist,
                                 uint16_t* second_dist) {
    if (!bf || nq < 0 || (nq > 0 && !q)) {
        orb_set_error("orb_bf_query_top2: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq == 0) return ORB_OK;
    if (bf->globalBase + bf->n > 0x7FFFFFFFll) {
        orb_set_error("orb_bf_query_top2 returns row numbers as int32_t (-1 = none): rows up to %lld do not fit; the packed entry points carry 32 bits",
                      (long long)(bf->globalBase + bf->n - 1));
        return ORB_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipSetDevice(bf->device));
    int rc = bf_ensure(bf->d_q, bf->qBytes, (size_t)nq * 32);
    if (!rc) rc = bf_ensure(bf->d_packed, bf->packedBytes, (size_t)nq * 8);
    if (!rc) rc = bf_ensure(bf->d_out, bf->outBytes, (size_t)nq * 8);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(bf->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, bf->stream));
    rc = orb_bf_query_packed_device(bf, (const uint8_t*)bf->d_q, nq, (uint64_t*)bf->d_packed, bf->stream);
    if (rc) return rc;
    int32_t* dIdx = (int32_t*)bf->d_out;
    uint16_t* dB = (uint16_t*)((uint8_t*)bf->d_out + (size_t)nq * 4);
    uint16_t* dS = dB + nq;
    rc = orb_bf_merge_packed_device(bf->device, (const uint64_t*)bf->d_packed, 1, nq, dIdx, dB, dS, bf->stream);
    if (rc) return rc;
    if (best_idx) HIP_TRY(hipMemcpyAsync(best_idx, dIdx, (size_t)nq * 4, hipMemcpyDeviceToHost, bf->stream));
    if (best_dist) HIP_TRY(hipMemcpyAsync(best_dist, dB, (size_t)nq * 2, hipMemcpyDeviceToHost, bf->stream));
    if (second_dist) HIP_TRY(hipMemcpyAsync(second_dist, dS, (size_t)nq * 2, hipMemcpyDeviceToHost, bf->stream));
    HIP_TRY(hipStreamSynchronize(bf->stream));
    return ORB_OK;
}
