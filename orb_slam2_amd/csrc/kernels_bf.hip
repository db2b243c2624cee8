// Brute-force Hamming top-2 over a descriptor database shard for gfx950.
//
// The unit of work is ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:460-476):
// popcount of a 256-bit XOR, i.e. 8 x (v_xor_b32 + accumulating v_bcnt_u32_b32).
// The scan applies the best / second-best rule of src/ORBmatcher.cc:53-76 to
// every database row in index order.
//
// Mapping: one lane holds QPL query descriptors in registers (8 VGPRs each).
// A workgroup copies TR database rows per tile into LDS with one coalesced
// 16-byte load per thread (two tiles in flight) and every wavefront reads the
// rows back as broadcast ds_read_b128 (all lanes the same address, no bank
// conflict), so a comparison is VGPR x VGPR and the row traffic rides the
// vector memory path.  (Rows streamed through scalar loads instead stall on the
// scalar data cache, which sustains about one 64-byte miss per ~130 cycles per
// CU.)  Rows are taken GROUP at a time; the top-2 state (best distance, best
// row, second distance per query) is only touched when some lane of the wave
// saw a distance below its current second best, which for all but the first few
// thousand rows of a chunk is rare.  The database is cut into chunks along
// blockIdx.y so that a shard fills the chip; per-chunk results are packed as
// best << 48 | second << 32 | global row and combined by the same merge kernel
// that combines the shards of several GPUs.
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


struct Row256 {  // one database row, the same for every lane of the wavefront
    uint4 lo, hi;
};

static __device__ __forceinline__ Row256 load_row(const uint32_t* __restrict__ p) {
    Row256 r;
    r.lo = ((const uint4*)p)[0];
    r.hi = ((const uint4*)p)[1];
    return r;
}

// popcount of the 256-bit XOR of a query (a) and a row (r): 8 x (v_xor_b32 + accumulating v_bcnt_u32_b32).
// One asm statement, for two reasons.  The compiler otherwise splits the accumulation into bcnt(x, 0) plus
// separate adds.  And the scalar no-op between a v_xor and the v_bcnt that reads its result matters: v_xor_b32
// is a double-rate instruction on gfx950 (a wavefront's 64 lanes in about 2 cycles), v_bcnt_u32_b32 a
// single-rate one (4 cycles), yet a wavefront that issues the v_bcnt straight after the v_xor costs the SIMD
// about 7.8 cycles per pair, whatever the distance between the two; with one scalar instruction after the
// v_xor the pair takes about 6.6, and a further one after the v_bcnt makes it 8.3 (tools/valu_peak.hip measures
// these orders).  Between separate asm statements the compiler adds such a no-op of its own, hence one statement.
// PATTERN selects the order (a tuning parameter; the product uses BF_PATTERN): 0 = no scalar instruction,
// 1 = s_nop 0 after every v_xor, 2 = s_nop 1 after every v_xor, 3 = every v_xor issued one pair ahead of
// the v_bcnt that reads it, s_nop 0 between.
#define BF_X(k, K) "v_xor_b32 %1, %" #k ", %" #K "\n\t"
#define BF_X2(k, K) "v_xor_b32 %2, %" #k ", %" #K "\n\t"
#define BF_B0 "v_bcnt_u32_b32 %0, %1, 0\n\t"
#define BF_B "v_bcnt_u32_b32 %0, %1, %0\n\t"
#define BF_B2 "v_bcnt_u32_b32 %0, %2, %0\n\t"
#define BF_N0 "s_nop 0\n\t"
#define BF_N1 "s_nop 1\n\t"
#define BF_INS(a, r)                                                                                                  \
    "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(r.lo.x), "v"(r.lo.y), \
        "v"(r.lo.z), "v"(r.lo.w), "v"(r.hi.x), "v"(r.hi.y), "v"(r.hi.z), "v"(r.hi.w)
template <int PATTERN>
static __device__ __forceinline__ uint32_t hamming256(const uint32_t (&a)[8], const Row256& r) {
    uint32_t d, t, u;
    if (PATTERN == 0) {
        asm(BF_X(3, 11) BF_B0 BF_X(4, 12) BF_B BF_X(5, 13) BF_B BF_X(6, 14) BF_B BF_X(7, 15) BF_B BF_X(8, 16) BF_B BF_X(9, 17)
                BF_B BF_X(10, 18) BF_B
            : "=&v"(d), "=&v"(t), "=&v"(u)
            : BF_INS(a, r));
    } else if (PATTERN == 1) {
        asm(BF_X(3, 11) BF_N0 BF_B0 BF_X(4, 12) BF_N0 BF_B BF_X(5, 13) BF_N0 BF_B BF_X(6, 14) BF_N0 BF_B BF_X(7, 15) BF_N0 BF_B
                BF_X(8, 16) BF_N0 BF_B BF_X(9, 17) BF_N0 BF_B BF_X(10, 18) BF_N0 BF_B
            : "=&v"(d), "=&v"(t), "=&v"(u)
            : BF_INS(a, r));
    } else if (PATTERN == 2) {
        asm(BF_X(3, 11) BF_N1 BF_B0 BF_X(4, 12) BF_N1 BF_B BF_X(5, 13) BF_N1 BF_B BF_X(6, 14) BF_N1 BF_B BF_X(7, 15) BF_N1 BF_B
                BF_X(8, 16) BF_N1 BF_B BF_X(9, 17) BF_N1 BF_B BF_X(10, 18) BF_N1 BF_B
            : "=&v"(d), "=&v"(t), "=&v"(u)
            : BF_INS(a, r));
    } else {
        asm(BF_X(3, 11) BF_X2(4, 12) BF_N0 BF_B0 BF_X(5, 13) BF_N0 BF_B2 BF_X2(6, 14) BF_N0 BF_B BF_X(7, 15) BF_N0 BF_B2
                BF_X2(8, 16) BF_N0 BF_B BF_X(9, 17) BF_N0 BF_B2 BF_X2(10, 18) BF_N0 BF_B BF_N0 BF_B2
            : "=&v"(d), "=&v"(t), "=&v"(u)
            : BF_INS(a, r));
    }
    return d;
}

static __device__ __forceinline__ void top2_update(uint32_t d, uint32_t row, uint32_t& bd, uint32_t& bi, uint32_t& sd) {
    // strict '<' in scan order: an equal distance never replaces the best, but becomes the second best
    if (d < sd) {
        if (d < bd) {
            sd = bd;
            bd = d;
            bi = row;
        } else {
            sd = d;
        }
    }
}

// One group of GROUP rows against the lane's QPL queries.
template <int QPL, int GROUP, int PATTERN>
static __device__ __forceinline__ void bf_group(const uint32_t (&qv)[QPL][8], const Row256 (&rw)[GROUP], uint32_t rowBase,
                                                uint32_t (&bd)[QPL], uint32_t (&bi)[QPL], uint32_t (&sd)[QPL]) {
    uint32_t d[GROUP][QPL];
    bool any = false;
#pragma unroll
    for (int g = 0; g < GROUP; g++) {
#pragma unroll
        for (int j = 0; j < QPL; j++) {
            d[g][j] = hamming256<PATTERN>(qv[j], rw[g]);
        }
    }
    // one compare per query against the smallest distance of the group
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        uint32_t m = d[0][j];
#pragma unroll
        for (int g = 1; g < GROUP; g++) m = min(m, d[g][j]);
        any |= m < sd[j];
    }
    if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
        for (int g = 0; g < GROUP; g++) {
#pragma unroll
            for (int j = 0; j < QPL; j++) top2_update(d[g][j], rowBase + g, bd[j], bi[j], sd[j]);
        }
    }
}

template <int QPL, int GROUP, int TR, int PATTERN>
__global__ __launch_bounds__(256) void k_bf_top2_lds(const uint32_t* __restrict__ q, int nq,
                                                     const uint32_t* __restrict__ db, long long nd,
                                                     long long rowsPerChunk, long long globalBase,
                                                     unsigned long long* __restrict__ partial) {
    static_assert(TR * 32 % (256 * 16) == 0, "tile must be a whole number of 16-byte loads per thread");
    constexpr int LPT = TR * 32 / (256 * 16);  // 16-byte loads per thread and tile
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[2][TR * 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long qb = ((long long)blockIdx.x * 4 + wave) * (64 * QPL);
    const bool active = qb < nq;   // inactive waves still help loading tiles and hit every barrier
    if (!active) qb = 0;
    uint32_t qv[QPL][8];
    uint32_t bd[QPL], sd[QPL], bi[QPL];
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        long long qi = qb + j * 64 + lane;
        if (qi >= nq) qi = nq - 1;
        const uint4* p = (const uint4*)(q + qi * 8);
        const uint4 lo = p[0], hi = p[1];
        qv[j][0] = lo.x; qv[j][1] = lo.y; qv[j][2] = lo.z; qv[j][3] = lo.w;
        qv[j][4] = hi.x; qv[j][5] = hi.y; qv[j][6] = hi.z; qv[j][7] = hi.w;
        bd[j] = BF_NONE;
        sd[j] = BF_NONE;
        bi[j] = 0xFFFFFFFFu;
    }
    const long long r0 = (long long)blockIdx.y * rowsPerChunk;
    const long long r1 = (r0 + rowsPerChunk < nd) ? r0 + rowsPerChunk : nd;
    const long long lastByte = nd * 32 - 16;  // clamp tile loads that run past the database
    const uint8_t* dbb = (const uint8_t*)db;
    uint4 stage[LPT];
    auto fetch = [&](long long tileRow) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            long long off = tileRow * 32 + ((long long)k * 256 + threadIdx.x) * 16;
            if (off > lastByte) off = lastByte;
            stage[k] = *(const uint4*)(dbb + off);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LPT; k++) ((uint4*)s_rows[buf])[k * 256 + threadIdx.x] = stage[k];
    };
    if (r0 < r1) {
        fetch(r0);
        stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (long long t = r0; t < r1; t += TR) {
        const bool more = t + TR < r1;
        if (more) fetch(t + TR);
        const int rowsHere = (int)((r1 - t < TR) ? r1 - t : TR);
        if (active) {
            const uint32_t* tile = s_rows[buf];
            int r = 0;
            for (; r + GROUP <= rowsHere; r += GROUP) {
                Row256 rw[GROUP];
#pragma unroll
                for (int g = 0; g < GROUP; g++) rw[g] = load_row(tile + (r + g) * 8);
                bf_group<QPL, GROUP, PATTERN>(qv, rw, (uint32_t)(t - r0) + r, bd, bi, sd);
            }
            for (; r < rowsHere; r++) {
                const Row256 row = load_row(tile + r * 8);
#pragma unroll
                for (int j = 0; j < QPL; j++)
                    top2_update(hamming256<PATTERN>(qv[j], row), (uint32_t)(t - r0) + r, bd[j], bi[j], sd[j]);
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!active) return;
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        const long long qi = qb + j * 64 + lane;
        if (qi < nq) {
            const unsigned long long idx =
                (bd[j] == BF_NONE) ? 0xFFFFFFFFull : (unsigned long long)(globalBase + r0 + bi[j]) & 0xFFFFFFFFull;
            partial[(long long)blockIdx.y * nq + qi] =
                ((unsigned long long)bd[j] << 48) | ((unsigned long long)sd[j] << 32) | idx;
        }
    }
}

// Tuning of the scan kernel (measured over 2..8 queries per lane, groups of 2..8 rows, tiles of 128 / 256 rows).
#define BF_QPL 2
#define BF_GROUP 4
#define BF_TILE_ROWS 128
#define BF_PATTERN 1
typedef void (*bf_kernel_t)(const uint32_t*, int, const uint32_t*, long long, long long, long long, unsigned long long*);
static const bf_kernel_t k_bf_scan = k_bf_top2_lds<BF_QPL, BF_GROUP, BF_TILE_ROWS, BF_PATTERN>;

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
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_bf_scan, 256, 0));
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
static int bf_chunks(int nq, int64_t nd, int resident, int qpl, int group, long long chunkLimit,
                     long long* rowsPerChunk) {
    const long long qBlocks = (nq + 256 * qpl - 1) / (256 * qpl);
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

// Scan + per-chunk merge into d_packed, partials in `partial` (room for chunkLimit * nq words).
static int bf_scan_into(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, unsigned long long* partial,
                        int chunks, long long rpc, hipStream_t st) {
    const int qBlocks = (nq + 256 * BF_QPL - 1) / (256 * BF_QPL);
    if (bf->n > 0) {
        hipLaunchKernelGGL(k_bf_scan, dim3(qBlocks, chunks), dim3(256), 0, st, (const uint32_t*)d_q, nq,
                           (const uint32_t*)bf->d_db, (long long)bf->n, rpc, (long long)bf->globalBase, partial);
    } else {
        HIP_TRY(hipMemsetAsync(partial, 0xFF, (size_t)nq * 8, st));
    }
    hipLaunchKernelGGL(k_bf_merge, dim3((nq + 255) / 256), dim3(256), 0, st, (const unsigned long long*)partial, chunks,
                       nq, (unsigned long long*)d_packed, (int32_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("brute-force launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_bf_query_packed_device(orb_bf* bf, const uint8_t* d_q, int nq, uint64_t* d_packed, void* stream) {
    if (!bf || nq < 0 || (nq > 0 && (!d_q || !d_packed))) {
        orb_set_error("orb_bf_query_packed_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(bf->device));
    long long rpc = BF_GROUP;
    const int chunks = bf_chunks(nq, bf->n, bf->residentBlocks, BF_QPL, BF_GROUP, 4096, &rpc);
    int rc = bf_ensure(bf->d_partial, bf->partialBytes, (size_t)chunks * nq * 8);
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
    // as many chunks as the fixed block has room for: ORB_BF_NOALLOC_MAX_QUERIES queries still get 16
    long long rpc = BF_GROUP;
    const int chunks = bf_chunks(nq, bf->n, bf->residentBlocks, BF_QPL, BF_GROUP, (long long)(BF_STATIC_WORDS / (unsigned)nq), &rpc);
    // every (chunk, query) word is written by the scan before the merge reads it: nothing to clear between calls
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
