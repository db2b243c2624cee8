// Development prototype (FP4 form): brute-force Hamming top-2 on the matrix cores.  Query bits become the FP4 values
// +1 / -1, database bits 1 / 0, so that the dot product of a row and a query is popcount(query) - distance;
// v_mfma_scale_f32_32x32x64_f8f6f4 makes 32 rows x 32 queries x 64 bits per instruction.  Stand-alone: generates
// data, runs the kernel, checks a sample of queries against a CPU scan, prints Gcmp/s, the engine clock during the
// launch (clock64 / wall_clock64) and the launch in engine cycles.
//   tools/bf_mfma4 <rows> <queries> <repetitions> [workgroups to aim for]
// Two kernels.  k_bf_mfma: every row block's products are filtered right after its matrix instructions (the
// library's kernel until the interleaved one).  Switches for measurements: -DNO_EXPAND / -DNO_EPI drop the row
// expansion / the top-2 filter (shares of the run time), -DBARE also the LDS operand reads and the barrier (the
// matrix instructions alone), -DPREX scans a database expanded once in global memory, -DPD=n lets the global loads
// run n tiles ahead, -DWIDE accumulates all row blocks of a tile before any is looked at, -DUSE_LUT expands
// through a table in LDS, -DPERM expands through v_perm_b32 (the library's form).  None of PREX, PD, WIDE, USE_LUT
// changed the rate by more than 2 %.
// k_bf_pipe (-DPIPE): the library's kernel -- filter and expansion issued between the matrix instructions, list
// updates per query block, thresholds shared by the two lanes of a query and, with -DGTHR=n (refresh every n tiles),
// by the chunks of a query.  -DP_NOFILT / -DP_NOEXP / -DP_NOSLOW take the filter, the expansion, the list updates
// out (timing only: results are wrong), -DTOURN updates the lists through a tournament on keys (no gain: it spills).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -mllvm -disable-machine-sink -DPERM -DPIPE -DGTHR=32
//        tools/bf_mfma4.hip -o tools/bf_mfma4
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef QB
#define QB 4  // blocks of 32 queries per wavefront
#endif
#ifndef TR
#define TR 64  // database rows per tile
#endif
#define ROWP 144  // bytes per expanded row in LDS (256 + 16: rows 16 apart meet on the same banks, not rows 1 apart)
#define SENT (-100000.f)

// 32 descriptor bits -> 32 FP4 (E2M1) values, 16 bytes.  Database rows: 1.0 (0x2) for a set bit, 0.0 for a clear one.
// Queries: +1.0 (0x2) for a set bit, -1.0 (0xA) for a clear one.
template <bool QUERY>
static __device__ __forceinline__ v4i expand32(uint32_t bits) {
    v4i o;
#pragma unroll
    for (int n = 0; n < 4; n++) {
#ifdef PERM
        // two bits -> one byte of two nibbles through v_perm_b32: the byte's four bit pairs are brought to the low bits
        // of four bytes (shifted copies ORed together: no kept bit is hit twice), and each selects a byte of the
        // table {0x00, 0x02, 0x20, 0x22}
        uint32_t m = (bits >> (8 * n)) & 255u;
        m |= m << 6;
        m |= m << 12;
        m = __builtin_amdgcn_perm(0u, 0x22200200u, m & 0x03030303u);
#else
        // the byte's bit i -> bit 1 of nibble i: halves, pairs, single bits apart, each step one shift-or and one mask
        uint32_t m = ((bits >> (8 * n)) & 255u) << 1;
        m = (m | (m << 12)) & 0x001E001Eu;
        m = (m | (m << 6)) & 0x06060606u;
        m = (m | (m << 3)) & 0x22222222u;
#endif
        if (QUERY) m = 0x22222222u | ((m ^ 0x22222222u) << 2);  // clear bit: sign bit (0x8) on top of 1.0
        o[n] = (int)m;
    }
    return o;
}

__global__ void k_expand(const uint32_t* __restrict__ db, long long n, v4i* __restrict__ dbx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // one word of the database per thread
    if (i < n * 8) dbx[i] = expand32<false>(db[i]);
}

__global__ __launch_bounds__(256, 2) void k_bf_mfma(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ db,
                                                 long long n, long long rowsPerChunk, long long globalBase,
                                                 unsigned long long* __restrict__ partial,
                                                 unsigned long long* __restrict__ clk) {
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rows[];  // two buffers of TR expanded rows
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, c = lane & 31;
    const long long q0 = (long long)blockIdx.x * (4 * QB * 32) + w * (QB * 32);
    const long long r0 = (long long)blockIdx.y * rowsPerChunk;
    const long long rEnd = std::min(n, r0 + rowsPerChunk);
    const int rows = (int)(rEnd - r0);
    if (rows <= 0) return;  // (whole workgroup)

    // the wavefront's queries as B operands: lane (c, h) holds bits [32 ks + 16 h, + 16) of query c of each block
    v4i bq[QB][4];
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const uint32_t word = qi < nq ? q[qi * 8 + 2 * ks + h] : 0u;
            bq[b][ks] = expand32<true>(word);
        }
    }
    int pcq[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
        pcq[b] = 0;
        if (qi < nq)
            for (int ks = 0; ks < 8; ks++) pcq[b] += __popc(q[qi * 8 + ks]);
    }
    float best[QB], sec[QB];
    int bidx[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) best[b] = sec[b] = SENT, bidx[b] = 0;

    // staging: thread = (row t / 4, word pair t % 4), TR / 64 rows per thread and tile
    const int srow = t >> 2, swp = t & 3;
    constexpr int SP = TR / 64;
#ifdef PREX
    struct Words { v4i w[SP][2]; };
#else
    struct Words { uint2 w[SP]; };
#endif
    auto fetch = [&](int tile) -> Words {
        Words o;
#pragma unroll
        for (int p = 0; p < SP; p++) {
            const long long r = r0 + (long long)tile * TR + srow + 64 * p;
#ifdef PREX
            const v4i z = {0, 0, 0, 0};
            const v4i* src = (const v4i*)db + r * 8 + 2 * swp;  // (db is the expanded database here: 8 x 16 bytes per row)
            o.w[p][0] = r < rEnd ? src[0] : z;
            o.w[p][1] = r < rEnd ? src[1] : z;
#else
            o.w[p] = r < rEnd ? *(const uint2*)(db + r * 8 + 2 * swp) : uint2{0u, 0u};
#endif
        }
        return o;
    };
    auto stash = [&](int buf, int tile, const Words& wds) {
#pragma unroll
        for (int p = 0; p < SP; p++) {
            uint8_t* d = s_rows + buf * (TR * ROWP) + (srow + 64 * p) * ROWP + swp * 32;
            const bool valid = r0 + (long long)tile * TR + srow + 64 * p < rEnd;
#ifdef PREX
            *(v4i*)(d + 0) = wds.w[p][0];
            *(v4i*)(d + 16) = wds.w[p][1];
#elif defined(USE_LUT)
            const uint32_t* lut = (const uint32_t*)(s_rows + 2 * TR * ROWP);
            const uint32_t wx = valid ? wds.w[p].x : 0u, wy = valid ? wds.w[p].y : 0u;
            *(v4i*)(d + 0) = v4i{(int)lut[wx & 255u], (int)lut[(wx >> 8) & 255u], (int)lut[(wx >> 16) & 255u], (int)lut[wx >> 24]};
            *(v4i*)(d + 16) = v4i{(int)lut[wy & 255u], (int)lut[(wy >> 8) & 255u], (int)lut[(wy >> 16) & 255u], (int)lut[wy >> 24]};
#else
            *(v4i*)(d + 0) = expand32<false>(valid ? wds.w[p].x : 0u);
            *(v4i*)(d + 16) = expand32<false>(valid ? wds.w[p].y : 0u);
#endif
        }
    };
#ifdef USE_LUT
    // byte value -> its eight bits as FP4 nibbles
    uint32_t* lutw = (uint32_t*)(s_rows + 2 * TR * ROWP);
    {
        uint32_t m = (uint32_t)t << 1;
        m = (m | (m << 12)) & 0x001E001Eu;
        m = (m | (m << 6)) & 0x06060606u;
        m = (m | (m << 3)) & 0x22222222u;
        lutw[t] = m;
    }
    __syncthreads();
#endif
    const int ntiles = (rows + TR - 1) / TR;
#ifndef PD
#define PD 1  // tiles the global loads run ahead of the tile being computed
#endif
    stash(0, 0, fetch(0));
    Words ring[PD];
#pragma unroll
    for (int k = 0; k < PD; k++) ring[k] = fetch(1 + k);  // (past the end: zero words, never stashed)
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        const bool more = tile + 1 < ntiles;
        const Words nextw = ring[0];  // tile + 1, requested PD tiles ago
#pragma unroll
        for (int k = 0; k + 1 < PD; k++) ring[k] = ring[k + 1];
        ring[PD - 1] = fetch(tile + 1 + PD);
        const uint8_t* base = s_rows + buf * (TR * ROWP) + c * ROWP + h * 16;
#ifdef WIDE
        constexpr int NRB = TR / 32;
        v16f accw[NRB][QB];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int rb = 0; rb < NRB; rb++) {
                const v4i a4 = *(const v4i*)(base + rb * 32 * ROWP + ks * 32);
                const v8i a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
#pragma unroll
                for (int b = 0; b < QB; b++) {
                    const v8i bb = {bq[b][ks][0], bq[b][ks][1], bq[b][ks][2], bq[b][ks][3], 0, 0, 0, 0};
                    if (ks == 0) {
                        const v16f zero = {};
                        accw[rb][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    } else {
                        accw[rb][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, accw[rb][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    }
                }
            }
        }
#endif
#pragma unroll
        for (int rb = 0; rb < TR / 32; rb++) {
#ifdef WIDE
            v16f (&acc)[QB] = accw[rb];
#else
            v16f acc[QB];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
#ifdef BARE
                v4i a4 = bq[0][ks];  // (no LDS read: the matrix instructions alone)
                a4[0] ^= tile << 16;  // (differs per tile, or the loop's products would be computed once)
#else
                const v4i a4 = *(const v4i*)(base + rb * 32 * ROWP + ks * 32);
#endif
                const v8i a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
#pragma unroll
                for (int b = 0; b < QB; b++) {
                    const v8i bb = {bq[b][ks][0], bq[b][ks][1], bq[b][ks][2], bq[b][ks][3], 0, 0, 0, 0};
                    if (ks == 0) {
                        const v16f zero = {};
                        acc[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    } else {
                        acc[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc[b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    }
                }
            }
#endif
            const int rowBase = tile * TR + rb * 32 + 4 * h;
#pragma unroll
            for (int b = 0; b < QB; b++) {
                const v16f s = acc[b];
                // (three-input maxima: eight instructions for the sixteen values when the file is built with
                // -fno-honor-nans; otherwise every accumulator first goes through a canonicalising v_max_f32 x, x)
#define MAX3(a, b, c) fmaxf(fmaxf(a, b), c)
                const float m0 = MAX3(s[0], s[1], s[2]), m1 = MAX3(s[3], s[4], s[5]), m2 = MAX3(s[6], s[7], s[8]);
                const float m3 = MAX3(s[9], s[10], s[11]), m4 = MAX3(s[12], s[13], s[14]);
                const float m = MAX3(MAX3(m0, m1, m2), MAX3(m3, m4, s[15]), m0);
#undef MAX3
#ifdef NO_EPI
                if (m == 12345) best[b] = m;
                if (false) {
#else
                if (__ballot(m > sec[b]) != 0ull) {
#endif
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int r = rowBase + (i & 3) + 8 * (i >> 2);
                        const float v = s[i];
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
                }
            }
        }
#ifndef NO_EXPAND
        if (more) stash(buf ^ 1, tile + 1, nextw);
#else
        if (more && tile == 0) stash(buf ^ 1, tile + 1, nextw);
#endif
#ifndef BARE
        __syncthreads();
#endif
    }
    if (t == 0) {  // engine clock while the kernel ran: clock64() ticks per 100 MHz wall-clock tick
        atomicAdd(&clk[0], clock64() - clk0);
        atomicAdd(&clk[1], wall_clock64() - wall0);
    }
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
        if (qi < nq) {
            const unsigned long long bd = best[b] == SENT ? 0xFFFFull : (unsigned long long)(pcq[b] - (int)best[b]);
            const unsigned long long sd = sec[b] == SENT ? 0xFFFFull : (unsigned long long)(pcq[b] - (int)sec[b]);
            const unsigned long long idx =
                best[b] == SENT ? 0xFFFFFFFFull : (unsigned long long)(globalBase + r0 + bidx[b]) & 0xFFFFFFFFull;
            partial[((long long)blockIdx.y * 2 + h) * nq + qi] = (bd << 48) | (sd << 32) | idx;
        }
    }
}

// -DPIPE: the same scan with the vector-ALU work placed in the shadow of the matrix instructions.  A matrix
// instruction keeps the SIMD's issue port for about half of its 32 cycles and leaves room for four vector
// instructions beside it (tools/mfma_valu_overlap.hip), but only if they are independent and stand next to it in the
// instruction stream: a wavefront issues in order.  So the tile's two row blocks use two sets of accumulators, and
// while the 16 matrix instructions of one block run, the top-2 filter of the other block's finished products and the
// expansion of the next tile's rows are issued between them, two to four at a time (scheduling barriers pin the order);
// the filter of all four query blocks ends in one branch per row block instead of four.
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifndef VPM
#define VPM 4  // vector instructions placed behind each matrix instruction
#endif
__global__ __launch_bounds__(256, 2) void k_bf_pipe(const uint32_t* __restrict__ q, int nq, const uint32_t* __restrict__ db,
                                                 long long n, long long rowsPerChunk, long long globalBase,
                                                 unsigned long long* __restrict__ partial,
                                                 unsigned long long* __restrict__ clk) {
    static_assert(QB == 4 && TR == 64, "the interleaving below is written for four query blocks and two row blocks");
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
    uint32_t* gthr = (uint32_t*)(clk + 2);  // one word per query, zero before the launch
    extern __shared__ __attribute__((aligned(16))) uint8_t s_rows[];  // two buffers of TR expanded rows
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, c = lane & 31;
    const long long q0 = (long long)blockIdx.x * (4 * QB * 32) + w * (QB * 32);
    const long long r0 = (long long)blockIdx.y * rowsPerChunk;
    const long long rEnd = std::min(n, r0 + rowsPerChunk);
    const int rows = (int)(rEnd - r0);
    if (rows <= 0) return;  // (whole workgroup)

    v4i bq[QB][4];
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bq[b][ks] = expand32<true>(qi < nq ? q[qi * 8 + 2 * ks + h] : 0u);
    }
    int pcq[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
        pcq[b] = 0;
        if (qi < nq)
            for (int ks = 0; ks < 8; ks++) pcq[b] += __popc(q[qi * 8 + ks]);
    }
    float best[QB], sec[QB];
    int bidx[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) best[b] = sec[b] = SENT, bidx[b] = 0;

    const int srow = t >> 2, swp = t & 3;  // staging: thread = (row, word pair)
    auto fetch = [&](int tile) -> uint2 {
        const long long r = r0 + (long long)tile * TR + srow;
        return r < rEnd ? *(const uint2*)(db + r * 8 + 2 * swp) : uint2{0u, 0u};
    };
    // every register of a row block's products, one by one (rare: only when the filter saw a candidate)
    bool hitB[QB] = {true, true, true, true};  // which query blocks the last filter pass flagged
    // The two lanes of a query (rows 0-3, 8-11, .. and rows 4-7, 12-15, .. of every block) keep separate lists; the
    // filter compares with the larger of their second-best values: a product that does not beat it has two earlier
    // rows at least as good, in one list or the other, and cannot be among the query's two best.
    float thr[QB];
#pragma unroll
    for (int b = 0; b < QB; b++) thr[b] = SENT;
    auto slow = [&](const v16f* acc, int rowBase) {
#pragma unroll
        for (int b = 0; b < QB; b++) {
            if (__ballot(hitB[b]) == 0ull) continue;
#ifdef TOURN
            if (rowBase >= 0 && rowBase + 28 <= rows - 4 * 1) {  // (all 16 rows of both lanes exist: wavefront-uniform test on the block)
                // the two largest of the sixteen by a tournament on keys 16 * product + (15 - register): the larger key
                // is the larger product and, among equal products, the earlier row
                float hi[8], lo[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float k0 = fmaf(acc[b][2 * j], 16.f, (float)(15 - 2 * j)), k1 = fmaf(acc[b][2 * j + 1], 16.f, (float)(14 - 2 * j));
                    hi[j] = fmaxf(k0, k1);
                    lo[j] = fminf(k0, k1);
                }
#pragma unroll
                for (int w2 = 4; w2 >= 1; w2 >>= 1) {
#pragma unroll
                    for (int j = 0; j < w2; j++) {
                        const float h1 = hi[j], h2 = hi[j + w2];
                        hi[j] = fmaxf(h1, h2);
                        lo[j] = fmaxf(fmaxf(fminf(h1, h2), lo[j]), lo[j + w2]);
                    }
                }
                const float v1 = floorf(hi[0] * 0.0625f), v2 = floorf(lo[0] * 0.0625f);
                const int slot = 15 - (int)fmaf(v1, -16.f, hi[0]);
                if (v1 > sec[b]) {
                    if (v1 > best[b]) {
                        sec[b] = fmaxf(best[b], v2);
                        best[b] = v1;
                        bidx[b] = rowBase + (slot & 3) + 8 * (slot >> 2);
                    } else {
                        sec[b] = v1;
                    }
                }
            } else
#endif
            {
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
            }
            const float own = fmaxf(sec[b], __shfl_xor(sec[b], 32));
            thr[b] = fmaxf(thr[b], own);
#ifdef GTHR
            const long long qi = q0 + b * 32 + c;
            if (h == 0 && qi < nq && own > -1000.f) atomicMax(&gthr[qi], __float_as_uint(own + 1024.f));
#endif
        }
    };
    // 16 matrix instructions (row block at aBase -> accW) with the filter of accR and the expansion of one word of the
    // next tile between them; returns whether any lane of the wavefront has a candidate in accR
    auto half = [&](const uint8_t* aBase, v16f* accW, const v16f* accR, uint32_t word, uint8_t* dst, bool doStash) -> bool {
        v4i aK[4];  // the row block's four k-steps, all requested before the first matrix instruction
#pragma unroll
        for (int ks = 0; ks < 4; ks++) aK[ks] = *(const v4i*)(aBase + ks * 32);
        float mm[5], mA = 0.f;
        bool hit = false;
        uint32_t ex = 0u;
        v4i o = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int ks = j >> 2, b = j & 3;
            const v4i a4 = aK[ks];
            const v8i a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
            const v8i bb = {bq[b][ks][0], bq[b][ks][1], bq[b][ks][2], bq[b][ks][3], 0, 0, 0, 0};
            if (ks == 0) {
                const v16f zero = {};
                accW[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else {
                accW[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, accW[b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
            const int eb = j >> 2, p = j & 3;  // filter: query block eb, quarter p; expansion: byte eb of the word
            const v16f s = accR[eb];
#define MAX3(a, b, c) fmaxf(fmaxf(a, b), c)
#ifndef P_NOFILT
            if (p == 0) {
                mm[0] = MAX3(s[0], s[1], s[2]);
                mm[1] = MAX3(s[3], s[4], s[5]);
            } else if (p == 1) {
                mm[2] = MAX3(s[6], s[7], s[8]);
                mm[3] = MAX3(s[9], s[10], s[11]);
            } else if (p == 2) {
                mm[4] = MAX3(s[12], s[13], s[14]);
                mA = MAX3(mm[0], mm[1], mm[2]);
            } else {
                const float m = fmaxf(mA, MAX3(mm[3], mm[4], s[15]));
#ifdef P_NOSLOW
                hitB[eb] = m > thr[eb] + 1e9f;  // (timing only: the list updates never run)
#else
                hitB[eb] = m > thr[eb];
#endif
                hit |= hitB[eb];
            }
#else
            if (j == 15) hit |= s[0] > sec[eb] + 1e9f;
#endif
#ifndef P_NOEXP
            if (p == 0) {
                ex = (word >> (8 * eb)) & 255u;
                ex |= ex << 6;
            } else if (p == 1) {
                ex |= ex << 12;
            } else if (p == 2) {
                ex &= 0x03030303u;
            } else {
                o[eb] = (int)__builtin_amdgcn_perm(0u, 0x22200200u, ex);
            }
#else
            if (j == 0) o[0] = (int)word;
#endif
#undef MAX3
        }
        // the order the scheduler is to build from the block: the operand reads, then one matrix instruction and up
        // to four vector ones in turn
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
        (void)doStash;
        *(v4i*)dst = o;  // (past the last tile: rows that are never read)
        return __ballot(hit) != 0ull;
    };

    const int ntiles = (rows + TR - 1) / TR;
    {
        const uint2 w0 = fetch(0);
        uint8_t* d = s_rows + srow * ROWP + swp * 32;
        *(v4i*)(d + 0) = expand32<false>(w0.x);
        *(v4i*)(d + 16) = expand32<false>(w0.y);
    }
    uint2 ring = fetch(1);
    __syncthreads();
    v16f acc0[QB], acc1[QB];
#pragma unroll
    for (int b = 0; b < QB; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[b][i] = SENT;  // (the first filter pass looks at these)
    for (int tile = 0; tile < ntiles; tile++) {
#ifdef GTHR
        // The chunks of a query run side by side and tell each other their second-best products (biased to be
        // positive, so that the bit patterns order like the values): a product below another chunk's second best
        // cannot be among the query's two best; one equal to it still can (it may be the earlier row), hence - 0.5.
        if ((tile & (GTHR - 1)) == GTHR - 1) {
#pragma unroll
            for (int b = 0; b < QB; b++) {
                const long long qi = q0 + b * 32 + c;
                if (qi < nq) {
                    const uint32_t g = __hip_atomic_load(&gthr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g != 0u) thr[b] = fmaxf(thr[b], __uint_as_float(g) - 1024.5f);
                }
            }
        }
#endif
        const int buf = tile & 1;
        const bool more = tile + 1 < ntiles;
        const uint2 nextw = ring;
        ring = fetch(tile + 2);
        const uint8_t* base = s_rows + buf * (TR * ROWP) + c * ROWP + h * 16;
        uint8_t* d = s_rows + (buf ^ 1) * (TR * ROWP) + srow * ROWP + swp * 32;
        SB();
        if (half(base, acc0, acc1, nextw.x, d, more)) slow(acc1, (tile - 1) * TR + 32 + 4 * h);
        SB();
        if (half(base + 32 * ROWP, acc1, acc0, nextw.y, d + 16, more)) slow(acc0, tile * TR + 4 * h);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < QB; b++) hitB[b] = true;
    slow(acc1, (ntiles - 1) * TR + 32 + 4 * h);
    if (t == 0) {
        atomicAdd(&clk[0], clock64() - clk0);
        atomicAdd(&clk[1], wall_clock64() - wall0);
    }
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const long long qi = q0 + b * 32 + c;
        if (qi < nq) {
            const unsigned long long bd = best[b] == SENT ? 0xFFFFull : (unsigned long long)(pcq[b] - (int)best[b]);
            const unsigned long long sd = sec[b] == SENT ? 0xFFFFull : (unsigned long long)(pcq[b] - (int)sec[b]);
            const unsigned long long idx =
                best[b] == SENT ? 0xFFFFFFFFull : (unsigned long long)(globalBase + r0 + bidx[b]) & 0xFFFFFFFFull;
            partial[((long long)blockIdx.y * 2 + h) * nq + qi] = (bd << 48) | (sd << 32) | idx;
        }
    }
}
#ifdef PIPE
#define SCAN_KERNEL k_bf_pipe
#else
#define SCAN_KERNEL k_bf_mfma
#endif

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int nq = argc > 2 ? atoi(argv[2]) : 20000;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<uint32_t> db((size_t)n * 8), q((size_t)nq * 8);
    for (auto& x : db) x = (uint32_t)rnd();
    for (int i = 0; i < nq; i++) {
        const long long src = (long long)(rnd() % (uint64_t)n);
        for (int k = 0; k < 8; k++) q[(size_t)i * 8 + k] = (i % 3 == 0) ? db[(size_t)src * 8 + k] : (uint32_t)rnd();
        if (i % 3 == 0) {  // a near duplicate of a database row: a few flipped bits
            const int flips = (int)(rnd() % 40);
            for (int f = 0; f < flips; f++) q[(size_t)i * 8 + (rnd() & 7)] ^= 1u << (rnd() & 31);
        }
    }
    for (int k = 0; k < 8; k++) {  // exact duplicates in the database: ties on the best row
        db[(size_t)(n - 5) * 8 + k] = db[(size_t)7 * 8 + k];
        q[k] = db[(size_t)7 * 8 + k];
    }
    uint32_t *dDb, *dQ;
    unsigned long long *dPart, *dClk;
    CHECK(hipMalloc((void**)&dClk, 16 + (size_t)nq * 4));
    CHECK(hipMalloc((void**)&dDb, db.size() * 4));
    CHECK(hipMalloc((void**)&dQ, q.size() * 4));
    CHECK(hipMemcpy(dDb, db.data(), db.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dQ, q.data(), q.size() * 4, hipMemcpyHostToDevice));
#ifdef PREX
    v4i* dDbx;
    CHECK(hipMalloc((void**)&dDbx, (size_t)n * 128));
    hipLaunchKernelGGL(k_expand, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, 0, dDb, n, dDbx);
    CHECK(hipDeviceSynchronize());
    const uint32_t* dScan = (const uint32_t*)dDbx;
#else
    const uint32_t* dScan = dDb;
#endif
    const int qGroups = (nq + 4 * QB * 32 - 1) / (4 * QB * 32);
    int chunks = std::max(1, (int)((argc > 4 ? atoi(argv[4]) : 2 * 512) / qGroups));
    long long rpc = (n + chunks - 1) / chunks;
    rpc = (rpc + TR - 1) / TR * TR;
    chunks = (int)((n + rpc - 1) / rpc);
    CHECK(hipMalloc((void**)&dPart, (size_t)2 * chunks * nq * 8));
    const size_t lds = 2 * TR * ROWP + 2048;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long clk[2] = {0, 0};
    for (int r = 0; r < reps + 1; r++) {
        CHECK(hipMemset(dClk, 0, 16 + (size_t)nq * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(SCAN_KERNEL, dim3(qGroups, chunks), dim3(256), lds, 0, dQ, nq, dScan, n, rpc, 0ll, dPart, dClk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = std::min(best, ms);
        CHECK(hipMemcpy(clk, dClk, 16, hipMemcpyDeviceToHost));
    }
    CHECK(hipGetLastError());
    const double mhz = clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0;
    printf("QB %d TR %d: %lld rows x %d queries, %d x %d workgroups: %.3f ms = %.0f Gcmp/s, clock64 at %.0f MHz (%.1f M cycles)\n", QB, TR,
           n, nq, qGroups, chunks, best, (double)n * nq / (best * 1e-3) / 1e9, mhz, best * 1e-3 * mhz);
    std::vector<unsigned long long> part((size_t)2 * chunks * nq);
    CHECK(hipMemcpy(part.data(), dPart, part.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    const int step = std::max(1, nq / 96);
    for (int i = 0; i < nq; i += step) {
        // device: merge the partial words as the library's merge kernel does
        uint32_t bd = 0xFFFF, sd = 0xFFFF, bi = 0xFFFFFFFFu;
        for (int s = 0; s < 2 * chunks; s++) {
            const unsigned long long p = part[(size_t)s * nq + i];
            const uint32_t d = (uint32_t)(p >> 48), d2 = (uint32_t)(p >> 32) & 0xFFFFu, idx = (uint32_t)p;
            if (d == 0xFFFF) continue;
            if (d < bd || (d == bd && idx < bi)) { sd = bd; bd = d; bi = idx; } else if (d < sd) sd = d;
            if (d2 < sd) sd = d2;
        }
        // host scan
        uint32_t rb = 0xFFFF, rs = 0xFFFF, ri = 0xFFFFFFFFu;
        for (long long r = 0; r < n; r++) {
            uint32_t d = 0;
            for (int k = 0; k < 8; k++) d += (uint32_t)__builtin_popcount(db[(size_t)r * 8 + k] ^ q[(size_t)i * 8 + k]);
            if (d < rb) { rs = rb; rb = d; ri = (uint32_t)r; } else if (d < rs) rs = d;
        }
        if (bd != rb || sd != rs || bi != ri) {
            if (bad < 5) printf("query %d: device (%u, %u, %u) host (%u, %u, %u)\n", i, bd, sd, bi, rb, rs, ri);
            bad++;
        }
    }
    printf("%d of %d checked queries differ\n", bad, (nq + step - 1) / step);
    return bad ? 1 : 0;
}
