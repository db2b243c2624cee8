// Micro-benchmark: sustained VALU rates on gfx950 for the instructions of the
// Hamming inner loop (8 x v_xor_b32 + 8 x accumulating v_bcnt_u32_b32 per
// 256-bit comparison), to find the ceiling the brute-force matcher is priced
// against.  Every loop keeps its operands in registers or LDS (no global
// traffic) and changes its scalar operand every step so nothing can be hoisted.
//
// The modes separate the questions a single "xor+bcnt" number cannot answer:
//   - does an instruction's rate depend on where its operands come from
//     (SGPR x VGPR, VGPR x VGPR, a literal)?
//   - does it depend on whether the result feeds the next instruction of the
//     same kind (a chain) or is consumed elsewhere?
//   - what does the exact loop of the scan kernel (rows broadcast from LDS,
//     two queries per lane) sustain without any top-2 bookkeeping?
// The engine clock during each run is reported from clock64() / wall_clock64().
// Build: python tools/gen_valu_banks.py && hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

// valu_peak_banks.inc is written by tools/gen_valu_banks.py.
#define CHAINS 8
#define STEPS 16

enum {
    M_PAIR_SV = 0,    // t = s ^ x[c] (SGPR x VGPR, result not chained); y[c] = bcnt(t) + y[c]
    M_XOR_CHAIN_SV,   // x[c] = s ^ x[c]
    M_BCNT_CHAIN,     // y[c] = bcnt(x[c]) + y[c]
    M_FMA,            // v_fma_f32
    M_PKFMA,          // v_pk_fma_f32
    M_PAIR_VV,        // t = z[c] ^ x[c] (VGPR x VGPR); y[c] = bcnt(t) + y[c]
    M_XOR_FREE_SV,    // t[c] = s ^ x[c], results never read by the loop
    M_XOR_FREE_VV,    // t[c] = z[c] ^ x[c]
    M_BCNT_FREE,      // t[c] = bcnt(x[c]) + z[c], results never read by the loop
    M_XOR_CHAIN_VV,   // x[c] = z[c] ^ x[c]
    M_LDS_PAIR,       // the scan kernel's loop: ds_read_b128 x2 of a broadcast row, 2 queries x 8 x (xor + bcnt)
    M_AND_OR,         // v_and_or_b32 chain (a three-operand integer op)
    M_ADD_CHAIN,      // v_add_u32 chain
    M_COUNT
};

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(const unsigned* __restrict__ in, unsigned* __restrict__ out, int iters,
                                              unsigned s0, unsigned long long* __restrict__ clocks) {
    __shared__ __attribute__((aligned(16))) unsigned s_rows[64 * 8];
    unsigned x[CHAINS], y[CHAINS], z[CHAINS], t[CHAINS];
    float f[CHAINS], g2[CHAINS];
    for (int c = 0; c < CHAINS; c++) {
        x[c] = in[(threadIdx.x + c * 256) & 4095];
        z[c] = in[(threadIdx.x * 3 + c * 256 + 7) & 4095];
        y[c] = 0;
        t[c] = 0;
        f[c] = (float)(x[c] & 255) * 1e-3f;
        g2[c] = f[c] + 1.f;
    }
    for (int i = threadIdx.x; i < 64 * 8; i += 256) s_rows[i] = in[i];
    __syncthreads();
    unsigned s = s0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == M_LDS_PAIR) {
            // 16 rows per iteration; x = query 0, z = query 1 (8 dwords each), y[0], y[1] the distances (summed up)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint4 lo = ((const uint4*)s_rows)[((it + r) & 63) * 2], hi = ((const uint4*)s_rows)[((it + r) & 63) * 2 + 1];
                const unsigned row[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    unsigned a, b;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a) : "v"(row[k]), "v"(x[k]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[0]) : "v"(a), "v"(y[0]));
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(b) : "v"(row[k]), "v"(z[k]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[1]) : "v"(b), "v"(y[1]));
                }
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (MODE == M_PAIR_SV) {
                    unsigned u;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u) : "s"(s), "v"(x[c]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[c]) : "v"(u), "v"(y[c]));
                } else if (MODE == M_XOR_CHAIN_SV) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[c]) : "s"(s), "v"(x[c]));
                } else if (MODE == M_BCNT_CHAIN) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[c]));
                } else if (MODE == M_FMA) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[c]) : "v"(g2[c]), "v"(f[c]));
                } else if (MODE == M_PKFMA) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a = {f[c], g2[c]};
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a));
                    f[c] = a.x;
                    g2[c] = a.y;
                } else if (MODE == M_PAIR_VV) {
                    unsigned u;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u) : "v"(z[c]), "v"(x[c]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[c]) : "v"(u), "v"(y[c]));
                } else if (MODE == M_XOR_FREE_SV) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[c]) : "s"(s), "v"(x[c]));
                } else if (MODE == M_XOR_FREE_VV) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[c]) : "v"(z[c]), "v"(x[c]));
                } else if (MODE == M_BCNT_FREE) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(t[c]) : "v"(x[c]), "v"(z[c]));
                } else if (MODE == M_XOR_CHAIN_VV) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[c]) : "v"(z[c]), "v"(x[c]));
                } else if (MODE == M_AND_OR) {
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(y[c]) : "v"(x[c]), "v"(z[c]));
                } else if (MODE == M_ADD_CHAIN) {
                    asm volatile("v_add_u32 %0, %1, %0" : "+v"(y[c]) : "v"(x[c]));
                }
            }
            s += 0x9e3779b9u;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned r = 0;
    for (int c = 0; c < CHAINS; c++) r += x[c] + y[c] + t[c] + (unsigned)f[c] + (unsigned)g2[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = w1 - w0;
    }
}

// Fixed-register instruction blocks: does the rate of the double-rate v_xor_b32 depend on which VGPR banks
// (register number mod 4) its two sources and its destination sit in, and on how soon v_bcnt reads its result?
#include "valu_peak_banks.inc"
#define FIXED_CLOBBERS                                                                                                  \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
        "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
template <int MODE>
__global__ __launch_bounds__(256) void k_fixed(const unsigned* __restrict__ in, unsigned* __restrict__ out, int iters,
                                               unsigned s0, unsigned long long* __restrict__ clocks) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) asm volatile(BLK_XOR_SRC_SAME_BANK ::: FIXED_CLOBBERS);
            if (MODE == 1) asm volatile(BLK_XOR_SRC_DIFF_BANK ::: FIXED_CLOBBERS);
            if (MODE == 2) asm volatile(BLK_XOR_SRC_DIFF_DST_SAME ::: FIXED_CLOBBERS);
            if (MODE == 3) asm volatile(BLK_PAIR_SAME_BANK ::: FIXED_CLOBBERS);
            if (MODE == 4) asm volatile(BLK_PAIR_DIFF_BANK ::: FIXED_CLOBBERS);
            if (MODE == 5) asm volatile(BLK_PAIR_DIFF_BANK_PIPELINED ::: FIXED_CLOBBERS);
            if (MODE == 6) asm volatile(BLK_PAIR_DIFF_BANK_ONE_CHAIN ::: FIXED_CLOBBERS);
            if (MODE == 7) asm volatile(BLK_PAIR_SAME_BANK_ONE_CHAIN ::: FIXED_CLOBBERS);
            if (MODE == 8) asm volatile(BLK_PAIR_NOP_BETWEEN ::: FIXED_CLOBBERS);
            if (MODE == 9) asm volatile(BLK_PAIR_NOP_EVERYWHERE ::: FIXED_CLOBBERS);
            if (MODE == 10) asm volatile(BLK_PAIR_SKEWED ::: FIXED_CLOBBERS);
            if (MODE == 11) asm volatile(BLK_PAIR_SKEWED_NOP ::: FIXED_CLOBBERS);
            if (MODE == 12) asm volatile(BLK_PAIR_2X2 ::: FIXED_CLOBBERS);
            if (MODE == 13) asm volatile(BLK_BCNT_ONLY ::: FIXED_CLOBBERS);
            if (MODE == 14) asm volatile(BLK_BCNT_NOP ::: FIXED_CLOBBERS);
            if (MODE == 15) asm volatile(BLK_XOR_NOP ::: FIXED_CLOBBERS);
            if (MODE == 16) asm volatile(BLK_PAIR_FILL_SADD ::: FIXED_CLOBBERS, "s20", "scc");
            if (MODE == 17) asm volatile(BLK_PAIR_FILL_SMOV ::: FIXED_CLOBBERS, "s20", "scc");
            if (MODE == 18) asm volatile(BLK_PAIR_FILL_SCMP ::: FIXED_CLOBBERS, "s20", "scc");
            if (MODE == 19) asm volatile(BLK_PAIR_FILL_VNOP ::: FIXED_CLOBBERS, "s20", "scc");
            if (MODE == 20) asm volatile(BLK_PAIR_FILL_SSLEEP ::: FIXED_CLOBBERS, "s20", "scc");
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned r;
    asm volatile("v_add_u32 %0, v56, v57" : "=v"(r)::FIXED_CLOBBERS);
    out[blockIdx.x * 256 + threadIdx.x] = r + in[threadIdx.x] + s0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = w1 - w0;
    }
}

template <int MODE, bool FIXED = false>
static void run(const char* name, double opsPerIter, unsigned* in, unsigned* out, int blocks, unsigned long long* d_clk,
                int wallKHz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e30f;
    unsigned long long clk[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        if (FIXED)
            hipLaunchKernelGGL(k_fixed<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 12345u, d_clk);
        else
            hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 12345u, d_clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            (void)hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        }
    }
    const double laneops = (double)blocks * 256 * iters * opsPerIter;
    const double T = laneops / best / 1e9;
    // clock64() ticks per wall_clock64() tick x wall clock rate: the rate clock64() counts at while the loop runs
    const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * wallKHz / 1e3 : 0.0;
    printf("%-44s %8.3f ms  %6.2f T lane-instr/s = %5.2f lanes/clk/SIMD at 2.4 GHz   clock64 at %.0f MHz\n", name, best, T,
           T * 1e12 / (256.0 * 4 * 2.4e9), mhz);
    fflush(stdout);
}

int main(int argc, char** argv) {
    unsigned *in, *out;
    unsigned long long* d_clk;
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8;
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_clk, 16);
    (void)hipMemset(in, 0x5a, 4096 * 4);
    int wallKHz = 0, clkKHz = 0, cus = 0;
    (void)hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
    (void)hipDeviceGetAttribute(&clkKHz, hipDeviceAttributeClockRate, 0);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("blocks=%d x 256 threads, %d CUs, clock rate attribute %d kHz, wall clock %d kHz\n", blocks, cus, clkKHz, wallKHz);
    const double per = (double)STEPS * CHAINS;
    run<M_PAIR_SV>("xor(SGPR,VGPR) free + bcnt chain", 2 * per, in, out, blocks, d_clk, wallKHz);
    run<M_PAIR_VV>("xor(VGPR,VGPR) free + bcnt chain", 2 * per, in, out, blocks, d_clk, wallKHz);
    run<M_XOR_CHAIN_SV>("xor(SGPR,VGPR) chain", per, in, out, blocks, d_clk, wallKHz);
    run<M_XOR_CHAIN_VV>("xor(VGPR,VGPR) chain", per, in, out, blocks, d_clk, wallKHz);
    run<M_XOR_FREE_SV>("xor(SGPR,VGPR) free", per, in, out, blocks, d_clk, wallKHz);
    run<M_XOR_FREE_VV>("xor(VGPR,VGPR) free", per, in, out, blocks, d_clk, wallKHz);
    run<M_BCNT_CHAIN>("bcnt chain", per, in, out, blocks, d_clk, wallKHz);
    run<M_BCNT_FREE>("bcnt free", per, in, out, blocks, d_clk, wallKHz);
    run<M_LDS_PAIR>("scan loop: LDS row, 2 queries x 8 x (xor+bcnt)", 16 * 32.0, in, out, blocks, d_clk, wallKHz);
    run<M_AND_OR>("v_and_or_b32 chain", per, in, out, blocks, d_clk, wallKHz);
    run<M_ADD_CHAIN>("v_add_u32 chain", per, in, out, blocks, d_clk, wallKHz);
    run<M_FMA>("v_fma_f32", per, in, out, blocks, d_clk, wallKHz);
    run<M_PKFMA>("v_pk_fma_f32 (2 fma per lane)", per, in, out, blocks, d_clk, wallKHz);
    const double blk = 8 * 16.0;
    run<0, true>("fixed regs: xor, sources in one bank", blk, in, out, blocks, d_clk, wallKHz);
    run<1, true>("fixed regs: xor, sources in two banks", blk, in, out, blocks, d_clk, wallKHz);
    run<2, true>("fixed regs: xor, two banks, dst = src0's bank", blk, in, out, blocks, d_clk, wallKHz);
    run<3, true>("fixed regs: xor(one bank) + bcnt, 8 chains", blk, in, out, blocks, d_clk, wallKHz);
    run<4, true>("fixed regs: xor(two banks) + bcnt, 8 chains", blk, in, out, blocks, d_clk, wallKHz);
    run<5, true>("fixed regs: 4 xor(two banks), then 4 bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<6, true>("fixed regs: xor(two banks) + bcnt, 1 chain", blk, in, out, blocks, d_clk, wallKHz);
    run<7, true>("fixed regs: xor(one bank) + bcnt, 1 chain", blk, in, out, blocks, d_clk, wallKHz);
    run<8, true>("fixed regs: xor, s_nop, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<9, true>("fixed regs: xor, s_nop, bcnt, s_nop", blk, in, out, blocks, d_clk, wallKHz);
    run<10, true>("fixed regs: xor, bcnt of the previous xor", blk, in, out, blocks, d_clk, wallKHz);
    run<11, true>("fixed regs: xor, s_nop, bcnt of the previous xor", blk, in, out, blocks, d_clk, wallKHz);
    run<12, true>("fixed regs: 2 xor, 2 bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<13, true>("fixed regs: bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<14, true>("fixed regs: bcnt, s_nop", blk, in, out, blocks, d_clk, wallKHz);
    run<15, true>("fixed regs: xor, s_nop", blk, in, out, blocks, d_clk, wallKHz);
    run<16, true>("fixed regs: xor, s_add_u32, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<17, true>("fixed regs: xor, s_mov_b32, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<18, true>("fixed regs: xor, s_cmp_lg_u32, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<19, true>("fixed regs: xor, v_nop, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    run<20, true>("fixed regs: xor, s_sleep 0, bcnt", blk, in, out, blocks, d_clk, wallKHz);
    return 0;
}
