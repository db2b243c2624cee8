// Sustained issue rate of single VALU / LDS opcodes on gfx950: a loop of blocks of 16 independent
// instructions with fixed registers (tools/gen_valu_rates.py writes the blocks), 8 wavefronts per SIMD.
// Reported as lanes per clock and SIMD at the clock the run measured: 16 = one wavefront instruction per
// 4 cycles (single rate), 32 = one per 2 cycles (double rate).
// Build: python tools/gen_valu_rates.py && hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "valu_rates.inc"
#define RATE_CLOBBERS                                                                                                   \
    "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
        "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",   \
        "s20", "s21", "vcc", "scc", "memory"

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(unsigned* __restrict__ out, int iters, unsigned long long* __restrict__ clocks) {
    __shared__ unsigned s_mem[1024];
    s_mem[threadIdx.x] = threadIdx.x;
    __syncthreads();
    asm volatile("v_lshlrev_b32 v31, 2, %0" ::"v"(threadIdx.x & 63) : "v31");
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            RATE_DISPATCH(MODE)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned r;
    asm volatile("v_add_u32 %0, v48, v49" : "=v"(r)::RATE_CLOBBERS);
    out[blockIdx.x * 256 + threadIdx.x] = r + s_mem[(threadIdx.x + 1) & 255];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = w1 - w0;
    }
}

static unsigned* d_out;
static unsigned long long* d_clk;
static int g_blocks = 2048, g_wallKHz = 100000;

template <int MODE>
static void run() {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    float best = 1e30f;
    unsigned long long clk[2] = {0, 1};
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(g_blocks), dim3(256), 0, 0, d_out, iters, d_clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            (void)hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        }
    }
    const double mhz = (double)clk[0] / (double)clk[1] * g_wallKHz / 1e3;
    const double instr = (double)g_blocks * 4 * iters * 8 * 16;  // wavefront instructions
    const double perSimdPerClk = instr * 64 / (256.0 * 4) / (best * 1e-3 * mhz * 1e6);
    printf("%-28s %8.3f ms  %5.2f lanes/clk/SIMD = %4.2f cycles per wavefront instruction (clock %4.0f MHz)\n", kRateNames[MODE],
           best, perSimdPerClk, 64.0 / perSimdPerClk, mhz);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    (void)hipMalloc(&d_out, (size_t)g_blocks * 256 * 4);
    (void)hipMalloc(&d_clk, 16);
    (void)hipDeviceGetAttribute(&g_wallKHz, hipDeviceAttributeWallClockRate, 0);
    printf("blocks=%d x 256 threads\n", g_blocks);
#define RUN(k) run<k>();
    RATE_RUN_ALL(RUN)
    return 0;
}
