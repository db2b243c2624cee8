// Micro-benchmark: sustained VALU rates on gfx950 for the Hamming inner loop's
// instructions, to calibrate the peak the brute-force matcher is priced
// against.  Every loop keeps its operands in registers (no memory traffic) and
// changes its scalar operands every iteration so nothing can be hoisted.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHAINS 8
#define STEPS 16

// MODE 0: v_xor_b32 (VGPR x SGPR) + accumulating v_bcnt_u32_b32 pairs   (2 ops per step)
// MODE 1: v_xor_b32 only                                              (1 op per step)
// MODE 2: v_bcnt_u32_b32 only                                         (1 op per step)
// MODE 3: v_fma_f32                                                   (1 op per step)
// MODE 4: v_pk_fma_f32                                                (1 op per step, 2 lanes-worth)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(const unsigned* __restrict__ in, unsigned* __restrict__ out, int iters,
                                              unsigned s0) {
    unsigned x[CHAINS], y[CHAINS];
    float f[CHAINS], g2[CHAINS];
    for (int c = 0; c < CHAINS; c++) {
        x[c] = in[(threadIdx.x + c * 256) & 4095];
        y[c] = 0;
        f[c] = (float)(x[c] & 255) * 1e-3f;
        g2[c] = f[c] + 1.f;
    }
    unsigned s = s0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (MODE == 0) {
                    unsigned t;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "s"(s), "v"(x[c]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[c]) : "v"(t), "v"(y[c]));
                } else if (MODE == 1) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[c]) : "s"(s), "v"(x[c]));
                } else if (MODE == 2) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[c]));
                } else if (MODE == 3) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[c]) : "v"(g2[c]), "v"(f[c]));
                } else {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a = {f[c], g2[c]};
                    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a));
                    f[c] = a.x;
                    g2[c] = a.y;
                }
            }
            s += 0x9e3779b9u;
        }
    }
    unsigned r = 0;
    for (int c = 0; c < CHAINS; c++) r += x[c] + y[c] + (unsigned)f[c] + (unsigned)g2[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, int opsPerStep, unsigned* in, unsigned* out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 12345u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double laneops = (double)blocks * 256 * iters * STEPS * CHAINS * opsPerStep;
    const double T = laneops / best / 1e9;
    printf("%-34s %8.3f ms  %6.2f T lane-instr/s = %5.1f lanes/clk/SIMD at 2.4 GHz\n", name, best, T,
           T * 1e12 / (256.0 * 4 * 2.4e9));
}

int main(int argc, char** argv) {
    unsigned *in, *out;
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8;
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMemset(in, 0x5a, 4096 * 4);
    printf("blocks=%d x 256 threads\n", blocks);
    run<0>("v_xor_b32 + v_bcnt_u32_b32 (acc)", 2, in, out, blocks);
    run<1>("v_xor_b32", 1, in, out, blocks);
    run<2>("v_bcnt_u32_b32 (acc)", 1, in, out, blocks);
    run<3>("v_fma_f32", 1, in, out, blocks);
    run<4>("v_pk_fma_f32 (2 fma per lane)", 1, in, out, blocks);
    return 0;
}
