// Do vector-ALU instructions run beside the FP4 matrix instruction on gfx950, or after it?
//
// Each wavefront runs a loop of 8 v_mfma_scale_f32_32x32x64_f8f6f4 on four accumulators in turn (FP4 operands: 8 passes, 32 cycles
// each at full rate) and K independent v_max3_f32 on registers the matrix instructions do not touch, for K = 0 .. 96,
// with 1, 2 or 3 wavefronts per SIMD.  If the two kinds overlap, the time per iteration stays at the matrix
// instructions' 256 cycles until K x 4 cycles exceed what is left beside their issue slots; if they do not, it
// grows by 4 cycles per vector instruction from the start.  Prints cycles per iteration and SIMD (clock64) and the
// engine clock during the run.
//
//   hipcc -O3 --offload-arch=gfx950 -fno-honor-nans tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int K, int NM>
__global__ __launch_bounds__(256, 3) void k_mix(int iters, float* __restrict__ out, unsigned long long* __restrict__ clk) {
    const int t = threadIdx.x;
    v8i a = {0x22222222 ^ t, 0x2a2a2a2a, 0x02020202, 0x20202020, 0, 0, 0, 0};
    v8i b = {0x2a2a2a2a, 0x22222222 ^ (t << 3), 0x0a0a0a0a, 0x20202020, 0, 0, 0, 0};
    v16f acc[NM > 0 ? NM : 1];
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); m++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[m][i] = 0.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (float)(t + i);
    const float y = (float)t * 0.5f, z = (float)blockIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        // each matrix instruction is followed by its share of the vector instructions, in this order in the code
        // (the scheduling barriers pin it): a lone wavefront can then overlap them too, if the hardware does
        constexpr int GROUPS = 8;  // matrix instructions per iteration, on NM accumulators in turn
#pragma unroll
        for (int m = 0; m < GROUPS; m++) {
            if (NM > 0) acc[m % (NM > 0 ? NM : 1)] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m % (NM > 0 ? NM : 1)], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = m * K / GROUPS; k < (m + 1) * K / GROUPS; k++) {
                // 16 independent chains
                float r;
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x[k & 15]), "v"(y), "v"(z));
                x[k & 15] = r;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); m++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[m][i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[(size_t)blockIdx.x * 256 + t] = s;
    if (t == 0) {
        atomicAdd(&clk[0], c1 - c0);
        atomicAdd(&clk[1], w1 - w0);
        atomicAdd(&clk[2], 1ull);
    }
}

template <int K, int NM>
static void run(int wavesPerSimd, int iters, float* dOut, unsigned long long* dClk) {
    const int blocks = 256 * wavesPerSimd;  // one 4-wavefront workgroup per CU and wavefront-per-SIMD step
    unsigned long long clk[3];
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipMemset(dClk, 0, 24));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<K, NM>), dim3(blocks), dim3(256), 0, 0, iters, dOut, dClk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(clk, dClk, 24, hipMemcpyDeviceToHost));
    }
    const double cyclesPerWave = (double)clk[0] / (double)clk[2];
    const double mhz = 100.0 * (double)clk[0] / (double)clk[1];
    // per SIMD: the wavefronts of a SIMD run side by side, so an iteration of all of them takes cyclesPerWave / iters
    printf("%d accumulator sets (8 MFMA per iteration when > 0) + %3d VALU per iteration, %d wavefront(s) per SIMD: %7.1f cycles per iteration and wavefront, %7.1f per "
           "wavefront-iteration of the SIMD, %.3f ms, clock %4.0f MHz\n",
           NM, K, wavesPerSimd, cyclesPerWave / iters, cyclesPerWave / iters / wavesPerSimd, best, mhz);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* dOut;
    unsigned long long* dClk;
    CHECK(hipMalloc((void**)&dOut, (size_t)256 * 3 * 256 * 4));
    CHECK(hipMalloc((void**)&dClk, 24));
    for (int w = 1; w <= 3; w++) {
        run<0, 4>(w, iters, dOut, dClk);
        run<16, 4>(w, iters, dOut, dClk);
        run<32, 4>(w, iters, dOut, dClk);
        run<48, 4>(w, iters, dOut, dClk);
        run<64, 4>(w, iters, dOut, dClk);
        run<96, 4>(w, iters, dOut, dClk);
        run<24, 4>(w, iters, dOut, dClk);
        run<40, 4>(w, iters, dOut, dClk);
        run<128, 4>(w, iters, dOut, dClk);
        run<32, 0>(w, iters, dOut, dClk);
        run<64, 0>(w, iters, dOut, dClk);
    }
    CHECK(hipGetLastError());
    return 0;
}
