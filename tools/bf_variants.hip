// Development aid (not part of the library): times tuning variants of the brute-force scan kernel of
// orb_slam2_amd/csrc/kernels_bf.hip - queries per lane, rows per group, rows per LDS tile, instruction
// order of the 256-bit comparison - on one database, and prints a checksum of the packed results so that
// equality between variants is visible.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf_variants.hip -o tools/bf_variants
#include <cstdarg>
#include <cstdio>

#include "../orb_slam2_amd/csrc/kernels_bf.hip"
#include "bf_valu_scan.inc"

void orb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

__global__ void k_fill(uint32_t* p, long long n, uint32_t seed) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (uint32_t)(z >> 32);
    }
}
__global__ void k_sum(const unsigned long long* p, int n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    atomicAdd(out, s);
}

static uint32_t *d_q, *d_db;
static unsigned long long *d_partial, *d_packed, *d_sum;
static int nq = 100000;
static long long nd = 10000000;

template <int QPL, int GROUP, int TR, int PATTERN>
static void run(const char* name) {
    int perCu = 0, cus = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_bf_top2_lds<QPL, GROUP, TR, PATTERN>, 256, 0);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    long long rpc = GROUP;
    const int chunks = bf_chunks(nq, nd, perCu * cus, 256 * QPL, GROUP, 4096, &rpc);
    const int qBlocks = (nq + 256 * QPL - 1) / (256 * QPL);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_bf_top2_lds<QPL, GROUP, TR, PATTERN>), dim3(qBlocks, chunks), dim3(256), 0, 0, d_q, nq, d_db, nd, rpc,
                           0ll, d_partial);
        hipLaunchKernelGGL(k_bf_merge, dim3((nq + 255) / 256), dim3(256), 0, 0, d_partial, chunks, nq, d_packed,
                           (int32_t*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipMemset(d_sum, 0, 8);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, d_packed, nq, d_sum);
    unsigned long long sum = 0;
    (void)hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost);
    printf("%-22s wg/CU %d chunks %3d rows/chunk %8lld  %7.1f ms  %6.0f Gcmp/s  checksum %llu\n", name, perCu, chunks, rpc, best,
           (double)nq * nd / best / 1e6, sum);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) nq = atoi(argv[1]);
    if (argc > 2) nd = atoll(argv[2]);
    (void)hipMalloc(&d_q, (size_t)nq * 32);
    (void)hipMalloc(&d_db, (size_t)nd * 32);
    (void)hipMalloc(&d_partial, (size_t)4096 * nq * 8);
    (void)hipMalloc(&d_packed, (size_t)nq * 8);
    (void)hipMalloc(&d_sum, 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_q, (long long)nq * 8, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d_db, nd * 8, 2u);
    (void)hipDeviceSynchronize();
#define RUN(q, g, t, p) run<q, g, t, p>("q" #q " g" #g " t" #t " p" #p)
    RUN(2, 4, 128, 0);
    RUN(2, 4, 128, 1);
    RUN(2, 4, 128, 2);
    RUN(2, 4, 128, 3);
    RUN(2, 8, 128, 1);
    RUN(2, 8, 256, 1);
    RUN(4, 4, 128, 1);
    RUN(4, 4, 256, 1);
    RUN(4, 8, 256, 1);
    RUN(4, 2, 128, 1);
    RUN(3, 4, 128, 1);
    RUN(1, 8, 128, 1);
    RUN(4, 4, 128, 3);
    RUN(2, 8, 128, 3);
    return 0;
}
