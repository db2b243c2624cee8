// Development aid: what the pieces of a one-frame call cost on this machine, each as wall time of
// (enqueue + hipStreamSynchronize), median of many repetitions.  Build: hipcc --offload-arch=gfx950 -O2
// tools/latency_probe.hip -o tools/latency_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ void k_empty(int* p) {
    if (p && threadIdx.x == 9999) *p = 1;
}
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

static double median_us(const std::function<void()>& f, int reps = 400) {
    std::vector<double> t;
    for (int i = 0; i < 20; i++) f();
    for (int i = 0; i < reps; i++) {
        auto a = std::chrono::steady_clock::now();
        f();
        auto b = std::chrono::steady_clock::now();
        t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t IN = 640 * 480, OUT = 64 * 1024;
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint8_t *hIn, *hOut, *dIn, *dOut;
    CHECK(hipHostMalloc((void**)&hIn, IN, hipHostMallocDefault));
    CHECK(hipHostMalloc((void**)&hOut, OUT, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&dIn, IN));
    CHECK(hipMalloc((void**)&dOut, OUT));
    memset(hIn, 7, IN);
    std::vector<uint8_t> src(IN, 3);

    printf("host memcpy 300 KB                      %7.1f us\n", median_us([&] { memcpy(hIn, src.data(), IN); }));
    printf("sync only                               %7.1f us\n", median_us([&] { (void)hipStreamSynchronize(st); }));
    printf("1 empty kernel + sync                   %7.1f us\n", median_us([&] {
               hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr);
               (void)hipStreamSynchronize(st);
           }));
    for (int n : {4, 10, 20})
        printf("%2d dependent empty kernels + sync       %7.1f us\n", n, median_us([&] {
                   for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr);
                   (void)hipStreamSynchronize(st);
               }));
    printf("H2D 300 KB (copy engine) + sync         %7.1f us\n", median_us([&] {
               (void)hipMemcpyAsync(dIn, hIn, IN, hipMemcpyHostToDevice, st);
               (void)hipStreamSynchronize(st);
           }));
    printf("H2D 300 KB in 4 bands + sync            %7.1f us\n", median_us([&] {
               for (int b = 0; b < 4; b++) (void)hipMemcpyAsync(dIn + b * IN / 4, hIn + b * IN / 4, IN / 4, hipMemcpyHostToDevice, st);
               (void)hipStreamSynchronize(st);
           }));
    printf("kernel reads 300 KB of pinned + sync    %7.1f us\n", median_us([&] {
               hipLaunchKernelGGL(k_copy16, dim3(75), dim3(256), 0, st, (const uint4*)hIn, (uint4*)dIn, (int)(IN / 16));
               (void)hipStreamSynchronize(st);
           }));
    printf("D2H 64 KB (copy engine) + sync          %7.1f us\n", median_us([&] {
               (void)hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, st);
               (void)hipStreamSynchronize(st);
           }));
    printf("kernel writes 64 KB to pinned + sync    %7.1f us\n", median_us([&] {
               hipLaunchKernelGGL(k_copy16, dim3(16), dim3(256), 0, st, (const uint4*)dOut, (uint4*)hOut, (int)(OUT / 16));
               (void)hipStreamSynchronize(st);
           }));
    printf("H2D + kernel + D2H + sync               %7.1f us\n", median_us([&] {
               (void)hipMemcpyAsync(dIn, hIn, IN, hipMemcpyHostToDevice, st);
               hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr);
               (void)hipMemcpyAsync(hOut, dOut, OUT, hipMemcpyDeviceToHost, st);
               (void)hipStreamSynchronize(st);
           }));
    printf("read kernel + kernel + write kernel + sync %4.1f us\n", median_us([&] {
               hipLaunchKernelGGL(k_copy16, dim3(75), dim3(256), 0, st, (const uint4*)hIn, (uint4*)dIn, (int)(IN / 16));
               hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr);
               hipLaunchKernelGGL(k_copy16, dim3(16), dim3(256), 0, st, (const uint4*)dOut, (uint4*)hOut, (int)(OUT / 16));
               (void)hipStreamSynchronize(st);
           }));
    return 0;
}
