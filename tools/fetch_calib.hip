// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this device for the access widths the ORB
// kernels use: each kernel streams a buffer of known size once (512 MiB, beyond the on-die cache),
// with 4-byte or 16-byte loads per lane, and writes a 4-byte or 16-byte result per lane.
// Build: hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o tools/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- tools/fetch_calib   (WRITE_SIZE in its own pass)
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_read4(const unsigned* __restrict__ p, size_t n, unsigned* __restrict__ out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read16(const uint4* __restrict__ p, size_t n, unsigned* __restrict__ out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write4(unsigned* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void k_write16(uint4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = (size_t)512 << 20;
    void *buf = nullptr, *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const dim3 grid(256 * 16), block(256);
    k_read4<<<grid, block>>>((const unsigned*)buf, bytes / 4, (unsigned*)out);
    k_read16<<<grid, block>>>((const uint4*)buf, bytes / 16, (unsigned*)out);
    k_write4<<<grid, block>>>((unsigned*)buf, bytes / 4);
    k_write16<<<grid, block>>>((uint4*)buf, bytes / 16);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
