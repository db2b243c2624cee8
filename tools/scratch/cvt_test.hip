#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main() {
    const int n = 256 * 64 + 64;
    float* h = new float[n];
    for (int i = 0; i < 256 * 64; i++) h[i] = i / 64.0f;   // steps of 1/64 over [0,256)
    float extra[] = {-1.f, -0.5f, -0.f, 255.4f, 255.5f, 255.6f, 256.f, 300.f, 1e9f, 0.49999997f, 0.5f, 0.50000006f, 1.5f, 2.5f, 3.5f, 254.5f};
    for (int i = 0; i < 64; i++) h[256 * 64 + i] = extra[i % 16];
    float* d; unsigned* o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(d, o, n);
    unsigned* r = new unsigned[n];
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        long e = lrintf(h[i]); if (e < 0) e = 0; if (e > 255) e = 255;
        if ((long)r[i] != e) { if (bad < 20) printf("x=%.9g got %u want %ld\n", h[i], r[i], e); bad++; }
    }
    printf("mismatches vs saturate(lrintf): %d of %d\n", bad, n);
    return 0;
}
