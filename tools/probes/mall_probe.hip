// Read bandwidth as a function of the working-set size: where the Infinity Cache (256 MiB, memory side) stops helping.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_probe tools/probes/mall_probe.hip && tools/probes/mall_probe
// Each pass streams the whole buffer once with 16-byte loads (grid-stride, 2048 x 256 threads, 4 loads in flight per
// thread); passes are separate launches, so the L2s are written back / invalidated in between and only the memory-side
// cache can serve a repeat.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) rd(const float4* __restrict__ p, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    float4 a = {0, 0, 0, 0};
    for (; i + 3 * st < n; i += 4 * st) {
        const float4 v0 = p[i], v1 = p[i + st], v2 = p[i + 2 * st], v3 = p[i + 3 * st];
        a.x += v0.x + v1.x + v2.x + v3.x; a.y += v0.y + v1.y + v2.y + v3.y;
        a.z += v0.z + v1.z + v2.z + v3.z; a.w += v0.w + v1.w + v2.w + v3.w;
    }
    for (; i < n; i += st) { const float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
__global__ void __launch_bounds__(256) wr(float4* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i < n; i += st) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    float* out; hipMalloc(&out, 4);
    const size_t mb[] = {8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t m : mb) {
        const size_t bytes = m << 20, n = bytes / 16;
        float4* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
        for (int w = 0; w < 3; ++w) rd<<<2048, 256>>>(p, n, out);
        const int reps = (int)(8192 / m) + 4;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) rd<<<2048, 256>>>(p, n, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // producer -> consumer: a write pass followed by a read pass of the same buffer (what a layer boundary does)
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) { wr<<<2048, 256>>>(p, n); rd<<<2048, 256>>>(p, n, out); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("%5zu MiB  repeat-read %7.0f GB/s   write+read pair %7.0f GB/s (bytes moved / time)\n", m,
               bytes * (double)reps / ms * 1e-6, 2.0 * bytes * (double)reps / ms2 * 1e-6);
        hipFree(p);
    }
    return 0;
}
