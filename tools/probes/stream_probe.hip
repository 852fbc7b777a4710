// Probe: chip-wide HBM read streaming at the conv kernel's occupancy (2 workgroups x 8 waves per CU), every wave
// keeping D one-KiB loads in flight, (a) as global_load_lds_dwordx4 (LDS-DMA), (b) as global_load_dwordx4 to registers.
//   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int N> __device__ __forceinline__ void vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D>
__global__ void __launch_bounds__(512) k_dma(const unsigned char* g, size_t per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* src = g + ((size_t)blockIdx.x * 8 + wave) * per_wave + lane * 16;
    const int n = (int)(per_wave >> 10);
    unsigned char* dst = smem + wave * (D * 1024);
#pragma unroll
    for (int i = 0; i < D; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    for (int i = D; i < n; i += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            vm<D - 1>();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(i + j) * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        }
    }
    vm<0>();
    if (sink && lane == 0 && blockIdx.x == 0xffffff) sink[0] = dst[0];
}

template <int D>
__global__ void __launch_bounds__(512) k_reg(const unsigned char* g, size_t per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* src = (const uint4*)(g + ((size_t)blockIdx.x * 8 + wave) * per_wave + lane * 16);
    const int n = (int)(per_wave >> 10);
    uint4 r[D];
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = src[(size_t)i * 64];
    for (int i = D; i < n; i += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
            r[j] = src[(size_t)(i + j) * 64];
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K>
static void run(const char* name, K kern, int D, int lds, const unsigned char* g, size_t per_wave, unsigned* sink, int wgs) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, per_wave, sink);
    hipEventRecord(a, 0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, per_wave, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s D=%2d wgs=%d: %.2f TB/s\n", name, D, wgs, 5.0 * wgs * 8 * per_wave / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t per_wave = 512 << 10;
    const int wgs = 512;
    unsigned char* g; unsigned* sink;
    hipMalloc(&g, per_wave * 8 * wgs); hipMemset(g, 1, per_wave * 8 * wgs); hipMalloc(&sink, 64);
    const int lds = 81920;      // two workgroups per CU, like the conv kernel
    run("dma", k_dma<1>, 1, lds, g, per_wave, sink, wgs);
    run("dma", k_dma<2>, 2, lds, g, per_wave, sink, wgs);
    run("dma", k_dma<4>, 4, lds, g, per_wave, sink, wgs);
    run("dma", k_dma<8>, 8, lds, g, per_wave, sink, wgs);
    run("reg", k_reg<1>, 1, lds, g, per_wave, sink, wgs);
    run("reg", k_reg<2>, 2, lds, g, per_wave, sink, wgs);
    run("reg", k_reg<4>, 4, lds, g, per_wave, sink, wgs);
    run("reg", k_reg<8>, 8, lds, g, per_wave, sink, wgs);
    run("reg", k_reg<16>, 16, lds, g, per_wave, sink, wgs);
    // half the waves (the conv kernel's 4 activation waves per workgroup) -- same total bytes over fewer waves
    return 0;
}
