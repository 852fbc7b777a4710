// Probe: what one CU can pull out of the L2 (hits) at the conv kernels' occupancy (2 workgroups x 8 waves per CU), every wave
// keeping D one-KiB loads in flight, as LDS-DMA (global_load_lds_dwordx4) and as register loads (global_load_dwordx4).
// mode 0: every workgroup walks the SAME `span` bytes (the weight stream of a convolution: one copy per XCD L2);
// mode 1: every workgroup walks its own `span` bytes (512 x span must fit the 8 x 4 MiB of L2).
//   hipcc --offload-arch=gfx950 -O3 l2_stream_probe.hip -o l2_stream_probe && ./l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int N> __device__ __forceinline__ void vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D>
__global__ void __launch_bounds__(512) k_dma(const unsigned char* g, size_t span, size_t wg_stride, int reps, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = g + (size_t)blockIdx.x * wg_stride + lane * 16;
    const int n = (int)(span >> 13);                          // KiB pieces per wave and pass (8 waves share the span)
    unsigned char* dst = smem + wave * (D * 1024);
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n; i += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                vm<D - 1>();
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((size_t)(i + j) * 8 + wave) * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
            }
        }
    }
    vm<0>();
    if (sink && lane == 0 && blockIdx.x == 0xffffff) sink[0] = dst[0];
}

template <int D>
__global__ void __launch_bounds__(512) k_reg(const unsigned char* g, size_t span, size_t wg_stride, int reps, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = g + (size_t)blockIdx.x * wg_stride + lane * 16;
    const int n = (int)(span >> 13);
    uint4 r[D];
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) r[j] = uint4{0, 0, 0, 0};
    for (int rp = 0; rp < reps; ++rp) {
        for (int i = 0; i < n; i += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                acc ^= r[j].x ^ r[j].w;
                r[j] = *(const uint4*)(base + ((size_t)(i + j) * 8 + wave) * 1024);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) acc ^= r[j].x ^ r[j].w;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename K>
static void run(const char* name, K kern, int D, const unsigned char* g, size_t span, size_t wg_stride, int reps, unsigned* sink) {
    const int lds = 8 * D * 1024, wgs = 512;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, span, wg_stride, reps, sink);
    hipEventRecord(a, 0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, span, wg_stride, reps, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 3.0 * wgs * (double)span * reps;
    printf("%-10s D=%d span %7zu KiB %s: %7.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz\n", name, D, span >> 10,
           wg_stride ? "per WG " : "shared ", bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e0 / 1e0 / 1.0);
}

int main() {
    unsigned char* g; unsigned* sink;
    const size_t total = 64u << 20;
    hipMalloc(&g, total); hipMemset(g, 1, total); hipMalloc(&sink, 64);
    for (int mode = 0; mode < 2; ++mode) {
        for (size_t span : {(size_t)32 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
            if (mode == 1 && span * 512 > (32u << 20)) continue;
            const size_t stride = mode ? span : 0;
            const int reps = (int)((64u << 20) / span / 4) + 1;
            run("lds-dma", k_dma<1>, 1, g, span, stride, reps, sink);
            run("lds-dma", k_dma<2>, 2, g, span, stride, reps, sink);
            run("lds-dma", k_dma<4>, 4, g, span, stride, reps, sink);
            run("register", k_reg<1>, 1, g, span, stride, reps, sink);
            run("register", k_reg<2>, 2, g, span, stride, reps, sink);
            run("register", k_reg<4>, 4, g, span, stride, reps, sink);
        }
    }
    return 0;
}
