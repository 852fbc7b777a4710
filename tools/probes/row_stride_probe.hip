// Probe: first-touch (HBM) read rate of the activation operand of a pointwise convolution as the K loop walks it, at the
// conv kernels' occupancy (2 workgroups x 8 waves per CU, each wave D one-KiB LDS-DMA pieces in flight).
//   rows:    a tile is 128 pixel rows of `rowbytes` (= 2 K) bytes; K step k fetches bytes [128 k, 128 k + 128) of every row
//            (a piece = 8 rows x 128 B), i.e. 128-byte granules at a stride of rowbytes      -- the NHWC layout of the trunk
//   blocked: the same bytes stored [pixel block of 8][K / 64][8][64 channels]: a piece is 1 KiB contiguous
//   hipcc --offload-arch=gfx950 -O3 row_stride_probe.hip -o row_stride_probe && ./row_stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int N> __device__ __forceinline__ void vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, bool BLOCKED>
__global__ void __launch_bounds__(512) k_walk(const unsigned char* g, int rowbytes, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ksteps = rowbytes / 128;
    const unsigned char* tile = g + (size_t)blockIdx.x * 128 * rowbytes;
    unsigned char* dst = smem + wave * (D * 1024);
    int issued = 0;
    for (int k = 0; k < ksteps; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                        // 16 pieces per K step, two per wave
            const int piece = wave * 2 + i;
            const unsigned char* src;
            if (BLOCKED) src = tile + ((size_t)piece * ksteps + k) * 1024 + lane * 16;
            else src = tile + (size_t)(piece * 8 + (lane >> 3)) * rowbytes + k * 128 + (lane & 7) * 16;
            if (issued >= D) vm<D - 1>();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (issued % D) * 1024), 16, 0, 0);
            ++issued;
        }
    }
    vm<0>();
    if (sink && lane == 0 && blockIdx.x == 0xffffff) sink[0] = dst[0];
}

template <typename K>
static void run(const char* name, K kern, int D, const unsigned char* g, size_t total, int rowbytes, unsigned* sink) {
    const int lds = 8 * D * 1024;
    const int wgs = (int)(total / ((size_t)128 * rowbytes));
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, rowbytes, sink);
    (void)hipEventRecord(a, 0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, rowbytes, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-8s D=%d row %5d B (%4d tiles): %6.2f TB/s\n", name, D, rowbytes, wgs, 3.0 * total / ms / 1e9);
}

int main() {
    unsigned char* g; unsigned* sink;
    const size_t total = (size_t)1 << 30;                    // 1 GiB: far beyond the 256 MB Infinity Cache
    (void)hipMalloc(&g, total); (void)hipMemset(g, 1, total); (void)hipMalloc(&sink, 64);
    for (int rb : {256, 512, 1024, 2048, 4096}) {
        run("rows", k_walk<2, false>, 2, g, total, rb, sink);
        run("rows", k_walk<4, false>, 4, g, total, rb, sink);
        run("blocked", k_walk<2, true>, 2, g, total, rb, sink);
        run("blocked", k_walk<4, true>, 4, g, total, rb, sink);
    }
    return 0;
}
