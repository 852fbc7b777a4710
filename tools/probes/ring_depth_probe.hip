// Probe: the DMA side of the ring convolution kernel alone (no MFMA, no fragment reads): a 128-row activation tile walked
// in 128-byte K steps (first touch: HBM) plus a 128-row weight tile per step (L2-resident), 4 one-KiB pieces per wave and
// step, ONE workgroup barrier per step, S tiles of look-ahead.  What does the barrier-coupled pipeline stream at, and what
// does a deeper ring buy?   (2 workgroups x 8 waves per CU for S <= 2, LDS permitting)
//   hipcc --offload-arch=gfx950 -O3 ring_depth_probe.hip -o ring_depth_probe && ./ring_depth_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int N> __device__ __forceinline__ void vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int S, bool BARRIER>
__global__ void __launch_bounds__(512) k_ring(const unsigned char* g, const unsigned char* w, int rowbytes, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ksteps = rowbytes / 128;
    const unsigned char* tile = g + (size_t)blockIdx.x * 128 * rowbytes;
    auto issue = [&](int k) {
        unsigned char* dst = smem + (k % S) * 32768 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;                  // 0..15 activation rows, 16..31 weight rows
            const unsigned char* src = piece < 16
                ? tile + (size_t)(piece * 8 + (lane >> 3)) * rowbytes + k * 128 + (lane & 7) * 16
                : w + (size_t)((piece - 16) * 8 + (lane >> 3)) * rowbytes + k * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    for (int k = 0; k < S - 1 && k < ksteps; ++k) issue(k);
    for (int k = 0; k < ksteps; ++k) {
        if (k + S - 1 < ksteps) { issue(k + S - 1); vm<(S - 1) * 4>(); }
        else vm<0>();
        if (BARRIER) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_sleep(4);                         // stand-in for the MFMA clusters of the step (~256 cycles)
        if (BARRIER && S == 2) __builtin_amdgcn_s_barrier(); // a 2-slot ring frees its slot only behind a second barrier here
    }
    if (sink && lane == 0 && blockIdx.x == 0xffffff) sink[0] = smem[0];
}

template <typename K>
static void run(const char* name, K kern, int S, const unsigned char* g, const unsigned char* w, size_t total, int rowbytes, unsigned* sink) {
    const int lds = S * 32768;
    const int wgs = (int)(total / ((size_t)128 * rowbytes));
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, w, rowbytes, sink);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, 0, g, w, rowbytes, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-12s S=%d (%3d KiB LDS) row %5d B: %6.2f TB/s of activations, %5.2f us per K step and workgroup slot\n", name, S, lds >> 10,
           rowbytes, 3.0 * total / ms / 1e9, ms * 1e3 / 3.0 / ((double)wgs / (256.0 * (lds > 81920 ? 1 : 2)) * (rowbytes / 128)));
}

int main() {
    unsigned char *g, *w; unsigned* sink;
    const size_t total = (size_t)1 << 30;
    (void)hipMalloc(&g, total); (void)hipMemset(g, 1, total); (void)hipMalloc(&w, 128 * 4096); (void)hipMemset(w, 1, 128 * 4096);
    (void)hipMalloc(&sink, 64);
    for (int rb : {512, 2048}) {
        run("barrier", k_ring<2, true>, 2, g, w, total, rb, sink);
        run("barrier", k_ring<3, true>, 3, g, w, total, rb, sink);
        run("barrier", k_ring<4, true>, 4, g, w, total, rb, sink);
        run("no barrier", k_ring<2, false>, 2, g, w, total, rb, sink);
        run("no barrier", k_ring<3, false>, 3, g, w, total, rb, sink);
    }
    return 0;
}
