// Probe: how many global_load_lds_dwordx4 (1 KiB each) can a CU keep in flight?  W waves of one workgroup each issue
// 32 pieces from COLD memory (HBM latency) and stamp the clock after every issue: the piece at which the per-issue
// cost jumps from ~20 cycles to the memory latency is the queue depth.
//   hipcc --offload-arch=gfx950 dma_depth_probe.hip -o dma_depth_probe && ./dma_depth_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ void k(const unsigned char* g, unsigned long long* out, size_t region) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* src = g + (size_t)blockIdx.x * region + (size_t)wave * (32 << 12) + lane * 16;
    unsigned long long t[33];
    __builtin_amdgcn_s_barrier();
    t[0] = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 4096),
                                         (__attribute__((address_space(3))) void*)(smem + ((wave * 32 + i) & 127) * 1024), 16, 0, 0);
        t[i + 1] = __builtin_readcyclecounter();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long te = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 33; ++i) out[wave * 40 + i] = t[i] - t[0];
        out[wave * 40 + 33] = te - t[0];
    }
}
int main() {
    unsigned char* g; unsigned long long* o;
    const size_t region = (size_t)4 << 20;
    hipMalloc(&g, region * 256 * 4); hipMemset(g, 1, region * 256 * 4); hipMalloc(&o, 8 * 40 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    std::vector<unsigned long long> r(8 * 40);
    int rep = 0;
    for (int blocks : {1, 256})
        for (int waves : {1, 2, 4, 8}) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 131072, 0, g + (size_t)(rep++ % 4) * region * 256, o, region);
            hipDeviceSynchronize();
            hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
            printf("blocks %3d waves %d: all landed after %llu cycles; cumulative issue time of wave 0 after piece 1,2,4,8,12,16,20,24,28,32:", blocks, waves, r[33]);
            for (int i : {1, 2, 4, 8, 12, 16, 20, 24, 28, 32}) printf(" %llu", r[i]);
            printf("\n");
        }
    return 0;
}
