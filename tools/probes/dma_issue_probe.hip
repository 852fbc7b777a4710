// Probe: what does issuing global_load_lds_dwordx4 cost the issuing wave, and what rate does a CU sustain?
// One workgroup of W waves on one CU; each wave issues P pieces (1 KiB each, L2-resident source) back to back.
// Prints cycles from first issue to last issue (issue cost) and to vmcnt(0) (completion), per configuration.
//   hipcc --offload-arch=gfx950 dma_issue_probe.hip -o dma_issue_probe && ./dma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
template <int P>
__global__ void k(const unsigned char* g, unsigned long long* out, int stride_wave, int blocks_busy) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* src = g + (size_t)blockIdx.x * (1 << 20) + (size_t)wave * stride_wave + lane * 16;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < P; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * P + i) * 1024), 16, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = t2 - t0; }
}
template <int P> void run(const unsigned char* g, unsigned long long* o, int waves, int blocks) {
    hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    std::vector<unsigned long long> r(32);
    for (int rep = 0; rep < 3; ++rep) {     // last repetition: source resident in L2
        hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(64 * waves), 16 * P * 1024 > 160000 ? 160000 : 16 * P * 1024, 0, g, o, P * 1024, blocks);
        hipDeviceSynchronize();
    }
    hipMemcpy(r.data(), o, 256, hipMemcpyDeviceToHost);
    unsigned long long mi = 0, mc = 0;
    for (int w = 0; w < waves; ++w) { if (r[2 * w] > mi) mi = r[2 * w]; if (r[2 * w + 1] > mc) mc = r[2 * w + 1]; }
    printf("blocks %4d waves %2d pieces/wave %2d: issue %6llu cycles (%5.1f / piece / wave), complete %6llu cycles -> %5.1f B/clk/CU\n",
           blocks, waves, P, mi, (double)mi / P, mc, (double)waves * P * 1024 / mc);
}
int main() {
    unsigned char* g; unsigned long long* o;
    hipMalloc(&g, (size_t)512 << 20); hipMemset(g, 1, (size_t)512 << 20); hipMalloc(&o, 4096);
    for (int blocks : {1, 256}) {
        for (int waves : {1, 4, 8}) { run<4>(g, o, waves, blocks); run<8>(g, o, waves, blocks); run<16>(g, o, waves, blocks); }
    }
    return 0;
}
