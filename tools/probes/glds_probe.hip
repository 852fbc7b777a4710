// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes, and does the LDS base (M0) reach
// beyond 64 KiB on gfx950?   hipcc --offload-arch=gfx950 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ void k(const unsigned* g, unsigned* out, int base) {
    const int lane = threadIdx.x;
    for (int i = threadIdx.x; i < 40000; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned char* src = (const unsigned char*)g + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + base), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 40000; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    unsigned *g, *o;
    std::vector<unsigned> h(256), r(40000);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    hipMalloc(&g, 1024); hipMalloc(&o, 160000);
    hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int base : {0, 4096, 61440, 66560, 98304, 147456}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160000, 0, g, o, base);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), o, 160000, hipMemcpyDeviceToHost);
        int first = -1, cnt = 0, ok = 1;
        for (int i = 0; i < 40000; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; ++cnt; }
        for (int i = 0; i < 256 && first >= 0; ++i) ok &= r[first + i] == 0x1000u + i;
        printf("base %6d: err=%d first changed dword %d (byte %d), %d dwords changed, lane-linear=%d\n", base, (int)e, first, first * 4, cnt, ok);
    }
    return 0;
}
