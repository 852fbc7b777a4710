// Probe: raw_buffer_load_lds (buffer_load_dwordx4 ... lds): lane-linear LDS destination? out-of-range voffset -> zeros?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ void k(const unsigned* g, unsigned nbytes, unsigned* out, int oob_lane) {
    const int lane = threadIdx.x;
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    unsigned voff = lane * 16;
    if (lane == oob_lane) voff = 0xfffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    unsigned *g, *o;
    std::vector<unsigned> h(256), r(1024);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    (void)hipMalloc(&g, 1024); (void)hipMalloc(&o, 4096);
    (void)hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, g, 1024u, o, 5);
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    printf("err=%d\n", (int)e);
    int ok = 1;
    for (int i = 0; i < 256; ++i) {
        unsigned want = (i / 4 == 5) ? 0u : 0x1000u + i;
        if (r[256 + i] != want) { ok = 0; printf("dword %d: got %x want %x\n", i, r[256 + i], want); if (i > 40) break; }
    }
    printf("lane-linear with OOB lane zero-filled: %d ; untouched before=%x after=%x\n", ok, r[255], r[512]);
    return 0;
}
