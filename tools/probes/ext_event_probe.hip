// What do the start / stop events of hipExtLaunchKernelGGL measure when they belong to DIFFERENT launches?
// hipcc --offload-arch=gfx950 tools/probes/ext_event_probe.hip -o /tmp/ext_event_probe && /tmp/ext_event_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t r0, r1, r2, eA, eB, eA2, eB2;
    for (hipEvent_t* e : {&r0, &r1, &r2, &eA, &eB, &eA2, &eB2}) hipEventCreate(e);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(r0, st);
        hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, eA, eA2, 0, 200000LL, (int*)nullptr);   // ~100 us at 2 GHz (s_memtime ticks differ)
        hipEventRecord(r1, st);
        hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, eB2, eB, 0, 400000LL, (int*)nullptr);
        hipEventRecord(r2, st);
        hipStreamSynchronize(st);
        float a = 0, b = 0, ab = 0, rr = 0, r01 = 0, r12 = 0, a2b = 0;
        hipError_t e1 = hipEventElapsedTime(&a, eA, eA2), e2 = hipEventElapsedTime(&b, eB2, eB), e3 = hipEventElapsedTime(&ab, eA, eB);
        hipError_t e4 = hipEventElapsedTime(&a2b, eA2, eB);
        hipEventElapsedTime(&rr, r0, r2); hipEventElapsedTime(&r01, r0, r1); hipEventElapsedTime(&r12, r1, r2);
        printf("kernel A %.1f us (%d) | kernel B %.1f us (%d) | startA -> stopB %.1f us (%d) | stopA -> stopB %.1f us (%d) | records r0->r1 %.1f r1->r2 %.1f r0->r2 %.1f us\n",
               a * 1e3, (int)e1, b * 1e3, (int)e2, ab * 1e3, (int)e3, a2b * 1e3, (int)e4, r01 * 1e3, r12 * 1e3, rr * 1e3);
    }
    return 0;
}
