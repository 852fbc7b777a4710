// Probe: what v_mfma_f32_16x16x32_f16 rate does a wave reach in the shape of this library's image-resident loops -- 56 independent
// accumulator tiles (4 A fragments x 14 B fragments), one wave per SIMD -- (a) with the B fragments in registers, (b) with one
// ds_read_b128 per four MFMAs seven fragments ahead (block_img / conv_img3's bi_pipe), on one CU and on the whole chip (clock under
// load), with random operands.   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_rate_probe tools/probes/mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int LDSR, int WPE>
__global__ void __launch_bounds__(256 * WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
rate_kernel(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((u32x4*)smem)[i] = src[i];         // 64 KB of operand bits
    __syncthreads();
    u32x4 a[4], b[14];
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = src[(threadIdx.x + 64 * f) & 4095];
#pragma unroll
    for (int g = 0; g < 14; ++g) b[g] = src[(threadIdx.x * 3 + 17 * g) & 4095];
    f32x4 acc[4][14];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 14; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lbase = (unsigned)(lane * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (LDSR) {
            // ring of 7 fragments ahead: the reads of group g + 7 are issued before the MFMAs of group g
#pragma unroll
            for (int g = 0; g < 14; ++g) {
                const int gn = (g + 7) % 14;
                asm volatile("ds_read_b128 %0, %1" : "=v"(b[gn]) : "v"((lbase + (unsigned)(gn * 1024 + (it & 3) * 16384)) & 65535u));
                if (g == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");      // (hand-counted like bi_pipe: the fragments of this half are in)
                if (g == 7) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[f][g]) : "v"(a[f]), "v"(b[g]));   // (the library's form: AGPR accumulators)
            }
        } else {
#pragma unroll
            for (int g = 0; g < 14; ++g)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[f][g]) : "v"(a[f]), "v"(b[g]));   // (the library's form: AGPR accumulators)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 14; ++g) s += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int LDSR, int WPE>
static void run(const char* what, int grid, const u32x4* src, float* out, unsigned long long* ticks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)rate_kernel<LDSR, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rate_kernel<LDSR, WPE>), dim3(grid), dim3(256 * WPE), 65536, 0, src, out, ticks, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t = 0;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 56.0;                   // MFMAs per wave
    const double per_simd = n * WPE;                         // MFMAs per SIMD
    printf("%-58s grid %4d: %8.1f us | %6.2f ns per MFMA and SIMD = %5.2f cycles at 2.4 GHz | %5.2f s_memtime ticks per MFMA and SIMD | %6.0f TFLOP/s\n", what, grid,
           ms * 1e3, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, (double)t / per_simd, (double)grid * 4 * per_simd * 16384.0 / (ms * 1e-3) * 1e-12);
}

int main() {
    u32x4* src; float* out; unsigned long long* ticks;
    std::vector<unsigned> h(4096 * 4);
    srand(7);
    for (auto& v : h) {                                      // random fp16 pairs in [-2, 2)
        auto r16 = [] { const float x = (rand() / (float)RAND_MAX) * 4.f - 2.f; _Float16 hx = (_Float16)x; unsigned short u; __builtin_memcpy(&u, &hx, 2); return (unsigned)u; };
        v = r16() | (r16() << 16);
    }
    (void)hipMalloc(&src, 65536); (void)hipMalloc(&out, 4 * 512 * 1024); (void)hipMalloc(&ticks, 64);
    (void)hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int grid : {1, 64, 128, 256}) {
        run<0, 1>("registers only, 1 wave per SIMD", grid, src, out, ticks, iters);
        run<1, 1>("one ds_read_b128 per 4 MFMAs (7 ahead), 1 wave per SIMD", grid, src, out, ticks, iters);
    }
    return 0;
}
