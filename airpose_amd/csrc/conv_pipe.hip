// Software-pipelined implicit-GEMM convolution for gfx950: the throughput path of the trunk.
//
// Same math and LDS image as conv_igemm.hip (see there for the reference lines it replaces), but the
// operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR round trip) into an
// S-deep ring, so S-1 K-steps of loads stay in flight under the MFMAs:
//
//   per K step:  s_waitcnt vmcnt((S-2)*LPW)   tile kt has landed (this wave's pieces)
//                s_barrier                    ... everybody's pieces; ring slot (kt-1)%S is free
//                issue tile kt+S-1            LPW x global_load_lds per wave, 1 KiB each
//                2 x (ds_read_b128 frags, FM*FN MFMA)
//
// LDS-DMA writes lane-linearly (wave-uniform base + lane*16), so the XOR swizzle of the 16-byte
// chunk index lives on the SOURCE address: lane l of a piece (8 rows x 128 B) fetches chunk
// (l & 7) ^ (l >> 3) of row l >> 3 and the MFMA side reads chunk c at position c ^ (row & 7).
// Predication (3x3 halo, ragged M) is a pointer select to a 16-byte zero line.
// Waits are counted by hand (inline asm) and the barrier is the raw s_barrier: __syncthreads()
// would drain the DMA queue (vmcnt(0)) on every step.
#include "ap_common.h"
#include "kernels.h"

namespace {

template <typename T> struct Elem2;
template <> struct Elem2<bf16_t> { static constexpr int EPC = 8; };
template <> struct Elem2<float>  { static constexpr int EPC = 4; };

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int FM, int FN>
__device__ __forceinline__ void mma_chunk2(const u32x4 (&xf)[FM], const u32x4 (&wf)[FN], f32x4 (&acc)[FM][FN]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[fm]), acc[fm][fn], 0, 0, 0);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const uint32_t wv = t == 0 ? wf[fn].x : t == 1 ? wf[fn].y : t == 2 ? wf[fn].z : wf[fn].w;
                    const uint32_t xv = t == 0 ? xf[fm].x : t == 1 ? xf[fm].y : t == 2 ? xf[fm].z : xf[fm].w;
                    acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        __builtin_bit_cast(float, wv), __builtin_bit_cast(float, xv), acc[fm][fn], 0, 0, 0);
                }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int S>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) conv_pipe_kernel(const ConvArgs p) {
    constexpr int EPC = Elem2<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    constexpr int FM = BM / WAVES_M / 16, FN = BN / WAVES_N / 16;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;
    constexpr int LPW = ROWS / 8 / NW;          // 1-KiB DMA pieces per wave per tile
    constexpr int CLD = BN + 4;
    static_assert(ROWS % (8 * NW) == 0 && BM % 8 == 0, "tile rows must split into 8-row pieces per wave");
    static_assert(S >= 2 && (S - 2) * LPW < 64, "vmcnt immediate");
    static_assert(BM * CLD * 4 <= S * STAGE, "epilogue tile must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile % p.ntiles;

    // ---------------------------------------------------------------- DMA source state
    const int prow = lane >> 3;                              // row within an 8-row piece
    const int pchunk = (lane & 7) ^ prow;                    // source chunk (swizzle on the source side)
    const unsigned char* src[LPW];                           // row base + chunk, bytes
    int hi0[LPW], wi0[LPW];
    const int HoWo = p.Ho * p.Wo;
    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    const unsigned char* zg = (const unsigned char*)p.zero;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int row = (wave * LPW + i) * 8 + prow;
        if ((wave * LPW + i) * 8 < BM) {                     // activation row (wave-uniform test)
            const int m = bm * BM + row;
            if (m < p.M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                hi0[i] = ho * p.stride - p.pad;
                wi0[i] = wo * p.stride - p.pad;
                src[i] = xg + (((size_t)n * p.H * p.W + (ptrdiff_t)hi0[i] * p.W + wi0[i]) * p.ldx + pchunk * EPC) * sizeof(T);
            } else {
                hi0[i] = -0x40000000;
                wi0[i] = 0;
                src[i] = zg;
            }
        } else {                                             // weight row: always valid (rows padded to 128)
            hi0[i] = 0;
            wi0[i] = 0;
            src[i] = wg + ((size_t)(bn * BN + row - BM) * p.wld + pchunk * EPC) * sizeof(T);
        }
    }
    const int cpb = p.Cin / BK;
    const int KT = p.KH * p.KW * cpb;
    int r = 0, s = 0, cb = 0, ktl = 0;                       // tap / chunk / index of the tile being ISSUED

    auto issue_tile = [&](int stage) {
        const ptrdiff_t xoff = (((ptrdiff_t)r * p.W + s) * p.ldx + cb * BK) * (ptrdiff_t)sizeof(T);
        const ptrdiff_t woff = (ptrdiff_t)ktl * BK * (ptrdiff_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int piece = wave * LPW + i;
            const unsigned char* g;
            if (piece * 8 < BM) {
                const bool ok = (unsigned)(hi0[i] + r) < (unsigned)p.H && (unsigned)(wi0[i] + s) < (unsigned)p.W;
                g = ok ? src[i] + xoff : zg;
            } else {
                g = src[i] + woff;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + piece * 1024),
                                             16, 0, 0);
        }
        ++ktl;
        if (++cb == cpb) { cb = 0; if (++s == p.KW) { s = 0; ++r; } }
    };

    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g4 = lane >> 4;
    const int sw0 = ((g4 ^ (lr & 7)) << 4), sw1 = (((4 + g4) ^ (lr & 7)) << 4);
    const int xfrag = (wm * (BM / WAVES_M) + lr) * 128;
    const int wfrag = BM * 128 + (wn * (BN / WAVES_N) + lr) * 128;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: S-1 tiles in flight
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < KT) issue_tile(t);

    int cs = 0, is = S - 1;                                  // stage being computed / issued
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + S - 2 < KT) wait_vmcnt<(S - 2) * LPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < KT) issue_tile(is);
        if (++is == S) is = 0;
        const unsigned char* sb = smem + cs * STAGE;
        if (++cs == S) cs = 0;
        u32x4 xf[FM], wf[FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) xf[fm] = *(const u32x4*)(sb + xfrag + fm * 16 * 128 + sw0);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) wf[fn] = *(const u32x4*)(sb + wfrag + fn * 16 * 128 + sw0);
        mma_chunk2<T, FM, FN>(xf, wf, acc);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) xf[fm] = *(const u32x4*)(sb + xfrag + fm * 16 * 128 + sw1);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) wf[fn] = *(const u32x4*)(sb + wfrag + fn * 16 * 128 + sw1);
        mma_chunk2<T, FM, FN>(xf, wf, acc);
    }
    __syncthreads();                                         // all MFMA reads done before the ring is reused

    // ---------------------------------------------------------------- epilogue (as conv_igemm.hip)
    float* ct = (float*)smem;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int chl = wn * (BN / WAVES_N) + fn * 16 + g4 * 4;
        const int ch = bn * BN + chl;
        const float4 sc = *(const float4*)(p.scale + ch);
        const float4 sh = *(const float4*)(p.shift + ch);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int px = wm * (BM / WAVES_M) + fm * 16 + lr;
            float4 v;
            v.x = acc[fm][fn][0] * sc.x + sh.x;
            v.y = acc[fm][fn][1] * sc.y + sh.y;
            v.z = acc[fm][fn][2] * sc.z + sh.z;
            v.w = acc[fm][fn][3] * sc.w + sh.w;
            *(float4*)(ct + px * CLD + chl) = v;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / EPC;
    constexpr int NIT = BM * CPR / NT;
    static_assert(BM * CPR % NT == 0, "epilogue chunks must divide evenly");
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    u32x4 rv[NIT];
    if (rg) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = tid + it * NT, px = q / CPR, cc = q - px * CPR;
            const int m = bm * BM + px, ch = bn * BN + cc * EPC;
            const bool ok = m < p.M && ch < p.Cout;
            rv[it] = *(const u32x4*)(ok ? rg + (size_t)m * p.ldr + ch : (const T*)p.zero);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = tid + it * NT, px = q / CPR, cc = q - px * CPR;
        const int m = bm * BM + px, ch = bn * BN + cc * EPC;
        if (m >= p.M || ch >= p.Cout) continue;
        const float* sp = ct + px * CLD + cc * EPC;
        if constexpr (sizeof(T) == 2) {
            float4 a = *(const float4*)sp, b = *(const float4*)(sp + 4);
            if (rg) {
                float lo, hi;
                unpack_bf16x2(rv[it][0], lo, hi); a.x += lo; a.y += hi;
                unpack_bf16x2(rv[it][1], lo, hi); a.z += lo; a.w += hi;
                unpack_bf16x2(rv[it][2], lo, hi); b.x += lo; b.y += hi;
                unpack_bf16x2(rv[it][3], lo, hi); b.z += lo; b.w += hi;
            }
            if (p.relu) {
                a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
            }
            u32x4 o;
            o[0] = pack_bf16x2(a.x, a.y); o[1] = pack_bf16x2(a.z, a.w);
            o[2] = pack_bf16x2(b.x, b.y); o[3] = pack_bf16x2(b.z, b.w);
            *(u32x4*)(yg + (size_t)m * p.ldy + ch) = o;
        } else {
            float4 a = *(const float4*)sp;
            if (rg) {
                // (copy the lanes out first: bit_cast applied directly to a vector element mis-compiles)
                const uint32_t r0 = rv[it].x, r1 = rv[it].y, r2 = rv[it].z, r3 = rv[it].w;
                a.x += __builtin_bit_cast(float, r0); a.y += __builtin_bit_cast(float, r1);
                a.z += __builtin_bit_cast(float, r2); a.w += __builtin_bit_cast(float, r3);
            }
            if (p.relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            *(float4*)(yg + (size_t)m * p.ldy + ch) = a;
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int S>
hipError_t launch_pipe(ConvArgs a, hipStream_t st) {
    static bool attr_set = false;
    auto kern = conv_pipe_kernel<T, BM, BN, WM, WN, S>;
    constexpr int lds = S * (BM + BN) * 128;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(64 * WM * WN), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// cfg: 0 = 256x128 (8 waves, 3 stages), 1 = 128x128 (4 waves, 4 stages), 2 = 128x64 (4 waves, 4 stages),
//      3 = 256x64 (8 waves, 3 stages)
hipError_t ap_launch_conv_pipe(const ConvArgs& a, int is_bf16, int cfg, hipStream_t st) {
    if (!a.zero) return hipErrorInvalidValue;
    if (is_bf16) {
        switch (cfg) {
            case 0: return launch_pipe<bf16_t, 256, 128, 4, 2, 3>(a, st);
            case 1: return launch_pipe<bf16_t, 128, 128, 2, 2, 4>(a, st);
            case 2: return launch_pipe<bf16_t, 128, 64, 2, 2, 4>(a, st);
            case 3: return launch_pipe<bf16_t, 256, 64, 4, 2, 3>(a, st);
        }
    } else {
        switch (cfg) {
            case 0: return launch_pipe<float, 256, 128, 4, 2, 3>(a, st);
            case 1: return launch_pipe<float, 128, 128, 2, 2, 4>(a, st);
            case 2: return launch_pipe<float, 128, 64, 2, 2, 4>(a, st);
            case 3: return launch_pipe<float, 256, 64, 4, 2, 3>(a, st);
        }
    }
    return hipErrorInvalidValue;
}
