// Stem and pooling kernels of the ResNet-50 trunk for gfx950.
//   stem   conv 7x7 stride 2 pad 3 (3 -> 64) + BN + ReLU   model_copenet.py:57-60, 163-165
//   pool   MaxPool2d(3, stride 2, pad 1)                    model_copenet.py:61, 166
//   avg    AvgPool2d(7, stride 1) + view(B, -1)             model_copenet.py:66, 173-174
// The stem reads the caller's NCHW fp32 crop directly (input contract) and
// writes NHWC in the trunk's storage type, so no separate layout-conversion pass exists.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int IMG = 224, SO = 112, PO = 56, SC = 64;

// ------------------------------------------------------------------------------------------------
// Direct (VALU, exact fp32 FMA chain) stem: one thread = one output pixel x 64 channels.
// Used by the fp32 parity path; the bf16 throughput path uses the MFMA stem below.
template <typename T>
__global__ void __launch_bounds__(256) stem_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ y) {
    constexpr int TS = 16, PS = 2 * TS + 5, PLD = PS + 1;       // 37-wide patch, padded rows
    __shared__ __attribute__((aligned(16))) float wsm[147 * 64];
    __shared__ float patch[3 * PS * PLD];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
    for (int i = tid; i < 147 * 64; i += 256) wsm[i] = w[i];
    const float* xin = x + (size_t)n * 3 * IMG * IMG;
    for (int i = tid; i < 3 * PS * PS; i += 256) {
        const int c = i / (PS * PS), rem = i - c * PS * PS, py = rem / PS, px = rem - py * PS;
        const int iy = 2 * ty0 - 3 + py, ix = 2 * tx0 - 3 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)IMG && (unsigned)ix < (unsigned)IMG) v = xin[(c * IMG + iy) * IMG + ix];
        patch[(c * PS + py) * PLD + px] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    // k order = (c, r, s): the order in which a direct NCHW convolution accumulates
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 7; ++r)
            for (int s = 0; s < 7; ++s) {
                const float xv = patch[(c * PS + 2 * ty + r) * PLD + 2 * tx + s];
                const float4* wr = (const float4*)(wsm + ((r * 7 + s) * 3 + c) * 64);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 wv = wr[q];
                    acc[4 * q + 0] = fmaf(xv, wv.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(xv, wv.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(xv, wv.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(xv, wv.w, acc[4 * q + 3]);
                }
            }
    const int oy = ty0 + ty, ox = tx0 + tx;
    if (oy >= SO || ox >= SO) return;
    T* dst = y + (((size_t)n * SO + oy) * SO + ox) * SC;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * q + e] * scale[4 * q + e] + shift[4 * q + e], 0.f);
        if constexpr (sizeof(T) == 2) {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(dst + 4 * q) = o;
        } else if constexpr (ElemKind<T>::KIND == K_SPLIT) {
            if (q & 1) {                                    // groups of 8 channels: [8 hi | 8 lo]
                float v8[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v8[e] = fmaxf(acc[4 * (q - 1) + e] * scale[4 * (q - 1) + e] + shift[4 * (q - 1) + e], 0.f);
                    v8[4 + e] = v[e];
                }
                u32x4 hi, lo;
                split8_pack(v8, hi, lo);
                *(u32x4*)(dst + 4 * (q - 1)) = hi;
                *(u32x4*)(dst + 4 * q) = lo;
            }
        } else {
            *(float4*)(dst + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// Kernel rows in the order 0, 2, 4, 6, 1, 3, 5 (both MFMA stems of the 16-bit kinds, fused and unfused: same fp32 summation order,
// so they stay bit-identical).  Conv row fm at kernel row kb reads input row 2 fm + kb: within one parity the rows of step kb + 2
// are the rows of step kb moved up by one conv row, so the fused strip kernel keeps its five fragments in registers and reads
// ONE new input-row fragment per step instead of five (15 instead of 35 per strip; its K loop is bound by LDS reads).
__device__ constexpr int stem_kb_of(int step) { return step < 4 ? 2 * step : 2 * (step - 4) + 1; }
template <int I, int N, typename F> __device__ __forceinline__ void stem_sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        stem_sfor<I + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA stem (bf16 throughput path).  One workgroup = 16x16 conv outputs x 64 channels.
// GEMM view: M = pixels, N = 64, K' = 7 kernel rows x 32 (7 taps x 4 channel slots, the 4th slot and an 8th tap are
// zero) = 224.  The input patch sits in LDS as bf16 [iy][ix][4] (channel-interleaved, 8 bytes per pixel), so the 8
// consecutive k' a lane feeds to v_mfma_f32_16x16x32_bf16 are two neighbouring pixels = ONE aligned ds_read_b128
// (with 3 channels per pixel the same operand was four unaligned ds_read_b32 and the kernel was LDS-bound: PMC showed
// 65 % of its LDS cycles as bank conflicts).  No im2col buffer; k' slots without a tap meet zero weights.
constexpr int SP = 37;                       // patch rows / cols for a 16x16 output tile
constexpr int SPW = 38;                      // patch row stride in pixels (even: every fragment 16-byte aligned)
constexpr int SWLD = AP_STEM_WLD;            // packed weight row stride (bf16): 480 B.  ds_read_b128 serves a wave in the lane groups
                                             // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, .. (MI355X_MICROARCH.md, LDS): with rows of 464 B
                                             // (232, rounds 2-5: chosen for contiguous groups of 16) every group met a bank twice --
                                             // PMC: a third of the persistent kernel's LDS cycles -- with 480 B none does
constexpr int SKB = 7;                       // 7 x 32 = 224 padded K: one k-block per kernel row

__global__ void __launch_bounds__(256) stem_mfma_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int n_split, const bf16_t* __restrict__ wpk,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, bf16_t* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) bf16_t wsm[64 * SWLD];
    __shared__ __attribute__((aligned(16))) bf16_t patch[SP * SPW * 4 + 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, ty0 = blockIdx.y * 16, tx0 = blockIdx.x * 16;
    {   // weights: 64 x 400 B
        const u32x4* src = (const u32x4*)wpk;
        u32x4* dst = (u32x4*)wsm;
        for (int i = tid; i < 64 * SWLD * 2 / 16; i += 256) dst[i] = src[i];
    }
    const float* xin = (n < n_split ? x0 + (size_t)n * 3 * IMG * IMG : x1 + (size_t)(n - n_split) * 3 * IMG * IMG);
    for (int i = tid; i < 3 * SP * SP; i += 256) {
        const int c = i / (SP * SP), rem = i - c * SP * SP, py = rem / SP, px = rem - py * SP;
        const int iy = 2 * ty0 - 3 + py, ix = 2 * tx0 - 3 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)IMG && (unsigned)ix < (unsigned)IMG) v = xin[(c * IMG + iy) * IMG + ix];
        patch[(py * SPW + px) * 4 + c] = f32_to_bf16(v);
    }
    for (int i = tid; i < SP * SPW; i += 256) patch[i * 4 + 3] = 0;          // 4th channel slot
    if (tid < 32) patch[SP * SPW * 4 + tid] = 0;
    if (tid < SP * 3) patch[((tid / 3) * SPW + SP) * 4 + tid % 3] = 0;       // pad column
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int step = 0; step < SKB; ++step) {                // k-block = kernel row; lane group g = taps 2g, 2g+1
        const int kb = stem_kb_of(step);
        u32x4 wf[4], xf[4];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) wf[fn] = *(const u32x4*)(wsm + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int ty = wave * 4 + fm;
            xf[fm] = *(const u32x4*)(patch + ((2 * ty + kb) * SPW + 2 * lr + 2 * g) * 4);
        }
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn)
                acc[fm][fn] = ap_mfma16(
                    __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[fm]), acc[fm][fn]);
    }
    // epilogue: lane = pixel (tile row wave*4+fm, col lr), channels fn*16 + g*4 .. +3
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        const int ch = fn * 16 + g * 4;
        const float4 sc = *(const float4*)(scale + ch), sh = *(const float4*)(shift + ch);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int oy = ty0 + wave * 4 + fm, ox = tx0 + lr;
            uint2 o;
            o.x = pack_bf16x2(fmaxf(acc[fm][fn][0] * sc.x + sh.x, 0.f), fmaxf(acc[fm][fn][1] * sc.y + sh.y, 0.f));
            o.y = pack_bf16x2(fmaxf(acc[fm][fn][2] * sc.z + sh.z, 0.f), fmaxf(acc[fm][fn][3] * sc.w + sh.w, 0.f));
            *(uint2*)(y + (((size_t)n * SO + oy) * SO + ox) * SC + ch) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 MFMA stem (bf16x2 parity mode): the same tile program as stem_mfma_kernel with every operand as a
// (hi, lo) bf16 pair held in two planes -- two weight images [64][240] and two input patches -- and each product as
// hi*hi + hi*lo + lo*hi on the bf16 matrix pipe (the lo*lo term, 2^-18 relative, is dropped: the planes are separate
// fragments here, so the three MFMAs are explicit).  Output: split-bf16 pairs, NHWC.
__global__ void __launch_bounds__(256) stem_mfma_split_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                              int n_split, const bf16_t* __restrict__ wpk_hi,
                                                              const bf16_t* __restrict__ wpk_lo,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, bsplit_t* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stem_lds[];
    constexpr int WEL = 64 * SWLD, PEL = SP * SPW * 4 + 32;
    bf16_t* wsm_hi = (bf16_t*)stem_lds;
    bf16_t* wsm_lo = wsm_hi + WEL;
    bf16_t* patch_hi = wsm_lo + WEL;
    bf16_t* patch_lo = patch_hi + PEL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // one workgroup = one column of 7 tiles of an image: the 58 KB of weights are fetched once per 7 tiles
    const int n = blockIdx.z, tx0 = blockIdx.x * 16;
    {
        const u32x4 *sh_ = (const u32x4*)wpk_hi, *sl_ = (const u32x4*)wpk_lo;
        u32x4 *dh = (u32x4*)wsm_hi, *dl = (u32x4*)wsm_lo;
        for (int i = tid; i < WEL * 2 / 16; i += 256) { dh[i] = sh_[i]; dl[i] = sl_[i]; }
    }
    const float* xin = (n < n_split ? x0 + (size_t)n * 3 * IMG * IMG : x1 + (size_t)(n - n_split) * 3 * IMG * IMG);
    const int lr = lane & 15, g = lane >> 4;
  for (int ty0 = 0; ty0 < SO; ty0 += 16) {
    if (ty0) __syncthreads();                               // every wave is done with the previous tile's patch
    for (int i = tid; i < 3 * SP * SP; i += 256) {
        const int c = i / (SP * SP), rem = i - c * SP * SP, py = rem / SP, px = rem - py * SP;
        const int iy = 2 * ty0 - 3 + py, ix = 2 * tx0 - 3 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)IMG && (unsigned)ix < (unsigned)IMG) v = xin[(c * IMG + iy) * IMG + ix];
        const bf16_t hi = f32_to_bf16(v);
        patch_hi[(py * SPW + px) * 4 + c] = hi;
        patch_lo[(py * SPW + px) * 4 + c] = f32_to_bf16(v - bf16_to_f32(hi));
    }
    for (int i = tid; i < SP * SPW; i += 256) { patch_hi[i * 4 + 3] = 0; patch_lo[i * 4 + 3] = 0; }
    if (tid < 32) { patch_hi[SP * SPW * 4 + tid] = 0; patch_lo[SP * SPW * 4 + tid] = 0; }
    if (tid < SP * 3) {
        patch_hi[((tid / 3) * SPW + SP) * 4 + tid % 3] = 0;
        patch_lo[((tid / 3) * SPW + SP) * 4 + tid % 3] = 0;
    }
    __syncthreads();

    f32x4 acc[4][4];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < SKB; ++kb) {
        u32x4 wh[4], wl[4], xh[4], xl[4];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            wh[fn] = *(const u32x4*)(wsm_hi + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
            wl[fn] = *(const u32x4*)(wsm_lo + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
        }
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int ty = wave * 4 + fm;
            xh[fm] = *(const u32x4*)(patch_hi + ((2 * ty + kb) * SPW + 2 * lr + 2 * g) * 4);
            xl[fm] = *(const u32x4*)(patch_lo + ((2 * ty + kb) * SPW + 2 * lr + 2 * g) * 4);
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                for (int fn = 0; fn < 4; ++fn)
                    acc[fm][fn] = ap_mfma16(
                        __builtin_bit_cast(bf16x8, pass == 2 ? wl[fn] : wh[fn]),
                        __builtin_bit_cast(bf16x8, pass == 1 ? xl[fm] : xh[fm]), acc[fm][fn]);
    }
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        const int ch = fn * 16 + g * 4;
        const float4 sc = *(const float4*)(scale + ch), sh = *(const float4*)(shift + ch);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int oy = ty0 + wave * 4 + fm, ox = tx0 + lr;
            // a group of 8 channels = this lane's 4 and those of lane ^ 16 (g ^ 1): the even-g lane stores the 8 hi parts,
            // the odd-g lane the 8 lo parts
            float own[4] = {fmaxf(acc[fm][fn][0] * sc.x + sh.x, 0.f), fmaxf(acc[fm][fn][1] * sc.y + sh.y, 0.f),
                            fmaxf(acc[fm][fn][2] * sc.z + sh.z, 0.f), fmaxf(acc[fm][fn][3] * sc.w + sh.w, 0.f)};
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float other = __shfl_xor(own[e], 16);
                v8[e] = (g & 1) ? other : own[e];
                v8[4 + e] = (g & 1) ? own[e] : other;
            }
            u32x4 hi, lo;
            split8_pack(v8, hi, lo);
            const int ch8 = fn * 16 + (g >> 1) * 8;
            *(u32x4*)(y + (((size_t)n * SO + oy) * SO + ox) * SC + ch8 + (g & 1) * 4) = (g & 1) ? lo : hi;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused MFMA stem + 3x3/2 max-pool (bf16 throughput path): conv1 -> bn1 -> relu -> maxpool in one kernel,
// the 112x112x64 stem activation (1.6 MB/image) never touches HBM.
// One workgroup = 7 waves = one image x a strip of 2 pooled rows (5 conv rows, 15 input rows); wave w owns the
// 16-pixel column segment w of all 5 conv rows (5 x 4 accumulator fragments).  Vertical 3-max in registers,
// horizontal 3-max through a 28 KiB LDS tile.  Post-ReLU values are
// >= 0, so padding positions may be treated as 0 and bf16 rounding (monotonic) commutes with max: the result
// is bit-identical to stem_mfma_kernel followed by maxpool_kernel.
#ifndef AP_STEM_PASSES
#define AP_STEM_PASSES 1
#endif
// timing-only builds (results wrong): 1 no input loads | 2 no weight loads | 4 no MFMAs | 8 no output stores | 16 no horizontal
// pooling pass | 32 no epilogue (BatchNorm, ReLU, vertical maximum, pooled rows to LDS) | 128 no fragment reads in the K loop
#ifndef STEM_ABLATE
#define STEM_ABLATE 0
#endif
constexpr int FPW = 232;                                   // patch row stride in pixels (230 used, even)
constexpr int FROWS = 15;                                  // input rows of a strip
// The compute part of a strip (one of the 7 waves that own a 16-column segment each): K loop over the 7 kernel rows, then BatchNorm +
// ReLU + the vertical 3-maximum of the two pooled rows into vm (XOR-swizzled 128-byte rows).  Shared by the strip kernel (vm lies
// over the dead patch: OVERLAY, a workgroup barrier in between) and the persistent kernel (vm is a buffer of its own).
// row0_valid: conv row 2 py0 - 1 exists (not the first strip of an image)
template <bool OVERLAY>
__device__ __forceinline__ void stem_strip_compute(const bf16_t* patch, const bf16_t* wsm, const float* sbn, bf16_t* vm, int wave,
                                                   int lane, bool row0_valid) {
    const int lr = lane & 15, g = lane >> 4;
    const int xo = wave * 16 + lr;                          // conv column of this lane
    // AP_STEM_PASSES = 1: all 64 output channels in one pass over K (80 accumulator registers, 124 VGPRs, no
    // spill as long as the K-block loop is NOT unrolled: unrolling makes hipcc pre-compute every fragment address
    // and spill accumulators to scratch).  = 2: two passes of 32 channels (79 VGPRs), measured 15 % slower.
    constexpr int NH = AP_STEM_PASSES, FNH = 4 / NH;
    static_assert(NH == 1, "the pooled rows may reuse the patch's LDS: a second pass would read a clobbered patch");
#pragma unroll 1
    for (int half = 0; half < NH; ++half) {
        f32x4 acc[5][FNH];
#pragma unroll
        for (int fm = 0; fm < 5; ++fm)
#pragma unroll
            for (int fn = 0; fn < FNH; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k-block = kernel row (order: stem_kb_of); lane group g = taps 2g, 2g+1.  xf[(fm + rot) % 5] holds input row 2 fm + kb
        u32x4 xf[5];
        const bf16_t* xbase = patch + (2 * xo + 2 * g) * 4;
        const bf16_t* wbase = wsm + (half * FNH * 16 + lr) * SWLD + g * 8;
        stem_sfor<0, SKB>([&](auto ST) {
            constexpr int step = decltype(ST)::value, kb = stem_kb_of(step);
            constexpr bool first = step == 0 || step == 4;   // first step of a parity: all five rows; then one new row per step
            constexpr int rot = first ? 0 : (step < 4 ? step : step - 4);
            u32x4 wf[FNH];
#pragma unroll
            for (int fn = 0; fn < FNH; ++fn) {
                if (STEM_ABLATE & 128) wf[fn] = u32x4{0x3c003c00u + kb, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                else wf[fn] = *(const u32x4*)(wbase + fn * 16 * SWLD + kb * 32);
            }
            if constexpr (first) {
#pragma unroll
                for (int fm = 0; fm < 5; ++fm) {
                    if (STEM_ABLATE & 128) xf[fm] = u32x4{0x3c003c00u, 0x3c003c00u + kb, 0x3c003c00u, 0x3c003c00u};
                    else xf[fm] = *(const u32x4*)(xbase + (2 * fm + kb) * FPW * 4);
                }
            } else {
                // rows 2 fm + kb for fm = 0..3 are the previous step's rows of fm + 1; the slot the previous fm = 0 left takes row 8 + kb
                if (STEM_ABLATE & 128) xf[(4 + rot) % 5] = u32x4{0x3c003c00u, 0x3c003c00u + kb, 0x3c003c00u, 0x3c003c00u};
                else xf[(4 + rot) % 5] = *(const u32x4*)(xbase + (8 + kb) * FPW * 4);
            }
#pragma unroll
            for (int fm = 0; fm < 5; ++fm)
#pragma unroll
                for (int fn = 0; fn < FNH; ++fn) {
                    if (STEM_ABLATE & 4) asm volatile("" : "+v"(acc[fm][fn]) : "v"(wf[fn]), "v"(xf[(fm + rot) % 5]));
                    else acc[fm][fn] = ap_mfma16(
                        __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[(fm + rot) % 5]), acc[fm][fn]);
                }
            __builtin_amdgcn_sched_barrier(0);               // one kernel row at a time (hoisted reads of later rows spill the accumulators)
        });
        if constexpr (OVERLAY) __syncthreads();              // every wave is done with the patch: vm may overwrite it
#pragma unroll
        for (int fn = 0; fn < FNH; ++fn) {
            if (STEM_ABLATE & 32) { asm volatile("" ::"v"(acc[0][fn]), "v"(acc[1][fn]), "v"(acc[2][fn]), "v"(acc[3][fn]), "v"(acc[4][fn])); continue; }
            const int ch = (half * FNH + fn) * 16 + g * 4;
            const float4 sc = *(const float4*)(sbn + ch), sh = *(const float4*)(sbn + 64 + ch);
            float v[5][4];
#pragma unroll
            for (int fm = 0; fm < 5; ++fm) {
                v[fm][0] = fmaxf(acc[fm][fn][0] * sc.x + sh.x, 0.f);
                v[fm][1] = fmaxf(acc[fm][fn][1] * sc.y + sh.y, 0.f);
                v[fm][2] = fmaxf(acc[fm][fn][2] * sc.z + sh.z, 0.f);
                v[fm][3] = fmaxf(acc[fm][fn][3] * sc.w + sh.w, 0.f);
            }
            if (!row0_valid) { v[0][0] = v[0][1] = v[0][2] = v[0][3] = 0.f; }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint2 o;
                o.x = pack_bf16x2(fmaxf(fmaxf(v[2 * pr][0], v[2 * pr + 1][0]), v[2 * pr + 2][0]),
                                  fmaxf(fmaxf(v[2 * pr][1], v[2 * pr + 1][1]), v[2 * pr + 2][1]));
                o.y = pack_bf16x2(fmaxf(fmaxf(v[2 * pr][2], v[2 * pr + 1][2]), v[2 * pr + 2][2]),
                                  fmaxf(fmaxf(v[2 * pr][3], v[2 * pr + 1][3]), v[2 * pr + 2][3]));
                // rows of 128 B: 16-byte chunk index XOR-swizzled by the pixel (a lane's 16 pixels would all hit one bank)
                *(uint2*)(vm + ((pr * SO + xo) * SC + ((((ch >> 3) ^ (xo & 7)) << 3) | (ch & 7)))) = o;
            }
        }
    }
}

// The pooling pass of a strip by NT threads (t0 = this thread's index among them): horizontal 3-maximum over vm, 16-byte stores of
// the two pooled rows py0, py0 + 1 of image n; rng: the caller's fp16 range sentinel (the pooled maxima carry an overflow)
template <int NT>
__device__ __forceinline__ void stem_strip_pool(const bf16_t* vm, bf16_t* __restrict__ y, int n, int py0, int t0, uint32_t& rng) {
    // horizontal 3-maximum on the PACKED 16-bit values: every pooled candidate is a post-ReLU value (>= +0), and for non-negative
    // bf16 / fp16 the integer order of the bit patterns is the order of the values, so one v_pk_max_i16 per two channels and
    // neighbour replaces the unpack (conversion), two fp32 maxima and the re-packing (52 -> 8 VALU per 8-channel item)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    auto pkmax = [](uint32_t a, uint32_t b) -> uint32_t {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
    };
    for (int i = t0; i < 2 * PO * 8; i += NT) {             // (pr, px, 8-channel chunk)
        if (STEM_ABLATE & 16) break;
        const int c8 = i & 7, t = i >> 3, px = t % PO, pr = t / PO;
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int cx = 2 * px + dx;
            if (cx < 0) continue;
            const u32x4 q = *(const u32x4*)(vm + ((pr * SO + cx) * SC + ((c8 ^ (cx & 7)) << 3)));
            o.x = pkmax(o.x, q.x); o.y = pkmax(o.y, q.y); o.z = pkmax(o.z, q.z); o.w = pkmax(o.w, q.w);
        }
        ap_rng_note2(rng, o.x, o.y); ap_rng_note2(rng, o.z, o.w);
        if (STEM_ABLATE & 8) asm volatile("" ::"v"(o)); else
        *(u32x4*)(y + (((size_t)n * PO + py0 + pr) * PO + px) * SC + c8 * 8) = o;
    }
}

__global__ void __launch_bounds__(448, 4) stem_pool_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int n_split, const bf16_t* __restrict__ wpk,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, bf16_t* __restrict__ y, int* range_flag) {
    constexpr int WBYTES = 64 * SWLD * 2, PBYTES = (FROWS * FPW * 4 + 32) * 2;
    constexpr int VBYTES = 2 * SO * SC * 2;
    constexpr int PVBYTES = PBYTES > VBYTES ? PBYTES : VBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WBYTES + PVBYTES];   // 58 KB: two workgroups per CU
    __shared__ __attribute__((aligned(16))) float sbn[128];   // BatchNorm scale | shift (read from L2 with the prologue batch)
    bf16_t* wsm = (bf16_t*)lds;
    bf16_t* patch = (bf16_t*)(lds + WBYTES);
    bf16_t* vm = (bf16_t*)(lds + WBYTES);                    // [2][112][64] vertically pooled rows, over the dead patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int strip = blockIdx.x, n = blockIdx.y, py0 = strip * 2;
    // Every global load of the prologue is issued before the first one is consumed (weights: 5 x 16 B per thread from
    // L2; input strip: 2 x 3 float4 per thread from HBM): as rolled load -> wait -> store loops these were 4 + 2
    // dependent memory round trips per workgroup, about half of its lifetime.
    constexpr int WIT = (WBYTES / 16 + 447) / 448, XIT = (FROWS * 56 + 447) / 448;
    const float* xin = (n < n_split ? x0 + (size_t)n * 3 * IMG * IMG : x1 + (size_t)(n - n_split) * 3 * IMG * IMG);
    const int iy0 = 4 * py0 - 5;
    u32x4 wreg[WIT];
    float4 xin0[XIT], xin1[XIT], xin2[XIT];
    const float bnv = tid < 64 ? scale[tid] : shift[tid & 63];
    // (addresses are clamped instead of predicated: a load inside a divergent branch makes hipcc wait for it at the
    //  branch's end, which re-serialises the round trips; out-of-range values are zeroed when they are consumed)
#pragma unroll
    for (int k = 0; k < WIT; ++k) {
        const int i = tid + k * 448;
        if (STEM_ABLATE & 2) wreg[k] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        else wreg[k] = ((const u32x4*)wpk)[i < WBYTES / 16 ? i : WBYTES / 16 - 1];
    }
#pragma unroll
    for (int k = 0; k < XIT; ++k) {                         // (row, 4 pixels) of the three channel planes
        const int i0 = tid + k * 448, i = i0 < FROWS * 56 ? i0 : FROWS * 56 - 1;
        const int x4 = i % 56, row = i / 56, iy = iy0 + row, iyc = iy < 0 ? 0 : iy > IMG - 1 ? IMG - 1 : iy;
        const float* src = xin + (size_t)iyc * IMG + 4 * x4;
        if (STEM_ABLATE & 1) { xin0[k] = xin1[k] = xin2[k] = make_float4(0.5f, 0.25f, 0.125f, 1.f); continue; }
        xin0[k] = *(const float4*)src;
        xin1[k] = *(const float4*)(src + (size_t)IMG * IMG);
        xin2[k] = *(const float4*)(src + (size_t)2 * IMG * IMG);
    }
    if (tid < 128) sbn[tid] = bnv;
#pragma unroll
    for (int k = 0; k < WIT; ++k) {
        const int i = tid + k * 448;
        if (i < WBYTES / 16) ((u32x4*)wsm)[i] = wreg[k];
    }
    uint32_t rng_in = 0u;                                    // fp16 range sentinel over the converted crop values
#pragma unroll
    for (int k = 0; k < XIT; ++k) {                         // three channel planes -> 4 x [c0 c1 c2 0]
        const int i = tid + k * 448, x4 = i % 56, row = i / 56;
        if (i >= FROWS * 56) continue;
        const bool inside = (unsigned)(iy0 + row) < (unsigned)IMG;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v0 = inside ? xin0[k] : zero4, v1 = inside ? xin1[k] : zero4, v2 = inside ? xin2[k] : zero4;
        uint2* d = (uint2*)(patch + (row * FPW + 4 * x4 + 3) * 4);
        d[0] = make_uint2(pack_bf16x2(v0.x, v1.x), pack_bf16x2(v2.x, 0.f));
        d[1] = make_uint2(pack_bf16x2(v0.y, v1.y), pack_bf16x2(v2.y, 0.f));
        d[2] = make_uint2(pack_bf16x2(v0.z, v1.z), pack_bf16x2(v2.z, 0.f));
        d[3] = make_uint2(pack_bf16x2(v0.w, v1.w), pack_bf16x2(v2.w, 0.f));
        // fp16 storage: a crop value beyond +-65504 becomes +-inf HERE, and what it turns into downstream (NaN with the sign bit set) is
        // cleared by the ReLU before any epilogue's sentinel could see it -- the conversion of the input is watched like a store
        ap_rng_note4(rng_in, d[0].x, d[0].y, d[1].x, d[1].y, false);
        ap_rng_note4(rng_in, d[2].x, d[2].y, d[3].x, d[3].y, false);
    }
    for (int i = tid; i < FROWS * 8; i += 448) {           // left 3 / right 5 pad pixels of every row
        const int q = i % 8, row = i / 8;
        const int px = q < 3 ? q : 227 + (q - 3);
        *(uint2*)(patch + (row * FPW + px) * 4) = make_uint2(0u, 0u);
    }
    if (tid < 32) patch[FROWS * FPW * 4 + tid] = 0;
    __syncthreads();

    stem_strip_compute<true>(patch, wsm, sbn, vm, wave, lane, py0 > 0);
    __syncthreads();
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h)
    stem_strip_pool<448>(vm, y, n, py0, tid, rng);
    ap_rng_note(rng, rng_in);
    ap_rng_flush(range_flag, rng);
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the fused stem + max-pool (round 6): one workgroup per CU walks a contiguous range of strips (an image at the
// size of a pass), its 12 waves split by role.
//   waves 0-7   compute.  Wave w owns the conv columns 14 w - 1 .. 14 w + 14 -- the three conv columns of each of its 7 pooled
//               columns, so the horizontal 3-maximum is finished in registers (DPP row shifts) and stored from there; eight waves
//               = two per SIMD (seven 16-column waves leave one SIMD half empty: the two extra columns cost no time).  A strip is
//               4 conv rows, not 5: the conv row a strip shares with the one above (the top row of its upper pooling window) is
//               CARRIED in registers from the previous strip of the same walk (its BatchNorm output, 16 values per lane), which
//               takes a fifth off the MFMAs and the epilogue.  A walk that starts inside an image computes the strip above its
//               first one without storing it.  Every conv output is the same MFMA sequence on the same operands as in the
//               strip kernel and every maximum sees the same values: same bits.
//   waves 8-11  feed: convert the 13 crop rows of strip i + 1 into the other patch buffer (their global loads were issued an
//               iteration earlier and have had a whole strip to land) and issue the loads of strip i + 2.
// ONE workgroup barrier per strip.  The weights are read once per workgroup instead of once per strip, the rows a strip shares with
// the one above come out of the L2 the same workgroup filled two strips earlier, and of what a strip kernel serialises (load round
// trip -> conversion -> K loop -> epilogue -> pooling pass through LDS -> stores) only the K loop and the epilogue are left on the
// compute waves' path.  What bounds it (cycle stamps, tools/probes/stem_trace.py; PMC): the SIMD's VALU issue port -- MFMAs (16
// cycles each), the epilogue's and the feeding waves' VALU instructions hardly overlap, so the kernel is written to need few of
// each: packed BatchNorm (v_pk_fma_f32), ReLU after the vertical maximum, no accumulator zeroing (constant C operand in the first
// K step), the input's fp16 range sentinel on the fp32 values, 32-bit address arithmetic in the feeding waves.
constexpr int S2_CW = 8, S2_CT = S2_CW * 64, S2_HT = 256, S2_NT = S2_CT + S2_HT;
constexpr int S2_ROWS = 13;                                                            // crop rows of a 4-conv-row strip
constexpr int S2_WBYTES = 64 * SWLD * 2, S2_PBYTES = (S2_ROWS * FPW * 4 + 32) * 2;
constexpr int S2_LDS = S2_WBYTES + 2 * S2_PBYTES + 128 * 4;                           // 78 KB
constexpr int S2_XIT = (S2_ROWS * 56 + S2_HT - 1) / S2_HT;                            // (row, 4 pixels) items per feeding thread
__global__ void __launch_bounds__(S2_NT) stem_pool2_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int n_split,
                                                           const bf16_t* __restrict__ wpk, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, bf16_t* __restrict__ y, int* range_flag,
                                                           int total, int per, unsigned long long* dbg) {
#ifdef AP_TRACE   // cycle stamps of workgroup 3, strips 8-11 of its range: wave 0 (compute) slots 0.., wave 8 (feed) slots 64.. (tools/probes/stem_trace.py)
#define S2STAMP(base, slot) do { if (dbg && blockIdx.x == 3 && lane == 0 && it >= 8 && it < 12) dbg[(base) + (it - 8) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2STAMP(base, slot) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    bf16_t* wsm = (bf16_t*)lds2;
    unsigned char* pbase = lds2 + S2_WBYTES;
    float* sbn = (float*)(pbase + 2 * S2_PBYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Lfirst = blockIdx.x * per, nstore = (total - Lfirst < per ? total - Lfirst : per);
    if (nstore <= 0) return;
    const int warm = Lfirst % (PO / 2) ? 1 : 0;             // the walk starts inside an image: the strip above it runs first, unstored
    const int L0 = Lfirst - warm, nit = nstore + warm;
    for (int i = tid; i < S2_WBYTES / 16; i += S2_NT) ((u32x4*)wsm)[i] = ((const u32x4*)wpk)[i];
    if (tid < 128) sbn[tid] = tid < 64 ? scale[tid] : shift[tid - 64];
    for (int i = tid; i < 2 * S2_ROWS * 8; i += S2_NT) {    // left 3 / right 5 pad pixels of every row of both patches: written once
        const int q = i % 8, row = (i / 8) % S2_ROWS, b = i / (8 * S2_ROWS);
        const int px = q < 3 ? q : 227 + (q - 3);
        *(uint2*)((bf16_t*)(pbase + b * S2_PBYTES) + (row * FPW + px) * 4) = make_uint2(0u, 0u);
    }
    if (tid < 64) ((bf16_t*)(pbase + (tid >> 5) * S2_PBYTES))[S2_ROWS * FPW * 4 + (tid & 31)] = 0;

    if (wave >= S2_CW) {
        // ---- feeding waves: crop rows 8 s - 3 .. 8 s + 9 of strip s in
        const int ht = tid - S2_CT;
        float4 xr[S2_XIT][3];
        float rng_abs = 0.f;                                 // fp16 range sentinel: largest |crop value| converted (NaN propagates)
        uint32_t ioff[S2_XIT], poff[S2_XIT];                 // (row, 4-pixel group) of the item / its byte offset in the patch
#pragma unroll
        for (int k = 0; k < S2_XIT; ++k) {
            const int i0 = ht + k * S2_HT, i = i0 < S2_ROWS * 56 ? i0 : S2_ROWS * 56 - 1;
            ioff[k] = (uint32_t)(i / 56) << 16 | (uint32_t)(i % 56);
            poff[k] = (uint32_t)(((i / 56) * FPW + 4 * (i % 56) + 3) * 8);
        }
        auto issue = [&](int L) {
            const int n = L / (PO / 2), iy0 = 8 * (L - n * (PO / 2)) - 3;
            const char* xin = (const char*)(n < n_split ? x0 + (size_t)n * 3 * IMG * IMG : x1 + (size_t)(n - n_split) * 3 * IMG * IMG);
#pragma unroll
            for (int k = 0; k < S2_XIT; ++k) {
                const int x4 = ioff[k] & 0xffff, iy = iy0 + (int)(ioff[k] >> 16), iyc = iy < 0 ? 0 : iy > IMG - 1 ? IMG - 1 : iy;
                const uint32_t off = (uint32_t)(iyc * IMG + 4 * x4) * 4u;
                if (STEM_ABLATE & 1) { xr[k][0] = xr[k][1] = xr[k][2] = make_float4(0.5f, 0.25f, 0.125f, 1.f); continue; }
                xr[k][0] = *(const float4*)(xin + off);
                xr[k][1] = *(const float4*)(xin + (off + (uint32_t)(IMG * IMG * 4)));
                xr[k][2] = *(const float4*)(xin + (off + (uint32_t)(2 * IMG * IMG * 4)));
            }
        };
        auto fill = [&](int L, unsigned char* patch) {
            const int n = L / (PO / 2), strip = L - n * (PO / 2), iy0 = 8 * strip - 3;
            const bool edge = strip == 0 || strip == PO / 2 - 1;   // only these strips have rows outside the image
#pragma unroll
            for (int k = 0; k < S2_XIT; ++k) {
                if (ht + k * S2_HT >= S2_ROWS * 56) continue;
                float4 v0 = xr[k][0], v1 = xr[k][1], v2 = xr[k][2];
                if (edge && (unsigned)(iy0 + (int)(ioff[k] >> 16)) >= (unsigned)IMG) v0 = v1 = v2 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef AP_F16
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v0.x), "v"(v0.y));
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v0.z), "v"(v0.w));
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v1.x), "v"(v1.y));
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v1.z), "v"(v1.w));
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v2.x), "v"(v2.y));
                asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(rng_abs) : "v"(v2.z), "v"(v2.w));
#endif
                // a lane's four pixels are 32 contiguous bytes at 24 mod 32: 8 + 16 + 8 bytes (the 16-byte store is aligned)
                unsigned char* d = patch + poff[k];
                *(uint2*)d = make_uint2(pack_bf16x2(v0.x, v1.x), pack_bf16x2(v2.x, 0.f));
                *(u32x4*)(d + 8) = u32x4{pack_bf16x2(v0.y, v1.y), pack_bf16x2(v2.y, 0.f), pack_bf16x2(v0.z, v1.z), pack_bf16x2(v2.z, 0.f)};
                *(uint2*)(d + 24) = make_uint2(pack_bf16x2(v0.w, v1.w), pack_bf16x2(v2.w, 0.f));
            }
        };
        issue(L0);
        fill(L0, pbase);
        if (nit > 1) issue(L0 + 1);
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            if (wave == S2_CW) S2STAMP(64, 0);
#ifdef AP_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave == S2_CW) S2STAMP(64, 1);
#endif
            if (it + 1 < nit) {
                fill(L0 + it + 1, pbase + ((it + 1) & 1) * S2_PBYTES);
                if (wave == S2_CW) S2STAMP(64, 2);
                if (it + 2 < nit) issue(L0 + it + 2);
            }
            if (wave == S2_CW) S2STAMP(64, 3);
            __syncthreads();
            if (wave == S2_CW) S2STAMP(64, 4);
        }
#ifdef AP_F16
        // a crop value of 65520 or more (or a NaN) leaves the fp16 range at the conversion: what it turns into downstream (NaN with the
        // sign bit set) is cleared by the ReLU before any epilogue's sentinel could see it, so the input is watched here
        if (range_flag && !(rng_abs < 65520.f)) __hip_atomic_store(range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    } else {
        // ---- compute waves
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        auto pkmax = [](uint32_t a, uint32_t b) -> uint32_t {
            return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
        };
        const int lr = lane & 15, g = lane >> 4;
        // conv column of this lane; the pool's left border (column -1, lane 0 of wave 0) computes column 0 a second time -- the
        // maximum of (c0, c0, c1) is what the padded window gives; column 112 (lane 15 of wave 7) is never used
        const int xo = wave == 0 && lr == 0 ? 0 : 14 * wave - 1 + lr;
        const bool centre = (lr & 1) && lr < 14;             // lanes 1, 3, .. 13: pooled column 7 w + lr / 2 = conv columns xo - 1 .. xo + 1
        const int px = 7 * wave + (lr >> 1);
        uint32_t rng = 0u;                                   // fp16 range sentinel over the pooled maxima
        f32x2 carry[4][2];                                   // BatchNorm output of the previous strip's last conv row (channels fn * 16 + g * 4 ..)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) carry[fn][0] = carry[fn][1] = f32x2{0.f, 0.f};
        const uint32_t xoff = (uint32_t)((2 * xo + 2 * g) * 8);                       // byte offsets: patch pixel 2 xo + tap, weights row lr
        const bf16_t* wbase = wsm + lr * SWLD + g * 8;
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            const int L = L0 + it, n = L / (PO / 2), strip = L - n * (PO / 2);
            if (wave == 0) S2STAMP(0, 0);
            // K loop: kernel rows in the order 0, 2, 4, 6, 1, 3, 5 (stem_kb_of: the summation order of every 16-bit stem); conv row fm at
            // kernel row kb reads patch row 2 fm + kb, so within a parity one new row fragment per step (xf rotates)
            f32x4 acc[4][4];
            const unsigned char* xb = pbase + (it & 1) * S2_PBYTES + xoff;
            // K loop: kernel rows in the order 0, 2, 4, 6, 1, 3, 5 (stem_kb_of: the summation order of every 16-bit stem); conv row fm at
            // kernel row kb reads patch row 2 fm + kb, so within a parity a step needs ONE new row fragment (the others move up by one
            // conv row).  The fragments of step k + 1 are read while the MFMAs of step k run: weights in two register sets, the row
            // fragments in a ring of five (four in use, the fifth receives the next step's new row)
            u32x4 xf[5], wf[2][4];
            auto ldw = [&](int kb, u32x4 (&w)[4]) {
#pragma unroll
                for (int fn = 0; fn < 4; ++fn) w[fn] = *(const u32x4*)(wbase + fn * 16 * SWLD + kb * 32);
            };
            ldw(stem_kb_of(0), wf[0]);
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) xf[fm] = *(const u32x4*)(xb + (2 * fm + stem_kb_of(0)) * FPW * 8);
            stem_sfor<0, SKB>([&](auto ST) {
                constexpr int step = decltype(ST)::value;
                constexpr bool first = step == 0 || step == 4;
                constexpr int rot = first ? 0 : (step < 4 ? step : step - 4);
                if constexpr (step + 1 < SKB) {
                    constexpr int kn = stem_kb_of(step + 1);
                    ldw(kn, wf[(step + 1) & 1]);
                    if constexpr (step + 1 != 4) xf[(4 + rot) % 5] = *(const u32x4*)(xb + (6 + kn) * FPW * 8);
                }
#pragma unroll
                for (int fm = 0; fm < 4; ++fm)
#pragma unroll
                    for (int fn = 0; fn < 4; ++fn)
                        acc[fm][fn] = ap_mfma16(__builtin_bit_cast(bf16x8, wf[step & 1][fn]), __builtin_bit_cast(bf16x8, xf[(fm + rot) % 5]),
                                                step == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[fm][fn]);
                if constexpr (step + 1 == 4) {               // the second parity starts from four fresh rows
#pragma unroll
                    for (int fm = 0; fm < 4; ++fm) xf[fm] = *(const u32x4*)(xb + (2 * fm + stem_kb_of(4)) * FPW * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (wave == 0) S2STAMP(0, 1);
            // epilogue: BatchNorm (v_pk_fma_f32 = two v_fma_f32), then the two pooling windows over conv rows (carried, 0, 1) and
            // (1, 2, 3) with the ReLU folded into the shared row: max(relu a, relu b, relu c) = max3(a, b, max(c, 0)).  The first strip
            // of an image has no row above it: a carried 0 changes no maximum (the shared row's term is >= 0).  Maxima as
            // instructions: fmaxf() costs a canonicalising v_max_f32 x, x per operand in this mode.
            bf16_t* yrow = y + ((size_t)n * PO + 2 * strip) * PO * SC;
            const bool store = it >= warm;
            if (strip == 0) {
#pragma unroll
                for (int fn = 0; fn < 4; ++fn) carry[fn][0] = carry[fn][1] = f32x2{0.f, 0.f};
            }
            f32x4 bsc[4], bsh[4];                            // BatchNorm scale / shift of this lane's channels: one batch of LDS reads
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                bsc[fn] = *(const f32x4*)(sbn + fn * 16 + g * 4);
                bsh[fn] = *(const f32x4*)(sbn + 64 + fn * 16 + g * 4);
            }
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                if (STEM_ABLATE & 32) { asm volatile("" ::"v"(acc[0][fn]), "v"(acc[1][fn]), "v"(acc[2][fn]), "v"(acc[3][fn])); continue; }
                const int ch = fn * 16 + g * 4;
                const f32x2 sc0 = {bsc[fn][0], bsc[fn][1]}, sc1 = {bsc[fn][2], bsc[fn][3]};
                const f32x2 sh0 = {bsh[fn][0], bsh[fn][1]}, sh1 = {bsh[fn][2], bsh[fn][3]};
                f32x2 v0[4], v1[4];
#pragma unroll
                for (int fm = 0; fm < 4; ++fm) {
                    v0[fm] = __builtin_elementwise_fma(f32x2{acc[fm][fn][0], acc[fm][fn][1]}, sc0, sh0);
                    v1[fm] = __builtin_elementwise_fma(f32x2{acc[fm][fn][2], acc[fm][fn][3]}, sc1, sh1);
                }
                auto relu = [](float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; };
                auto max3 = [](float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
                const float t[4] = {relu(v0[1][0]), relu(v0[1][1]), relu(v1[1][0]), relu(v1[1][1])};
                float m[2][4];
                m[0][0] = max3(carry[fn][0][0], v0[0][0], t[0]); m[0][1] = max3(carry[fn][0][1], v0[0][1], t[1]);
                m[0][2] = max3(carry[fn][1][0], v1[0][0], t[2]); m[0][3] = max3(carry[fn][1][1], v1[0][1], t[3]);
                m[1][0] = max3(v0[2][0], v0[3][0], t[0]); m[1][1] = max3(v0[2][1], v0[3][1], t[1]);
                m[1][2] = max3(v1[2][0], v1[3][0], t[2]); m[1][3] = max3(v1[2][1], v1[3][1], t[3]);
                carry[fn][0] = v0[3]; carry[fn][1] = v1[3];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const uint2 o = make_uint2(pack_bf16x2(m[pr][0], m[pr][1]), pack_bf16x2(m[pr][2], m[pr][3]));
                    // horizontal 3-maximum on the packed values (>= +0: integer order = value order; a neighbour beyond the DPP row
                    // bound enters as 0: the pooling pass's rule, stem_strip_pool)
                    uint2 h;
                    h.x = pkmax(pkmax(o.x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o.x, 0x111, 0xf, 0xf, true)),
                                (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o.x, 0x101, 0xf, 0xf, true));
                    h.y = pkmax(pkmax(o.y, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o.y, 0x111, 0xf, 0xf, true)),
                                (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o.y, 0x101, 0xf, 0xf, true));
                    if (centre && store) {
                        ap_rng_note2(rng, h.x, h.y);
                        if (STEM_ABLATE & 8) asm volatile("" ::"v"(h)); else
                        *(uint2*)(yrow + ((size_t)pr * PO + px) * SC + ch) = h;
                    }
                }
            }
            if (wave == 0) S2STAMP(0, 2);
            __syncthreads();
            if (wave == 0) S2STAMP(0, 3);
        }
        ap_rng_flush(range_flag, rng);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused split-bf16 MFMA stem + 3x3/2 max-pool (bf16x2 parity mode): stem_pool_kernel's strip program with every operand as a
// (hi, lo) pair -- two weight images, two input patches, three MFMAs per product in stem_mfma_split_kernel's order (so the
// fp32 accumulators hold the same bits) -- the vertically pooled rows in fp32 and the split rounding AFTER the 3x3 maximum:
// v -> hi + lo (hi = bf16(v), lo = bf16(v - hi)) is monotonic in v (checked exhaustively over an exponent range on the host)
// and post-ReLU values are >= 0, so the maximum of the rounded values is the rounded maximum; packed the way
// maxpool_split_kernel packs it (see the store below) the result equals the split stem followed by that kernel bit for bit.  The 112 x 112 x 64 pairs (3.2 MB per image) never touch HBM; the two-kernel path was
// 11 % of a parity-mode step.  LDS 114 KiB: one workgroup (7 waves) per CU.
__global__ void __launch_bounds__(448) stem_pool_split_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                              int n_split, const bf16_t* __restrict__ wpk_hi,
                                                              const bf16_t* __restrict__ wpk_lo, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, bsplit_t* __restrict__ y) {
    constexpr int WBYTES = 64 * SWLD * 2, PEL = FROWS * FPW * 4 + 32, PBYTES = PEL * 2;
    constexpr int VBYTES = 2 * SO * SC * 4;                  // [2][112][64] fp32
    constexpr int PVBYTES = 2 * PBYTES > VBYTES ? 2 * PBYTES : VBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    bf16_t* wsm_hi = (bf16_t*)lds;
    bf16_t* wsm_lo = (bf16_t*)(lds + WBYTES);
    bf16_t* patch_hi = (bf16_t*)(lds + 2 * WBYTES);
    bf16_t* patch_lo = patch_hi + PEL;
    float* vm = (float*)(lds + 2 * WBYTES);                  // vertically pooled rows, over the dead patches
    float* sbn = (float*)(lds + 2 * WBYTES + PVBYTES);       // BatchNorm scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int strip = blockIdx.x, n = blockIdx.y, py0 = strip * 2;
    constexpr int XIT = (FROWS * 56 + 447) / 448;
    const float* xin = (n < n_split ? x0 + (size_t)n * 3 * IMG * IMG : x1 + (size_t)(n - n_split) * 3 * IMG * IMG);
    const int iy0 = 4 * py0 - 5;
    float4 xin0[XIT], xin1[XIT], xin2[XIT];
#pragma unroll
    for (int k = 0; k < XIT; ++k) {                         // (row, 4 pixels) of the three channel planes; addresses clamped
        const int i0 = tid + k * 448, i = i0 < FROWS * 56 ? i0 : FROWS * 56 - 1;
        const int x4 = i % 56, row = i / 56, iy = iy0 + row, iyc = iy < 0 ? 0 : iy > IMG - 1 ? IMG - 1 : iy;
        const float* src = xin + (size_t)iyc * IMG + 4 * x4;
        xin0[k] = *(const float4*)src;
        xin1[k] = *(const float4*)(src + (size_t)IMG * IMG);
        xin2[k] = *(const float4*)(src + (size_t)2 * IMG * IMG);
    }
    if (tid < 128) sbn[tid] = tid < 64 ? scale[tid] : shift[tid & 63];
    for (int i = tid; i < WBYTES / 16; i += 448) {
        ((u32x4*)wsm_hi)[i] = ((const u32x4*)wpk_hi)[i];
        ((u32x4*)wsm_lo)[i] = ((const u32x4*)wpk_lo)[i];
    }
    auto split2 = [](float a, float b, uint32_t& hi, uint32_t& lo) {      // two values -> packed hi pair, packed lo pair
        const bf16_t ha = f32_to_bf16(a), hb = f32_to_bf16(b);
        hi = (uint32_t)ha | ((uint32_t)hb << 16);
        lo = (uint32_t)f32_to_bf16(a - bf16_to_f32(ha)) | ((uint32_t)f32_to_bf16(b - bf16_to_f32(hb)) << 16);
    };
#pragma unroll
    for (int k = 0; k < XIT; ++k) {                         // three channel planes -> 4 x [c0 c1 c2 0], hi and lo
        const int i = tid + k * 448, x4 = i % 56, row = i / 56;
        if (i >= FROWS * 56) continue;
        const bool inside = (unsigned)(iy0 + row) < (unsigned)IMG;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v0 = inside ? xin0[k] : zero4, v1 = inside ? xin1[k] : zero4, v2 = inside ? xin2[k] : zero4;
        const int o = (row * FPW + 4 * x4 + 3) * 4;
        const float c0[4] = {v0.x, v0.y, v0.z, v0.w}, c1[4] = {v1.x, v1.y, v1.z, v1.w}, c2[4] = {v2.x, v2.y, v2.z, v2.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h01, l01, h2z, l2z;
            split2(c0[e], c1[e], h01, l01);
            split2(c2[e], 0.f, h2z, l2z);
            *(uint2*)(patch_hi + o + e * 4) = make_uint2(h01, h2z);
            *(uint2*)(patch_lo + o + e * 4) = make_uint2(l01, l2z);
        }
    }
    for (int i = tid; i < FROWS * 8; i += 448) {           // left 3 / right 5 pad pixels of every row
        const int q = i % 8, row = i / 8;
        const int px = q < 3 ? q : 227 + (q - 3);
        *(uint2*)(patch_hi + (row * FPW + px) * 4) = make_uint2(0u, 0u);
        *(uint2*)(patch_lo + (row * FPW + px) * 4) = make_uint2(0u, 0u);
    }
    if (tid < 32) { patch_hi[FROWS * FPW * 4 + tid] = 0; patch_lo[FROWS * FPW * 4 + tid] = 0; }
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    const int xo = wave * 16 + lr;                          // conv column of this lane
    const bool row0_valid = py0 > 0;                        // conv row 2*py0-1 exists
    f32x4 acc[5][4];
#pragma unroll
    for (int fm = 0; fm < 5; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = 0; kb < SKB; ++kb) {                      // k-block = kernel row; lane group g = taps 2g, 2g+1
        u32x4 wh[4], wl[4], xh[5], xl[5];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
            wh[fn] = *(const u32x4*)(wsm_hi + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
            wl[fn] = *(const u32x4*)(wsm_lo + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
        }
#pragma unroll
        for (int fm = 0; fm < 5; ++fm) {
            xh[fm] = *(const u32x4*)(patch_hi + ((2 * fm + kb) * FPW + 2 * xo + 2 * g) * 4);
            xl[fm] = *(const u32x4*)(patch_lo + ((2 * fm + kb) * FPW + 2 * xo + 2 * g) * 4);
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)                 // hi*hi, w_hi*x_lo, w_lo*x_hi: stem_mfma_split_kernel's order
#pragma unroll
            for (int fm = 0; fm < 5; ++fm)
#pragma unroll
                for (int fn = 0; fn < 4; ++fn)
                    acc[fm][fn] = ap_mfma16(
                        __builtin_bit_cast(bf16x8, pass == 2 ? wl[fn] : wh[fn]),
                        __builtin_bit_cast(bf16x8, pass == 1 ? xl[fm] : xh[fm]), acc[fm][fn]);
    }
    __syncthreads();                                         // every wave is done with the patches: vm may overwrite them
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
        const int ch = fn * 16 + g * 4;
        const float4 sc = *(const float4*)(sbn + ch), sh = *(const float4*)(sbn + 64 + ch);
        float v[5][4];
#pragma unroll
        for (int fm = 0; fm < 5; ++fm) {
            v[fm][0] = fmaxf(acc[fm][fn][0] * sc.x + sh.x, 0.f);
            v[fm][1] = fmaxf(acc[fm][fn][1] * sc.y + sh.y, 0.f);
            v[fm][2] = fmaxf(acc[fm][fn][2] * sc.z + sh.z, 0.f);
            v[fm][3] = fmaxf(acc[fm][fn][3] * sc.w + sh.w, 0.f);
        }
        if (!row0_valid) { v[0][0] = v[0][1] = v[0][2] = v[0][3] = 0.f; }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float4 o;
            o.x = fmaxf(fmaxf(v[2 * pr][0], v[2 * pr + 1][0]), v[2 * pr + 2][0]);
            o.y = fmaxf(fmaxf(v[2 * pr][1], v[2 * pr + 1][1]), v[2 * pr + 2][1]);
            o.z = fmaxf(fmaxf(v[2 * pr][2], v[2 * pr + 1][2]), v[2 * pr + 2][2]);
            o.w = fmaxf(fmaxf(v[2 * pr][3], v[2 * pr + 1][3]), v[2 * pr + 2][3]);
            // rows of 256 B: the 16-byte chunk index (4 channels) XOR-swizzled by the pixel
            *(float4*)(vm + ((pr * SO + xo) * SC + ((((ch >> 2) ^ (xo & 15)) << 2)))) = o;
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * PO * 8; i += 448) {           // (pr, px, 8-channel group)
        const int c8 = i & 7, t = i >> 3, px = t % PO, pr = t / PO;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 0.f;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int cx = 2 * px + dx;
            if (cx < 0) continue;
            const float4 a = *(const float4*)(vm + ((pr * SO + cx) * SC + (((2 * c8) ^ (cx & 15)) << 2)));
            const float4 b = *(const float4*)(vm + ((pr * SO + cx) * SC + (((2 * c8 + 1) ^ (cx & 15)) << 2)));
            m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
            m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
        }
        // split, rebuild hi + lo, split again: the two-kernel path rounds every conv output to a pair, takes the maximum of
        // the VALUES and packs that again -- where lo rounded up to half an ulp of hi (value = the midpoint of two hi
        // neighbours) the second packing picks the even neighbour, the same value as another pair.  One packing here would
        // give equal values in a few different bit patterns (seen as 2e-6 relative after 52 more layers)
        u32x4 hi, lo;
        split8_pack(m, hi, lo);
        split8_unpack(hi, lo, m);
        split8_pack(m, hi, lo);
        u32x4* dst = (u32x4*)(y + (((size_t)n * PO + py0 + pr) * PO + px) * SC + c8 * 8);
        dst[0] = hi; dst[1] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int total, int* range_flag) {
    constexpr int EPC = 16 / sizeof(T), CPP = SC / EPC;          // chunks per pixel
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cc = idx % CPP, pix = idx / CPP;
    const int ow = pix % PO, t = pix / PO, oh = t % PO, n = t / PO;
    float m[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
    for (int r = 0; r < 3; ++r) {
        const int ih = 2 * oh - 1 + r;
        if ((unsigned)ih >= (unsigned)SO) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = 2 * ow - 1 + s;
            if ((unsigned)iw >= (unsigned)SO) continue;
            const uint4 v = *(const uint4*)(x + (((size_t)n * SO + ih) * SO + iw) * SC + cc * EPC);
            if constexpr (sizeof(T) == 2) {
                float lo, hi;
                unpack_bf16x2(v.x, lo, hi); m[0] = fmaxf(m[0], lo); m[1] = fmaxf(m[1], hi);
                unpack_bf16x2(v.y, lo, hi); m[2] = fmaxf(m[2], lo); m[3] = fmaxf(m[3], hi);
                unpack_bf16x2(v.z, lo, hi); m[4] = fmaxf(m[4], lo); m[5] = fmaxf(m[5], hi);
                unpack_bf16x2(v.w, lo, hi); m[6] = fmaxf(m[6], lo); m[7] = fmaxf(m[7], hi);
            } else {
                m[0] = fmaxf(m[0], __builtin_bit_cast(float, v.x));
                m[1] = fmaxf(m[1], __builtin_bit_cast(float, v.y));
                m[2] = fmaxf(m[2], __builtin_bit_cast(float, v.z));
                m[3] = fmaxf(m[3], __builtin_bit_cast(float, v.w));
            }
        }
    }
    uint4 o;
    if constexpr (sizeof(T) == 2) {
        o.x = pack_bf16x2(m[0], m[1]); o.y = pack_bf16x2(m[2], m[3]);
        o.z = pack_bf16x2(m[4], m[5]); o.w = pack_bf16x2(m[6], m[7]);
        uint32_t rng = 0u;                                   // fp16 range sentinel (ap_common.h): the maxima carry the stem's overflow
        ap_rng_note2(rng, o.x, o.y); ap_rng_note2(rng, o.z, o.w);
        ap_rng_flush(range_flag, rng);
    } else {
        o.x = __builtin_bit_cast(uint32_t, m[0]); o.y = __builtin_bit_cast(uint32_t, m[1]);
        o.z = __builtin_bit_cast(uint32_t, m[2]); o.w = __builtin_bit_cast(uint32_t, m[3]);
    }
    *(uint4*)(y + (size_t)pix * SC + cc * EPC) = o;
}

// ------------------------------------------------------------------------------------------------
// split-bf16 pools: one thread = one group of 8 channels (32 bytes: 8 hi | 8 lo)
__global__ void __launch_bounds__(256) maxpool_split_kernel(const bsplit_t* __restrict__ x, bsplit_t* __restrict__ y, int total) {
    constexpr int GPP = SC / 8;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int gc = idx % GPP, pix = idx / GPP;
    const int ow = pix % PO, t = pix / PO, oh = t % PO, n = t / PO;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int r = 0; r < 3; ++r) {
        const int ih = 2 * oh - 1 + r;
        if ((unsigned)ih >= (unsigned)SO) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = 2 * ow - 1 + s;
            if ((unsigned)iw >= (unsigned)SO) continue;
            const u32x4* src = (const u32x4*)(x + (((size_t)n * SO + ih) * SO + iw) * SC + gc * 8);
            float v[8];
            split8_unpack(src[0], src[1], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    // hi + lo is exact in fp32 and packing it again returns the same parts: the maximum is one of the inputs, bit for bit
    u32x4 hi, lo;
    split8_pack(m, hi, lo);
    u32x4* dst = (u32x4*)(y + (size_t)pix * SC + gc * 8);
    dst[0] = hi; dst[1] = lo;
}

// AvgPool2d(7) over the 49 pixels of an image, all three storage kinds.  One workgroup = 32 consecutive channel groups (16-byte
// pieces: 4 fp32 / 8 bf16 / 8 split pairs) x 8 pixel partitions: thread (part, gl) sums pixels part, part + 8, ... of its
// group in that order, the 8 partial sums meet in LDS and are added in partition order -- a fixed association, the same on
// every run.  (One thread per group walking all 49 pixels left the chip at 4 waves per CU and the stage at 0.6-1.5 TB/s.)
// range_flag (optional, fp16 storage): a non-finite pooled value -- an activation that left the fp16 range somewhere in the stack
// reaches the last layer as inf / NaN, and a sum with a non-finite term is non-finite -- sets *range_flag (host-mapped word)
template <int EPC, typename LOAD>
__device__ __forceinline__ void avgpool_body(LOAD&& load, float* __restrict__ y, int C, int n_img, int* range_flag = nullptr) {
    __shared__ float part_sum[8][32][EPC];
    const int gl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int gpr = C / EPC, tiles = gpr / 32, n = blockIdx.x / tiles, g0 = (blockIdx.x % tiles) * 32;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    float v[7][EPC];
#pragma unroll
    for (int j = 0; j < 7; ++j) {                           // pixels part + 8 j (49 = 6 x 8 + 1: partition 0 has a seventh)
        const int p = part + 8 * j;
        load(n, p < 49 ? p : 48, g0 + gl, v[j]);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (part + 8 * j < 49) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += v[j][e];
        }
#pragma unroll
    for (int e = 0; e < EPC; ++e) part_sum[part][gl][e] = s[e];
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += part_sum[k][gl][e];
        float* dst = y + (size_t)n * C + (size_t)(g0 + gl) * EPC;
        bool bad = false;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            dst[e] = s[e] / 49.0f;
            bad |= !(fabsf(s[e]) <= 3.0e38f);               // inf or NaN
        }
        if (range_flag && bad) __hip_atomic_store(range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(256) avgpool_split_kernel(const bsplit_t* __restrict__ x, float* __restrict__ y, int C, int n_img) {
    avgpool_body<8>([&](int n, int p, int g, float (&v)[8]) {
        const u32x4* q = (const u32x4*)(x + ((size_t)n * 49 + p) * C + (size_t)g * 8);
        split8_unpack(q[0], q[1], v);
    }, y, C, n_img);
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int n_img, int* range_flag) {
    constexpr int EPC = 16 / sizeof(T);
    avgpool_body<EPC>([&](int n, int p, int g, float (&v)[EPC]) {
        const uint4 q = *(const uint4*)(x + ((size_t)n * 49 + p) * C + (size_t)g * EPC);
        if constexpr (sizeof(T) == 2) {
            unpack_bf16x2(q.x, v[0], v[1]); unpack_bf16x2(q.y, v[2], v[3]);
            unpack_bf16x2(q.z, v[4], v[5]); unpack_bf16x2(q.w, v[6], v[7]);
        } else {
            v[0] = __builtin_bit_cast(float, q.x); v[1] = __builtin_bit_cast(float, q.y);
            v[2] = __builtin_bit_cast(float, q.z); v[3] = __builtin_bit_cast(float, q.w);
        }
    }, y, C, n_img, range_flag);
}

// ------------------------------------------------------------------------------------------------
// Input contract of the network on the GPU: aerialpeople_crop.__getitem__
// (copenet/src/copenet/dsets/aerialpeople.py:125-141,174) + resize_with_pad (utils/utils.py:214-235):
//   img = frame[:, :, ::-1] / 255 -> crop -> cv2.resize (float image: INTER_LINEAR, half-pixel centres, float
//   coefficients, border clamp) to int(scale*w) x int(scale*h), scale = 224 / max(h, w) -> zero padding to
//   224x224, centred -> CHW float -> Normalize(mean, std).   One thread per output pixel, all three channels.
__global__ void preprocess_kernel(const unsigned char* __restrict__ frames, size_t frame_stride, int H, int W, int bgr,
                                  const int* __restrict__ crop, float* __restrict__ out, float* __restrict__ scale_out,
                                  int* __restrict__ pad_out) {
    const int i = blockIdx.y, px = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = crop[i * 4 + 0], y1 = crop[i * 4 + 1], x0 = crop[i * 4 + 2], x1 = crop[i * 4 + 3];
    const int h = y1 - y0, w = x1 - x0;
    const double scale = 224.0 / (double)(h > w ? h : w);
    const int dw = (int)(scale * w), dh = (int)(scale * h);
    const int pad_top = (224 - dh) / 2, pad_left = (224 - dw) / 2;
    if (px == 0) {
        scale_out[i] = (float)scale;
        pad_out[i * 2 + 0] = pad_left;
        pad_out[i * 2 + 1] = pad_top;
    }
    if (px >= 224 * 224) return;
    const int oy = px / 224, ox = px - oy * 224, dy = oy - pad_top, dx = ox - pad_left;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float v[3] = {0.f, 0.f, 0.f};
    if (dy >= 0 && dy < dh && dx >= 0 && dx < dw) {
        // cv::resize, INTER_LINEAR: scale_x = 1 / (dw / w) in double, fx = (float)((dx + 0.5) * scale_x - 0.5)
        const double sxd = 1.0 / ((double)dw / (double)w), syd = 1.0 / ((double)dh / (double)h);
        float fx = (float)((dx + 0.5) * sxd - 0.5), fy = (float)((dy + 0.5) * syd - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= sx; fy -= sy;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= w - 1) { fx = 0.f; sx = w - 1; }
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= h - 1) { fy = 0.f; sy = h - 1; }
        const int sx1 = sx + 1 < w ? sx + 1 : sx, sy1 = sy + 1 < h ? sy + 1 : sy;
        const unsigned char* f = frames + (size_t)i * frame_stride;
        const unsigned char* p00 = f + ((size_t)(y0 + sy) * W + x0 + sx) * 3;
        const unsigned char* p01 = f + ((size_t)(y0 + sy) * W + x0 + sx1) * 3;
        const unsigned char* p10 = f + ((size_t)(y0 + sy1) * W + x0 + sx) * 3;
        const unsigned char* p11 = f + ((size_t)(y0 + sy1) * W + x0 + sx1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = bgr ? 2 - c : c;
            const float a = p00[cs] / 255.f, b = p01[cs] / 255.f, cc = p10[cs] / 255.f, d = p11[cs] / 255.f;
            const float top = a * (1.f - fx) + b * fx, bot = cc * (1.f - fx) + d * fx;
            v[c] = top * (1.f - fy) + bot * fy;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((size_t)i * 3 + c) * 224 + oy) * 224 + ox] = (v[c] - mean[c]) / stdv[c];
}

}  // namespace

#ifndef AP_F16
hipError_t ap_launch_preprocess(const unsigned char* frames, size_t frame_stride, int n, int H, int W, int bgr,
                                const int* crop, float* out, float* scale_out, int* pad_out, hipStream_t st) {
    hipLaunchKernelGGL(preprocess_kernel, dim3((224 * 224 + 255) / 256, n), dim3(256), 0, st, frames, frame_stride, H, W,
                       bgr, crop, out, scale_out, pad_out);
    return hipGetLastError();
}
#endif

hipError_t ap_launch_stem_conv(const float* x, const float* w, const float* scale, const float* shift, void* y,
                               int n_img, int kind, hipStream_t st) {
    dim3 grid(SO / 16, SO / 16, n_img);
    if (kind == K_BF16)
        hipLaunchKernelGGL(stem_direct_kernel<bf16_t>, grid, dim3(256), 0, st, x, w, scale, shift, (bf16_t*)y);
#ifndef AP_F16                                               // (the fp16 set carries the 16-bit kind only)
    else if (kind == K_SPLIT)
        hipLaunchKernelGGL(stem_direct_kernel<bsplit_t>, grid, dim3(256), 0, st, x, w, scale, shift, (bsplit_t*)y);
    else if (kind == K_F32)
        hipLaunchKernelGGL(stem_direct_kernel<float>, grid, dim3(256), 0, st, x, w, scale, shift, (float*)y);
#endif
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t ap_launch_stem_conv_mfma(const float* x0, const float* x1, int n_split, const void* w_packed,
                                    const float* scale, const float* shift, void* y, int n_img, hipStream_t st) {
    hipLaunchKernelGGL(stem_mfma_kernel, dim3(SO / 16, SO / 16, n_img), dim3(256), 0, st, x0, x1, n_split,
                       (const bf16_t*)w_packed, scale, shift, (bf16_t*)y);
    return hipGetLastError();
}

#ifndef AP_F16
hipError_t ap_launch_stem_conv_mfma_split(const float* x0, const float* x1, int n_split, const void* w_hi, const void* w_lo,
                                          const float* scale, const float* shift, void* y, int n_img, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    constexpr int lds = (2 * 64 * SWLD + 2 * (SP * SPW * 4 + 32)) * 2;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)stem_mfma_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(stem_mfma_split_kernel, dim3(SO / 16, 1, n_img), dim3(256), lds, st, x0, x1, n_split,
                       (const bf16_t*)w_hi, (const bf16_t*)w_lo, scale, shift, (bsplit_t*)y);
    return hipGetLastError();
}
#endif

hipError_t ap_launch_stem_pool(const float* x0, const float* x1, int n_split, const void* w_packed, const float* scale,
                                const float* shift, void* y_pooled, int n_img, int* range_flag, int form, hipStream_t st,
                                unsigned long long* dbg) {
    if (form == 2) {                                         // persistent form: one workgroup per CU, contiguous ranges of strips
        static int n_cu_dev[AP_MAX_DEVICES] = {};
        int dev = 0;
        hipError_t e = ap_current_device(&dev);
        if (e != hipSuccess) return e;
        if (!n_cu_dev[dev]) {
            int n = 0;
            e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)stem_pool2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS);
            if (e != hipSuccess) return e;
            n_cu_dev[dev] = n > 0 ? n : 1;
        }
        const int total = n_img * (PO / 2);
        const int per = (total + n_cu_dev[dev] - 1) / n_cu_dev[dev];
        const int grid = (total + per - 1) / per;
        hipLaunchKernelGGL(stem_pool2_kernel, dim3(grid), dim3(S2_NT), S2_LDS, st, x0, x1, n_split, (const bf16_t*)w_packed, scale,
                           shift, (bf16_t*)y_pooled, range_flag, total, per, dbg);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(stem_pool_kernel, dim3(PO / 2, n_img), dim3(448), 0, st, x0, x1, n_split,
                       (const bf16_t*)w_packed, scale, shift, (bf16_t*)y_pooled, range_flag);
    return hipGetLastError();
}

#ifndef AP_F16
hipError_t ap_launch_stem_pool_split(const float* x0, const float* x1, int n_split, const void* w_hi, const void* w_lo,
                                     const float* scale, const float* shift, void* y_pooled, int n_img, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    constexpr int pb = 2 * (FROWS * FPW * 4 + 32) * 2, vb = 2 * SO * SC * 4;
    constexpr int lds = 2 * 64 * SWLD * 2 + (pb > vb ? pb : vb) + 128 * 4;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)stem_pool_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(stem_pool_split_kernel, dim3(PO / 2, n_img), dim3(448), lds, st, x0, x1, n_split, (const bf16_t*)w_hi,
                       (const bf16_t*)w_lo, scale, shift, (bsplit_t*)y_pooled);
    return hipGetLastError();
}
#endif

hipError_t ap_launch_maxpool(const void* x, void* y, int n_img, int kind, int* range_flag, hipStream_t st) {
    if (kind == K_BF16) {
        const int total = n_img * PO * PO * (SC / 8);
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)x,
                           (bf16_t*)y, total, range_flag);
    }
#ifndef AP_F16                                               // (the fp16 set carries the 16-bit kind only)
    else if (kind == K_SPLIT) {
        const int total = n_img * PO * PO * (SC / 8);
        hipLaunchKernelGGL(maxpool_split_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const bsplit_t*)x,
                           (bsplit_t*)y, total);
    } else if (kind == K_F32) {
        const int total = n_img * PO * PO * (SC / 4);
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)x,
                           (float*)y, total, range_flag);
    }
#endif
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t ap_launch_avgpool(const void* x, float* y, int n_img, int C, int kind, int* range_flag, hipStream_t st) {
    const int epc = kind == K_F32 ? 4 : 8;
    if (n_img <= 0 || C % (32 * epc)) return hipErrorInvalidValue;          // workgroup = 32 groups of epc channels
    const dim3 grid((unsigned)(n_img * (C / epc / 32)));
    if (kind == K_BF16) hipLaunchKernelGGL(avgpool_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, y, C, n_img, range_flag);
#ifndef AP_F16                                               // (the fp16 set carries the 16-bit kind only)
    else if (kind == K_SPLIT) hipLaunchKernelGGL(avgpool_split_kernel, grid, dim3(256), 0, st, (const bsplit_t*)x, y, C, n_img);
    else if (kind == K_F32) hipLaunchKernelGGL(avgpool_kernel<float>, grid, dim3(256), 0, st, (const float*)x, y, C, n_img, range_flag);
#endif
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

AP_NS_END
