// Stem and pooling kernels of the ResNet-50 trunk for gfx950.
//   stem   conv 7x7 stride 2 pad 3 (3 -> 64) + BN + ReLU   model_copenet.py:57-60, 163-165
//   pool   MaxPool2d(3, stride 2, pad 1)                    model_copenet.py:61, 166
//   avg    AvgPool2d(7, stride 1) + view(B, -1)             model_copenet.py:66, 173-174
// The stem reads the caller's NCHW fp32 crop directly (input contract) and
// writes NHWC in the trunk's storage type, so no separate layout-conversion pass exists.
#include "ap_common.h"
#include "kernels.h"

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
        } else {
            *(float4*)(dst + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA stem (bf16 throughput path).  One workgroup = 16x16 conv outputs x 64 channels.
// GEMM view: M = pixels, N = 64, K' = 7 kernel rows x 24 (21 = 7 taps x 3 channels, 3 zero pad) = 168 -> 192.
// The input patch sits in LDS as bf16 [iy][ix][c] (channel-interleaved), so the 8 consecutive k' a lane feeds
// to v_mfma_f32_16x16x32_bf16 are 8 consecutive bf16 of one patch row: no im2col buffer.  k' >= 21 within a
// kernel row reads the neighbouring pixels' (finite) data against zero weights.
constexpr int SP = 37;                       // patch rows / cols for a 16x16 output tile
constexpr int SPW = 38;                      // patch row stride in pixels (even: keeps every 8-run dword aligned)
constexpr int SWLD = 200;                    // packed weight row stride (bf16): 400 B, conflict-free b128 reads
constexpr int SKB = 6;                       // 6 x 32 = 192 padded K

__global__ void __launch_bounds__(256) stem_mfma_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int n_split, const bf16_t* __restrict__ wpk,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, bf16_t* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) bf16_t wsm[64 * SWLD];
    __shared__ __attribute__((aligned(16))) bf16_t patch[SP * SPW * 3 + 32];
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
        patch[(py * SPW + px) * 3 + c] = f32_to_bf16(v);
    }
    if (tid < 32) patch[SP * SPW * 3 + tid] = 0;
    if (tid < SP * 3) patch[((tid / 3) * SPW + SP) * 3 + tid % 3] = 0;       // pad column
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < SKB; ++kb) {
        const int k0 = kb * 32 + g * 8;                     // first of this lane's 8 k'
        const int r = k0 / 24, t0 = k0 - r * 24;            // kernel row, offset inside the 24-wide row slot
        u32x4 wf[4], xf[4];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) wf[fn] = *(const u32x4*)(wsm + (fn * 16 + lr) * SWLD + kb * 32 + g * 8);
        if (r < 7) {
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) {
                const int ty = wave * 4 + fm;
                const uint32_t* pp = (const uint32_t*)(patch + ((2 * ty + r) * SPW + 2 * lr) * 3 + t0);
                xf[fm].x = pp[0]; xf[fm].y = pp[1]; xf[fm].z = pp[2]; xf[fm].w = pp[3];
            }
        } else {
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) xf[fm] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int fn = 0; fn < 4; ++fn)
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[fm]), acc[fm][fn], 0, 0, 0);
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
template <typename T>
__global__ void __launch_bounds__(256) maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int total) {
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
    } else {
        o.x = __builtin_bit_cast(uint32_t, m[0]); o.y = __builtin_bit_cast(uint32_t, m[1]);
        o.z = __builtin_bit_cast(uint32_t, m[2]); o.w = __builtin_bit_cast(uint32_t, m[3]);
    }
    *(uint4*)(y + (size_t)pix * SC + cc * EPC) = o;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, int C, int total) {
    constexpr int EPC = 16 / sizeof(T);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cpr = C / EPC, cc = idx % cpr, n = idx / cpr;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    const T* src = x + (size_t)n * 49 * C + cc * EPC;
    for (int p = 0; p < 49; ++p) {
        const uint4 v = *(const uint4*)(src + (size_t)p * C);
        if constexpr (sizeof(T) == 2) {
            float lo, hi;
            unpack_bf16x2(v.x, lo, hi); s[0] += lo; s[1] += hi;
            unpack_bf16x2(v.y, lo, hi); s[2] += lo; s[3] += hi;
            unpack_bf16x2(v.z, lo, hi); s[4] += lo; s[5] += hi;
            unpack_bf16x2(v.w, lo, hi); s[6] += lo; s[7] += hi;
        } else {
            s[0] += __builtin_bit_cast(float, v.x); s[1] += __builtin_bit_cast(float, v.y);
            s[2] += __builtin_bit_cast(float, v.z); s[3] += __builtin_bit_cast(float, v.w);
        }
    }
    float* dst = y + (size_t)n * C + cc * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) dst[e] = s[e] / 49.0f;
}

}  // namespace

hipError_t ap_launch_stem_conv(const float* x, const float* w, const float* scale, const float* shift, void* y,
                               int n_img, int is_bf16, hipStream_t st) {
    dim3 grid(SO / 16, SO / 16, n_img);
    if (is_bf16)
        hipLaunchKernelGGL(stem_direct_kernel<bf16_t>, grid, dim3(256), 0, st, x, w, scale, shift, (bf16_t*)y);
    else
        hipLaunchKernelGGL(stem_direct_kernel<float>, grid, dim3(256), 0, st, x, w, scale, shift, (float*)y);
    return hipGetLastError();
}

hipError_t ap_launch_stem_conv_mfma(const float* x0, const float* x1, int n_split, const void* w_packed,
                                    const float* scale, const float* shift, void* y, int n_img, hipStream_t st) {
    hipLaunchKernelGGL(stem_mfma_kernel, dim3(SO / 16, SO / 16, n_img), dim3(256), 0, st, x0, x1, n_split,
                       (const bf16_t*)w_packed, scale, shift, (bf16_t*)y);
    return hipGetLastError();
}

hipError_t ap_launch_maxpool(const void* x, void* y, int n_img, int is_bf16, hipStream_t st) {
    if (is_bf16) {
        const int total = n_img * PO * PO * (SC / 8);
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)x,
                           (bf16_t*)y, total);
    } else {
        const int total = n_img * PO * PO * (SC / 4);
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)x,
                           (float*)y, total);
    }
    return hipGetLastError();
}

hipError_t ap_launch_avgpool(const void* x, float* y, int n_img, int C, int is_bf16, hipStream_t st) {
    if (is_bf16) {
        const int total = n_img * (C / 8);
        hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)x, y, C,
                           total);
    } else {
        const int total = n_img * (C / 4);
        hipLaunchKernelGGL(avgpool_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)x, y, C,
                           total);
    }
    return hipGetLastError();
}
