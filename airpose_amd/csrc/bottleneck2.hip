// Identity bottleneck of layer1 (256 -> 64 -> 64 -> 256, 56 x 56) as ONE persistent kernel, second cut (bf16, gfx950).
//
//   out = relu(bn3(conv3(relu(bn2(conv2_3x3(relu(bn1(conv1(x))))))) + x)          Bottleneck.forward, model_copenet.py:27-45
//
// Same tiling as bottleneck.hip (14 x 14 output pixels, 16 x 16 halo, one workgroup of 8 waves per CU walking tiles), cut
// differently inside the tile:
//   * ALL weights stay in LDS: W1 (32 KB) and W2 (72 KB) for the whole kernel, W3 (32 KB) half resident (chunks 0, 1 in two
//     8-KB slots) and half (chunks 2, 3) fetched by LDS-DMA into the t1 buffer while that is dead (conv3 time).  The first cut
//     re-streamed all 136 KB of weights per tile through the texture path -- more than the 128 KB of x it read; this one
//     moves 32 KB per tile (four 1-KB pieces per wave).
//   * a wave owns two halo ROWS (16 pixels each, lane = column) from the x load to the store:
//       - x of its rows: 2 x 8 MFMA B fragments = 64 VGPRs, loaded global -> registers one tile AHEAD into a second set of
//         64, one instruction per MFMA group (nothing of x ever crosses LDS);
//       - conv1 on them, BN + ReLU, zero outside the image, t1 to LDS (the only tensor the waves exchange: conv2 needs the
//         rows above / below and the columns left / right);
//       - conv2 for the same two rows as OUTPUT rows (B fragments gathered from t1 with the tap's shift; the first / last halo
//         row and column give junk that is computed and never stored), BN + ReLU in registers;
//       - the packed result is the B fragment of conv3 (weight rows permuted so that a lane owns 8 consecutive channels per
//         fragment pair, conv_pair.hip's trick), conv3 in four chunks of 64 channels, + identity = THE SAME x FRAGMENT REGISTERS
//         conv1 consumed (B-fragment k of the x row and the 8 output channels a lane holds in chunk k/2, pair k&1 are the same
//         16 bytes of the same pixel), ReLU, 16-byte stores.
//   * 4 barriers per tile (t1 region free / t1 ready / conv2 done + W3 chunks 0,1 landed / W3 chunks 2,3 landed); no barrier
//     inside conv1 (64 MFMAs per wave), conv2 (144) or a conv3 chunk (16).
//   * every vmcnt hand-counted: per tile and wave the queue sees exactly  16 x loads | 2 DMA | 8 stores | 2 DMA | 8 stores
//     (stores of junk lanes carry an out-of-range buffer offset and are dropped by the hardware instead of being masked, so the
//     count never depends on the data).
// Results are bit-identical to conv1 -> conv2 -> conv3(+identity) through conv_pipe.hip (same K order per output element,
// same epilogue expression, same bf16 rounding points): tests/test_gpu_parity.py.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

#ifndef B2_SAFE
#define B2_SAFE 0
#endif
// timing-only builds (results wrong): 1 no x loads after the prologue | 2 no stores | 4 no W3 DMA | 8 no MFMAs | 16 no epilogue math |
// 32 every store is issued out of range (dropped: no write traffic) | 64 stores as 8 consecutive lanes per 128-byte line
#ifndef B2_ABLATE
#define B2_ABLATE 0
#endif


namespace {

constexpr int B2_TS = 14;
// LDS map (bytes), per variant.  DS = first block of layer1: cin = 64, conv3 carries the folded downsample branch as a second
// K segment of 64 (W3 rows of 128), no identity
template <bool DS, bool TAIL = false> struct B2Map {
    static constexpr int CIN = DS ? 64 : 256;
    static constexpr int NK1 = CIN / 32;                     // K groups of conv1 = x fragments per pixel row
    static constexpr int CH3 = DS ? 16384 : 8192;            // one W3 chunk (64 channels x K3): [K3 / 64 halves][64 rows][128 B]
    static constexpr int NP3 = CH3 / 8192;                   // DMA pieces per wave and chunk
    static constexpr int W1 = 0;                             // [CIN / 64 K chunks][64 rows][128 B]
    static constexpr int W2 = CIN * 128;                     // [9 taps][64 rows][128 B]
    static constexpr int T1 = W2 + 73728;                    // [16 x 16 halo pixels][128 B]; W3 chunks 2, 3 during conv3
    static constexpr int W3 = T1 + 32768;                    // W3 chunks 0, 1
    static constexpr int TAB = W3 + 2 * CH3;                 // s1 h1 s2 h2 (64 floats each), s3 h3 (256 each)
    static constexpr int TOTAL = TAB + 3072 + (TAIL ? 1024 : 0);   // tail variant: + s1n h1n (128 floats each)
    // vector-memory operations of a wave per tile: NX loads, 16 stores, 4 NP3 DMA pieces; tail variant: + 8 DMA pieces (the next
    // block's conv1 weights in eight 8-KB units) and 8 stores (its output)
    static constexpr int Q = 2 * NK1 + 4 * NP3 + 16 + (TAIL ? 16 : 0);
    static_assert(!(DS && TAIL), "the tail variant is the identity block");
    static_assert(2 * CH3 <= 32768 && TOTAL <= 160 * 1024 && Q - 2 <= 63, "chunks 2, 3 fit the t1 buffer; LDS per CU; vmcnt is 6 bits");
};
constexpr int T_S1 = 0, T_H1 = 256, T_S2 = 512, T_H2 = 768, T_S3 = 1024, T_H3 = 2048, T_S1N = 3072, T_H1N = 3584;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B2_SAFE ? 0 : N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_read_f32x4(uint32_t addr) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ void lds_write_b128(uint32_t addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ u32x4 gload_b128(const unsigned char* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    return r;
}
// SGPR base + 32-bit lane offset (one VGPR per address instead of two; the launcher keeps every tensor below 4 GB)
template <int OFF> __device__ __forceinline__ u32x4 gload_b128_s(uint32_t off, const unsigned char* base) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(off), "s"(base), "n"(OFF) : "memory");
    return r;
}
template <int OFF> __device__ __forceinline__ void gstore_b128_s(uint32_t off, unsigned char* base, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ::"v"(off), "v"(v), "s"(base), "n"(OFF) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
__device__ __forceinline__ void mm(f32x4& c, const u32x4& w, const u32x4& x) {
    if (B2_ABLATE & 8) asm volatile("" : "+v"(c) : "v"(w), "v"(x));
    else c = ap_mfma16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm(AP_CVTPK_ASM " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t relu_pk_bf16(uint32_t u) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(u));
    return r;
}
// channel (within its 64) of row rho of a 64-row weight tile: fragment f = rho >> 4, row i of it; lane group g4 of fragment
// pair q = f >> 1 owns channels q*32 + g4*8 .. + 7 (conv_pair.hip: pr_row_channel)
__host__ __device__ __forceinline__ int b2_row_channel(int rho) {
    const int f = rho >> 4, i = rho & 15;
    return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);
}
// BN (+ identity) + ReLU + bf16 of the 8 consecutive channels a lane holds in fragments (2q, 2q+1): the expression of the
// stand-alone kernels' epilogue (fma, add, max, round), two values per instruction
__device__ __forceinline__ u32x4 bn8(const f32x4& lo, const f32x4& hi, const f32x4& s0, const f32x4& s1, const f32x4& h0,
                                     const f32x4& h1, const u32x4* res, uint32_t& rng) {
    if (B2_ABLATE & 16) {
        u32x4 o = res ? *res : u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        asm volatile("" : "+v"(o) : "v"(lo), "v"(hi), "v"(s0), "v"(s1), "v"(h0), "v"(h1));
        return o;
    }
    f32x2 v0 = __builtin_elementwise_fma(lo.xy, s0.xy, h0.xy), v1 = __builtin_elementwise_fma(lo.zw, s0.zw, h0.zw);
    f32x2 v2 = __builtin_elementwise_fma(hi.xy, s1.xy, h1.xy), v3 = __builtin_elementwise_fma(hi.zw, s1.zw, h1.zw);
    if (res) {
        const uint32_t r0 = (*res).x, r1 = (*res).y, r2 = (*res).z, r3 = (*res).w;
#ifdef AP_F16
        { float a_ = v0.x, b_ = v0.y; ap_res_add2(a_, b_, r0); v0 = f32x2{a_, b_}; }
        { float a_ = v1.x, b_ = v1.y; ap_res_add2(a_, b_, r1); v1 = f32x2{a_, b_}; }
        { float a_ = v2.x, b_ = v2.y; ap_res_add2(a_, b_, r2); v2 = f32x2{a_, b_}; }
        { float a_ = v3.x, b_ = v3.y; ap_res_add2(a_, b_, r3); v3 = f32x2{a_, b_}; }
#else
        { float a_, b_; unpack_bf16x2(r0, a_, b_); v0 += f32x2{a_, b_}; }
        { float a_, b_; unpack_bf16x2(r1, a_, b_); v1 += f32x2{a_, b_}; }
        { float a_, b_; unpack_bf16x2(r2, a_, b_); v2 += f32x2{a_, b_}; }
        { float a_, b_; unpack_bf16x2(r3, a_, b_); v3 += f32x2{a_, b_}; }
#endif
    }
    u32x4 o;
    o.x = relu_pk_bf16(cvt_pk_bf16(v0.x, v0.y)); o.y = relu_pk_bf16(cvt_pk_bf16(v1.x, v1.y));
    o.z = relu_pk_bf16(cvt_pk_bf16(v2.x, v2.y)); o.w = relu_pk_bf16(cvt_pk_bf16(v3.x, v3.y));
    ap_rng_note2(rng, o.x, o.y); ap_rng_note2(rng, o.z, o.w);   // fp16 range sentinel (ap_common.h)
    return o;
}

template <bool DS, bool TAIL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
bneck2_kernel(const BneckArgs a) {
    using M = B2Map<DS, TAIL>;
    constexpr int L_W1 = M::W1, L_W2 = M::W2, L_T1 = M::T1, L_W3 = M::W3, L_TAB = M::TAB, CIN = M::CIN, NK1 = M::NK1, NX = 2 * M::NK1;
    constexpr int CH3 = M::CH3, K3 = DS ? 128 : 64, Q = M::Q;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the bf16 set

    // ---------------------------------------------------------------- once per workgroup: W1, W2, BatchNorm tables -> LDS
    // tile = 64 rows x 128 B, row rho = channel b2_row_channel(rho), 16-byte chunk c at position c ^ (rho & 7)
    {
        const unsigned char* w1 = (const unsigned char*)a.w1;
        const unsigned char* w2 = (const unsigned char*)a.w2;
        for (int idx = tid; idx < CIN * 8; idx += 512) {
            const int kc = idx >> 9, rho = (idx >> 3) & 63, c = (idx & 7) ^ (rho & 7);
            *(u32x4*)(smem + L_W1 + idx * 16) = *(const u32x4*)(w1 + ((size_t)b2_row_channel(rho) * CIN + kc * 64 + c * 8) * 2);
        }
        for (int idx = tid; idx < 4608; idx += 512) {
            const int tap = idx >> 9, rho = (idx >> 3) & 63, c = (idx & 7) ^ (rho & 7);
            *(u32x4*)(smem + L_W2 + idx * 16) = *(const u32x4*)(w2 + ((size_t)b2_row_channel(rho) * 576 + tap * 64 + c * 8) * 2);
        }
        float* tb = (float*)(smem + L_TAB);
        if (tid < 64) {
            tb[T_S1 / 4 + tid] = a.s1[tid]; tb[T_H1 / 4 + tid] = a.h1[tid];
            tb[T_S2 / 4 + tid] = a.s2[tid]; tb[T_H2 / 4 + tid] = a.h2[tid];
        }
        if (tid < 256) { tb[T_S3 / 4 + tid] = a.s3[tid]; tb[T_H3 / 4 + tid] = a.h3[tid]; }
        if constexpr (TAIL) { if (tid < 128) { tb[T_S1N / 4 + tid] = a.s1n[tid]; tb[T_H1N / 4 + tid] = a.h1n[tid]; } }
    }
    // W3 by LDS-DMA: chunk cc = channels cc*64 .. +63 (rows permuted); per 64-wide K half one 1-KB piece (8 rows) per wave
    const int prow = lane >> 3;
    const unsigned char* w3src =
        (const unsigned char*)a.w3 + ((size_t)b2_row_channel(wave * 8 + prow) * K3 + (((lane & 7) ^ prow) * 8)) * 2;
    auto dma3 = [&](auto CC) {
        constexpr int cc = CC;
        if (B2_ABLATE & 4) return;
        constexpr int off = cc < 2 ? L_W3 + cc * CH3 : L_T1 + (cc - 2) * CH3;
#pragma unroll
        for (int kh = 0; kh < K3 / 64; ++kh)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w3src + cc * (64 * K3 * 2) + kh * 128),
                                             (__attribute__((address_space(3))) void*)(smem + off + kh * 8192 + wave * 1024), 16, 0, 0);
    };
    // tail variant: conv1 of the next block, W1' [128][256], in eight 8-KB units of the W3 chunk shape: unit u = output channels
    // (u / 4) * 64 .. + 63 (rows permuted like every tile here), K columns (u % 4) * 64 .. + 63; one 1-KB piece per wave and unit.
    // Units pass through the six 8-KB slots S0, S1 (the W3 slots) and T0 .. T3 (the t1 buffer) behind the W3 chunks
    const uint32_t w1noff = (uint32_t)((b2_row_channel(wave * 8 + prow) * 256 + (((lane & 7) ^ prow) * 8)) * 2);
    auto dmaU = [&](auto UU, auto SL) {                      // SL: 0, 1 = S0, S1; 2 .. 5 = T0 .. T3
        constexpr int u = UU, sl = SL;
        if (B2_ABLATE & 4) return;
        constexpr int off = sl < 2 ? L_W3 + sl * 8192 : L_T1 + (sl - 2) * 8192;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const unsigned char*)a.w1n + w1noff + (u / 4) * (64 * 256 * 2) + (u % 4) * 128),
                                         (__attribute__((address_space(3))) void*)(smem + off + wave * 1024), 16, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---------------------------------------------------------------- lane constants: derived per tile from the lane id
    // (lane_consts below) -- kept in registers across the tile loop they cost ~20 VGPRs the kernel does not have
    struct LaneC { uint32_t aW1, aW2, aW3, aW3b, tw, tab, tbc[3]; };
    auto lane_consts = [&](int ll, LaneC& c) {
        const int lr = ll & 15, g4 = ll >> 4;
        const uint32_t fb = lds0 + lr * 128 + (uint32_t)((g4 ^ (lr & 7)) << 4);
        c.aW1 = fb + L_W1; c.aW2 = fb + L_W2;                // A fragment f of a tile: + f*2048; K half 1: ^ 64
        c.aW3 = fb + L_W3; c.aW3b = fb + L_T1;
        c.tw = fb + L_T1 + wave * 4096;                      // t1 of halo row 2*wave (+2048: the next row), pair 1: ^ 64
#pragma unroll
        for (int j = 0; j < 3; ++j) {                        // t1 B fragment of column lr + dc (clamped), row 0, K half 0
            int cx = lr + j - 1;
            cx = cx < 0 ? 0 : (cx > 15 ? 15 : cx);
            c.tbc[j] = lds0 + L_T1 + cx * 128 + ((g4 ^ (cx & 7)) << 4);
        }
        c.tab = lds0 + L_TAB + g4 * 32;
    };
    int rowoff[4];                                           // halo rows 2*wave - 1 .. 2*wave + 2, clamped (junk rows only)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int r = 2 * wave - 1 + k;
        r = r < 0 ? 0 : (r > 15 ? 15 : r);
        rowoff[k] = r * 2048;
    }

    const int H = a.H, W = a.W;
    const unsigned char* xg = (const unsigned char*)a.x;
    unsigned char* yg = (unsigned char*)a.y;
    // every kernel-argument load completes here (a scalar load pending inside the loop would share lgkmcnt with the counted
    // fragment reads, and scalar loads return out of order)
    const int tpi = a.tiles_per_img, tlx = a.tiles_x, total = a.total;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, (int)((uint32_t)a.N * H * W * 512u), 0x00020000);   // (< 0xffff0000: launcher)
    const int yeven = TAIL ? a.y_even : 0;
    const auto t1rsrc = __builtin_amdgcn_make_buffer_rsrc(TAIL ? (unsigned char*)a.t1n : yg, 0, (int)((uint32_t)a.N * H * W * 256u), 0x00020000);
    asm volatile("" ::"s"(tpi), "s"(tlx), "s"(total), "s"(H), "s"(W), "s"(xg), "s"(yg), "s"(a.w3), "s"(a.w1n), "s"(yeven));

    // x of the wave's two halo rows of tile T as B fragments: xs[g*NK1 + k] = channels k*32 + g4*8 .. + 7 of pixel (row g, col lr).
    // xptr: the two row pointers; xone<i>: one 16-byte piece per lane (fragment i/2 of row i%2; fragments 2m, 2m+1 of a pixel share a 128-byte line)
    auto xptr = [&](int T, uint32_t (&xp)[2], int lr, int g4) {
        const int n = T / tpi, r = T - n * tpi, ty = r / tlx, tx = r - ty * tlx;
        int cx = tx * B2_TS - 1 + lr;
        cx = cx < 0 ? 0 : (cx >= W ? W - 1 : cx);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int cy = ty * B2_TS - 1 + 2 * wave + g;
            cy = cy < 0 ? 0 : (cy >= H ? H - 1 : cy);
            xp[g] = ((((uint32_t)n * H + cy) * W + cx) * CIN + g4 * 8) * 2;          // (unsigned: up to 4 GB of x)
        }
    };
    auto xone = [&](u32x4 (&xs)[NX], uint32_t (&xp)[2], auto II) {      // load i: fragment i/2 of row i%2
        constexpr int i = decltype(II)::value, k = i / 2, g = i % 2;
        if (B2_ABLATE & 1) return;
        xs[g * NK1 + k] = gload_b128_s<k * 64>(xp[g], xg);
    };
    // where the output rows of tile T go.  Stores go through a buffer descriptor over y: lanes that have
    // nothing to store -- junk columns (lane 0 / 15), junk rows (halo row 0 / 15) -- carry an offset behind
    // the end of the buffer and the hardware drops them, so every wave ISSUES the same 16 stores per tile whatever it holds
    // (the hand-counted vmcnt waits depend on that) without a branch or a dump line
    struct OutRows { uint32_t off[2], off1[2]; };            // off1: the pixel's row of t1n (tail variant)
    constexpr uint32_t OOB = 0xffff0000u;
    auto optr = [&](int T, OutRows& o, int lr, int g4) {
        const int n = T / tpi, r = T - n * tpi, ty = r / tlx, tx = r - ty * tlx;
        const int hx = tx * B2_TS - 1 + lr;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int R = 2 * wave + g, hy = ty * B2_TS - 1 + R;
            const bool ok = R >= 1 && R <= 14 && lr >= 1 && lr <= 14 && !(B2_ABLATE & 32);
            // y_even: only the pixels a stride-2 reader of y samples
            o.off[g] = ok && !(yeven && ((hy | hx) & 1)) ? ((((uint32_t)n * H + hy) * W + hx) * 256 + g4 * 8) * 2 : OOB;
            if constexpr (TAIL) o.off1[g] = ok ? ((((uint32_t)n * H + hy) * W + hx) * 128 + g4 * 8) * 2 : OOB;
        }
    };
    auto rdA = [&](u32x4 (&w)[4], uint32_t addr) {
        w[0] = lds_read_b128<0>(addr); w[1] = lds_read_b128<2048>(addr);
        w[2] = lds_read_b128<4096>(addr); w[3] = lds_read_b128<6144>(addr);
    };

    // scale / shift of the 8 channels a lane holds in fragment pair q: t = {s[0..3], s[4..7], h[0..3], h[4..7]}
    auto rdT = [&](f32x4 (&t)[4], uint32_t tabq, auto OS, auto OH) {
        constexpr int os = decltype(OS)::value, oh = decltype(OH)::value;
        t[0] = lds_read_f32x4<os>(tabq); t[1] = lds_read_f32x4<os + 16>(tabq);
        t[2] = lds_read_f32x4<oh>(tabq); t[3] = lds_read_f32x4<oh + 16>(tabq);
    };
    // ---------------------------------------------------------------- one tile.  xc: its x fragments (requested a tile ago),
    // xn: the set the next tile's are requested into
#ifdef AP_TRACE   // cycle stamps of waves 0 and 4 of workgroup 0, sixth tile (24 slots each): tools/probes/bneck2_trace.py
    int tile_no = 0;
    unsigned long long stamps[17];
#define B2STAMP(i) do { if (a.dbg && blockIdx.x == 0 && tile_no == 5 && (wave & 3) == 0) { \
        stamps[i] = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } } while (0)
#else
#define B2STAMP(i) do { } while (0)
#endif
    auto tile = [&](int T, int Tn, u32x4 (&xc)[NX], u32x4 (&xn)[NX]) {
        // the lane id passes through an empty asm per tile: what derives from it (36 addresses in conv2 alone) is then
        // recomputed where it is used (one VALU each) instead of being hoisted out of the tile loop and spilled
        int ll = lane;
        asm volatile("" : "+v"(ll));
        LaneC lc;
        lane_consts(ll, lc);
        const uint32_t aW1 = lc.aW1, aW2 = lc.aW2, aW3 = lc.aW3, aW3b = lc.aW3b, tw = lc.tw, tab = lc.tab;
        const uint32_t tbc[3] = {lc.tbc[0], lc.tbc[1], lc.tbc[2]};
        const int lr = ll & 15, g4 = ll >> 4;
        const int n = T / tpi, r = T - n * tpi, ty = r / tlx, tx = r - ty * tlx;
        const int hx = tx * B2_TS - 1 + lr;
        const bool colin = hx >= 0 && hx < W;
        bool inimg[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int hy = ty * B2_TS - 1 + 2 * wave + g;
            inimg[g] = colin && hy >= 0 && hy < H;
        }
        OutRows opp;
        optr(T, opp, lr, g4);
        // The x fragments of the NEXT tile are requested ONE instruction at a time behind the first NX MFMA groups of this tile
        // (conv1, then conv2): a burst of 128 KB per CU at the tile start left HBM idle for the rest of the tile, and bursts of
        // 4 x 8 waves still stalled the issuing waves on a full memory pipe.  The queue of a wave per tile is
        //     NX x loads | DMA | 8 stores | DMA | 8 stores                                         (Q operations)
        // and conv1 waits for its fragments pair by pair (they were requested one tile ago).
        // (Measured and not kept: the 16 output pieces held in registers and stored one per MFMA group during the next tile's
        // conv1 / conv2 -- the first block, which has the registers, ran 4-10 % slower: what the 822 MB of output cost is the
        // write bandwidth itself (3.5 TB/s store-only in this kernel's skeleton), not the moment the stores are issued.)
        uint32_t xp[2];
        xptr(Tn, xp, lr, g4);
        B2STAMP(0);
        auto xwait = [](u32x4 (&xc)[NX], auto KK) {      // (the array as a parameter: asm operands on a captured array reference do not compile)
            constexpr int k = decltype(KK)::value;
            wait_vmcnt<Q - 2 - k>();                         // behind loads 2k, 2k+1 of the last tile: Q - 2k - 2 of that tile, k of this one
            asm volatile("" : "+v"(xc[k]), "+v"(xc[NK1 + k]));
        };

        f32x4 acc[8];
        u32x4 wf[2][4];
        // ------------------------------------------------------------ conv1: NK1 K groups of 32, 4 row fragments x 2 pixel rows
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        rdA(wf[0], aW1);
        sfor<0, NK1>([&](auto K) {
            constexpr int k = K;
            if constexpr (k + 1 < NK1) {
                rdA(wf[(k + 1) & 1], (aW1 + ((k + 1) >> 1) * 8192) ^ (((k + 1) & 1) ? 64u : 0u));
                wait_lgkmcnt<4>();
            } else wait_lgkmcnt<0>();
            xwait(xc, K);
#pragma unroll
            for (int f = 0; f < 4; ++f) { mm(acc[f], wf[k & 1][f], xc[k]); mm(acc[4 + f], wf[k & 1][f], xc[NK1 + k]); }
            __builtin_amdgcn_sched_barrier(0);
            xone(xn, xp, K);
        });
        B2STAMP(1);
        // every wave is done with W3 chunks 2, 3 of the last tile (they lie in the t1 region)
        __builtin_amdgcn_s_barrier();
        B2STAMP(2);
        f32x4 tq[4];
        sfor<0, 2>([&](auto Q) {                             // (one table read serves both pixel rows)
            constexpr int q = Q;
            rdT(tq, tab, std::integral_constant<int, T_S1 + q * 128>{}, std::integral_constant<int, T_H1 + q * 128>{});
            wait_lgkmcnt<0>();
            sfor<0, 2>([&](auto GG) {
                constexpr int g = GG;
                u32x4 o = bn8(acc[g * 4 + 2 * q], acc[g * 4 + 2 * q + 1], tq[0], tq[1], tq[2], tq[3], nullptr, rng);
                if (!inimg[g]) o = u32x4{0u, 0u, 0u, 0u};    // conv2 pads t1 with zeros, not with conv1 of zeros
                lds_write_b128<g * 2048>(q ? tw ^ 64u : tw, o);
            });
        });
        wait_lgkmcnt<0>();
        B2STAMP(3);
        __builtin_amdgcn_s_barrier();                        // t1 complete
        B2STAMP(4);

        // ------------------------------------------------------------ conv2: 9 taps x 2 K halves; output rows = the same two
        // halo rows (row 0 / 15 and column 0 / 15 of the halo give junk)
        u32x4 bfr[2][2];
        auto rdB = [&](u32x4 (&b)[2], auto TT, auto SS) {
            constexpr int t = TT, s = SS, dr = t / 3 - 1, dc = t % 3 - 1;
            b[0] = lds_read_b128<0>((tbc[dc + 1] + rowoff[dr + 1]) ^ (s ? 64u : 0u));
            b[1] = lds_read_b128<0>((tbc[dc + 1] + rowoff[dr + 2]) ^ (s ? 64u : 0u));
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        rdA(wf[0], aW2);
        rdB(bfr[0], I0{}, I0{});
        sfor<0, 18>([&](auto J) {
            constexpr int j = J;
            if constexpr (j + 1 < 18) {
                constexpr int t1 = (j + 1) >> 1, s1 = (j + 1) & 1;
                rdA(wf[(j + 1) & 1], (aW2 + t1 * 8192) ^ (s1 ? 64u : 0u));
                rdB(bfr[(j + 1) & 1], std::integral_constant<int, t1>{}, std::integral_constant<int, s1>{});
                wait_lgkmcnt<6>();
            } else wait_lgkmcnt<0>();
#pragma unroll
            for (int f = 0; f < 4; ++f) { mm(acc[f], wf[j & 1][f], bfr[j & 1][0]); mm(acc[4 + f], wf[j & 1][f], bfr[j & 1][1]); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NK1 + j < NX) xone(xn, xp, std::integral_constant<int, NK1 + j>{});
        });
        B2STAMP(5);
        u32x4 y[4];                                          // t2 of the two rows = conv3's B fragments (row g, K half q)
        sfor<0, 2>([&](auto Q) {
            constexpr int q = Q;
            rdT(tq, tab, std::integral_constant<int, T_S2 + q * 128>{}, std::integral_constant<int, T_H2 + q * 128>{});
            wait_lgkmcnt<0>();
            y[q] = bn8(acc[2 * q], acc[2 * q + 1], tq[0], tq[1], tq[2], tq[3], nullptr, rng);
            y[2 + q] = bn8(acc[4 + 2 * q], acc[4 + 2 * q + 1], tq[0], tq[1], tq[2], tq[3], nullptr, rng);
        });

        // ------------------------------------------------------------ conv3 in four chunks of 64 channels
        // own pieces of W3 chunks 0, 1 (requested in the last tile, behind them: 8 stores, NX x loads); the barrier also says
        // every wave is done reading t1
        B2STAMP(6);
        wait_vmcnt<(TAIL ? 4 : 8) + NX>();                  // (tail variant: behind them the 4 stores of the last conv1' chunk)
        __builtin_amdgcn_s_barrier();
        B2STAMP(7);
        dma3(I2{}); dma3(I3{});
        if constexpr (TAIL) { dmaU(I0{}, I4{}); dmaU(I1{}, I5{}); }      // units 0, 1 -> T2, T3
        sfor<0, 4>([&](auto CC) {
            constexpr int cc = CC;
            if constexpr (cc == 1) B2STAMP(8);
            if constexpr (cc == 3) B2STAMP(11);
            if constexpr (cc == 2) {
                B2STAMP(9);
                wait_vmcnt<TAIL ? 10 : 8>();                 // own pieces of chunks 2, 3; behind them the stores of chunks 0, 1 (+ units 0, 1)
                __builtin_amdgcn_s_barrier();                // ... everybody's; and the slots of chunks 0, 1 are free
                if constexpr (TAIL) { dmaU(I2{}, I0{}); dmaU(I3{}, I1{}); }   // units 2, 3 -> S0, S1
                else { dma3(I0{}); dma3(I1{}); }             // for the next tile
                B2STAMP(10);
            }
            const uint32_t base = cc < 2 ? aW3 + cc * CH3 : aW3b + (cc - 2) * CH3;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // K groups of 32: 0, 1 = t2 (the packed conv2 result); DS: 2, 3 = x of the pixel (the folded downsample branch,
            // model_copenet.py:41-42 -- the x fragments conv1 consumed)
            constexpr int NG = K3 / 32;
            auto grp = [&](uint32_t b, auto G) -> uint32_t { constexpr int g = decltype(G)::value; return (b + (g >> 1) * 8192) ^ ((g & 1) ? 64u : 0u); };
            rdA(wf[0], base);
            rdA(wf[1], base ^ 64u);
            sfor<0, NG>([&](auto S) {
                constexpr int sg = S;
                if constexpr (sg + 1 < NG) wait_lgkmcnt<4>(); else wait_lgkmcnt<0>();     // younger: the next group's fragments
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if constexpr (sg < 2) { mm(acc[f], wf[sg & 1][f], y[sg]); mm(acc[4 + f], wf[sg & 1][f], y[2 + sg]); }
                    else { mm(acc[f], wf[sg & 1][f], xc[sg - 2]); mm(acc[4 + f], wf[sg & 1][f], xc[NK1 + sg - 2]); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (sg + 2 < NG) rdA(wf[sg & 1], grp(base, std::integral_constant<int, sg + 2>{}));
            });
            if constexpr (cc == 1) B2STAMP(13);              // chunk 1 in detail: MFMAs issued | tables of pair 0 | pair 0 done | tables of pair 1
            sfor<0, 2>([&](auto Q) {
                constexpr int q = Q;
                constexpr int to = cc * 256 + q * 128;
                rdT(tq, tab, std::integral_constant<int, T_S3 + to>{}, std::integral_constant<int, T_H3 + to>{});
                wait_lgkmcnt<0>();
                if constexpr (cc == 1) B2STAMP(14 + 2 * q);
                sfor<0, 2>([&](auto GG) {
                    constexpr int g = GG;
                    const u32x4 o = bn8(acc[g * 4 + 2 * q], acc[g * 4 + 2 * q + 1], tq[0], tq[1], tq[2], tq[3],
                                        DS ? nullptr : &xc[DS ? 0 : g * 8 + cc * 2 + q], rng);
                    if constexpr (TAIL) xc[g * 8 + cc * 2 + q] = o;      // the block output replaces x: conv1' operand (same fragment)
                    if (B2_ABLATE & 2) asm volatile("" ::"v"(o), "v"(opp.off[g]));
                    else if (B2_ABLATE & 64) {               // shape experiment (data of the wrong pixels): 8 CONSECUTIVE lanes = one 128-byte line
                        const int R = 2 * wave + g, hy = ty * B2_TS - 1 + R, p = (ll >> 3) + 8 * q;
                        const bool ok = R >= 1 && R <= 14 && p >= 1 && p <= 14;
                        const uint32_t off = ok ? ((((uint32_t)n * H + hy) * W + tx * B2_TS - 1 + p) * 256) * 2 + (ll & 7) * 16 : 0xffff0000u;
                        __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, off + cc * 128, 0, 0);
                    } else __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, opp.off[g] + cc * 128 + q * 64, 0, 0);
                });
                if constexpr (cc == 1 && q == 0) B2STAMP(15);
            });
        });
        // ------------------------------------------------------------ tail variant: conv1 of the next block (256 -> 128) on the
        // block output held in xc, two chunks of 64 channels x two halves of K (four K groups of 32 = two units each).
        // Queue of the wave since chunks 2, 3 were requested:  c2 c3 u0 u1 | 8 st | u2 u3 | 8 st | u4 u5 | u6 u7 | 4 st | c0' c1' | 4 st
        if constexpr (TAIL) {
            auto grp = [&](uint32_t b, auto G) -> uint32_t { constexpr int g = decltype(G)::value; return (b + (g >> 1) * 8192) ^ ((g & 1) ? 64u : 0u); };
            auto sub = [&](auto KB, uint32_t base) {         // K groups kb .. kb + 3: units at base, base + 8192
                constexpr int kb = decltype(KB)::value;
                rdA(wf[0], base);
                rdA(wf[1], base ^ 64u);
                sfor<0, 4>([&](auto S) {
                    constexpr int sg = S;
                    if constexpr (sg + 1 < 4) wait_lgkmcnt<4>(); else wait_lgkmcnt<0>();
#pragma unroll
                    for (int f = 0; f < 4; ++f) { mm(acc[f], wf[sg & 1][f], xc[kb + sg]); mm(acc[4 + f], wf[sg & 1][f], xc[NK1 + kb + sg]); }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (sg + 2 < 4) rdA(wf[sg & 1], grp(base, std::integral_constant<int, sg + 2>{}));
                });
            };
            auto epi = [&](auto CB) {                        // BN + ReLU + 16-bit of chunk cb, 16-byte stores to t1n
                constexpr int cb = decltype(CB)::value;
                sfor<0, 2>([&](auto Q) {
                    constexpr int q = Q;
                    constexpr int to = cb * 256 + q * 128;
                    rdT(tq, tab, std::integral_constant<int, T_S1N + to>{}, std::integral_constant<int, T_H1N + to>{});
                    wait_lgkmcnt<0>();
                    sfor<0, 2>([&](auto GG) {
                        constexpr int g = GG;
                        const u32x4 o = bn8(acc[g * 4 + 2 * q], acc[g * 4 + 2 * q + 1], tq[0], tq[1], tq[2], tq[3], nullptr, rng);
                        if (B2_ABLATE & 2) asm volatile("" ::"v"(o), "v"(opp.off1[g]));
                        else __builtin_amdgcn_raw_buffer_store_b128(o, t1rsrc, opp.off1[g] + cb * 128 + q * 64, 0, 0);
                    });
                });
            };
            wait_vmcnt<18>();                                // own pieces of units 0, 1
            __builtin_amdgcn_s_barrier();                    // ... everybody's; T0, T1 (W3 chunks 2, 3) are free
            dmaU(I4{}, I2{}); dmaU(I5{}, I3{});              // units 4, 5 -> T0, T1
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            sub(I0{}, aW3b + 16384);                         // chunk A, K groups 0-3 (T2, T3)
            wait_vmcnt<10>();                                // units 2, 3
            __builtin_amdgcn_s_barrier();                    // T2, T3 free
            dmaU(I6{}, I4{}); dmaU(I7{}, I5{});              // units 6, 7 -> T2, T3
            sub(I4{}, aW3);                                  // chunk A, K groups 4-7 (S0, S1)
            epi(I0{});
            wait_vmcnt<6>();                                 // units 4, 5
            __builtin_amdgcn_s_barrier();                    // S0, S1 free
            dma3(I0{}); dma3(I1{});                          // W3 chunks 0, 1 of the next tile
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            sub(I0{}, aW3b);                                 // chunk B, K groups 0-3 (T0, T1)
            wait_vmcnt<6>();                                 // units 6, 7
            __builtin_amdgcn_s_barrier();
            sub(I4{}, aW3b + 16384);                         // chunk B, K groups 4-7 (T2, T3)
            epi(I1{});
        }
#ifdef AP_TRACE
        B2STAMP(12);
        if (a.dbg && blockIdx.x == 0 && tile_no == 5 && (wave & 3) == 0 && lane == 0)
            for (int i = 0; i < 17; ++i) a.dbg[(wave >> 2) * 24 + i] = stamps[i];
        ++tile_no;
#endif
    };

    // ---------------------------------------------------------------- prologue + tile loop (two x register sets, alternating)
    // Round r gives tile r*G + t to the workgroup whose slot is t.  Workgroups are dealt to the 8 XCDs round-robin
    // (blockIdx % 8), each XCD has its own L2, and neighbouring tiles share halo rows / columns: slot = (b % 8) * (G / 8) + b / 8
    // puts 32 CONSECUTIVE tiles (two whole images at 256 workgroups) on one XCD per round, so the halo is re-read from that
    // XCD's L2 instead of from HBM
    const int G = gridDim.x;
    int T = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    u32x4 xa[NX], xb[NX];
    {
        uint32_t xp0[2];
        xptr(T, xp0, lr, g4);
        if (B2_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { xa[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; xb[i] = xa[i]; }
        }
        sfor<0, NX>([&](auto II) { xone(xa, xp0, II); });
    }
    dma3(I0{}); dma3(I1{});
    wait_vmcnt<0>();
    __syncthreads();
    while (true) {
        int Tn = T + G;
        bool more = Tn < total;
        tile(T, more ? Tn : T, xa, xb);
        if (!more) break;
        T = Tn; Tn = T + G; more = Tn < total;
        tile(T, more ? Tn : T, xb, xa);
        if (!more) break;
        T = Tn;
    }
    wait_vmcnt<0>();                                         // no LDS-DMA may be in flight when the LDS is released
    ap_rng_flush(a.range_flag, rng);
}

}  // namespace

// layer1 bottleneck, second cut: ds = 0 identity block (cin = 256), ds = 1 first block (cin = 64, W3 rows = [conv3 | downsample])
hipError_t ap_launch_bneck2(BneckArgs a, int ds, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    if (a.H % B2_TS || a.W % B2_TS || a.N <= 0) return hipErrorInvalidValue;
    if ((size_t)a.N * a.H * a.W * 512 >= 0xffff0000ull) return hipErrorInvalidValue;      // 32-bit offsets, out-of-range marker
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!n_cu_dev[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)bneck2_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B2Map<false>::TOTAL);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)bneck2_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B2Map<true>::TOTAL);
        if (e != hipSuccess) return e;
        constexpr int lds_t = B2Map<false, true>::TOTAL;
        e = hipFuncSetAttribute((const void*)bneck2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_t);
        if (e != hipSuccess) return e;
        n_cu_dev[dev] = n;
    }
    a.tiles_x = a.W / B2_TS;
    a.tiles_per_img = a.tiles_x * (a.H / B2_TS);
    a.total = a.N * a.tiles_per_img;
    const int grid = a.total < n_cu_dev[dev] ? a.total : n_cu_dev[dev];
    if (a.w1n && (ds || !a.s1n || !a.h1n || !a.t1n)) return hipErrorInvalidValue;
    if ((size_t)a.N * a.H * a.W * 256 >= 0xffff0000ull) return hipErrorInvalidValue;
    constexpr int lds_ds = B2Map<true>::TOTAL, lds_id = B2Map<false>::TOTAL, lds_tail = B2Map<false, true>::TOTAL;
    if (ds) hipLaunchKernelGGL((bneck2_kernel<true, false>), dim3(grid), dim3(512), lds_ds, st, a);
    else if (a.w1n) hipLaunchKernelGGL((bneck2_kernel<false, true>), dim3(grid), dim3(512), lds_tail, st, a);
    else hipLaunchKernelGGL((bneck2_kernel<false, false>), dim3(grid), dim3(512), lds_id, st, a);
    return hipGetLastError();
}

AP_NS_END
