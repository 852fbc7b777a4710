// conv3 of an identity bottleneck and conv1 of the NEXT bottleneck as ONE pixel-local kernel (bf16, gfx950).
//
//   block k :  out = relu(bn3(conv3(t2)) + x)            Bottleneck.forward, model_copenet.py:38-45
//   block k+1: t1' = relu(bn1(conv1(out)))               model_copenet.py:29-31
//
// Both convolutions are 1x1: a pixel's t1' depends on that pixel only, so `out` is written to HBM once (it is the next
// block's identity) and never read back for conv1 -- the 4P-channel tensor makes one HBM trip less per block boundary
// (layer2: 411 MB of 1.44 GB per pair at 512 images, layer3: 206 MB of 719 MB).
//
// Structure (MI355X-first, not a two-GEMM translation): a workgroup is FOUR waves that share nothing but the weight
// stream; each wave owns 16 pixels and keeps everything of them in registers:
//   * the t2 rows of its pixels as MFMA B fragments (P/32 x 4 VGPRs), loaded once per tile;
//   * conv3 in chunks of 128 output channels: acc3[8] (16 channels x 16 pixels each, v_mfma_f32_16x16x32_bf16);
//   * the chunk's epilogue (BN, + identity, ReLU, bf16) in registers; the packed result IS the B fragment of the second
//     GEMM: an MFMA D fragment holds 4 channels per lane, two neighbouring fragments 8 -- the 8 consecutive K values a
//     B fragment wants -- because the weight ROWS of a tile are permuted at pack time (row f*16 + i of a tile is channel
//     (f>>1)*32 + (i>>2)*8 + (f&1)*4 + (i&3) of its 128): a lane then owns 8 CONSECUTIVE channels per fragment pair, so the
//     identity load, the `out` store and the conv1 operand are all one aligned 16-byte piece per lane (16 pixel rows x 64 B
//     per wave instruction), with no LDS round trip and no cross-lane traffic;
//   * acc1[N1/16] for conv1, accumulated chunk by chunk in the K order of the stand-alone kernels (results are
//     bit-identical to conv3 followed by conv1 through conv_pipe.hip: same K-step order, same k-slot assignment, same
//     epilogue expression).
// Weights: ONE linear stream of 16-KiB tiles (128 rows x 64 K, swizzle baked in by pair_pack_kernel) in consumption order,
// fetched by LDS-DMA into a 4-slot ring, 4 pieces per wave and tile issued between the MFMA groups; one raw s_barrier per
// tile, every wait counted by hand (vmcnt retires in order; loads, LDS-DMA pieces and stores of a wave share the counter:
// a wait for "at most N younger operations" is exact when only DMA pieces follow and conservative when epilogue loads /
// stores sit among them).  -DPR_SAFE=1 turns every counted vmcnt into vmcnt(0) for cross-checking.
// Two workgroups (8 waves, <= 256 VGPRs) per CU: the waves of a SIMD belong to different workgroups and drift apart, so one
// wave's epilogue (VALU + HBM latency) runs under the other's MFMA groups.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

#ifndef PR_SAFE
#define PR_SAFE 0
#endif

namespace {

constexpr int PR_TILE = 16384;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PR_SAFE ? 0 : N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);                       // register-only MFMAs ignore "memory": keep them below the wait
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// (fp32 table rows are read as f32x4 at once: __builtin_bit_cast applied to ONE ELEMENT of an integer vector mis-compiles
// with this hipcc -- every element but the first comes out wrong; conv_pipe.hip's epilogue carries the same note)
template <int OFF> __device__ __forceinline__ f32x4 lds_read_f32x4(uint32_t addr) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// register load the compiler does not count (a load it counts is answered with vmcnt(0) beside LDS-DMA and would drain the
// weight ring): the destination is valid only behind the hand-placed wait that names it
// cache policy of the activation traffic: bit 0 = streaming (nt) loads of t2 / identity / x2 (each byte is read once; measured
// +0.3 % on the whole bench, 5 interleaved pairs, profiles/r04_pair_nt_ab.txt), bit 1 = nt stores (measured -0.5 %: off)
#ifndef PR_NT
#define PR_NT 1
#endif
template <int OFF> __device__ __forceinline__ u32x4 gload_b128(const unsigned char* p) {
    u32x4 r;
    if (PR_NT & 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void gstore_b128(unsigned char* p, const u32x4& v) {
    if (PR_NT & 2) __builtin_nontemporal_store(v, (u32x4*)p);
    else *(u32x4*)p = v;
}
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& w, const u32x4& x, const f32x4& c) {
    return ap_mfma16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
// two fp32 -> one dword of two bf16 (round to nearest even: the instruction the compiler emits for a (__bf16) cast, but
// once per PAIR -- the cast gives one v_cvt_pk per value plus shifts / ors to merge)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm(AP_CVTPK_ASM " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ReLU on two packed bf16: as signed 16-bit integers negative values (sign bit set, -0 included) are below zero, positive
// ones keep their order -- max(x, 0) per half; rounding is monotonic and keeps the sign, so relu-then-round == round-then-relu
__device__ __forceinline__ uint32_t relu_pk_bf16(uint32_t u) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(u));
    return r;
}

// channel (within its 128) of row rho of a weight tile: lane group g4 of fragment pair q = f >> 1 owns channels
// q*32 + g4*8 .. + 7 (fragment f = 2q + e holds e*4 .. e*4 + 3 of them)
__host__ __device__ __forceinline__ int pr_row_channel(int rho) {
    const int f = rho >> 4, i = rho & 15;
    return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);
}

// weight stream: for every 128-channel chunk nb of conv3: P/64 tiles of conv3 (K steps of 64), then for each 64-deep half
// kh of the chunk and each 128-row half hn of conv1: one tile of conv1.  Tile = 128 rows x 128 B, 16-byte chunk c of row
// rho at position c ^ (rho & 7); thread = one 16-byte chunk.
__global__ void __launch_bounds__(256) pair_pack_kernel(const bf16_t* __restrict__ w3, const bf16_t* __restrict__ w1,
                                                        unsigned char* __restrict__ dst, int KA, int C3, int N1) {
    const int KP = KA / 64, HN = N1 / 128, SPC = KP + 2 * HN, T = (C3 / 128) * SPC;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * 1024) return;
    const int t = idx >> 10, rho = (idx >> 3) & 127, c = (idx & 7) ^ (rho & 7);
    const int nb = t / SPC, j = t - nb * SPC, chl = pr_row_channel(rho);
    const bf16_t* src;
    if (j < KP) {
        src = w3 + (size_t)(nb * 128 + chl) * KA + j * 64 + c * 8;
    } else {
        const int g = j - KP, kh = g / HN, hn = g - kh * HN;
        src = w1 + (size_t)(hn * 128 + chl) * C3 + nb * 128 + kh * 64 + c * 8;
    }
    *(u32x4*)(dst + (size_t)idx * 16) = *(const u32x4*)src;
}

// Timing-only builds (results WRONG, times valid): -DPR_ABLATE=<bits>
//   1 no identity loads | 2 no stores | 4 no weight DMA after the prologue | 8 no MFMAs | 16 no epilogue arithmetic |
//   32 return behind the prologue | 64 identity loads as 8 rows x 128 B per instruction (shape experiment) | 128 stores likewise
#ifndef PR_ABLATE
#define PR_ABLATE 0
#endif

// P: channels of t2 (first K segment); P2: channels of the second K segment (the folded downsample branch of a stage's first
// block, model_copenet.py:41-42,97-102: x of the block sampled at the strided pixel, 1x1; 0 = identity block); C3: conv3
// output channels; N1: conv1 width of the next block (0 = none: conv3 alone); RES: the block input is added before the ReLU
// PG: pixel groups of 16 per wave.  Only PG = 1 is instantiated (two workgroups per CU, <= 256 registers); PG = 2 (32 pixels per
// wave, a whole SIMD's register file, one workgroup per CU) measured 4-10 % slower on the layer3 shapes and its knob was retired
// in round 5 -- layer3's identity blocks run on block_img.hip, which takes that idea to 224 pixels per wave.
template <int P, int P2, int C3, int N1, bool RES, int NW, int S, int D, bool IDB, int PG>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3 - PG, 3 - PG))) conv_pair_kernel(const PairArgs p) {
    constexpr int KA = P + P2, KP = KA / 64, HN = N1 / 128, KG = 2 * HN, NB = C3 / 128, SPC = KP + KG;
    constexpr int NXF = KA / 32, NXF1 = P / 32, NT = 64 * NW, LPW = 16 / NW, RING = S * PR_TILE;
    static_assert(!(IDB && !RES) && (RES || P2 > 0) && P % 64 == 0 && P2 % 64 == 0, "identity look-ahead needs an identity");
    constexpr int TAB3 = RING, TAB1 = TAB3 + 2 * C3 * 4;
    static_assert(S >= 3 && S - 1 <= SPC && NB % 2 == 0 && (LPW == 4 || LPW == 2), "tail waits / chunk pairs / piece placement");
    static_assert((S & (S - 1)) == 0, "ring offsets wrap with a mask");
    static_assert(D == 8 || D == 16, "fragment registers: half a tile or a whole tile ahead");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    static_assert(PG == 1 || PG == 2, "one or two groups of 16 pixels per wave");
    bool mok[PG], sok[PG];                                   // sok: the pixel's row of `out` is stored
    size_t mc[PG];
    const unsigned char *t2p[PG], *resp[PG], *x2p[PG];
    unsigned char *outp[PG], *t1p[PG];
    const unsigned char* wnext = (const unsigned char*)p.wstream + (size_t)(wave * LPW) * 1024 + lane * 16;   // next tile to issue
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int m = blockIdx.x * (16 * NW * PG) + (wave * PG + pg) * 16 + lr;
        mok[pg] = m < p.M;
        mc[pg] = mok[pg] ? (size_t)m : (size_t)(p.M - 1);   // ragged tail: loads clamped, stores masked
        // NHWC rows, or (p.*_tiled) the fragment-tiled layout [M/16][C/8][16 pixels][8 channels]: the 16-byte piece of pixel m,
        // channel group c8 sits at ((m >> 4) * (C / 8) + c8) * 256 + (m & 15) * 16 -- a wave instruction (16 pixels x the 4
        // channel groups of its lane groups) then touches ONE contiguous KiB instead of 16 rows x 64 B
        t2p[pg] = p.t2_tiled ? (const unsigned char*)p.t2 + (((mc[pg] >> 4) * (P / 8) + g4) * 256 + (mc[pg] & 15) * 16)
                             : (const unsigned char*)p.t2 + (mc[pg] * P + g4 * 8) * 2;
        resp[pg] = p.res_tiled ? (const unsigned char*)p.res + (((mc[pg] >> 4) * (C3 / 8) + g4) * 256 + (mc[pg] & 15) * 16)
                               : (const unsigned char*)p.res + (mc[pg] * C3 + g4 * 8) * 2;
        sok[pg] = mok[pg];
        if (p.out_even) {                                    // `out` is read by a stride-2 1x1 only (the next block's downsample branch,
            const int hw = p.Ho * p.Wo, rem = (int)mc[pg] % hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;   // model_copenet.py:97-102)
            sok[pg] = mok[pg] && !((ho | wo) & 1);
        }
        x2p[pg] = t2p[pg];                                   // second K segment: pixel (ho*stride2, wo*stride2) of image n in x2
        if constexpr (P2 > 0) {
            const int hw = p.Ho * p.Wo, n = (int)mc[pg] / hw, rem = (int)mc[pg] - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
            x2p[pg] = (const unsigned char*)p.x2 + ((((size_t)n * p.H2 + (size_t)ho * p.stride2) * p.W2 + (size_t)wo * p.stride2) * P2 + g4 * 8) * 2;
        }
        outp[pg] = p.out_tiled ? (unsigned char*)p.out + ((((size_t)m >> 4) * (C3 / 8) + g4) * 256 + ((size_t)m & 15) * 16)
                               : (unsigned char*)p.out + ((size_t)m * C3 + g4 * 8) * 2;
        t1p[pg] = (unsigned char*)p.t1n + ((size_t)m * N1 + g4 * 8) * 2;
    }
    // every kernel-argument load completes here: a scalar load the compiler believes pending inside the loop costs an
    // s_waitcnt lgkmcnt(0) in front of each DMA instruction, which also drains the fragment reads in flight
    asm volatile("" ::"s"(p.wstream), "s"(p.t2), "s"(p.res), "s"(p.out), "s"(p.t1n), "s"(p.M), "s"(p.x2));
    // byte strides of a 32-channel step (one B fragment) and of a 128-channel chunk in the two layouts
    const int t2_fs = p.t2_tiled ? 1024 : 64;
    const int res_fs = p.res_tiled ? 1024 : 64, res_cs = p.res_tiled ? 4096 : 256;
    const int out_fs = p.out_tiled ? 1024 : 64, out_cs = p.out_tiled ? 4096 : 256;

    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // A fragment of tile row f*16 + lr, K half s: chunk s*4 + g4 at position (s*4 + g4) ^ (lr & 7): byte ^ 64 for s = 1
    const uint32_t fb0 = lds0 + lr * 128 + ((g4 ^ (lr & 7)) << 4);
    const uint32_t tb3 = lds0 + TAB3 + g4 * 32, tb1 = lds0 + TAB1 + g4 * 32;

    int so = 0, si = (S - 1) * PR_TILE;                      // ring byte offsets: tile of this step / slot of the tile issued in it
    auto piece = [&](int i, bool issue) {                   // one 1-KiB piece of the tile S-1 steps ahead
        if (issue)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wnext + i * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + si + (wave * LPW + i) * 1024), 16, 0, 0);
    };
    auto mm = [&](f32x4& c, const u32x4& w, const u32x4& x) {
        if (PR_ABLATE & 8) asm volatile("" : "+v"(c) : "v"(w), "v"(x)); else c = mfma16(w, x, c);
    };
    // Fragment k of a tile (k = 8 s + f: row fragment f, K half s) lives in wf[k % D].  The fragment reads run D fragments
    // AHEAD of the MFMAs and straight across tile boundaries and barriers: the barrier of step t also covers tile t+1 (each
    // wave waits for its pieces of tile t+1 before it), so fragments of tile t+1 are requested while tile t is multiplied --
    // an LDS read has D - 4 .. D MFMAs of cover instead of a restart of the read pipeline behind every barrier (the first
    // version, four fragments of cover, spent 1 900 cycles per tile on 256 cycles of MFMAs with no memory traffic at all).
    u32x4 wf[D];
    auto rd4 = [&](auto G, uint32_t a) {                     // fragments 4G .. 4G+3 of the tile at a (= fb0 + slot offset)
        constexpr int g = G;
        const uint32_t b = g >= 2 ? a ^ 64u : a;
        constexpr int o = (g & 1) * 8192;
        wf[(4 * g) % D] = lds_read_b128<o>(b); wf[(4 * g + 1) % D] = lds_read_b128<o + 2048>(b);
        wf[(4 * g + 2) % D] = lds_read_b128<o + 4096>(b); wf[(4 * g + 3) % D] = lds_read_b128<o + 6144>(b);
    };
    // one weight tile: 16 MFMAs (8 row fragments x 2 K halves) on each 16-pixel group of this wave (acc[pg]: its 8 accumulators,
    // bb[pg][0 / 1]: its B fragments of the two K halves); the wave's LPW DMA pieces go out between the MFMA groups (a piece costs
    // its wave ~100 cycles of issue: under the matrix pipe, not in front of it).
    // pre: the next tile exists (its fragments are requested as this tile's are consumed)
    auto step = [&](f32x4* const (&acc)[PG], const u32x4* const (&bb)[PG], bool issue, bool pre) {
        const uint32_t a0 = fb0 + so, an = fb0 + ((so + PR_TILE) & (RING - 1));
        sfor<0, 4>([&](auto G) {
            constexpr int g = G;
            if (pre) wait_lgkmcnt<D - 4>(); else wait_lgkmcnt<0>();     // (last tile: nothing younger follows its fragments)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) mm(acc[pg][(4 * g + f) & 7], wf[(4 * g + f) % D], bb[pg][g < 2 ? 0 : 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LPW == 4) piece(g, issue);
            else if constexpr (g & 1) piece(g >> 1, issue);
            // refill the four registers just consumed: D = 16 -> the same fragments of the next tile; D = 8 -> the other K
            // half of this tile (groups 0, 1) or the first K half of the next tile (groups 2, 3)
            if constexpr (D == 16) { if (pre) rd4(G, an); }
            else if constexpr (g < 2) rd4(std::integral_constant<int, g + 2>{}, a0);
            else { if (pre) rd4(std::integral_constant<int, g - 2>{}, an); }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (issue) wnext += PR_TILE;
        so = (so + PR_TILE) & (RING - 1);
        si = (si + PR_TILE) & (RING - 1);
    };
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the bf16 set
    // BN + (identity) + ReLU + bf16 of the 8 consecutive channels a lane holds in fragments (2q, 2q+1); sc / sh: their tables
    auto bn8 = [&](const f32x4& lo, const f32x4& hi, const f32x4& s0, const f32x4& s1, const f32x4& h0, const f32x4& h1,
                   const u32x4* res) -> u32x4 {
        if (PR_ABLATE & 16) {
            u32x4 o = res ? *res : u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            asm volatile("" : "+v"(o) : "v"(lo), "v"(hi), "v"(s0), "v"(s1), "v"(h0), "v"(h1));
            return o;
        }
        // two values per instruction (v_pk_fma_f32 / v_pk_add_f32: the same IEEE fma / add per component as the scalar
        // epilogue of the stand-alone kernels), one v_cvt_pk + one packed integer max per pair
#ifdef PR_SCALAR_EPI                                         // A/B build: one v_fma_f32 / v_add_f32 per value instead of the packed forms
        f32x2 v0, v1, v2, v3;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v0.x) : "v"(lo.x), "v"(s0.x), "v"(h0.x)); asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v0.y) : "v"(lo.y), "v"(s0.y), "v"(h0.y));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v1.x) : "v"(lo.z), "v"(s0.z), "v"(h0.z)); asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v1.y) : "v"(lo.w), "v"(s0.w), "v"(h0.w));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v2.x) : "v"(hi.x), "v"(s1.x), "v"(h1.x)); asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v2.y) : "v"(hi.y), "v"(s1.y), "v"(h1.y));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v3.x) : "v"(hi.z), "v"(s1.z), "v"(h1.z)); asm("v_fma_f32 %0, %1, %2, %3" : "=v"(v3.y) : "v"(hi.w), "v"(s1.w), "v"(h1.w));
        if (res) {
            const uint32_t r0 = (*res).x, r1 = (*res).y, r2 = (*res).z, r3 = (*res).w;
            float a_, b_;
            unpack_bf16x2(r0, a_, b_); asm("v_add_f32 %0, %0, %1" : "+v"(v0.x) : "v"(a_)); asm("v_add_f32 %0, %0, %1" : "+v"(v0.y) : "v"(b_));
            unpack_bf16x2(r1, a_, b_); asm("v_add_f32 %0, %0, %1" : "+v"(v1.x) : "v"(a_)); asm("v_add_f32 %0, %0, %1" : "+v"(v1.y) : "v"(b_));
            unpack_bf16x2(r2, a_, b_); asm("v_add_f32 %0, %0, %1" : "+v"(v2.x) : "v"(a_)); asm("v_add_f32 %0, %0, %1" : "+v"(v2.y) : "v"(b_));
            unpack_bf16x2(r3, a_, b_); asm("v_add_f32 %0, %0, %1" : "+v"(v3.x) : "v"(a_)); asm("v_add_f32 %0, %0, %1" : "+v"(v3.y) : "v"(b_));
        }
#else
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
#endif
        u32x4 o;
        o.x = relu_pk_bf16(cvt_pk_bf16(v0.x, v0.y)); o.y = relu_pk_bf16(cvt_pk_bf16(v1.x, v1.y));
        o.z = relu_pk_bf16(cvt_pk_bf16(v2.x, v2.y)); o.w = relu_pk_bf16(cvt_pk_bf16(v3.x, v3.y));
        ap_rng_note2(rng, o.x, o.y); ap_rng_note2(rng, o.z, o.w);   // fp16 range sentinel
        return o;
    };
    auto load_identity = [&](int nb, u32x4 (&r)[PG][4]) {    // 4 x 16 B per lane and pixel group: channels nb*128 + q*32 + g4*8 .. + 7
        if constexpr (!RES) return;
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            if (PR_ABLATE & 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[pg][q] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                continue;
            }
            if (PR_ABLATE & 64) {                            // same 4 KiB of the group's 16 pixels, full 128-byte lines per row
                const unsigned char* rq = (const unsigned char*)p.res + ((size_t)(mc[pg] - lr + (lane >> 3)) * C3 + nb * 128) * 2 + (lane & 7) * 16;
                r[pg][0] = gload_b128<0>(rq); r[pg][1] = gload_b128<128>(rq);
                rq += (size_t)8 * C3 * 2;
                r[pg][2] = gload_b128<0>(rq); r[pg][3] = gload_b128<128>(rq);
                continue;
            }
            const unsigned char* rp = resp[pg] + nb * res_cs;
            r[pg][0] = gload_b128<0>(rp); r[pg][1] = gload_b128<0>(rp + res_fs); r[pg][2] = gload_b128<0>(rp + 2 * res_fs);
            r[pg][3] = gload_b128<0>(rp + 3 * res_fs);
        }
    };

    // ---------------------------------------------------------------- prologue: t2 fragments, (identity of chunk 0,)
    // tiles 0 .. S-2, the first D fragments of tile 0
    u32x4 xf[PG][NXF];
    sfor<0, PG * NXF>([&](auto II) {
        constexpr int pg = II / NXF, I = II % NXF;
        if constexpr (I < NXF1) xf[pg][I] = gload_b128<0>(t2p[pg] + I * t2_fs);
        else xf[pg][I] = gload_b128<(I - NXF1) * 64>(x2p[pg]);
    });
    // identity pieces of a chunk, then its packed result (= conv1 operand).  IDB: two sets, the next chunk's identity is
    // requested a whole chunk ahead; otherwise one set, requested at the chunk's first step
    u32x4 ra[PG][4], rb[PG][4];                              // (rb is used with IDB only)
    if constexpr (IDB) load_identity(0, ra);
#pragma unroll
    for (int t = 0; t < S - 1; ++t) {
        si = t * PR_TILE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) piece(i, true);
        wnext += PR_TILE;
    }
    si = (S - 1) * PR_TILE;
    f32x4 acc1[PG][HN > 0 ? HN * 8 : 1];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int i = 0; i < HN * 8; ++i) acc1[pg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // BatchNorm tables into LDS while those requests fly (read in the epilogues by inline-asm ds_read: a load the compiler
    // counts would be fenced against the LDS-DMA writes of the ring with vmcnt(0)).  The compiler waits for its table loads
    // with vmcnt(0), which covers the t2 fragments, the identity and the first tiles as well
    {
        float* t3 = (float*)(smem + TAB3);
        float* t1 = (float*)(smem + TAB1);
        for (int i = tid; i < C3; i += NT) { t3[i] = p.s3[i]; t3[C3 + i] = p.h3[i]; }
        for (int i = tid; i < N1; i += NT) { t1[i] = p.s1[i]; t1[N1 + i] = p.h1[i]; }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int i = 0; i < NXF; ++i) asm volatile("" : "+v"(xf[pg][i]));
    __syncthreads();
    rd4(std::integral_constant<int, 0>{}, fb0);
    rd4(std::integral_constant<int, 1>{}, fb0);
    if constexpr (D == 16) { rd4(std::integral_constant<int, 2>{}, fb0); rd4(std::integral_constant<int, 3>{}, fb0); }

    if (PR_ABLATE & 32) {                                    // timing build: the prologue alone
        wait_lgkmcnt<0>();
        asm volatile("" ::"v"(wf[0]), "v"(wf[D - 1]), "v"(xf[0][0]), "v"(xf[PG - 1][NXF - 1]), "v"(ra[0][0]), "v"(outp[0]), "v"(t1p[0]), "v"(resp[0]));
        return;
    }
    // one 128-channel chunk of conv3 + its share of conv1; cur: this chunk's identity / result, nxt: the next chunk's identity
#ifdef AP_TRACE   // cycle stamps of wave 0 of workgroups 0 and 300, chunk 1 (32 slots each): tools/probes/pair_trace.py
#define PRSTAMP(i) do { if (p.dbg && nb == 1 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) { \
        const unsigned long long t_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (lane == 0) p.dbg[(blockIdx.x ? 32 : 0) + (i)] = t_; } } while (0)
#else
#define PRSTAMP(i) do { } while (0)
#endif
    auto chunk = [&](int nb, u32x4 (&cur)[PG][4], u32x4 (&nxt)[PG][4]) {
        const bool lastc = nb == NB - 1;
        f32x4 acc3[PG][8];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc3[pg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        sfor<0, SPC>([&](auto J) {
            constexpr int j = J;
            // own pieces of the NEXT step's tile have landed (its fragments are requested during this step): S-3 younger
            // tiles may stay in flight, fewer at the very end; every other operation in the queue only makes the wait
            // conservative.  After the barrier everybody's pieces of that tile have landed and the slot read last step is free.
            constexpr int rem = SPC - 1 - j;                 // steps after this one in the chunk
            constexpr int yl = rem - 1 < S - 3 ? (rem - 1 > 0 ? rem - 1 : 0) : S - 3;
            PRSTAMP(3 * j);
            // (measured, tools/probes/pair_trace.py: counting the identity loads / the next chunk's operations into these waits moves
            //  the 1.2-2.5k-cycle stall of step 1 into the epilogue's wait for the identity and changes nothing: the loads themselves
            //  take that long under this kernel's own HBM traffic)
            if (lastc) { if constexpr (rem > 0) wait_vmcnt<LPW * yl>(); } else wait_vmcnt<LPW * (S - 3)>();
            PRSTAMP(3 * j + 1);
            __builtin_amdgcn_s_barrier();
            PRSTAMP(3 * j + 2);
            const bool issue = !(PR_ABLATE & 4) && !(lastc && rem < S - 1);
            const bool pre = !(lastc && rem == 0);
            if constexpr (j < KP) {
                f32x4* accs[PG];
                const u32x4* bbs[PG];
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) { accs[pg] = acc3[pg]; bbs[pg] = &xf[pg][2 * j]; }
                step(accs, bbs, issue, pre);
            } else {
                constexpr int g = j - KP, kh = g / (HN > 0 ? HN : 1), hn = g % (HN > 0 ? HN : 1);
                f32x4* accs[PG];
                const u32x4* bbs[PG];
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) { accs[pg] = &acc1[pg][hn * 8]; bbs[pg] = &cur[pg][2 * kh]; }
                step(accs, bbs, issue, pre);
            }
            if constexpr (j == 0) {                          // behind this step's DMA pieces
                if constexpr (IDB) { if (!lastc) load_identity(nb + 1, nxt); }
                else load_identity(nb, cur);
            }
            if constexpr (j == KP - 1) {
                PRSTAMP(28);
                // ---------------------------------------------------- conv3 epilogue of chunk nb, in registers.
                // Certainly younger than this chunk's identity loads -- IDB: the DMA pieces of steps 1 .. SPC-1 of the previous
                // chunk (S-1 <= SPC: every step of a chunk that is not the last one issues), of the prologue for chunk 0;
                // otherwise: the pieces of steps 1 .. KP-1 of this chunk (those that were issued)
                if constexpr (RES) {
                    if constexpr (IDB) {
                        if (nb == 0) wait_vmcnt<LPW * (S - 1)>(); else wait_vmcnt<(LPW * (SPC - 1) < 60 ? LPW * (SPC - 1) : 60)>();
                    } else {
                        constexpr int nl = KP - 1 < SPC - S ? KP - 1 : (SPC - S > 0 ? SPC - S : 0);
                        constexpr int ys = LPW * (KP - 1) < 60 ? LPW * (KP - 1) : 60;
                        if (lastc) wait_vmcnt<(LPW * nl < 60 ? LPW * nl : 60)>(); else wait_vmcnt<ys>();
                    }
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) asm volatile("" : "+v"(cur[pg][0]), "+v"(cur[pg][1]), "+v"(cur[pg][2]), "+v"(cur[pg][3]));
                }
                const uint32_t ta = tb3 + nb * 512;
                sfor<0, 4>([&](auto Q) {
                    constexpr int q = Q;
                    const f32x4 s0 = lds_read_f32x4<q * 128>(ta), s1 = lds_read_f32x4<q * 128 + 16>(ta);
                    const f32x4 h0 = lds_read_f32x4<C3 * 4 + q * 128>(ta), h1 = lds_read_f32x4<C3 * 4 + q * 128 + 16>(ta);
                    wait_lgkmcnt<0>();
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) {
                        cur[pg][q] = bn8(acc3[pg][2 * q], acc3[pg][2 * q + 1], s0, s1, h0, h1, RES ? &cur[pg][q] : nullptr);
                        if (PR_ABLATE & 2) asm volatile("" ::"v"(cur[pg][q]), "v"(outp[pg]));   // (timing build: keep the value live)
                        else if (PR_ABLATE & 128) {
                            unsigned char* oq = (unsigned char*)p.out + ((size_t)(mc[pg] - lr + (lane >> 3) + (q >> 1) * 8) * C3 + nb * 128) * 2 + (lane & 7) * 16 + (q & 1) * 128;
                            *(u32x4*)oq = cur[pg][q];
                        } else if (sok[pg]) gstore_b128(outp[pg] + nb * out_cs + q * out_fs, cur[pg][q]);
                    }
                });
                PRSTAMP(29);
            }
            if constexpr (j == SPC - 1) PRSTAMP(30);
        });
    };
    for (int nb = 0; nb < NB; nb += 2) {
        if constexpr (IDB) { chunk(nb, ra, rb); chunk(nb + 1, rb, ra); }
        else { chunk(nb, ra, rb); chunk(nb + 1, ra, rb); }
    }
    // ---------------------------------------------------------------- conv1 epilogue: BN + ReLU + bf16, 16-byte stores
    sfor<0, HN * 4>([&](auto I) {
        constexpr int hn = I / 4, q = I % 4;
        const uint32_t ta = tb1 + (hn * 128 + q * 32) * 4;
        const f32x4 s0 = lds_read_f32x4<0>(ta), s1 = lds_read_f32x4<16>(ta), h0 = lds_read_f32x4<N1 * 4>(ta), h1 = lds_read_f32x4<N1 * 4 + 16>(ta);
        wait_lgkmcnt<0>();
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            const u32x4 o = bn8(acc1[pg][hn * 8 + 2 * q], acc1[pg][hn * 8 + 2 * q + 1], s0, s1, h0, h1, nullptr);
            if (PR_ABLATE & 2) asm volatile("" ::"v"(o), "v"(t1p[pg]));
            else if (mok[pg]) gstore_b128(t1p[pg] + (hn * 128 + q * 32) * 2, o);
        }
    });
    ap_rng_flush(p.range_flag, rng);
}

template <int P, int P2, int C3, int N1, bool RES, int NW, int S, int D, bool IDB, int PG = 1>
hipError_t launch_pair(const PairArgs& a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    auto kern = conv_pair_kernel<P, P2, C3, N1, RES, NW, S, D, IDB, PG>;
    constexpr int lds = S * PR_TILE + (2 * C3 + 2 * N1) * 4;
    static_assert(NW == 8 || PG == 2 || lds <= 81920, "two workgroups per CU");
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.M + 16 * NW * PG - 1) / (16 * NW * PG)), dim3(64 * NW), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// identity pairs: (P, 4P, N1) in (128,512,128) (128,512,256) (256,1024,256); stage-first blocks (conv3 + folded downsample,
// P2 = channels of the block input): (128 | 256, 512, 128) with the next conv1, (256 | 512, 1024, 0) conv3 alone
bool ap_conv_pair_supported(int P, int P2, int C3, int N1) {
    if (P2 == 0) return C3 == 4 * P && ((P == 128 && (N1 == 128 || N1 == 256)) || (P == 256 && N1 == 256));
    return (P == 128 && P2 == 256 && C3 == 512 && N1 == 128) || (P == 256 && P2 == 512 && C3 == 1024 && N1 == 0);
}

size_t ap_conv_pair_stream_bytes(int P, int P2, int C3, int N1) {
    return (size_t)(C3 / 128) * ((P + P2) / 64 + 2 * (N1 / 128)) * PR_TILE;
}

hipError_t ap_launch_pair_pack(const void* w3, const void* w1, void* dst, int P, int P2, int C3, int N1, hipStream_t st) {
    if (!ap_conv_pair_supported(P, P2, C3, N1) || (N1 > 0 && !w1)) return hipErrorInvalidValue;
    const size_t chunks = ap_conv_pair_stream_bytes(P, P2, C3, N1) / 16;
    hipLaunchKernelGGL(pair_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w3,
                       (const bf16_t*)w1, (unsigned char*)dst, P + P2, C3, N1);
    return hipGetLastError();
}

hipError_t ap_launch_conv_pair(const PairArgs& a, int P, int P2, int C3, int N1, hipStream_t st) {
    if (a.M <= 0 || !a.t2 || !a.wstream || !a.out || (N1 > 0 && !a.t1n) || (P2 == 0 && !a.res) || (P2 > 0 && !a.x2))
        return hipErrorInvalidValue;
    // four waves per workgroup, two workgroups per CU, 4-slot ring, half a tile of fragment look-ahead.  Measured and not
    // kept (tools/pair_bench.py, 256 images): eight waves x one workgroup per CU (half the weight DMA per MFMA) is 4-10 %
    // slower; a whole tile of fragment look-ahead (64 registers) times the same
    if (P2 == 0 && C3 == 4 * P) {
        if (P == 128 && N1 == 128) return launch_pair<128, 0, 512, 128, true, 4, 4, 8, true>(a, st);
        if (P == 128 && N1 == 256) return launch_pair<128, 0, 512, 256, true, 4, 4, 8, true>(a, st);
        if (P == 256 && N1 == 256) return launch_pair<256, 0, 1024, 256, true, 4, 4, 8, false>(a, st);
    }
    if (P == 128 && P2 == 256 && C3 == 512 && N1 == 128) return launch_pair<128, 256, 512, 128, false, 4, 4, 8, false>(a, st);
    if (P == 256 && P2 == 512 && C3 == 1024 && N1 == 0) return launch_pair<256, 512, 1024, 0, false, 4, 4, 8, false>(a, st);
    return hipErrorInvalidValue;
}

AP_NS_END
