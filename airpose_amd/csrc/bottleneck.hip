// Whole-bottleneck fusion for the 64-plane stage (layer1) of the trunk, bf16, gfx950.
//
// Reference: Bottleneck.forward, copenet/src/copenet/models/model_copenet.py:27-47
//   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + (downsample(x) | x))
// At 56x56 the three convs of a layer1 block are HBM-bound when run as separate GEMMs (the 64-channel
// intermediates and the 256-channel block input are written and re-read: 3.3 GB per block at 512 images).
// These kernels keep both intermediates in LDS: a workgroup owns a 14x14 output tile of one image and runs
//
//   phase 1  mid1[16x16 halo px][64] = relu(bn1(W1 . x))         K = CIN in 64-channel steps
//   phase 2  mid2[14x14 px][64]      = relu(bn2(W2 * mid1))      9 taps x K 64, weights streamed per tap
//   phase 3  y[14x14 px][256]        = relu(bn3(W3 . mid2) + x)  4 passes of 64 output channels through an LDS stage,
//                                                               16-byte coalesced stores
//   (downsample block: K3 = 128 = [mid2 | x] with the downsample conv and both BN scales folded into W3,
//    pack_c3_ds in api.hip; x is the resident phase-1 tile, no identity add)
//
// so HBM sees the block input once (+ the 1.31x halo) and the output once.  The result equals the three-convolution
// path bit for bit (same operands, same bf16 rounding points, same K order per output element).
// MFMA operands as in conv_pipe.hip: weights are the A operand (rows = channels), pixels the B operand; every LDS
// image is rows of 128 B (64 bf16) with the 16-byte chunk index XOR-swizzled by (row & 7), written lane-linearly by
// global_load_lds with the swizzle on the SOURCE address.  Pixel rows are indexed on a 16-wide grid (row = y*16 + x)
// so a 16-pixel MFMA column block is one tile row (columns 14, 15 are junk that is never stored).
// All LDS traffic is inline asm and every s_waitcnt is counted by hand (a compiler-visible LDS access after an LDS-DMA
// would be answered with vmcnt(0)); -DAP_BNECK_SAFE_X -DAP_BNECK_SAFE_W turn the counted waits of the two DMA roles
// into vmcnt(0) for cross-checking: results must be bit-identical.
// Two kernels: bneck256_kernel (identity blocks) and bneck64ds_kernel (the downsample block), both persistent.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int TS = 14;                                       // output tile edge

template <int N> __device__ __forceinline__ void vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vmx() {   // x-wave / weight-wave waits of the persistent kernel (debug switches)
#ifdef AP_BNECK_SAFE_X
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    vm<N>();
#endif
}
template <int N> __device__ __forceinline__ void vmw() {
#ifdef AP_BNECK_SAFE_W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    vm<N>();
#endif
}
template <int N> __device__ __forceinline__ void lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void bar() { __builtin_amdgcn_s_barrier(); }

template <int OFF> __device__ __forceinline__ u32x4 rd128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ u32x2 rd64(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ void wr64(uint32_t addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

__device__ __forceinline__ void mma(f32x4& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// N fragment reads at addr + OFF0 + k*2048 (16 rows of 128 B per MFMA block)
template <int N, int OFF0> __device__ __forceinline__ void rd_blocks(uint32_t addr, u32x4 (&o)[N]) {
    static_assert(N <= 8, "unroll");
    if constexpr (N > 0) o[0] = rd128<OFF0>(addr);
    if constexpr (N > 1) o[1] = rd128<OFF0 + 2048>(addr);
    if constexpr (N > 2) o[2] = rd128<OFF0 + 4096>(addr);
    if constexpr (N > 3) o[3] = rd128<OFF0 + 6144>(addr);
    if constexpr (N > 4) o[4] = rd128<OFF0 + 8192>(addr);
    if constexpr (N > 5) o[5] = rd128<OFF0 + 10240>(addr);
    if constexpr (N > 6) o[6] = rd128<OFF0 + 12288>(addr);
    if constexpr (N > 7) o[7] = rd128<OFF0 + 14336>(addr);
}

// one 64-deep contraction step of a [2 channel blocks] x [4 pixel blocks] wave tile.
// wa0/wa1: weight fragment addresses of the two 32-deep halves; xa0/xa1: pixel fragment addresses (blocks 0..2 at
// OFF + j*2048), xb0/xb1: the same for block 3 (a clamped duplicate of block 2 in the 3-row waves)
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// hook(0) runs once the fragment reads are queued, hook(1) between the two MFMA groups (slots for global stores)
template <int OFF, typename F = NoHook>
__device__ __forceinline__ void step_2x4(uint32_t wa0, uint32_t wa1, uint32_t xa0, uint32_t xa1, uint32_t xb0, uint32_t xb1,
                                         f32x4 (&acc)[2][4], F&& hook = NoHook{}) {
    u32x4 a0[2], a1[2], b0[4], b1[4];
    rd_blocks<2, 0>(wa0, a0);
    { u32x4 t[3]; rd_blocks<3, OFF>(xa0, t); b0[0] = t[0]; b0[1] = t[1]; b0[2] = t[2]; }
    b0[3] = rd128<OFF + 6144>(xb0);
    rd_blocks<2, 0>(wa1, a1);
    { u32x4 t[3]; rd_blocks<3, OFF>(xa1, t); b1[0] = t[0]; b1[1] = t[1]; b1[2] = t[2]; }
    b1[3] = rd128<OFF + 6144>(xb1);
    hook(0);
    lgkm<6>();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc[i][j], a0[i], b0[j]);
    __builtin_amdgcn_sched_barrier(0);
    hook(1);
    __builtin_amdgcn_sched_barrier(0);
    lgkm<0>();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc[i][j], a1[i], b1[j]);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ u32x2 bn_relu_pack(const f32x4& a, const float4& sc, const float4& sh, bool keep) {
    float v0 = fmaxf(a[0] * sc.x + sh.x, 0.f), v1 = fmaxf(a[1] * sc.y + sh.y, 0.f);
    float v2 = fmaxf(a[2] * sc.z + sh.z, 0.f), v3 = fmaxf(a[3] * sc.w + sh.w, 0.f);
    u32x2 o;
    o.x = keep ? pack_bf16x2(v0, v1) : 0u;
    o.y = keep ? pack_bf16x2(v2, v3) : 0u;
    return o;
}

// =====================================================================================================================
// Identity blocks (CIN = 256): persistent, wave-specialised.
//
// In-kernel cycle stamps of a first, one-tile-per-workgroup version showed a workgroup waiting ~10k cycles for its
// first operands and ~1.5k cycles per pass for an identity tile fetched a second time; with one workgroup per CU
// (LDS) nothing covers them.  Here a workgroup loops over tiles (grid = #CUs) and
//   * the first two 64-channel steps of the NEXT tile's x halo are fetched into the (then idle) step ring during
//     phases 2-3; steps 2 and 3 land in the mid1 / mid2 buffers, which are idle during phase 1;
//   * the identity values are captured from the x steps in LDS during phase 1 (chunk kc of x is exactly what pass
//     nc = kc adds), in accumulator layout: no second read of x at all;
//   * DMA issue is split by wave: waves 0-3 stream the weight chunks (L2-resident, short latency), waves 4-7 the x
//     steps (HBM, long latency).  s_waitcnt vmcnt retires in order PER WAVE, so the weight waits of phases 2-3 never
//     queue behind the long-latency prefetch; the other role needs no wait at all, the barrier publishes;
//   * BatchNorm constants live in LDS (3 KiB) to keep the 64 identity registers under the 256-VGPR budget.
// LDS map (bytes): x step ring 2 x 32 KiB | mid1 / stage / x step 2 (32 KiB) | mid2 / x step 3 (32 KiB) |
// weight ring 3 x 8 KiB | constants.  mid1 rows 256, 257 (read by junk columns only) alias the first mid2 rows.
constexpr int P_XR0 = 0, P_XR1 = 32768, P_M1 = 65536, P_M2 = 98304, P_WR = 131072, P_CONST = 155648, P_TOTAL = 158720;
constexpr int C_S1 = 0, C_H1 = 64, C_S2 = 128, C_H2 = 192, C_S3 = 256, C_H3 = 512;   // float offsets in the constant block

template <int OFF> __device__ __forceinline__ void wr128(uint32_t addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ float4 as_f4(const u32x4& r) {
    const uint32_t a = r.x, b = r.y, c = r.z, d = r.w;       // (copy the lanes out first: bit_cast on a vector element mis-compiles)
    return make_float4(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c),
                       __builtin_bit_cast(float, d));
}
template <int OFF> __device__ __forceinline__ float4 rd_f4(uint32_t addr) {
    const u32x4 r = rd128<OFF>(addr);
    lgkm<0>();
    const uint32_t a = r.x, b = r.y, c = r.z, d = r.w;
    return make_float4(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c),
                       __builtin_bit_cast(float, d));
}

__global__ void __launch_bounds__(512) bneck256_kernel(const BneckArgs p) {
    constexpr int CIN = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const bool xw = wave >= 4;                               // DMA role: x steps (waves 4-7) or weight chunks (0-3)
    const int wq = wave & 3;
    const int G = gridDim.x;
#ifdef AP_TRACE   // cycle stamps of wave 0 / wave 4 of workgroup 0, third tile (40 slots each)
    int stamp_i = 0, tile_no = 0;
#define PSTAMP() do { if (p.dbg && blockIdx.x == 0 && tile_no == 2 && (tid == 0 || tid == 256)) \
        p.dbg[(tid ? 40 : 0) + stamp_i] = __builtin_readcyclecounter(); ++stamp_i; } while (0)
#define FSTAMP(k) do { if (nc == 1 && p.dbg && blockIdx.x == 0 && tile_no == 2 && (tid == 0 || tid == 256)) \
        p.dbg[(tid ? 40 : 0) + 26 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP() do {} while (0)
#define FSTAMP(k) do {} while (0)
#endif

    // compute roles: 2 channel-block pairs x 4 pixel-block groups.  Phase 1: halo rows 4*wn2 .. +3;
    // phases 2/3: output rows rb .. rb+nb-1 (4, 4, 3, 3)
    const int wm2 = wave & 1, wn2 = wave >> 1;
    const int rb = wn2 < 2 ? wn2 * 4 : 8 + (wn2 - 2) * 3;
    const int nb = wn2 < 2 ? 4 : 3;
    const uint32_t j3adj = nb == 4 ? 0u : (uint32_t)-2048;

    // ---------------------------------------------------------------- BatchNorm constants -> LDS (once per workgroup)
    if (tid < 192) {
        const float* src = tid < 16 ? p.s1 + tid * 4 : tid < 32 ? p.h1 + (tid - 16) * 4 : tid < 48 ? p.s2 + (tid - 32) * 4
                         : tid < 64 ? p.h2 + (tid - 48) * 4 : tid < 128 ? p.s3 + (tid - 64) * 4 : p.h3 + (tid - 128) * 4;
        const float4 v = *(const float4*)src;
        u32x4 u;
        u.x = __builtin_bit_cast(uint32_t, v.x); u.y = __builtin_bit_cast(uint32_t, v.y);
        u.z = __builtin_bit_cast(uint32_t, v.z); u.w = __builtin_bit_cast(uint32_t, v.w);
        wr128<0>(lds0 + P_CONST + tid * 16, u);
    }
    lgkm<0>();

    // ---------------------------------------------------------------- DMA sources
    const unsigned char* xg = (const unsigned char*)p.x;
    // x waves: piece i of a step covers halo rows (8*wq + i)*8 + prow = halo pixel (4*wq + (i >> 1), (i & 1)*8 + prow);
    // its byte offset is xoff0 + (i >> 1)*W*512 + (i & 1)*4096 (32-bit: the launcher rejects tensors of 4 GiB and more)
    uint32_t xoff0 = 0, xlive = 0;
    uint32_t xrc = (uint32_t)(prow << 8 | pchunk);           // laundered per tile, see wrc below
    auto setup_x = [&](int tile) {
        const int n = tile / p.tiles_per_img, trem = tile - n * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        xlive = 0;
        asm volatile("" : "+v"(xrc));
        const int prow = xrc >> 8, pchunk = xrc & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int h = (8 * wq + i) * 8 + prow;
            const int yy = ty * TS - 1 + (h >> 4), xx = tx * TS - 1 + (h & 15);
            const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            xlive |= ok ? (1u << i) : 0u;
        }
        xoff0 = (uint32_t)((((n * p.H + ty * TS - 1 + 4 * wq) * p.W + tx * TS - 1 + prow) * CIN + pchunk * 8) * 2);   // may wrap below 0: only used when live
    };
    const int wrow = 2 * wq * 8 + prow;                      // weight waves: rows wrow and wrow + 8 of a 64-row chunk
    // packed (row, chunk) of this lane's weight pieces.  Offsets are rebuilt at every issue from this one register,
    // which is laundered through an empty asm once per tile: hoisted out of the tile loop the 17 chunk addresses and
    // the per-piece x offsets spill, and scratch traffic would break the hand-counted vmcnt bookkeeping
    uint32_t wrc = (uint32_t)(wrow << 8 | pchunk);
    auto dma = [&](const unsigned char* src, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    auto issue_x1 = [&](int kc, int base, int i) {           // piece i (of 8 per x wave) of channels [64kc, 64kc+64) of the halo
        dma(((xlive >> i) & 1u) ? xg + (uint32_t)(xoff0 + (i >> 1) * p.W * 512 + (i & 1) * 4096 + kc * 128) : (const unsigned char*)p.zero,
            base + (8 * wq + i) * 1024);
    };
    auto issue_x = [&](int kc, int base) {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_x1(kc, base, i);
    };
    auto issue_w = [&](const void* w, uint32_t row_bytes, uint32_t col_bytes, int slot) {   // 2 pieces per weight wave
        const unsigned char* src = (const unsigned char*)w + (uint32_t)((wrc >> 8) * row_bytes + (wrc & 7) * 16 + col_bytes);
        dma(src, P_WR + slot * 8192 + 2 * wq * 1024);
        dma(src + 8 * row_bytes, P_WR + slot * 8192 + (2 * wq + 1) * 1024);
    };
    auto issue_w1 = [&](int kc, int slot) { issue_w(p.w1, CIN * 2, kc * 128, slot); };
    auto issue_tap = [&](int t, int slot) { issue_w(p.w2, 576 * 2, t * 128, slot); };
    auto issue_w3 = [&](int nc, int slot) { issue_w(p.w3, 64 * 2, nc * 8192, slot); };

    // ---------------------------------------------------------------- fragment / epilogue addresses (tile independent)
    // second 32-deep half of a 64-deep step: chunk index + 4, i.e. address ^ 64 (bases are 128-byte aligned)
    const uint32_t sw0 = (uint32_t)((g4 ^ (lr & 7)) << 4);
    const uint32_t wrow2 = (uint32_t)((32 * wm2 + lr) * 128);
    const uint32_t p1row = (uint32_t)((wn2 * 64 + lr) * 128);  // phase 1: halo rows (4*wn2 + j)*16 + lr
    uint32_t m1a[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) m1a[kx] = lds0 + P_M1 + (rb * 16 + lr + kx) * 128 + (uint32_t)((g4 ^ ((lr + kx) & 7)) << 4);
    uint32_t eoff[2], e1off[2], roff[2];                     // accumulator-layout byte offsets (8 B at channel 32*wm2+16*i+4*g4)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = 4 * wm2 + 2 * i + (g4 >> 1);
        eoff[i] = (uint32_t)((rb * 16 + lr) * 128 + ((chunk ^ (lr & 7)) << 4) + (g4 & 1) * 8);            // output grid
        e1off[i] = (uint32_t)((wn2 * 64 + lr) * 128 + ((chunk ^ (lr & 7)) << 4) + (g4 & 1) * 8);          // halo grid
        roff[i] = (uint32_t)(((rb + 1) * 16 + lr + 1) * 128 + ((chunk ^ ((lr + 1) & 7)) << 4) + (g4 & 1) * 8);   // centre px of the halo
    }
    const uint32_t cch = lds0 + P_CONST + (uint32_t)((32 * wm2 + 4 * g4) * 4);   // + 64*i bytes per channel block
    // coalesced store geometry: wave w owns the 196 valid 16-byte pieces w*196 .. w*196+195 of the [196 px][8 chunks]
    // stage (exactly four store instructions per wave and pass; the fourth carries 4 lanes)
    // Output stage, WAVE-PRIVATE: a wave's 2 x 4 accumulator tiles are [<= 4 rows x 16 px] x 32 channels = 64 B per pixel;
    // it transposes them through its own 4 KiB of the stage (rows of 64 B, 16-byte chunk index XOR (row >> 2) & 3) into
    // 16 bytes per lane and stores 64-byte pixel segments (the sibling wave of the channel pair stores the other half of
    // the 128-byte line).  No workgroup barrier between the accumulators and the stores.  Exactly four store
    // instructions per wave and pass (the wait counts rely on it): lanes past the wave's nb*14*4 pieces are masked,
    // lane 0 of an otherwise empty instruction repeats piece 0.
    const uint32_t stg = lds0 + P_M1 + wave * 4096;
    uint32_t sea[2];                                         // accumulator-layout write address of channel block i
#pragma unroll
    for (int i = 0; i < 2; ++i)
        sea[i] = stg + lr * 64 + (uint32_t)(((2 * i + (g4 >> 1)) ^ ((lr >> 2) & 3)) << 4) + (g4 & 1) * 8;
    uint32_t st_lds[4], st_off[4];
    uint32_t st_ok = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int idx = it * 64 + lane;
        const bool ok = idx < nb * TS * 4;
        if (!ok) idx = 0;
        st_ok |= (ok || lane == 0) ? (1u << it) : 0u;
        const int q = idx >> 2, c = idx & 3, j = q / TS, ox = q - j * TS, r = j * 16 + ox;
        st_lds[it] = stg + r * 64 + (uint32_t)((c ^ ((r >> 2) & 3)) << 4);
        st_off[it] = (uint32_t)((((rb + j) * p.W + ox) * 256 + 32 * wm2 + c * 8) * 2);
    }
    // A pass leaves its 64-channel output slice in sv (read back from the stage); the four global stores are issued
    // inside the NEXT compute step (a burst of stores right behind the stage blocks the wave: the write path drains
    // at the CU's HBM share, ~230 cycles per store instruction measured)
    u32x4 sv[4];
    auto put_stores = [&](int half, unsigned char* base) {
        if (half == 0) {
            *(u32x4*)(base + st_off[0]) = sv[0];
            *(u32x4*)(base + st_off[1]) = sv[1];
        } else {
            if (st_ok & 4u) *(u32x4*)(base + st_off[2]) = sv[2];
            if (st_ok & 8u) *(u32x4*)(base + st_off[3]) = sv[3];
        }
    };
    u32x2 rv[4][2][4];                                       // identity values of the tile, one set per 64-channel pass
    f32x4 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto p1_compute = [&](auto KC, int xbase, int slot, unsigned char* st_base = nullptr) {
        constexpr int kc = decltype(KC)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t ra = lds0 + xbase + roff[i];
            rv[kc][i][0] = rd64<0>(ra);
            rv[kc][i][1] = rd64<2048>(ra);
            rv[kc][i][2] = rd64<4096>(ra);
            rv[kc][i][3] = rd64<6144>(ra + j3adj);
        }
        const uint32_t wa = lds0 + P_WR + slot * 8192 + wrow2, xa = lds0 + xbase + p1row;
        if (kc == 0 && st_base)                              // the previous tile's last slice leaves under this step
            step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, acc,
                        [&](int half) { put_stores(half, st_base); });
        else
            step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, acc);
    };
    // The next tile's first two x steps (16 pieces per x wave) ride along with taps 0..8 and passes 0..2, at most two
    // per compute step: a CU keeps only ~32 LDS-DMA pieces in flight (tools/probes/dma_depth_probe.hip), so a burst
    // parks the issuing waves in the issue slot at the CU's share of the HBM rate.
    //   taps 0..7: step 0, piece t (between the MFMA groups);  tap 8, passes 0..2: step 1, two pieces each
    auto tap = [&](auto KY, auto KX, int slot) {
        constexpr int ky = decltype(KY)::value, kx = decltype(KX)::value, t = ky * 3 + kx;
        const uint32_t wa = lds0 + P_WR + slot * 8192 + wrow2;
        step_2x4<ky * 2048>(wa + sw0, (wa + sw0) ^ 64, m1a[kx], m1a[kx] ^ 64, m1a[kx] + j3adj, (m1a[kx] ^ 64) + j3adj, acc,
                            [&](int half) {
                                if (!xw) return;
                                if (t < 8) { if (half) issue_x1(0, P_XR0, t); }
                                else issue_x1(1, P_XR1, half);
                            });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ================================================================ prologue: the queue state a previous tile would leave
    int tile = xcd_remap(blockIdx.x, G);
    if (xw) { setup_x(tile); issue_x(0, P_XR0); issue_x(1, P_XR1); }
    else { issue_w1(1, 1); issue_w1(0, 0); }

    unsigned char* yprev = nullptr;                          // tile whose last output slice is still in sv
    for (bool first = true;; first = false) {
        const int n = tile / p.tiles_per_img, trem = tile - n * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int y0 = ty * TS, x0 = tx * TS;
        unsigned char* const ybase = (unsigned char*)p.y + (((size_t)n * p.H + y0) * p.W + x0) * 512;
        PSTAMP();
        asm volatile("" : "+v"(wrc));
        // ------------------------------------------------------------ phase 1
        bar();                                               // previous tile: stage and mid2 reads complete
        if (xw) { if (first) issue_x(2, P_M1); issue_x(3, P_M2); } else issue_w1(2, 2);   // (step 2 came in under pass 3)
        // vm<N>: N = operations this wave queued after the ones needed (x waves: 8 per step; weight waves: 2 per
        // chunk; everybody: 4 stores inside passes 1-3 and inside the next tile's first step)
        if (first) vm<0>(); else if (xw) vmx<36>(); else vmw<6>();
        bar();
        PSTAMP();
        zero_acc();
        p1_compute(I0{}, P_XR0, 0, yprev);
        if (xw) vmx<24>();                                   // x step 1 (its last piece left under pass 2)
        bar();
        if (!xw) issue_w1(3, 0);
        PSTAMP();
        p1_compute(I1{}, P_XR1, 1);
        if (xw) vmx<12>(); else vmw<6>();                      // x step 2, queued under the previous pass 3 (after it: step 3, 4 stores) / W1 chunk 2 (after it: 4 stores, chunk 3)
        bar();
        if (!xw) issue_tap(0, 1);
        PSTAMP();
        p1_compute(I2{}, P_M1, 2);
        if (xw) vmx<4>(); else vmw<2>();                       // x step 3 (after it: 4 stores) / W1 chunk 3 (after it: tap 0)
        bar();
        if (!xw) issue_tap(1, 2);
        PSTAMP();
        p1_compute(I3{}, P_M2, 0);
        bar();
        PSTAMP();
        // from here on the x waves fetch for the next tile (the last tile re-fetches its own: the queue bookkeeping stays)
        const int next = tile + G;
        const bool has_next = next < p.total;
        if (xw) setup_x(has_next ? next : tile);
        else issue_tap(2, 0);
        {   // epilogue 1 -> mid1 (halo pixels outside the image are conv2's zero padding: exactly 0)
            const bool xin = (unsigned)(x0 - 1 + lr) < (unsigned)p.W;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = rd_f4<C_S1 * 4>(cch + 64 * i), sh = rd_f4<C_H1 * 4>(cch + 64 * i);
                const uint32_t ea = lds0 + P_M1 + e1off[i];
                const int yb = y0 - 1 + 4 * wn2;
                wr64<0>(ea, bn_relu_pack(acc[i][0], sc, sh, xin && (unsigned)(yb + 0) < (unsigned)p.H));
                wr64<2048>(ea, bn_relu_pack(acc[i][1], sc, sh, xin && (unsigned)(yb + 1) < (unsigned)p.H));
                wr64<4096>(ea, bn_relu_pack(acc[i][2], sc, sh, xin && (unsigned)(yb + 2) < (unsigned)p.H));
                wr64<6144>(ea, bn_relu_pack(acc[i][3], sc, sh, xin && (unsigned)(yb + 3) < (unsigned)p.H));
            }
        }
        lgkm<0>();
        PSTAMP();
        // ------------------------------------------------------------ phase 2 (weight ring: tap t in slot (t+1)%3)
        zero_acc();
        if (!xw) vmw<4>(); bar(); tap(I0{}, I0{}, 1); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(3, 1); tap(I0{}, I1{}, 2); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(4, 2); tap(I0{}, I2{}, 0); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(5, 0); tap(I1{}, I0{}, 1); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(6, 1); tap(I1{}, I1{}, 2); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(7, 2); tap(I1{}, I2{}, 0); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(8, 0); tap(I2{}, I0{}, 1); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(0, 1); tap(I2{}, I1{}, 2); PSTAMP();
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(1, 2); tap(I2{}, I2{}, 0); PSTAMP();
        {   // epilogue 2 -> mid2
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = rd_f4<C_S2 * 4>(cch + 64 * i), sh = rd_f4<C_H2 * 4>(cch + 64 * i);
                const uint32_t ea = lds0 + P_M2 + eoff[i];
                wr64<0>(ea, bn_relu_pack(acc[i][0], sc, sh, true));
                wr64<2048>(ea, bn_relu_pack(acc[i][1], sc, sh, true));
                wr64<4096>(ea, bn_relu_pack(acc[i][2], sc, sh, true));
                if (nb == 4) wr64<6144>(ea, bn_relu_pack(acc[i][3], sc, sh, true));
            }
        }
        lgkm<0>();
        PSTAMP();
        // ------------------------------------------------------------ phase 3: 4 passes of 64 output channels
        auto pass = [&](auto NC, int slot) {
            constexpr int nc = decltype(NC)::value;
            FSTAMP(0);
            zero_acc();
            u32x4 cs[2], ch[2];                              // BatchNorm scale / shift of this slice (ready with the fragments)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                cs[i] = rd128<(C_S3 + 64 * nc) * 4>(cch + 64 * i);
                ch[i] = rd128<(C_H3 + 64 * nc) * 4>(cch + 64 * i);
            }
            const uint32_t wa = lds0 + P_WR + slot * 8192 + wrow2;
            const uint32_t m2a = m1a[0] + (P_M2 - P_M1);
            // stage of this pass: mid1's buffer, except for the last pass, which reuses mid2's (after one more barrier)
            // so that mid1's buffer can take x step 2 of the next tile a whole pass before the tile ends
            constexpr int STG = nc == 3 ? P_M2 - P_M1 : 0;
            if constexpr (nc > 0)                            // the previous slice leaves under this step's MFMAs
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, m2a, m2a ^ 64, m2a + j3adj, (m2a ^ 64) + j3adj, acc,
                            [&](int half) {
                                put_stores(half, ybase + (nc - 1) * 128);
                                if (!xw) return;
                                if (nc == 3) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i) issue_x1(2, P_M1, 4 * half + i);
                                } else {
                                    issue_x1(1, P_XR1, 2 + 2 * nc + half);
                                }
                            });
            else
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, m2a, m2a ^ 64, m2a + j3adj, (m2a ^ 64) + j3adj, acc,
                            [&](int half) { if (xw) issue_x1(1, P_XR1, 2 + half); });
            FSTAMP(1);
            if constexpr (nc == 3) bar();                    // every wave is done with mid2
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = as_f4(cs[i]), sh = as_f4(ch[i]);
                const uint32_t ea = sea[i] + STG;
                auto out = [&](const f32x4& a, const u32x2& r) {
                    float lo, hi;
                    float v0 = a[0] * sc.x + sh.x, v1 = a[1] * sc.y + sh.y, v2 = a[2] * sc.z + sh.z, v3 = a[3] * sc.w + sh.w;
                    unpack_bf16x2(r.x, lo, hi); v0 += lo; v1 += hi;
                    unpack_bf16x2(r.y, lo, hi); v2 += lo; v3 += hi;
                    u32x2 o;
                    o.x = pack_bf16x2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                    o.y = pack_bf16x2(fmaxf(v2, 0.f), fmaxf(v3, 0.f));
                    return o;
                };
                wr64<0>(ea, out(acc[i][0], rv[nc][i][0]));
                wr64<1024>(ea, out(acc[i][1], rv[nc][i][1]));
                wr64<2048>(ea, out(acc[i][2], rv[nc][i][2]));
                if (nb == 4) wr64<3072>(ea, out(acc[i][3], rv[nc][i][3]));
            }
            lgkm<0>();                                       // (own writes; the stage region is this wave's)
            FSTAMP(2);
            FSTAMP(3);
#pragma unroll
            for (int it = 0; it < 4; ++it) sv[it] = rd128<STG>(st_lds[it]);
            lgkm<0>();
            FSTAMP(4);
        };
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(2, 0); pass(I0{}, 1); PSTAMP();    // after W3(0): W3(1)
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w1(1, 1); pass(I1{}, 2); PSTAMP();    // after W3(1): W3(2)
        if (!xw) vmw<6>(); bar(); if (!xw) issue_w3(3, 2); pass(I2{}, 0); PSTAMP();    // after W3(2): W1'(1), 4 stores
        if (!xw) vmw<4>(); bar(); if (!xw) issue_w1(0, 0); pass(I3{}, 2); PSTAMP();    // after W3(3): 4 stores
        yprev = ybase + 3 * 128;
#ifdef AP_TRACE
        stamp_i = 0; ++tile_no;
#endif
        if (!has_next) break;
        tile = next;
    }
    put_stores(0, yprev);
    put_stores(1, yprev);
    vm<0>();                                                 // no LDS-DMA may be in flight when the LDS is released
}

// =====================================================================================================================
// Downsample block (CIN = 64, K3 = 128 = [mid2 | x]): the same persistent, wave-specialised scheme.  The x halo tile
// is one 32 KiB step that stays resident (it is conv3's second K segment), so two tile buffers alternate and the
// next tile's x rides along with taps 0..7; the weight ring has 4 slots (W1 | taps cycle through three | the 16 KiB
// W3 chunks take two).  LDS map: x tile A | x tile B | mid1 / stage | mid2 | weight ring 4 x 8 KiB | constants.
constexpr int D_XA = 0, D_XB = 32768, D_M1 = 65536, D_M2 = 98304, D_WR = 126976, D_CONST = 159744, D_TOTAL = 162816;

__global__ void __launch_bounds__(512) bneck64ds_kernel(const BneckArgs p) {
    constexpr int CIN = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g4 = lane >> 4;
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const bool xw = wave >= 4;                               // DMA role: x tiles (waves 4-7) or weight chunks (0-3)
    const int wq = wave & 3;
    const int G = gridDim.x;
    const int wm2 = wave & 1, wn2 = wave >> 1;
    const int rb = wn2 < 2 ? wn2 * 4 : 8 + (wn2 - 2) * 3;
    const int nb = wn2 < 2 ? 4 : 3;
    const uint32_t j3adj = nb == 4 ? 0u : (uint32_t)-2048;

    if (tid < 192) {                                         // BatchNorm constants -> LDS (once per workgroup)
        const float* src = tid < 16 ? p.s1 + tid * 4 : tid < 32 ? p.h1 + (tid - 16) * 4 : tid < 48 ? p.s2 + (tid - 32) * 4
                         : tid < 64 ? p.h2 + (tid - 48) * 4 : tid < 128 ? p.s3 + (tid - 64) * 4 : p.h3 + (tid - 128) * 4;
        const float4 v = *(const float4*)src;
        u32x4 u;
        u.x = __builtin_bit_cast(uint32_t, v.x); u.y = __builtin_bit_cast(uint32_t, v.y);
        u.z = __builtin_bit_cast(uint32_t, v.z); u.w = __builtin_bit_cast(uint32_t, v.w);
        wr128<0>(lds0 + D_CONST + tid * 16, u);
    }
    lgkm<0>();

    // ---------------------------------------------------------------- DMA sources (see bneck256_kernel)
    const unsigned char* xg = (const unsigned char*)p.x;
    uint32_t xoff0 = 0, xlive = 0;
    uint32_t xrc = (uint32_t)(prow << 8 | pchunk);
    auto setup_x = [&](int tile) {
        const int n = tile / p.tiles_per_img, trem = tile - n * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        xlive = 0;
        asm volatile("" : "+v"(xrc));
        const int prow = xrc >> 8, pchunk = xrc & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int h = (8 * wq + i) * 8 + prow;
            const int yy = ty * TS - 1 + (h >> 4), xx = tx * TS - 1 + (h & 15);
            const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            xlive |= ok ? (1u << i) : 0u;
        }
        xoff0 = (uint32_t)((((n * p.H + ty * TS - 1 + 4 * wq) * p.W + tx * TS - 1 + prow) * CIN + pchunk * 8) * 2);
    };
    const int wrow = 2 * wq * 8 + prow;
    uint32_t wrc = (uint32_t)(wrow << 8 | pchunk);
    auto dma = [&](const unsigned char* src, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    auto issue_x1 = [&](int base, int i) {                   // piece i (of 8 per x wave) of the 64-channel halo tile
        dma(((xlive >> i) & 1u) ? xg + (uint32_t)(xoff0 + (i >> 1) * p.W * (CIN * 2) + (i & 1) * (8 * CIN * 2)) : (const unsigned char*)p.zero,
            base + (8 * wq + i) * 1024);
    };
    auto issue_w = [&](const void* w, uint32_t row_bytes, uint32_t col_bytes, int slot) {   // 2 pieces per weight wave
        const unsigned char* src = (const unsigned char*)w + (uint32_t)((wrc >> 8) * row_bytes + (wrc & 7) * 16 + col_bytes);
        dma(src, D_WR + slot * 8192 + 2 * wq * 1024);
        dma(src + 8 * row_bytes, D_WR + slot * 8192 + (2 * wq + 1) * 1024);
    };
    auto issue_w1 = [&]() { issue_w(p.w1, CIN * 2, 0, 0); };
    auto issue_tap = [&](int t) { issue_w(p.w2, 576 * 2, t * 128, 1 + t % 3); };
    auto issue_w3 = [&](int nc, int u) { issue_w(p.w3, 128 * 2, nc * 64 * 256 + u * 128, ((nc & 1) ? 2 : 0) + u); };

    // ---------------------------------------------------------------- fragment / epilogue addresses
    const uint32_t sw0 = (uint32_t)((g4 ^ (lr & 7)) << 4);
    const uint32_t wrow2 = (uint32_t)((32 * wm2 + lr) * 128);
    const uint32_t p1row = (uint32_t)((wn2 * 64 + lr) * 128);
    uint32_t m1a[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) m1a[kx] = lds0 + D_M1 + (rb * 16 + lr + kx) * 128 + (uint32_t)((g4 ^ ((lr + kx) & 7)) << 4);
    uint32_t eoff[2], e1off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = 4 * wm2 + 2 * i + (g4 >> 1);
        eoff[i] = (uint32_t)((rb * 16 + lr) * 128 + ((chunk ^ (lr & 7)) << 4) + (g4 & 1) * 8);
        e1off[i] = (uint32_t)((wn2 * 64 + lr) * 128 + ((chunk ^ (lr & 7)) << 4) + (g4 & 1) * 8);
    }
    const uint32_t cch = lds0 + D_CONST + (uint32_t)((32 * wm2 + 4 * g4) * 4);
    // Output stage, WAVE-PRIVATE: a wave's 2 x 4 accumulator tiles are [<= 4 rows x 16 px] x 32 channels = 64 B per pixel;
    // it transposes them through its own 4 KiB of the stage (rows of 64 B, 16-byte chunk index XOR (row >> 2) & 3) into
    // 16 bytes per lane and stores 64-byte pixel segments (the sibling wave of the channel pair stores the other half of
    // the 128-byte line).  No workgroup barrier between the accumulators and the stores.  Exactly four store
    // instructions per wave and pass (the wait counts rely on it): lanes past the wave's nb*14*4 pieces are masked,
    // lane 0 of an otherwise empty instruction repeats piece 0.
    const uint32_t stg = lds0 + D_M1 + wave * 4096;
    uint32_t sea[2];                                         // accumulator-layout write address of channel block i
#pragma unroll
    for (int i = 0; i < 2; ++i)
        sea[i] = stg + lr * 64 + (uint32_t)(((2 * i + (g4 >> 1)) ^ ((lr >> 2) & 3)) << 4) + (g4 & 1) * 8;
    uint32_t st_lds[4], st_off[4];
    uint32_t st_ok = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int idx = it * 64 + lane;
        const bool ok = idx < nb * TS * 4;
        if (!ok) idx = 0;
        st_ok |= (ok || lane == 0) ? (1u << it) : 0u;
        const int q = idx >> 2, c = idx & 3, j = q / TS, ox = q - j * TS, r = j * 16 + ox;
        st_lds[it] = stg + r * 64 + (uint32_t)((c ^ ((r >> 2) & 3)) << 4);
        st_off[it] = (uint32_t)((((rb + j) * p.W + ox) * 256 + 32 * wm2 + c * 8) * 2);
    }
    u32x4 sv[4];
    auto put_stores = [&](int half, unsigned char* base) {
        if (half == 0) {
            *(u32x4*)(base + st_off[0]) = sv[0];
            *(u32x4*)(base + st_off[1]) = sv[1];
        } else {
            if (st_ok & 4u) *(u32x4*)(base + st_off[2]) = sv[2];
            if (st_ok & 8u) *(u32x4*)(base + st_off[3]) = sv[3];
        }
    };
    f32x4 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ================================================================ prologue: the queue state a previous tile would leave
    int tile = xcd_remap(blockIdx.x, G);
    int xcur = D_XA;                                         // buffer of the current tile's x halo; the other one takes the next
    if (xw) {
        setup_x(tile);
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_x1(D_XA, i);
    } else { issue_w1(); issue_tap(0); }

    unsigned char* yprev = nullptr;
    for (bool first = true;; first = false) {
        const int n = tile / p.tiles_per_img, trem = tile - n * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int y0 = ty * TS, x0 = tx * TS;
        unsigned char* const ybase = (unsigned char*)p.y + (((size_t)n * p.H + y0) * p.W + x0) * 512;
        const int xnext = xcur ^ (D_XA ^ D_XB);
        asm volatile("" : "+v"(wrc));
        // ------------------------------------------------------------ phase 1 (one 64-deep step)
        bar();                                               // previous tile: stage / mid2 / W3 reads complete
        if (!xw) { issue_tap(1); issue_tap(2); }
        // vm<N>: N = operations this wave queued after the ones needed
        if (first) vm<0>(); else if (xw) vmx<12>(); else vmw<10>();      // x tile (then: 12 stores) / W1 (then: T0, 4 stores, T1, T2)
        bar();
        zero_acc();
        {
            const uint32_t wa = lds0 + D_WR + wrow2, xa = lds0 + xcur + p1row;
            if (yprev)                                       // the previous tile's last slice leaves under this step
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, acc,
                            [&](int half) { put_stores(half, yprev); });
            else
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, xa + sw0, (xa + sw0) ^ 64, acc);
        }
        bar();
        const int next = tile + G;
        const bool has_next = next < p.total;
        if (xw) setup_x(has_next ? next : tile);             // (the last tile re-fetches its own: the bookkeeping stays)
        else issue_w3(0, 0);                                 // W1's slot
        {   // epilogue 1 -> mid1
            const bool xin = (unsigned)(x0 - 1 + lr) < (unsigned)p.W;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = rd_f4<C_S1 * 4>(cch + 64 * i), sh = rd_f4<C_H1 * 4>(cch + 64 * i);
                const uint32_t ea = lds0 + D_M1 + e1off[i];
                const int yb = y0 - 1 + 4 * wn2;
                wr64<0>(ea, bn_relu_pack(acc[i][0], sc, sh, xin && (unsigned)(yb + 0) < (unsigned)p.H));
                wr64<2048>(ea, bn_relu_pack(acc[i][1], sc, sh, xin && (unsigned)(yb + 1) < (unsigned)p.H));
                wr64<4096>(ea, bn_relu_pack(acc[i][2], sc, sh, xin && (unsigned)(yb + 2) < (unsigned)p.H));
                wr64<6144>(ea, bn_relu_pack(acc[i][3], sc, sh, xin && (unsigned)(yb + 3) < (unsigned)p.H));
            }
        }
        lgkm<0>();
        // ------------------------------------------------------------ phase 2 (tap t in ring slot 1 + t % 3)
        zero_acc();
        auto tap = [&](auto KY, auto KX) {
            constexpr int ky = decltype(KY)::value, kx = decltype(KX)::value, t = ky * 3 + kx;
            const uint32_t wa = lds0 + D_WR + (1 + t % 3) * 8192 + wrow2;
            step_2x4<ky * 2048>(wa + sw0, (wa + sw0) ^ 64, m1a[kx], m1a[kx] ^ 64, m1a[kx] + j3adj, (m1a[kx] ^ 64) + j3adj, acc,
                                [&](int half) { if (xw && t < 8 && half) issue_x1(xnext, t); });   // next tile's x, a piece per tap
        };
        if (!xw) vmw<14>(); bar(); tap(I0{}, I0{});                                   // after T0: 4 st, T1, T2, 4 st, W3(0)a
        if (!xw) vmw<8>(); bar(); if (!xw) issue_tap(3); tap(I0{}, I1{});             // after T1: T2, 4 st, W3(0)a
        if (!xw) vmw<8>(); bar(); if (!xw) issue_tap(4); tap(I0{}, I2{});             // after T2: 4 st, W3(0)a, T3
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(5); tap(I1{}, I0{});
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(6); tap(I1{}, I1{});
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(7); tap(I1{}, I2{});
        if (!xw) vmw<2>(); bar(); if (!xw) issue_tap(8); tap(I2{}, I0{});
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(0, 1); tap(I2{}, I1{});           // slot 1: tap 6 is done
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(1, 0); tap(I2{}, I2{});           // slot 2: tap 7 is done
        {   // epilogue 2 -> mid2
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = rd_f4<C_S2 * 4>(cch + 64 * i), sh = rd_f4<C_H2 * 4>(cch + 64 * i);
                const uint32_t ea = lds0 + D_M2 + eoff[i];
                wr64<0>(ea, bn_relu_pack(acc[i][0], sc, sh, true));
                wr64<2048>(ea, bn_relu_pack(acc[i][1], sc, sh, true));
                wr64<4096>(ea, bn_relu_pack(acc[i][2], sc, sh, true));
                if (nb == 4) wr64<6144>(ea, bn_relu_pack(acc[i][3], sc, sh, true));
            }
        }
        lgkm<0>();
        // ------------------------------------------------------------ phase 3: y = relu(W3 . [mid2 | x] + h3), 4 slices
        auto pass = [&](auto NC) {
            constexpr int nc = decltype(NC)::value;
            zero_acc();
            u32x4 cs[2], ch[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                cs[i] = rd128<(C_S3 + 64 * nc) * 4>(cch + 64 * i);
                ch[i] = rd128<(C_H3 + 64 * nc) * 4>(cch + 64 * i);
            }
            const uint32_t wa = lds0 + D_WR + ((nc & 1) ? 2 : 0) * 8192 + wrow2;
            const uint32_t m2a = m1a[0] + (D_M2 - D_M1);
            const uint32_t xa = m1a[1] - D_M1 + xcur;        // centre pixels of the resident x halo tile (row + 1, column + 1)
            if constexpr (nc > 0)
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, m2a, m2a ^ 64, m2a + j3adj, (m2a ^ 64) + j3adj, acc,
                            [&](int half) { put_stores(half, ybase + (nc - 1) * 128); });
            else
                step_2x4<0>(wa + sw0, (wa + sw0) ^ 64, m2a, m2a ^ 64, m2a + j3adj, (m2a ^ 64) + j3adj, acc);
            step_2x4<2048>(wa + 8192 + sw0, (wa + 8192 + sw0) ^ 64, xa, xa ^ 64, xa + j3adj, (xa ^ 64) + j3adj, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 sc = as_f4(cs[i]), sh = as_f4(ch[i]);
                const uint32_t ea = sea[i];
                wr64<0>(ea, bn_relu_pack(acc[i][0], sc, sh, true));
                wr64<1024>(ea, bn_relu_pack(acc[i][1], sc, sh, true));
                wr64<2048>(ea, bn_relu_pack(acc[i][2], sc, sh, true));
                if (nb == 4) wr64<3072>(ea, bn_relu_pack(acc[i][3], sc, sh, true));
            }
            lgkm<0>();                                       // (own writes; the stage region is this wave's)
#pragma unroll
            for (int it = 0; it < 4; ++it) sv[it] = rd128<0>(st_lds[it]);
            lgkm<0>();
        };
        if (!xw) vmw<2>(); bar(); if (!xw) issue_w3(1, 1); pass(I0{});                               // after W3(0)b: W3(1)a; slot 3: tap 8 is done
        if (!xw) vmw<0>(); bar(); if (!xw) { issue_w3(2, 0); issue_w3(2, 1); } pass(I1{});
        if (!xw) vmw<4>(); bar(); if (!xw) { issue_w3(3, 0); issue_w3(3, 1); } pass(I2{});          // after W3(2): 4 stores
        if (!xw) vmw<4>(); bar(); if (!xw) { issue_w1(); issue_tap(0); } pass(I3{});                 // after W3(3): 4 stores
        yprev = ybase + 3 * 128;
        if (!has_next) break;
        tile = next;
        xcur = xnext;
    }
    put_stores(0, yprev);
    put_stores(1, yprev);
    vm<0>();                                                 // no LDS-DMA may be in flight when the LDS is released
}

hipError_t launch_persistent(const BneckArgs& a, bool ds, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    int& n_cu = n_cu_dev[dev];
    if (!n_cu) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)bneck256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_TOTAL);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)bneck64ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, D_TOTAL);
        if (e != hipSuccess) return e;
        n_cu = n;
    }
    if ((size_t)a.N * a.H * a.W * 512 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit x offsets
    const int grid = a.total < n_cu ? a.total : n_cu;
    if (ds) hipLaunchKernelGGL(bneck64ds_kernel, dim3(grid), dim3(512), D_TOTAL, st, a);
    else hipLaunchKernelGGL(bneck256_kernel, dim3(grid), dim3(512), P_TOTAL, st, a);
    return hipGetLastError();
}

}  // namespace

// Fused layer1 bottleneck.  cin = 256 (identity blocks) or 64 with ds = 1 (block 0, downsample folded into W3).
hipError_t ap_launch_bneck64(BneckArgs a, int cin, int ds, hipStream_t st) {
    if (!a.zero || a.H % TS || a.W % TS || a.N <= 0) return hipErrorInvalidValue;
    a.tiles_x = a.W / TS;
    a.tiles_per_img = a.tiles_x * (a.H / TS);
    a.total = a.N * a.tiles_per_img;
    if (cin == 256 && !ds) return launch_persistent(a, false, st);
    if (cin == 64 && ds) return launch_persistent(a, true, st);
    return hipErrorInvalidValue;
}
