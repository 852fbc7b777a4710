// Phase-interleaved implicit-GEMM convolution for gfx950: 256-channel x (128..256)-pixel tiles, one workgroup
// of 8 waves per CU, two wave groups running half a phase apart (one in its MFMA cluster while the other
// reads LDS and issues DMA).  The deep contractions of the trunk (layer3/layer4 conv1, conv2, conv3+downsample;
// reference: Bottleneck.forward, copenet/src/copenet/models/model_copenet.py:27-47) run here; conv_pipe.hip keeps
// the residual layers and everything with fewer than 256 output channels.
//
// Same math and the same K order as conv_pipe.hip / conv_igemm.hip (taps outer, channels inner, 32-wide MFMA
// chunks ascending, v_mfma_f32_16x16x32_bf16 with the weights as the A operand), so results are bit-identical
// to every other tile configuration.
//
// Tile: 256 output channels x (32*FMW) pixels, FMW = 4..8 pixel fragments per wave.  Waves 2 (pixel halves:
// "groups") x 4 (64 channels each); a wave owns FMW x 4 accumulator fragments (FMW*16 VGPRs).  FMW is how the
// host fits a layer to the 256 CUs: its pixel rows are cut into rounds of one tile per CU whose heights differ
// by at most one fragment (100 352 rows x 256 channels = one round of 224-row tiles + one of 192-row tiles
// instead of 1.53 rounds of 256-row tiles).
//
// LDS: two K-tile buffers of 64 KiB: [A: 256 rows x 128 B | B: 256 rows x 128 B], BK = 64 bf16.  Rows are filled
// by LDS-DMA in 1-KiB pieces (8 rows) with the XOR swizzle on the SOURCE address (chunk c of row r sits at
// position c ^ (r & 7)); the MFMA side reads 16-byte fragments with the same XOR: no bank conflicts.
//
// K loop, per K-tile t (buffer t & 1), four phases; a wave's 4 x FMW x 2 MFMAs per K-tile run as
//   P1: a0 x b0    reads a0 (pixel frags 0-3, both 32-wide halves: 8 x b128) and b0 (channel frags 0-1: 4 x b128)
//   P2: a0 x b1    reads b1 (channel frags 2-3)
//   P3: a1 x b1    reads a1 (pixel frags 4 .. FMW-1) into the registers of a0
//   P4: a1 x b0    (b0 kept in registers)
// and every phase is  { ds_reads ; 2 DMA pieces ; s_barrier ; lgkmcnt(0) ; MFMA cluster ; s_barrier }.  Group 1
// runs one barrier behind group 0, so on every SIMD one wave is in its MFMA cluster while its partner is in the
// read/DMA part of the phase.
// DMA schedule (what a region's last reader allows): the regions of a buffer are free for the NEXT tile of the same
// parity as soon as their last ds_read has retired -- A rows 0-63 of each group (a0) after P1, B after P2, A rows
// 64-127 (a1) after P3 -- so tile t issues
//   P1: a1 rows of tile t+1      P2: a0 rows of tile t+2      P3: B rows 0-127 of t+2      P4: B rows 128-255 of t+2
// and ONE counted wait per K-tile, s_waitcnt vmcnt(6) at the end of P4's issue, retires everything up to P1's pieces:
// every piece has at least four phases (a whole K-tile of MFMAs) of flight time and three half-tiles stay in flight
// across the K-tile boundary.  Hazards: a wave's DMA into a region is issued after a barrier that every wave
// passed AFTER retiring its reads of that region (group 1 waits lgkmcnt(0) BEFORE the first barrier of a phase
// for exactly this reason; group 0's lgkmcnt(0) sits before its MFMA cluster, which precedes the second barrier);
// data are read one barrier after the last wave's vmcnt wait.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

// Timing-only builds for tuning (results are WRONG, times are valid): -DPH_ABLATE=<bits>
//   1 no DMA in the K loop | 2 no fragment reads | 4 no MFMAs | 8 no pre-barrier lgkmcnt wait of group 1 | 16 no s_setprio
#ifndef PH_ABLATE
#define PH_ABLATE 0
#endif
// K order of a multi-tap convolution: 0 = taps outer, channels inner (the order of every other conv kernel here:
// bit-identical results); 1 = channel chunks outer, taps inner (experiment: the nine taps of a chunk re-read the same
// input rows back to back, L2-resident)
#ifndef PH_ORDER
#define PH_ORDER 0
#endif

namespace {

constexpr int PH_BUF = 65536;        // one K-tile buffer
constexpr int PH_B_OFF = 32768;      // weight rows inside a buffer
constexpr int PH_LDS = 2 * PH_BUF;
constexpr bool PH_PREWAIT = !(PH_ABLATE & 8);
constexpr bool PH_DMA = !(PH_ABLATE & 1);

template <int N> __device__ __forceinline__ void ph_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void ph_wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // keep register-only MFMAs below the wait (they ignore "memory")
}
template <int OFF> __device__ __forceinline__ void ph_lds_read(u32x4& r, uint32_t addr) {
#if PH_ABLATE & 2
    asm volatile("" : "+v"(r) : "v"(addr));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
#endif
}
__device__ __forceinline__ f32x4 ph_mfma(const u32x4& w, const u32x4& x, const f32x4& c) {
#if PH_ABLATE & 4
    f32x4 r = c;
    asm volatile("" : "+v"(r) : "v"(w), "v"(x));
    return r;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ void ph_prio(int hi) {
#if !(PH_ABLATE & 16)
    if (hi) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
}

// position of a K-tile in the contraction: tap (r, s) and 64-channel chunk cb of the first segment, or chunk cb
// of the second segment (seg = 1: the folded 1x1 downsample over x2)
struct KCursor {
    int tap, r, s, cb, seg;
    int boff;                        // byte offset of the K-tile inside a packed weight row
    long long tapoff;                // byte offset of the tap's pixel relative to the row's (h0, w0) pixel
};

template <int FMW, bool TAPS, bool SEG2>
__global__ void __launch_bounds__(512, 2) conv_phase_kernel(const ConvArgs p) {
    static_assert(FMW >= 4 && FMW <= 8, "4..8 pixel fragments per wave");
    static_assert(!(TAPS && SEG2), "the folded downsample rides on a pointwise conv3");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int GROWS = 16 * FMW, TROWS = 32 * FMW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile - bm * p.ntiles;
    const int m0 = p.row0 + bm * TROWS;

    // ---------------------------------------------------------------- DMA source state
    // A rows of this thread: j = sub * 2 + i, piece = wave * 2 + i (0..15), group g = piece >> 3,
    // row inside the group rg = (piece & 7) * 8 + (lane >> 3) + sub * 64
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* zg = (const unsigned char*)p.zero;
    const unsigned char* src[4];
    uint32_t off2[SEG2 ? 4 : 1];
    uint32_t okmask[TAPS ? 2 : 1];   // TAPS: 16 tap bits per row (rows 0,1 in word 0; 2,3 in word 1); else bit j = row valid
    okmask[0] = 0;
    if constexpr (TAPS) okmask[1] = 0;
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sub = j >> 1, i = j & 1;
        const int piece = wave * 2 + i, g = piece >> 3;
        const int rg = (piece & 7) * 8 + prow + sub * 64;
        const int m = m0 + g * GROWS + rg;
        const bool valid = rg < GROWS && m < p.M;
        src[j] = zg;
        if constexpr (SEG2) off2[j] = 0;
        if constexpr (!TAPS) {
            if (valid) {
                src[j] = xg + ((size_t)m * p.ldx + pchunk * 8) * 2;
                okmask[0] |= 1u << j;
                if constexpr (SEG2) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    off2[j] = (uint32_t)(((((size_t)n * p.H2 + (size_t)ho * p.stride2) * p.W2 + (size_t)wo * p.stride2) * p.ldx2 +
                                          pchunk * 8) * 2);
                }
            }
        } else {
            if (valid) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
                src[j] = xg + (((ptrdiff_t)n * p.H * p.W + (ptrdiff_t)h0 * p.W + w0) * p.ldx + pchunk * 8) * 2;
                uint32_t bits = 0;
                for (int r = 0; r < p.KH; ++r)
                    for (int s = 0; s < p.KW; ++s)
                        if ((unsigned)(h0 + r) < (unsigned)p.H && (unsigned)(w0 + s) < (unsigned)p.W) bits |= 1u << (r * p.KW + s);
                okmask[j >> 1] |= bits << ((j & 1) * 16);
            }
        }
    }
    // weight rows: row = half * 128 + piece * 8 + prow of the tile's 256 channels; one per-lane base + uniform offsets
    const size_t wld2 = (size_t)p.wld * 2;
    const unsigned char* wbase = (const unsigned char*)p.w + ((size_t)bn * 256 + prow) * wld2 + pchunk * 16;

    const int cpb = p.Cin >> 6;
    const int ntap = TAPS ? p.KH * p.KW : 1;
    const int KT = ntap * cpb + (SEG2 ? (p.Cin2 >> 6) : 0);

    auto cursor_at = [&](int kt) {
        KCursor c;
        c.seg = 0; c.tap = 0; c.r = 0; c.s = 0; c.cb = kt; c.tapoff = 0;
        if constexpr (TAPS) {
            if (PH_ORDER == 0) { c.tap = kt / cpb; c.cb = kt - c.tap * cpb; }
            else { c.cb = kt / ntap; c.tap = kt - c.cb * ntap; }
            c.r = c.tap / p.KW;
            c.s = c.tap - c.r * p.KW;
            c.tapoff = ((long long)c.r * p.W + c.s) * p.ldx * 2;
        }
        if constexpr (SEG2) {
            if (kt >= cpb) { c.seg = 1; c.cb = kt - cpb; }
        }
        c.boff = ((c.seg ? ntap * cpb : c.tap * cpb) + c.cb) * 128;
        return c;
    };
    auto cursor_next = [&](KCursor& c) {
        if constexpr (TAPS) {
            if (PH_ORDER == 0) {
                if (++c.cb == cpb) {
                    c.cb = 0;
                    ++c.tap;
                    if (++c.s == p.KW) { c.s = 0; ++c.r; }
                }
            } else {
                ++c.tap;
                if (++c.s == p.KW) { c.s = 0; ++c.r; }
                if (c.tap == ntap) { c.tap = 0; c.r = 0; c.s = 0; ++c.cb; }
            }
            c.tapoff = ((long long)c.r * p.W + c.s) * p.ldx * 2;
        } else {
            ++c.cb;
        }
        if constexpr (SEG2) {
            if (c.seg == 0 && c.cb == cpb) { c.seg = 1; c.cb = 0; }
        }
        c.boff = ((c.seg ? ntap * cpb : c.tap * cpb) + c.cb) * 128;
    };
    // the two A pieces (i = 0, 1) of sub-tile `sub` at K position c into buffer `buf`
    auto issue_a = [&](int sub, const KCursor& c, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = sub * 2 + i;
            const int piece = wave * 2 + i;
            const unsigned char* ptr;
            if constexpr (TAPS) {
                const bool ok = (okmask[j >> 1] >> ((j & 1) * 16 + c.tap)) & 1u;
                ptr = ok ? src[j] + (c.tapoff + c.cb * 128) : zg;
            } else {
                const bool ok = (okmask[0] >> j) & 1u;
                if constexpr (SEG2) {
                    const unsigned char* q = c.seg ? (const unsigned char*)p.x2 + off2[j] : src[j];
                    ptr = ok ? q + c.cb * 128 : zg;
                } else {
                    ptr = ok ? src[j] + c.cb * 128 : zg;
                }
            }
            const int dst = buf * PH_BUF + (piece >> 3) * 16384 + sub * 8192 + (piece & 7) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr,
                                             (__attribute__((address_space(3))) void*)(smem + dst), 16, 0, 0);
        }
    };
    // the two B pieces of channel half `hh` of the K-tile at c into buffer `buf`
    auto issue_b = [&](int hh, const KCursor& c, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;
            const unsigned char* ptr = wbase + (size_t)(hh * 128 + piece * 8) * wld2 + c.boff;
            const int dst = buf * PH_BUF + PH_B_OFF + (hh * 128 + piece * 8) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr,
                                             (__attribute__((address_space(3))) void*)(smem + dst), 16, 0, 0);
        }
    };

    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g4 = lane >> 4;
    const uint32_t sw0 = (uint32_t)((g4 ^ (lr & 7)) << 4), sw1 = (uint32_t)(((4 + g4) ^ (lr & 7)) << 4);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t xrow = lds0 + (uint32_t)((wr * 128 + lr) * 128);
    const uint32_t wrow = lds0 + PH_B_OFF + (uint32_t)((wc * 64 + lr) * 128);
    const uint32_t xa0 = xrow + sw0, xa1 = xrow + sw1, wa0 = wrow + sw0, wa1 = wrow + sw1;
    f32x4 acc[FMW][4];
#pragma unroll
    for (int f = 0; f < FMW; ++f)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af[4][2], b0[2][2], b1[2][2];
#if PH_ABLATE & 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            af[i][h] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            b0[i & 1][h] = af[i][h]; b1[i & 1][h] = af[i][h];
        }
#endif

    // ---------------------------------------------------------------- prologue: tile 0 whole, tile 1 minus its a1 rows
    {
        KCursor c = cursor_at(0);
        issue_a(0, c, 0);
        issue_b(0, c, 0);
        issue_b(1, c, 0);
        issue_a(1, c, 0);
        if (KT > 1) {
            cursor_next(c);
            issue_a(0, c, 1);
            issue_b(0, c, 1);
            issue_b(1, c, 1);
            ph_wait_vmcnt<6>();
        } else {
            ph_wait_vmcnt<0>();
        }
    }
    KCursor c1 = cursor_at(KT > 1 ? 1 : 0);      // a1 rows of tile t+1
    KCursor c0 = cursor_at(KT > 2 ? 2 : 0);      // a0 rows of tile t+2
    KCursor c2 = c0;                             // tile t+2 as issued in P2 (its B rows follow in P3, P4)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const uint32_t so = (uint32_t)buf * PH_BUF;
        const bool more1 = kt + 1 < KT, more2 = kt + 2 < KT;
        // ------------------------------------------------------------ P1: a0 x b0
        ph_lds_read<0>(af[0][0], xa0 + so);     ph_lds_read<0>(af[0][1], xa1 + so);
        ph_lds_read<2048>(af[1][0], xa0 + so);  ph_lds_read<2048>(af[1][1], xa1 + so);
        ph_lds_read<4096>(af[2][0], xa0 + so);  ph_lds_read<4096>(af[2][1], xa1 + so);
        ph_lds_read<6144>(af[3][0], xa0 + so);  ph_lds_read<6144>(af[3][1], xa1 + so);
        ph_lds_read<0>(b0[0][0], wa0 + so);     ph_lds_read<0>(b0[0][1], wa1 + so);
        ph_lds_read<2048>(b0[1][0], wa0 + so);  ph_lds_read<2048>(b0[1][1], wa1 + so);
        if (PH_DMA && more1) { issue_a(1, c1, buf ^ 1); cursor_next(c1); }
        if (PH_PREWAIT && wr == 1) ph_wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        ph_wait_lgkmcnt<0>();
        ph_prio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[f][n] = ph_mfma(b0[n][h], af[f][h], acc[f][n]);
        ph_prio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------ P2: a0 x b1
        ph_lds_read<4096>(b1[0][0], wa0 + so);  ph_lds_read<4096>(b1[0][1], wa1 + so);
        ph_lds_read<6144>(b1[1][0], wa0 + so);  ph_lds_read<6144>(b1[1][1], wa1 + so);
        if (PH_DMA && more2) { c2 = c0; issue_a(0, c2, buf); cursor_next(c0); }
        if (PH_PREWAIT && wr == 1) ph_wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        ph_wait_lgkmcnt<0>();
        ph_prio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[f][2 + n] = ph_mfma(b1[n][h], af[f][h], acc[f][2 + n]);
        ph_prio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------ P3: a1 x b1
        if constexpr (FMW > 4) { ph_lds_read<8192>(af[0][0], xa0 + so);  ph_lds_read<8192>(af[0][1], xa1 + so); }
        if constexpr (FMW > 5) { ph_lds_read<10240>(af[1][0], xa0 + so); ph_lds_read<10240>(af[1][1], xa1 + so); }
        if constexpr (FMW > 6) { ph_lds_read<12288>(af[2][0], xa0 + so); ph_lds_read<12288>(af[2][1], xa1 + so); }
        if constexpr (FMW > 7) { ph_lds_read<14336>(af[3][0], xa0 + so); ph_lds_read<14336>(af[3][1], xa1 + so); }
        if (PH_DMA && more2) issue_b(0, c2, buf);
        if (PH_PREWAIT && wr == 1) ph_wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        ph_wait_lgkmcnt<0>();
        ph_prio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 4; f < FMW; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[f][2 + n] = ph_mfma(b1[n][h], af[f - 4][h], acc[f][2 + n]);
        ph_prio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------ P4: a1 x b0
        if (PH_DMA && more2) {
            issue_b(1, c2, buf);
            ph_wait_vmcnt<6>();                  // everything up to P1's pieces (a1 rows of tile kt+1) has landed
        } else {
            ph_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        ph_prio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 4; f < FMW; ++f)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[f][n] = ph_mfma(b0[n][h], af[f - 4][h], acc[f][n]);
        ph_prio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // matches group 1's extra barrier at the top

    // ---------------------------------------------------------------- epilogue: BN + ReLU straight from the accumulators
    // lane = (pixel lr, channels g4*4 .. +3) of a fragment; the four channel fragments of a wave complete each pixel's
    // 128-byte line.  Same expression as conv_pipe.hip's epilogue (bit-identical results).
    bf16_t* __restrict__ yg = (bf16_t*)p.y;
    auto store_tile = [&](auto relu_tag) {
        constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int ch = bn * 256 + wc * 64 + n * 16 + g4 * 4;
            const float4 sc = *(const float4*)(p.scale + ch);
            const float4 sh = *(const float4*)(p.shift + ch);
#pragma unroll
            for (int f = 0; f < FMW; ++f) {
                const int m = m0 + wr * GROWS + f * 16 + lr;
                float v0 = acc[f][n][0] * sc.x + sh.x, v1 = acc[f][n][1] * sc.y + sh.y;
                float v2 = acc[f][n][2] * sc.z + sh.z, v3 = acc[f][n][3] * sc.w + sh.w;
                if constexpr (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                if (m < p.M) {
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                    *(uint2*)(yg + (size_t)m * p.ldy + ch) = o;
                }
            }
        }
    };
    if (p.relu) store_tile(std::true_type{});
    else store_tile(std::false_type{});
}

template <int FMW, bool TAPS, bool SEG2>
hipError_t launch_phase(const ConvArgs& a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    auto kern = conv_phase_kernel<FMW, TAPS, SEG2>;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(512), PH_LDS, st, a);
    return hipGetLastError();
}

template <int FMW>
hipError_t launch_phase_fmw(const ConvArgs& a, bool taps, bool seg2, hipStream_t st) {
    if (taps) return launch_phase<FMW, true, false>(a, st);
    if (seg2) return launch_phase<FMW, false, true>(a, st);
    return launch_phase<FMW, false, false>(a, st);
}

}  // namespace

bool ap_conv_phase_supported(const ConvArgs& a, int is_bf16) {
    if (!is_bf16 || a.res) return false;
    if (a.Cout % 256 || a.Cin % 64 || a.Cin < 64) return false;
    if (a.KH != a.KW || a.KH * a.KW > 16) return false;
    if (a.x2 && (a.Cin2 % 64 || a.KH != 1 || a.stride != 1 || a.pad != 0)) return false;
    if (a.x2 && (size_t)a.N * a.H2 * a.W2 * a.ldx2 * 2 >= 0xffffffffull) return false;
    if (a.ldx % 8 || a.ldy % 4 || (a.x2 && a.ldx2 % 8) || a.wld % 8) return false;
    return true;
}

// One launch: pixel rows [row0, row0 + mtiles * 32 * fmw) (clipped at a.M) x all output channels, fmw = 4..8
hipError_t ap_launch_conv_phase(ConvArgs a, int fmw, int row0, int mtiles, hipStream_t st) {
    if (!a.zero || !ap_conv_phase_supported(a, 1) || mtiles <= 0 || row0 < 0) return hipErrorInvalidValue;
    a.row0 = row0;
    a.mtiles = mtiles;
    a.ntiles = a.Cout / 256;
    const bool taps = !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0);
    const bool seg2 = a.x2 != nullptr;
    switch (fmw) {
        case 4: return launch_phase_fmw<4>(a, taps, seg2, st);
        case 5: return launch_phase_fmw<5>(a, taps, seg2, st);
        case 6: return launch_phase_fmw<6>(a, taps, seg2, st);
        case 7: return launch_phase_fmw<7>(a, taps, seg2, st);
        case 8: return launch_phase_fmw<8>(a, taps, seg2, st);
    }
    return hipErrorInvalidValue;
}

// Fit a layer to the chip: the M pixel rows are cut into R rounds of (n_cu / ntiles) row tiles each, i.e. one tile per
// CU and round; the per-round heights (in 32-row units = one fragment per wave) differ by at most one.  Rounds of
// equal height become one launch.
hipError_t ap_conv_phase_auto(const ConvArgs& a, int n_cu, hipStream_t st) {
    const int ntiles = a.Cout / 256;
    int slots = n_cu / ntiles;                               // row tiles per round
    if (slots < 1) slots = 1;
    const long per_slot = ((long)a.M + slots - 1) / slots;   // rows one slot works through
    const int units = (int)((per_slot + 31) / 32);
    int R = (units + 7) / 8;
    int base = units / R, rem = units % R;                   // rem rounds of base+1 fragments, R-rem of base
    if (base < 4) { base = 4; rem = 0; R = (units + 3) / 4; }
    int row0 = 0;
    for (int pass = 0; pass < 2 && row0 < a.M; ++pass) {
        const int fmw = pass == 0 ? base + 1 : base;
        const int rounds = pass == 0 ? rem : R - rem;
        if (rounds == 0 || fmw > 8) continue;
        const long want_rows = (long)rounds * slots * 32 * fmw;
        const long rows = want_rows < (long)a.M - row0 ? want_rows : (long)a.M - row0;
        const int mtiles = (int)((rows + 32 * fmw - 1) / (32 * fmw));
        hipError_t e = ap_launch_conv_phase(a, fmw, row0, mtiles, st);
        if (e != hipSuccess) return e;
        row0 += mtiles * 32 * fmw;
    }
    if (row0 < a.M) {                                        // (cannot happen: units * 32 * slots >= M)
        const int mtiles = (a.M - row0 + 127) / 128;
        return ap_launch_conv_phase(a, 4, row0, mtiles, st);
    }
    return hipSuccess;
}
