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

AP_NS_BEGIN

// Timing-only builds for tuning (results WRONG, times valid): -DRP_ABLATE=<bits>
//   1 no DMA in the K loop | 2 no fragment reads in the K loop | 4 no MFMAs | 8 no barrier in the K loop
#ifndef RP_ABLATE
#define RP_ABLATE 0
#endif
// placement of a cluster's NP callbacks (DMA pieces) among its NM MFMAs: 0 = spread evenly (piece k after MFMA
// (k+1)*NM/(NP+1)), 1 = front-loaded (piece k after MFMA k+1), 2 = all before the first MFMA
#ifndef RP_PIECES
#define RP_PIECES 0
#endif

namespace {

template <typename T> using Elem2 = ElemKind<T>;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // keep register-only MFMAs below the wait (they ignore "memory")
}
// LDS fragment read the compiler does not count: hipcc answers a ds_read pending across the loop back-edge
// with lgkmcnt(0), which serialises the fragment double-buffering; the waits are placed by hand instead.
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 r;
#if RP_ABLATE & 2
    r = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(r) : "v"(addr));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
#endif
    return r;
}
__device__ __forceinline__ f32x4 rp_mfma_bf16(const u32x4& w, const u32x4& x, const f32x4& c) {
#if RP_ABLATE & 4
    f32x4 r = c;
    asm volatile("" : "+v"(r) : "v"(w), "v"(x));
    return r;
#else
    return ap_mfma16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c);
#endif
}

template <typename T, int FM, int FN>
__device__ __forceinline__ void mma_chunk2(const u32x4 (&xf)[FM], const u32x4 (&wf)[FN], f32x4 (&acc)[FM][FN]) {
    if constexpr (Elem2<T>::KIND == K_BF16) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fm][fn] = ap_mfma16(
                    __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[fm]), acc[fm][fn]);
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

// MFMA cluster with NP callbacks (one LDS-DMA piece each) spread evenly between the MFMAs
template <typename T, int FM, int FN, int NP, typename F>
__device__ __forceinline__ void mma_issue(const u32x4 (&xf)[FM], const u32x4 (&wf)[FN], f32x4 (&acc)[FM][FN], F&& piece) {
    constexpr int KIND = Elem2<T>::KIND;
    constexpr int NM = FM * FN;
    static_assert(NM >= NP + 1, "need more MFMAs than DMA pieces per cluster");
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int fm = j / FN, fn = j % FN;
        if constexpr (KIND == K_F32) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t wv = t == 0 ? wf[fn].x : t == 1 ? wf[fn].y : t == 2 ? wf[fn].z : wf[fn].w;
                const uint32_t xv = t == 0 ? xf[fm].x : t == 1 ? xf[fm].y : t == 2 ? xf[fm].z : xf[fm].w;
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    __builtin_bit_cast(float, wv), __builtin_bit_cast(float, xv), acc[fm][fn], 0, 0, 0);
            }
        } else {
            acc[fm][fn] = rp_mfma_bf16(wf[fn], xf[fm], acc[fm][fn]);
        }
        // piece k goes after MFMA number (k+1)*NM/(NP+1)  (all indices fold at compile time)
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((RP_PIECES == 0 ? (k + 1) * NM / (NP + 1) : RP_PIECES == 1 ? k + 1 : 1) == j + 1) {
                __builtin_amdgcn_sched_barrier(0);
                piece(k);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

#ifndef X2_DROP
#define X2_DROP 0      // experiment: 1 drops w_hi * x_lo (activations as plain bf16), 2 drops w_lo * x_hi (weights as plain bf16)
#endif
#ifndef X2_MINH
#define X2_MINH 0       // ... X2_DROP = 1 only in layers whose input has at least this many rows (0: everywhere)
#endif
// split-bf16 (planar) second cluster of a K step: w_hi * x_lo over all accumulators, then w_lo * x_hi (the first cluster
// is the plain mma_issue on the hi fragments), the two MFMAs of one accumulator FM*FN instructions apart
template <int FM, int FN, int NP, typename F>
__device__ __forceinline__ void mma_issue_cross(const u32x4 (&xh)[FM], const u32x4 (&xl)[FM], const u32x4 (&wh)[FN],
                                                const u32x4 (&wl)[FN], f32x4 (&acc)[FM][FN], F&& piece, bool drop_here = true) {
    constexpr int NM = 2 * FM * FN;
    static_assert(NM >= NP + 1, "need more MFMAs than DMA pieces per cluster");
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int jj = j % (FM * FN), fm = jj / FN, fn = jj % FN;
        if (!((X2_DROP == 1 && drop_here && j < FM * FN) || (X2_DROP == 2 && j >= FM * FN)))
            acc[fm][fn] = rp_mfma_bf16(j < FM * FN ? wh[fn] : wl[fn], j < FM * FN ? xl[fm] : xh[fm], acc[fm][fn]);
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((k + 1) * NM / (NP + 1) == j + 1) {
                __builtin_amdgcn_sched_barrier(0);
                piece(k);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int S, bool DIRECT>
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
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef AP_TRACE
#define AP_BSTAMP(k) do { if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) p.dbg[160 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define AP_BSTAMP(k) do {} while (0)
#endif
    AP_BSTAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile % p.ntiles;

    // ---------------------------------------------------------------- DMA source state
    const int prow = lane >> 3;                              // row within an 8-row piece
    const int pchunk = (lane & 7) ^ prow;                    // source chunk (swizzle on the source side)
    const unsigned char* src[LPW];                           // row base + chunk, bytes
    int hw0[LPW];                                            // (hi0 << 16) | (wi0 & 0xffff), both small signed
    const int HoWo = p.Ho * p.Wo;
    const bool pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    const unsigned char* zg = (const unsigned char*)p.zero;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int row = (wave * LPW + i) * 8 + prow;
        if ((wave * LPW + i) * 8 < BM) {                     // activation row (wave-uniform test)
            const int m = bm * BM + row;
            if (m < p.M && pointwise) {
                // 1x1 stride-1 conv (36 of the 52 trunk layers): output pixel m reads input pixel m, no decode
                hw0[i] = 0;
                src[i] = xg + ((size_t)m * p.ldx + pchunk * EPC) * sizeof(T);
            } else if (m < p.M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
                hw0[i] = (h0 << 16) | (w0 & 0xffff);
                src[i] = xg + (((size_t)n * p.H * p.W + (ptrdiff_t)h0 * p.W + w0) * p.ldx + pchunk * EPC) * sizeof(T);
            } else {
                hw0[i] = (int)0xc0000000;                    // hi0 = -16384: never passes the bounds test
                src[i] = zg;
            }
        } else {                                             // weight row: always valid (rows padded to 128)
            hw0[i] = 0;
            src[i] = wg + ((size_t)(bn * BN + row - BM) * p.wld + pchunk * EPC) * sizeof(T);
        }
    }
    const int cpb = p.Cin / BK;
    constexpr bool SEG2 = BN >= 128;                         // 64-channel tiles never see a folded downsample
    const int cpb2 = (SEG2 && p.x2) ? p.Cin2 / BK : 0;       // chunks of the second K segment (folded downsample)
    const int KT = p.KH * p.KW * cpb + cpb2;
    int r = 0, s = 0, cb = 0;                                // tap / channel chunk of the tile being ISSUED
    // second K segment: per-piece byte offsets into x2 at the strided pixel (1x1, always in bounds; 32-bit keeps
    // the kernel inside the 128-VGPR budget of two 8-wave workgroups per CU)
    uint32_t off2[SEG2 ? LPW : 1];
#pragma unroll
    for (int i = 0; i < (SEG2 ? LPW : 0); ++i) {
        off2[i] = 0xffffffffu;
        const int m = bm * BM + (wave * LPW + i) * 8 + prow;
        if (p.x2 && (wave * LPW + i) * 8 < BM && m < p.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            off2[i] = (uint32_t)(((((size_t)n * p.H2 + (size_t)ho * p.stride2) * p.W2 + (size_t)wo * p.stride2) * p.ldx2 +
                                  pchunk * EPC) * sizeof(T));
        }
    }
    // Per-piece running source pointers.  Address arithmetic is done once per TAP (every cpb K steps), a K
    // step inside a tap only adds the per-lane increment (0 for predicated-off lanes, which sit on the zero
    // line): the VALU work between the barrier and the MFMAs was what the 8 lock-stepped waves were waiting on.
    const unsigned char* cur[LPW];
    uint32_t live = 0xffffffffu;                             // bit i: piece i advances (0 = parked on the zero line)
#pragma unroll
    for (int i = 0; i < LPW; ++i) cur[i] = src[i];

    // per-tap pointer refresh for one piece (only when a new tap starts: wave-uniform branch)
    auto prep_piece = [&](int i) {
        if (cb == 0 && (wave * LPW + i) * 8 < BM) {
            if (r < p.KH) {
                const ptrdiff_t xoff = ((ptrdiff_t)r * p.W + s) * p.ldx * (ptrdiff_t)sizeof(T);
                const int h0 = hw0[i] >> 16, w0 = (int)(short)(hw0[i] & 0xffff);
                const bool ok = (unsigned)(h0 + r) < (unsigned)p.H && (unsigned)(w0 + s) < (unsigned)p.W;
                cur[i] = ok ? src[i] + xoff : zg;
                live = ok ? (live | (1u << i)) : (live & ~(1u << i));
            } else if constexpr (SEG2) {                     // entering the second K segment
                const bool ok = off2[i] != 0xffffffffu;
                cur[i] = ok ? (const unsigned char*)p.x2 + off2[i] : zg;
                live = ok ? (live | (1u << i)) : (live & ~(1u << i));
            }
        }
    };
    auto advance_tap = [&]() {
        if (++cb == (r < p.KH ? cpb : cpb2)) { cb = 0; if (r < p.KH && ++s == p.KW) { s = 0; ++r; } }
    };
    auto issue_piece = [&](int i, int stage) {               // one 1-KiB LDS-DMA piece of the tile
        const int piece = wave * LPW + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE + piece * 1024),
                                         16, 0, 0);
        cur[i] += ((live >> i) & 1u) ? (uint32_t)(BK * sizeof(T)) : 0u;
    };
    auto issue_tile = [&](int stage) {
#pragma unroll
        for (int i = 0; i < LPW; ++i) prep_piece(i);
        advance_tap();
#pragma unroll
        for (int i = 0; i < LPW; ++i) issue_piece(i, stage);
    };

    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g4 = lane >> 4;
    // chunks of a 128-byte row read by lane group g4: g4 and 4 + g4 (the two 32-deep halves of a bf16 K step / 16-deep of an
    // fp32 one); split-bf16: 2*g4 = hi parts of elements 8*g4 .. 8*g4+7, 2*g4 + 1 = their lo parts
    constexpr bool SPLIT = Elem2<T>::KIND == K_SPLIT;
    static_assert(!(SPLIT && DIRECT), "the split-bf16 kind uses the LDS-staged epilogue (8-channel groups)");
    const int ck0 = SPLIT ? 2 * g4 : g4, ck1 = SPLIT ? 2 * g4 + 1 : 4 + g4;
    const int sw0 = ((ck0 ^ (lr & 7)) << 4), sw1 = ((ck1 ^ (lr & 7)) << 4);
    const int xfrag = (wm * (BM / WAVES_M) + lr) * 128;
    const int wfrag = BM * 128 + (wn * (BN / WAVES_N) + lr) * 128;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DIRECT epilogue: the residual is fetched in the accumulator layout (4 channels of one pixel per lane and
    // fragment) before the K loop starts, so its latency hides under the whole loop
    constexpr int RW = sizeof(T) == 2 ? 2 : 4;               // dwords of 4 channels
    uint32_t rpre[DIRECT ? FM * FN * RW : 1];
    if constexpr (DIRECT) {
        if (p.res) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const int m = bm * BM + wm * (BM / WAVES_M) + fm * 16 + (lane & 15);
                    const int ch = bn * BN + wn * (BN / WAVES_N) + fn * 16 + (lane >> 4) * 4;
                    const bool ok = m < p.M && ch < p.Cout;
                    const T* rp = ok ? (const T*)p.res + (size_t)m * p.ldr + ch : (const T*)p.zero;
                    if constexpr (sizeof(T) == 2) {
                        const uint2 v = *(const uint2*)rp;
                        rpre[(fm * FN + fn) * 2] = v.x; rpre[(fm * FN + fn) * 2 + 1] = v.y;
                    } else {
                        const uint4 v = *(const uint4*)rp;
                        rpre[(fm * FN + fn) * 4] = v.x; rpre[(fm * FN + fn) * 4 + 1] = v.y;
                        rpre[(fm * FN + fn) * 4 + 2] = v.z; rpre[(fm * FN + fn) * 4 + 3] = v.w;
                    }
                }
        }
    }

    // residual tile prefetch in the epilogue's coalesced layout (16 B per thread and row chunk): issued before
    // the K loop so that its HBM latency is covered by the whole loop instead of stalling the epilogue
    constexpr int CPR = BN / EPC;                            // 16-byte output chunks per tile row
    constexpr int NIT = BM * CPR / NT;
    static_assert(BM * CPR % NT == 0, "epilogue chunks must divide evenly");
    const T* __restrict__ rg = (const T*)p.res;
    u32x4 rv[DIRECT ? 1 : NIT];
    // prologue: the whole ring (S tiles) in flight
    AP_BSTAMP(1);
#pragma unroll
    for (int t = 0; t < S; ++t)
        if (t < KT) issue_tile(t);
    AP_BSTAMP(2);
    // K loop.  One barrier per K step, between the two MFMA clusters.  Before it every wave has COMPLETED its
    // reads of tile kt (first half read one step earlier, second half waited with lgkmcnt(0)), so the barrier
    // both publishes tile kt+1 and frees slot kt%S, which is refilled with tile kt+S at once: S-1 K steps of
    // DMA latency cover.  Fragment reads of the next half step are always in flight under the current MFMAs;
    // the address refresh runs between the MFMAs of cluster 0, the DMA pieces between those of cluster 1
    // (the vector-memory path moves 64 B/clk/CU: a tile's 48 KiB are ~770 cycles of TA time that must run
    // UNDER the matrix pipe, not in front of it).
    u32x4 xf0[FM], wf0[FN], xf1[FM], wf1[FN];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t xa0 = lds0 + xfrag + sw0, xa1 = lds0 + xfrag + sw1, wa0 = lds0 + wfrag + sw0, wa1 = lds0 + wfrag + sw1;
    constexpr int NFR = FM + FN;                             // ds_reads per half K step
    auto load_frags = [&](uint32_t xa, uint32_t wa, u32x4 (&xf)[FM], u32x4 (&wf)[FN]) {
        static_assert(FM <= 4 && FN <= 4, "fragment unroll");
        if constexpr (FM > 0) xf[0] = lds_read_b128<0>(xa);
        if constexpr (FM > 1) xf[1] = lds_read_b128<2048>(xa);
        if constexpr (FM > 2) xf[2] = lds_read_b128<4096>(xa);
        if constexpr (FM > 3) xf[3] = lds_read_b128<6144>(xa);
        if constexpr (FN > 0) wf[0] = lds_read_b128<0>(wa);
        if constexpr (FN > 1) wf[1] = lds_read_b128<2048>(wa);
        if constexpr (FN > 2) wf[2] = lds_read_b128<4096>(wa);
        if constexpr (FN > 3) wf[3] = lds_read_b128<6144>(wa);
    };
    // tile 0 landed (tiles 1 .. S-1 may still be in flight)
    if (KT >= S) wait_vmcnt<(S - 1) * LPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    AP_BSTAMP(3);
    if constexpr (!SPLIT) load_frags(xa0, wa0, xf0, wf0);

#ifdef AP_TRACE
    const bool trace = p.dbg != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
#define AP_STAMP(k)                                                                                          \
    do {                                                                                                     \
        if (trace && kt >= 8 && kt < 16) p.dbg[((wave >> 2) * 8 + (kt - 8)) * 10 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define AP_STAMP(k) do {} while (0)
#endif
    int cs = 0;                                              // ring slot of tile kt
    for (int kt = 0; kt < KT; ++kt) {
        AP_STAMP(0);
        const uint32_t so = cs * STAGE;
        const bool refill = kt + S < KT;                     // tile kt+S goes into the slot of tile kt
        // split-bf16: the second cluster needs the hi fragments too, so they are read here, at the top of their own K step,
        // instead of under the previous step's second cluster
        if constexpr (SPLIT) load_frags(xa0 + so, wa0 + so, xf0, wf0);
        load_frags(xa1 + so, wa1 + so, xf1, wf1);            // second half of tile kt (split-bf16: its lo parts)
        AP_STAMP(1);
        wait_lgkmcnt<NFR>();                                 // first half (read one phase earlier) has landed
        AP_STAMP(2);
        mma_issue<T, FM, FN, LPW>(xf0, wf0, acc, [&](int i) { if (refill) prep_piece(i); });
        if (refill) advance_tap();
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(3);
        wait_lgkmcnt<0>();                                   // all of this wave's reads of tile kt are done
        AP_STAMP(4);
        if (kt + 1 < KT) {
            // tile kt+1 must have landed; the younger tiles kt+2 .. kt+S-1 may stay in flight
            const int younger = KT - 2 - kt;                 // tiles issued after kt+1
            if (younger >= S - 2) wait_vmcnt<(S - 2) * LPW>();
            else if (S > 3 && younger == 1) wait_vmcnt<LPW>();
            else wait_vmcnt<0>();
            AP_STAMP(5);
            if (!(RP_ABLATE & 8)) __builtin_amdgcn_s_barrier();
            AP_STAMP(6);
            if constexpr (!SPLIT) {
                const int ns = cs + 1 == S ? 0 : cs + 1;
                load_frags(xa0 + ns * STAGE, wa0 + ns * STAGE, xf0, wf0);   // first half of tile kt+1
            }
        }
        AP_STAMP(7);
        if constexpr (SPLIT)
            mma_issue_cross<FM, FN, LPW>(xf0, xf1, wf0, wf1, acc, [&](int i) { if (refill && !(RP_ABLATE & 1)) issue_piece(i, cs); }, p.H >= X2_MINH);
        else
            mma_issue<T, FM, FN, LPW>(xf1, wf1, acc, [&](int i) { if (refill && !(RP_ABLATE & 1)) issue_piece(i, cs); });
        if (++cs == S) cs = 0;
        __builtin_amdgcn_sched_barrier(0);
        AP_STAMP(8);
        AP_STAMP(9);
    }
#undef AP_STAMP
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the other kinds
    const uint32_t smask = p.relu ? 0xffffffffu : 0x7fff7fffu;
    if constexpr (DIRECT) {
        // ------------------------------------------------------------ epilogue straight from the accumulators:
        // lane = (pixel lr, channels g4*4..+3) per fragment; 8-byte (bf16) / 16-byte (fp32) stores, the four
        // fn fragments of a wave complete each pixel's 128-byte line
        T* __restrict__ yg = (T*)p.y;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int ch = bn * BN + wn * (BN / WAVES_N) + fn * 16 + g4 * 4;
            const float4 sc = *(const float4*)(p.scale + ch);
            const float4 sh = *(const float4*)(p.shift + ch);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m = bm * BM + wm * (BM / WAVES_M) + fm * 16 + lr;
                float v0 = acc[fm][fn][0] * sc.x + sh.x, v1 = acc[fm][fn][1] * sc.y + sh.y;
                float v2 = acc[fm][fn][2] * sc.z + sh.z, v3 = acc[fm][fn][3] * sc.w + sh.w;
                if (p.res) {
                    if constexpr (sizeof(T) == 2) {
                        float lo, hi;
                        unpack_bf16x2(rpre[(fm * FN + fn) * 2], lo, hi); v0 += lo; v1 += hi;
                        unpack_bf16x2(rpre[(fm * FN + fn) * 2 + 1], lo, hi); v2 += lo; v3 += hi;
                    } else {
                        v0 += __builtin_bit_cast(float, rpre[(fm * FN + fn) * 4]);
                        v1 += __builtin_bit_cast(float, rpre[(fm * FN + fn) * 4 + 1]);
                        v2 += __builtin_bit_cast(float, rpre[(fm * FN + fn) * 4 + 2]);
                        v3 += __builtin_bit_cast(float, rpre[(fm * FN + fn) * 4 + 3]);
                    }
                }
                if (p.relu) { v0 = ap_relu(v0); v1 = ap_relu(v1); v2 = ap_relu(v2); v3 = ap_relu(v3); }
                if (m < p.M && ch < p.Cout) {
                    if constexpr (sizeof(T) == 2) {
                        uint2 o;
                        o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                        ap_rng_note(rng, o.x & smask); ap_rng_note(rng, o.y & smask);
                        *(uint2*)(yg + (size_t)m * p.ldy + ch) = o;
                    } else {
                        *(float4*)(yg + (size_t)m * p.ldy + ch) = make_float4(v0, v1, v2, v3);
                    }
                }
            }
        }
        return;
    }
    AP_BSTAMP(4);
    // BatchNorm rows of the wave's channels, requested ahead of the barrier (conv_slab.hip)
    float4 scv[FN], shv[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int ch = bn * BN + wn * (BN / WAVES_N) + fn * 16 + g4 * 4;
        scv[fn] = *(const float4*)(p.scale + ch);
        shv[fn] = *(const float4*)(p.shift + ch);
    }
    __syncthreads();                                         // all MFMA reads done before the ring is reused

    if constexpr (!SPLIT && sizeof(T) == 2) {
        if (!rg && p.relu) {
            // no residual, ReLU (conv1 / conv2 of a bottleneck, the folded-downsample conv3): BatchNorm, rounding and the packed ReLU
            // BEFORE the stage, which then holds 16-bit values -- half the LDS traffic of the epilogue, and its second half is a
            // 16-byte copy (conv_slab.hip).  Same result: relu(round(v)) = round(relu(v)).
            constexpr int CLD16 = BN + 8;                    // row stride in 16-bit elements: 4 dwords of bank shift per row
            T* c16 = (T*)smem;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int chl = wn * (BN / WAVES_N) + fn * 16 + g4 * 4;
                const float4 sc = scv[fn], sh = shv[fn];
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const int px = wm * (BM / WAVES_M) + fm * 16 + lr;
                    uint2 o;
                    o.x = pack_bf16x2(acc[fm][fn][0] * sc.x + sh.x, acc[fm][fn][1] * sc.y + sh.y);
                    o.y = pack_bf16x2(acc[fm][fn][2] * sc.z + sh.z, acc[fm][fn][3] * sc.w + sh.w);
                    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.x));
                    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.y));
                    ap_rng_note2(rng, o.x, o.y);
                    *(uint2*)(c16 + px * CLD16 + chl) = o;
                }
            }
            __syncthreads();
            T* __restrict__ yq = (T*)p.y;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int px, cc;
                ap_epi_item(tid + it * NT, CPR, p.y_tiled != 0, px, cc);
                const int m = bm * BM + px, ch = bn * BN + cc * EPC;
                if (m >= p.M || ch >= p.Cout) continue;
                const u32x4 o = *(const u32x4*)(c16 + px * CLD16 + cc * EPC);
                *(u32x4*)(yq + (p.y_tiled ? ap_tiled_off((size_t)m, ch, p.Cout) : (size_t)m * p.ldy + ch)) = o;
            }
            ap_rng_flush(p.range_flag, rng);
            return;
        }
    }
    // ---------------------------------------------------------------- epilogue (as conv_igemm.hip)
    float* ct = (float*)smem;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int chl = wn * (BN / WAVES_N) + fn * 16 + g4 * 4;
        const float4 sc = scv[fn], sh = shv[fn];
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
    AP_BSTAMP(5);
    __syncthreads();
    AP_BSTAMP(6);
    T* __restrict__ yg = (T*)p.y;
    if constexpr (SPLIT) {
        // split-bf16: items of 8 channels = 32 bytes [8 hi | 8 lo] (fp32 output for the blend-shape contraction: 2 x 16 B)
        constexpr int CP8 = BN / 8, NI8 = BM * CP8 / NT;
        static_assert(BM * CP8 % NT == 0, "epilogue items must divide evenly");
        u32x4 rh[NI8], rl[NI8];
        if (rg) {
#pragma unroll
            for (int it = 0; it < NI8; ++it) {
                const int q = tid + it * NT, px = q / CP8, cc = q - px * CP8;
                const int m = bm * BM + px, ch = bn * BN + cc * 8;
                const bool ok = m < p.M && ch < p.Cout;
                const u32x4* rp = (const u32x4*)(ok ? rg + (size_t)m * p.ldr + ch : (const T*)p.zero);
                rh[it] = rp[0];
                rl[it] = (ok && ch + 4 < p.Cout) ? rp[1] : rp[0];
            }
        }
#pragma unroll
        for (int it = 0; it < NI8; ++it) {
            const int q = tid + it * NT, px = q / CP8, cc = q - px * CP8;
            const int m = bm * BM + px, ch = bn * BN + cc * 8;
            if (m >= p.M || ch >= p.Cout) continue;
            const float* sp = ct + px * CLD + cc * 8;
            const float4 a = *(const float4*)sp, b = *(const float4*)(sp + 4);
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const bool second = ch + 4 < p.Cout;             // (fp32 output: Cout may end on a multiple of 4)
            if (rg) {
                float r[8];
                if (p.out_f32) {
                    const uint32_t w[8] = {rh[it].x, rh[it].y, rh[it].z, rh[it].w, rl[it].x, rl[it].y, rl[it].z, rl[it].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (e < 4 || second) ? __builtin_bit_cast(float, w[e]) : 0.f;
                } else {
                    split8_unpack(rh[it], rl[it], r);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ap_relu(v[e]);
            }
            if (p.out_f32) {
                float* yp = (float*)yg + (size_t)m * p.ldy + ch;
                *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                if (second) *(float4*)(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                u32x4 hi, lo;
                split8_pack(v, hi, lo);
                u32x4* yp = (u32x4*)(yg + (size_t)m * p.ldy + ch);
                yp[0] = hi; yp[1] = lo;
            }
        }
    } else {
    if constexpr (!DIRECT) {
        if (rg) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int px, cc;
                ap_epi_item(tid + it * NT, CPR, sizeof(T) == 2 && p.y_tiled != 0, px, cc);
                const int m = bm * BM + px, ch = bn * BN + cc * EPC;
                const bool ok = m < p.M && ch < p.Cout;
                rv[it] = *(const u32x4*)(ok ? rg + (size_t)m * p.ldr + ch : (const T*)p.zero);
            }
        }
    }


#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int px, cc;
        ap_epi_item(tid + it * NT, CPR, sizeof(T) == 2 && p.y_tiled != 0, px, cc);
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
                a.x = ap_relu(a.x); a.y = ap_relu(a.y); a.z = ap_relu(a.z); a.w = ap_relu(a.w);
                b.x = ap_relu(b.x); b.y = ap_relu(b.y); b.z = ap_relu(b.z); b.w = ap_relu(b.w);
            }
            u32x4 o;
            o[0] = pack_bf16x2(a.x, a.y); o[1] = pack_bf16x2(a.z, a.w);
            o[2] = pack_bf16x2(b.x, b.y); o[3] = pack_bf16x2(b.z, b.w);
            ap_rng_note4(rng, o[0], o[1], o[2], o[3], p.relu != 0);
            *(u32x4*)(yg + (p.y_tiled ? ap_tiled_off((size_t)m, ch, p.Cout) : (size_t)m * p.ldy + ch)) = o;
        } else {
            float4 a = *(const float4*)sp;
            if (rg) {
                // (copy the lanes out first: bit_cast applied directly to a vector element mis-compiles)
                const uint32_t r0 = rv[it].x, r1 = rv[it].y, r2 = rv[it].z, r3 = rv[it].w;
                a.x += __builtin_bit_cast(float, r0); a.y += __builtin_bit_cast(float, r1);
                a.z += __builtin_bit_cast(float, r2); a.w += __builtin_bit_cast(float, r3);
            }
            if (p.relu) { a.x = ap_relu(a.x); a.y = ap_relu(a.y); a.z = ap_relu(a.z); a.w = ap_relu(a.w); }
            *(float4*)(yg + (size_t)m * p.ldy + ch) = a;
        }
    }
    }
    if constexpr (Elem2<T>::KIND == K_BF16) ap_rng_flush(p.range_flag, rng);
    AP_BSTAMP(7);
}

template <typename T, int BM, int BN, int WM, int WN, int S, bool DIRECT>
hipError_t launch_pipe(ConvArgs a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    auto kern = conv_pipe_kernel<T, BM, BN, WM, WN, S, DIRECT>;
    constexpr int ring = S * (BM + BN) * 128, epi = BM * (BN + 4) * 4;
    constexpr int lds = ring > epi ? ring : epi;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    if (a.x2 && BN < 128) return hipErrorInvalidValue;       // the 64-wide tiles carry no second K segment: refuse, never drop it
    // the second K segment addresses x2 with 32-bit byte offsets (0xffffffff = "row predicated off"): refuse a tensor
    // that does not fit instead of wrapping (fp32 at >= 1338 images of layer2.0 would)
    if (a.x2 && (size_t)a.N * a.H2 * a.W2 * a.ldx2 * sizeof(T) >= 0xffffffffull) return hipErrorInvalidValue;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(64 * WM * WN), lds, st, a);
    return hipGetLastError();
}

// cfg: 0 = 256x128 (8 waves, 3 stages), 1 = 128x128 (4 waves, 4 stages), 2 = 128x64 (4 waves, 4 stages),
//      3 = 256x64 (8 waves, 3 stages); +4 = same tiles with the register (LDS-free) epilogue;
//      8 = 128x128 / 9 = 128x64, 4 waves, 2 stages: small LDS footprint, 2-3 workgroups per CU
template <typename T>
hipError_t launch_pipe_cfg(const ConvArgs& a, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_pipe<T, 256, 128, 4, 2, 3, false>(a, st);
        case 1: return launch_pipe<T, 128, 128, 2, 2, 4, false>(a, st);
        case 2: return launch_pipe<T, 128, 64, 2, 2, 4, false>(a, st);
        case 3: return launch_pipe<T, 256, 64, 4, 2, 3, false>(a, st);
        case 4: return launch_pipe<T, 256, 128, 4, 2, 3, true>(a, st);
        case 5: return launch_pipe<T, 128, 128, 2, 2, 4, true>(a, st);
        case 6: return launch_pipe<T, 128, 64, 2, 2, 4, true>(a, st);
        case 7: return launch_pipe<T, 256, 64, 4, 2, 3, true>(a, st);
        case 8: return launch_pipe<T, 128, 128, 2, 2, 2, false>(a, st);
        case 9: return launch_pipe<T, 128, 64, 2, 2, 2, false>(a, st);
        case 10: return launch_pipe<T, 128, 128, 4, 2, 2, false>(a, st);
        case 11: return launch_pipe<T, 128, 128, 2, 4, 2, false>(a, st);
        case 12: return launch_pipe<T, 128, 64, 4, 2, 2, false>(a, st);
        case 13: return launch_pipe<T, 128, 64, 8, 1, 2, false>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// kind: K_F32 / K_BF16 / K_SPLIT (= AP_PREC_*).  The split-bf16 kind ships the two configurations the automatic choice
// uses (11, 12); every configuration exists for the other two.
hipError_t ap_launch_conv_pipe(const ConvArgs& a, int kind, int cfg, hipStream_t st) {
    if (!a.zero) return hipErrorInvalidValue;
    if (kind == K_BF16) return launch_pipe_cfg<bf16_t>(a, cfg, st);
#ifndef AP_F16                                               // (the fp16 set carries the 16-bit kind only)
    if (kind == K_F32) return launch_pipe_cfg<float>(a, cfg, st);
    if (kind == K_SPLIT) {
        if (cfg == 11) return launch_pipe<bsplit_t, 128, 128, 2, 4, 2, false>(a, st);
        if (cfg == 12) return launch_pipe<bsplit_t, 128, 64, 4, 2, 2, false>(a, st);
    }
#endif
    return hipErrorInvalidValue;
}

AP_NS_END
