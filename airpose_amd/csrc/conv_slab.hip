// Stride-1 3x3 convolution with the nine taps served from ONE LDS slab per 64-channel chunk (bf16, gfx950).
//
// Same tile program as conv_pipe.hip's <bf16,128,128,2,4,2> (8 waves, 64x32 per wave, one barrier per K step, two
// workgroups per CU, counted waits, LDS-staged epilogue; see there for the reference lines it replaces), but the
// activation operand is not re-fetched per tap.  The K loop runs channel chunk OUTER, tap INNER:
//
//   per 64-channel chunk:  a slab of the 130 + 2W input pixels [p0 - W - 1, p0 + 128 + W + 1) of the tile's 128 output
//                          pixels (linear NHW index: the 3x3 neighbours of pixel m are m + (r-1)W + (s-1)), 128 B per
//                          pixel, fetched ONCE by LDS-DMA (24 pieces of 8 rows, padded with zero rows) into one of two
//                          slab buffers -- the next chunk's slab arrives under the nine K steps of the current one;
//   per tap (K step):      the wave's four pixel fragments read slab rows base + (r-1)W + (s-1) (taps that fall outside
//                          the image read the slab's zero row instead: one v_cndmask per fragment), the 128 x 64 weight
//                          tile of that tap streams through a 2-slot ring as in the ring kernel.
//
// Operand DMA per chunk: 24 KiB of slab + 9 x 16 KiB of weights = 168 KiB instead of 9 x 32 KiB = 288 KiB (-42 %); the
// ring kernel's timing-only builds price the activation stream of these layers at 20-27 % of their time.
// The accumulation order differs from the ring kernel's (taps outer there), so results agree with it to fp32
// re-association, not bitwise; both are tested against an fp64 convolution of the same bf16 operands.
//
// LDS (80 KiB, two workgroups per CU): slab0 [0, 24K) | slab1 [24K, 48K) | weights0 [48K, 64K) | weights1 [64K, 80K);
// rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7) on the DMA source side.
#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

// Timing-only builds (results WRONG, times valid): -DSL_ABLATE=<bits>
//   1 no slab DMA in the K loop | 2 fragment addresses computed once | 4 weight rows walked sequentially (the ring kernel's
//   order) | 8 no weight DMA in the K loop | 16 slab pieces of the K loop issued, but from the zero line
#ifndef SL_ABLATE
#define SL_ABLATE 0
#endif
// cache policy bits of the slab loads (gfx950: 1 = sc0, 2 = nt, 16 = sc1); measured: no effect
#ifndef SL_NT
#define SL_NT 0
#endif

namespace {

constexpr int BM = 128, BN = 128, WAVES_M = 2, WAVES_N = 4, NT = 512, FM = 4, FN = 2;
constexpr int SLAB_ROWS = 192, SLAB_BYTES = SLAB_ROWS * 128, WT_BYTES = BN * 128;
constexpr int W_BASE = 2 * SLAB_BYTES;                      // weight ring after the two slabs
constexpr int LDS_BYTES = 2 * SLAB_BYTES + 2 * WT_BYTES;    // 81920
constexpr int ZROW = SLAB_ROWS - 1;                         // rows ZROW - 1 and ZROW stay zero (one per row parity: a b128 read is
                                                            // conflict-free when the 16 lanes of a group hit 16 different
                                                            // (row parity, chunk position) pairs)
constexpr int WLP = 3;                                      // weight DMA pieces per wave and K step (the third: waves 0, 1)
constexpr int CLD = BN + 4;
static_assert(BM * CLD * 4 <= LDS_BYTES, "the fp32 epilogue stage reuses the operand buffers");

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// MFMA with the accumulator tied in place.  With the nine taps unrolled the register allocator otherwise lets the eight
// accumulators migrate (v_mfma d, a, b, c with d != c), which costs ~25 registers and spills; spill reloads sit behind
// s_waitcnt vmcnt(0) and would drain the DMA queue every K step.  (Hazards the compiler no longer sees: consecutive MFMAs
// here never share an accumulator back to back with different operands, and the epilogue's first VALU read of an
// accumulator is fenced by s_nop below.)
__device__ __forceinline__ void mfma_acc(f32x4& c, const u32x4& w, const u32x4& x) {
    asm volatile(AP_MFMA16_ASM " %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}

// 8 MFMAs with NP callbacks spread evenly between them
template <int NP, typename F>
__device__ __forceinline__ void mma_issue(const u32x4 (&xf)[FM], const u32x4 (&wf)[FN], f32x4 (&acc)[FM][FN], F&& piece) {
    constexpr int NM = FM * FN;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int fm = j / FN, fn = j % FN;
        mfma_acc(acc[fm][fn], wf[fn], xf[fm]);
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((k + 1) * NM / (NP + 1) == j + 1) {
                __builtin_amdgcn_sched_barrier(0);
                piece(k);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) conv_slab_kernel(const ConvArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile % p.ntiles;
    const int W = p.W, HW = p.H * p.W;
    const int cpb = p.Cin / 64;

    // ---------------------------------------------------------------- DMA roles, split by wave.  vmcnt retires in order
    // per wave, so a slab piece (first touch of the activations: HBM / Infinity-Cache latency) queued in front of a weight
    // piece (L2 hit) makes the K step wait for the long-latency stream.  Waves 6 and 7 therefore issue the slab pieces (three
    // per K step on taps 0..7, for the NEXT chunk: 1 or 2 each) and wait for them once per chunk, at tap 8; waves 0..5 share
    // the 16 weight pieces of a K step (piece j on wave j % 6: three on waves 0..3, two on 4 and 5) and drain their queue
    // every step.  (A DMA piece costs its wave 75-140 cycles of issue; with all three slab pieces on one wave that wave's
    // second cluster took 700 cycles against 480 and every step waited for it at the barrier: tools/probes/slab_trace.py.)
    const bool srole = wave >= 6;                            // wave-uniform
    const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
    const unsigned char* zg = (const unsigned char*)p.zero;
    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    // the kernel-argument loads complete HERE on every path: a scalar load the compiler still believes pending inside the
    // K loop costs an s_waitcnt lgkmcnt(0) in front of every DMA instruction, which also drains the fragment reads in flight
    asm volatile("" ::"s"(zg), "s"(xg), "s"(wg));
    const int nrows = BM + 2 * W + 2;                        // slab rows that hold pixels (the rest stay zero)
    const int need = (nrows + 7) >> 3;                       // ... in 8-row pieces
    uint32_t doff[WLP];                                      // weight waves: running byte offsets of their rows into p.w
#pragma unroll
    for (int i = 0; i < WLP; ++i) {                          // weight rows are padded to multiples of 128: always valid
        const int piece = (wave + 6 * i) & 15;               // (slab waves, and i == 2 on waves 4, 5: unused)
        const int row = (piece & 15) * 8 + prow;
        doff[i] = (uint32_t)(((size_t)(bn * BN + row) * p.wld + pchunk * 8) * sizeof(T));
    }
    const uint32_t wstep = (uint32_t)p.Cin * sizeof(T);      // next tap, same chunk
    const uint32_t wwrap = 128u - 8u * wstep;                // tap 8 -> tap 0 of the next chunk (mod 2^32)
    // weight piece i of this wave for the tile of tap TT into ring slot `slot`, then on to the next tile's row segment
    auto issue_weights = [&](int slot, int i, bool last_tap) {
        const int piece = wave + 6 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + doff[i]),
                                         (__attribute__((address_space(3))) void*)(smem + W_BASE + slot * WT_BYTES + piece * 1024),
                                         16, 0, 0);
        doff[i] += (SL_ABLATE & 4) ? 128u : last_tap ? wwrap : wstep;
        asm volatile("" : "+v"(doff[i]));                    // keep the offset 32-bit (uniform base + VGPR offset addressing)
    };
    // Slab wave: piece j = slab rows 8j .. 8j+7 = input pixels p0 + 8j + prow.  Pixels outside [0, M) are clamped, not
    // predicated: only taps that fall outside their image would read such a row, and those read the zero row instead.
    // Rows >= nrows are never read either, except the zero row (the slab's last): it is written once, in the prologue,
    // and the K loop's DMA leaves it alone (the lanes of rows >= nrows are switched off: an LDS-DMA lane that is not
    // executed writes nothing).
    const int p0 = bm * BM - (W + 1);                        // input pixel of slab row 0
    const uint32_t rowbytes = (uint32_t)p.ldx * sizeof(T);
    const uint32_t lane_off = (uint32_t)pchunk * 16u;
    // Two VALU instructions per piece (the cycle stamps of tools/probes/slab_trace.py showed the slab wave's second cluster
    // at 850 cycles against 480 on the weight waves with the row index clamped and multiplied per piece: every K step
    // waited for it at the barrier): the byte offset is linear in the piece, and the clamp is on the offset.
    const int prow_rb = prow * (int)rowbytes;                // this lane's row within a piece, in bytes
    const int p0_rb = p0 * (int)rowbytes, hi_rb = (p.M - 1) * (int)rowbytes;         // (scalars)
    auto issue_slab = [&](int buf, int cb, int piece) {      // 8 rows of chunk cb
        int base = prow_rb, rb = (int)rowbytes;
        asm volatile("" : "+v"(base), "+s"(rb));             // computed where it is used (hoisted out of the unrolled loop the
                                                             // 24 offsets and products spill VGPRs and SGPRs)
        const int x = min(max(base + (p0_rb + piece * 8 * rb), 0), hi_rb);           // row offset, clamped into the tensor
        const uint32_t off = (uint32_t)x + lane_off + (uint32_t)cb * 128u;
        if ((SL_ABLATE & 16) && cb > 0) return;
        // the slab's last piece holds the two zero rows: its lanes of rows 6 and 7 never execute the DMA (an LDS-DMA lane that
        // is switched off writes nothing); rows past the last pixel row receive clamped garbage nobody reads
        if (piece != SLAB_ROWS / 8 - 1 || prow < 6)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + off),
                                             (__attribute__((address_space(3))) void*)(smem + buf * SLAB_BYTES + piece * 1024),
                                             16, 0, SL_NT);
    };
    auto issue_zero_row = [&](int buf) {                     // prologue only: the last piece of a slab from the zero line
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)zg,
                                         (__attribute__((address_space(3))) void*)(smem + buf * SLAB_BYTES + (SLAB_ROWS / 8 - 1) * 1024),
                                         16, 0, 0);
    };

    // ---------------------------------------------------------------- prologue DMA FIRST: slab 0 (+ the zero piece of slab 1, which
    // the K loop skips when the slab is short), weight tiles 0 and 1
    if (srole) {
        if (wave == 7) {
            issue_zero_row(0);
            issue_zero_row(1);
            wait_vmcnt<0>();                                 // (the same wave overwrites rows of that piece below)
        }
        for (int j = wave - 6; j < need; j += 2) issue_slab(0, 0, j);   // wave 7: the odd pieces, among them the last one
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            issue_weights(t, 0, false);
            issue_weights(t, 1, false);
            if (wave < 4) issue_weights(t, 2, false);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // (requested before the per-lane tap table below: its 8 integer divisions and 36 validity tests are ~800 instructions, 2 us,
    //  that used to run BEFORE the first operand request -- the request's own 1.5-us round trip then came on top, a third of a
    //  layer2 tile's lifetime.  Measured: neutral on its own -- the other workgroup of the CU covers it -- kept as the saner order)
    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g4 = lane >> 4;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // The slab row a fragment reads depends on (lane, fragment, tap) only, not on the chunk: all 36 are computed once per
    // tile (valid ? centre + (r-1)W + (s-1) : the zero row) and kept as bytes, the four fragments of a tap in one register.
    uint32_t pk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) pk[t] = 0u;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int px = wm * (BM / WAVES_M) + fm * 16 + lr;
        const int m = bm * BM + px;
        // pixel -> (row, column) of its image by multiplication with the launcher's reciprocals (exact for (M + tile) H W < 2^32:
        // __umulhi(m, ceil(2^32 / d)) = m / d there); hipcc's 32-bit division is ~25 instructions, eight of them per lane here
        int rem, ho;
        if (p.magic_hw) {
            rem = m - (int)__umulhi((unsigned)m, p.magic_hw) * HW;
            ho = (int)__umulhi((unsigned)rem, p.magic_w);
        } else {
            rem = m % HW;
            ho = rem / W;
        }
        const int wo = rem - ho * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, sx = t % 3;
            const bool ok = m < p.M && (unsigned)(ho + r - 1) < (unsigned)p.H && (unsigned)(wo + sx - 1) < (unsigned)W;
            const int nat = px + W + 1 + (r - 1) * W + (sx - 1);          // the row the tap would read
            const int row = ok ? nat : ZROW - 1 + (nat & 1);             // outside the image: the zero row of the same parity
            pk[t] |= (uint32_t)row << (fm * 8);
        }
    }
    const uint32_t g4s = (uint32_t)g4 << 4;
    const int c7 = lr + W + 1;                               // centre-tap row of every fragment of this lane, mod 8 (fragments are 16 rows apart)
    const uint32_t wa = lds0 + W_BASE + (wn * (BN / WAVES_N) + lr) * 128 + ((g4 ^ (lr & 7)) << 4);
    f32x4 acc[FM][FN];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_x = [&](const uint32_t (&xa)[FM], uint32_t flip, u32x4 (&xf)[FM]) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) xf[fm] = lds_read_b128<0>(xa[fm] ^ flip);
    };
    auto load_w = [&](uint32_t a, u32x4 (&wf)[FN]) {
        wf[0] = lds_read_b128<0>(a);
        wf[1] = lds_read_b128<2048>(a);
    };
    // chunk g4 of the 128-byte row (first K half) sits at position g4 ^ (row & 7); chunk 4 + g4 (second half) at that address ^ 64.
    // A lane whose tap falls outside the image reads a zero row of ITS OWN ROW'S PARITY at the position its own row would
    // have, (centre + d) & 7: bank = 32 * (row & 1) + 4 * position, and the 16 lanes of a ds_read_b128 group cover the 16
    // (parity, position) pairs exactly once for any tap shift -- only as long as a substituted row keeps both (with one zero
    // row at its own position 7 ^ g4, SQ_LDS_BANK_CONFLICT was 27 % of the LDS cycles)
    // (in FM pieces, one between each pair of MFMAs of the first cluster: the matrix pipe works on the wave's previous MFMA
    // while these issue; done in one block they left the pipe idle for ~100 cycles per K step on both waves of a SIMD at once)
    uint32_t a_f = 0, a_swz = 0;
    auto tap_addr_part = [&](int part, uint32_t f, int tap, uint32_t sbase, uint32_t (&xa)[FM]) {
        if (part == 0) {
            a_f = f;
            asm volatile("" : "+v"(a_f));                    // unpack HERE: hoisted out of the chunk loop the 36 addresses
                                                             // would be 36 registers instead of 9
            const int d = (tap / 3 - 1) * W + (tap % 3 - 1);
            a_swz = (((uint32_t)(c7 + d) & 7u) << 4) ^ g4s;
        }
        const uint32_t row = (a_f >> (part * 8)) & 0xffu;
        xa[part] = (sbase + (row << 7)) | a_swz;
        asm volatile("" : "+v"(xa[part]));                   // ... and finished here, between the MFMAs (left alone the compiler
                                                             // sinks the arithmetic to the reads behind the barrier)
    };
    auto tap_addr = [&](uint32_t f, int tap, uint32_t sbase, uint32_t (&xa)[FM]) {
#pragma unroll
        for (int part = 0; part < FM; ++part) tap_addr_part(part, f, tap, sbase, xa);
    };

    uint32_t xa[FM];                                         // fragment addresses of the tile whose first half is read next
    tap_addr(pk[0], 0, lds0, xa);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    u32x4 xf0[FM], wf0[FN], xf1[FM], wf1[FN];
    load_x(xa, 0u, xf0);
    load_w(wa, wf0);

#ifdef AP_TRACE
    // cycle stamps of workgroup 0, waves 0 and 7 (the slab wave), taps 0..7 of chunk 1: p.dbg[(w7 * 8 + tap) * 10 + k]
    const bool trace = p.dbg != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 7) && lane == 0;
#define SL_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (trace && cb == 1 && tap < 8) p.dbg[((wave == 7 ? 8 : 0) + tap) * 10 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define SL_STAMP(k) do {} while (0)
#endif
    // K loop: as the ring kernel's, one barrier per K step between the two MFMA clusters; the nine taps of a chunk are
    // unrolled so that everything tap-dependent is an immediate.  Weight tile kt+2 goes into the slot of tile kt after the
    // barrier of step kt.
    int ws = 0;                                              // ring slot of the weight tile of this step
    for (int cb = 0; cb < cpb; ++cb) {
        const uint32_t sb_cur = lds0 + (cb & 1) * SLAB_BYTES, sb_nxt = lds0 + ((cb & 1) ^ 1) * SLAB_BYTES;
        const bool more = cb + 1 < cpb;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool last = !more && tap == 8;             // last K step of the tile
            const bool refill = more || tap < 7;             // tile kt+2 exists
            SL_STAMP(0);
            load_x(xa, 64u, xf1);                            // second K half of tile kt (the addresses are consumed at issue:
                                                             // the first cluster overwrites them with the next tile's)
            load_w((wa + ws * WT_BYTES) ^ 64u, wf1);
            SL_STAMP(1);
            wait_lgkmcnt<FM + FN>();                         // first half (read one phase earlier) has landed
            SL_STAMP(2);
            mma_issue<FM>(xf0, wf0, acc, [&](int part) {
                if (!(SL_ABLATE & 2)) {
                    if (part == 0) tap_addr_part(0, pk[tap == 8 ? 0 : tap + 1], tap == 8 ? 0 : tap + 1, tap == 8 ? sb_nxt : sb_cur, xa);
                    else if (part == 1) tap_addr_part(1, 0, 0, tap == 8 ? sb_nxt : sb_cur, xa);
                    else if (part == 2) tap_addr_part(2, 0, 0, tap == 8 ? sb_nxt : sb_cur, xa);
                    else tap_addr_part(3, 0, 0, tap == 8 ? sb_nxt : sb_cur, xa);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            SL_STAMP(3);
            wait_lgkmcnt<0>();                               // all of this wave's reads of tile kt are done
            SL_STAMP(4);
            if (!last) {
                if (!srole || tap == 8) wait_vmcnt<0>();
                SL_STAMP(5);
                __builtin_amdgcn_s_barrier();
                SL_STAMP(6);
                load_x(xa, 0u, xf0);                         // first half of tile kt+1
                load_w(wa + (ws ^ 1) * WT_BYTES, wf0);
                SL_STAMP(7);
            }
            // one MFMA cluster for both roles (two copies cost the compiler 20 registers in accumulator copies); only the
            // DMA issue between the MFMAs branches on the role
            mma_issue<WLP>(xf1, wf1, acc, [&](int i) {
                if (!srole) {
                    if (refill && (i < 2 || wave < 4) && !(SL_ABLATE & 8)) issue_weights(ws, i, (tap + 2) % 9 == 8);
                } else if (more && tap < 8 && !(SL_ABLATE & 1)) {
                    // pieces 3 tap .. 3 tap + 2 of the next chunk's slab: wave 6 takes the even ones, wave 7 the odd ones
                    const int piece = 3 * tap + i;
                    if (((piece ^ wave) & 1) == 0 && piece < need) issue_slab((cb & 1) ^ 1, cb + 1, piece);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            SL_STAMP(8);
            ws ^= 1;
        }
    }
    // BatchNorm rows of the wave's channels: requested HERE, ahead of the settle / barrier below (requested where they are used
    // they were two dependent L2 round trips at the head of every tile's epilogue; the fragment registers are dead by now)
    __builtin_amdgcn_sched_barrier(0);                       // (not into the K loop: no register to spare there)
    float4 scv[FN], shv[FN];
    int tq = threadIdx.x;                                    // (the lane's channel offset from an opaque copy of the thread id: kept alive across
    asm volatile("" : "+v"(tq));                             //  the K loop it was the one register too many and spilled)
    const int lre = tq & 15, g4e = (tq & 63) >> 4;           // the epilogue's own copies of lr / g4
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int ch = bn * BN + wn * (BN / WAVES_N) + fn * 16 + g4e * 4;
        scv[fn] = *(const float4*)(p.scale + ch);
        shv[fn] = *(const float4*)(p.shift + ch);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMA results settle before the epilogue's VALU reads them
    if (!p.res && p.relu && p.y_tiled) {
        // Fragment-tiled output (t2 of a pair block: [M/16][C/8][16 pixels][8 channels]) straight from the accumulators: a lane holds
        // 4 consecutive channels of pixel lr per fragment, lanes g4 = 2k, 2k+1 the two halves of channel group k, so a wave's 8-byte
        // stores of one (fm, fn) fill two ADJACENT 256-byte micro-tiles completely -- 512 contiguous bytes per instruction.  No LDS
        // stage, no barrier: the epilogue of these layers is BatchNorm, rounding, packed ReLU and eight stores.
        T* __restrict__ yt = (T*)p.y;
        uint32_t rng = 0u;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int ch = bn * BN + wn * (BN / WAVES_N) + fn * 16 + g4e * 4;
            const float4 sc = scv[fn], sh = shv[fn];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m = bm * BM + wm * (BM / WAVES_M) + fm * 16 + lre;
                uint2 o;
                o.x = pack_bf16x2(acc[fm][fn][0] * sc.x + sh.x, acc[fm][fn][1] * sc.y + sh.y);
                o.y = pack_bf16x2(acc[fm][fn][2] * sc.z + sh.z, acc[fm][fn][3] * sc.w + sh.w);
                asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.x));
                asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.y));
                ap_rng_note2(rng, o.x, o.y);
                if (m < p.M && ch < p.Cout) *(uint2*)(yt + ap_tiled_off((size_t)m, ch, p.Cout) + (ch & 7)) = o;
            }
        }
        ap_rng_flush(p.range_flag, rng);
        return;
    }
    __syncthreads();                                         // all MFMA reads done before the buffers are reused

    // ---------------------------------------------------------------- epilogue (as conv_pipe.hip: fp32 stage in LDS,
    // BatchNorm + residual + ReLU, 16-byte coalesced stores)
    if (!p.res && p.relu) {
        // conv2 of a bottleneck (no residual, ReLU): BatchNorm, rounding and ReLU BEFORE the stage, on packed pairs -- the stage holds
        // 16-bit values (34 KB instead of 68, half the LDS traffic of the epilogue) and the second half is a 16-byte copy.  Same
        // result: rounding is monotonic and keeps the sign, so relu(round(v)) = round(relu(v)).
        constexpr int CLD16 = BN + 8;                        // 272-byte rows: 4 dwords of bank shift per row (tiled and NHWC item maps)
        T* c16 = (T*)smem;
        uint32_t rng = 0u;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int chl = wn * (BN / WAVES_N) + fn * 16 + g4e * 4;
            const float4 sc = scv[fn], sh = shv[fn];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int px = wm * (BM / WAVES_M) + fm * 16 + lre;
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
        constexpr int CPR = BN / 8, NIT = BM * CPR / NT;
        int et = tid;
        asm volatile("" : "+v"(et));
        T* __restrict__ yg = (T*)p.y;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int px, cc;
            ap_epi_item(et + it * NT, CPR, p.y_tiled != 0, px, cc);
            const int m = bm * BM + px, ch = bn * BN + cc * 8;
            if (m >= p.M || ch >= p.Cout) continue;
            const u32x4 o = *(const u32x4*)(c16 + px * CLD16 + cc * 8);
            *(u32x4*)(yg + (p.y_tiled ? ap_tiled_off((size_t)m, ch, p.Cout) : (size_t)m * p.ldy + ch)) = o;
        }
        ap_rng_flush(p.range_flag, rng);
        return;
    }
    float* ct = (float*)smem;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int chl = wn * (BN / WAVES_N) + fn * 16 + g4e * 4;
        const float4 sc = scv[fn], sh = shv[fn];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int px = wm * (BM / WAVES_M) + fm * 16 + lre;
            float4 v;
            v.x = acc[fm][fn][0] * sc.x + sh.x;
            v.y = acc[fm][fn][1] * sc.y + sh.y;
            v.z = acc[fm][fn][2] * sc.z + sh.z;
            v.w = acc[fm][fn][3] * sc.w + sh.w;
            *(float4*)(ct + px * CLD + chl) = v;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8, NIT = BM * CPR / NT;
    int et = tid;                                            // opaque copy: the item maps below are computed HERE, not hoisted above
    asm volatile("" : "+v"(et));                             // the K loop (one more live register there spills)
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the bf16 set
    u32x4 rv[NIT];
    if (rg) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int px, cc;
            ap_epi_item(et + it * NT, CPR, p.y_tiled != 0, px, cc);
            const int m = bm * BM + px, ch = bn * BN + cc * 8;
            const bool ok = m < p.M && ch < p.Cout;
            rv[it] = *(const u32x4*)(ok ? rg + (size_t)m * p.ldr + ch : (const T*)p.zero);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int px, cc;
        ap_epi_item(et + it * NT, CPR, p.y_tiled != 0, px, cc);
        const int m = bm * BM + px, ch = bn * BN + cc * 8;
        if (m >= p.M || ch >= p.Cout) continue;
        const float* sp = ct + px * CLD + cc * 8;
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
    }
    ap_rng_flush(p.range_flag, rng);
}

}  // namespace

// stride-1 3x3, pad 1, bf16, 64-channel chunks, rows short enough that the 130 + 2W pixel slab (+ its zero row) fits
bool ap_conv_slab_supported(const ConvArgs& a, int kind) {
    return kind == K_BF16 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.x2 && a.Cin % 64 == 0 &&
           a.Cin >= 64 && a.Ho == a.H && a.Wo == a.W && BM + 2 * a.W + 2 <= ZROW - 1 && a.ldx == a.Cin &&
           (long long)(a.M + 256) * a.ldx * 2 < 0x7fffffffll && (long long)a.wld * 2 * ((a.Cout + 127) / 128 * 128) < 0xffffffffll;
}

hipError_t ap_launch_conv_slab(ConvArgs a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    if (!a.zero || !ap_conv_slab_supported(a, K_BF16)) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    {
        const unsigned long long hw = (unsigned long long)a.H * a.W, lim = 1ull << 32;
        // (W == 1 or H W == 1: ceil(2^32 / 1) does not fit 32 bits and would truncate to 0 -> wrong rows; those shapes divide)
        const bool fits = ((unsigned long long)a.M + BM) * hw < lim && a.W > 1 && hw > 1;
        a.magic_hw = fits ? (unsigned)((lim + hw - 1) / hw) : 0u;
        a.magic_w = fits ? (unsigned)((lim + a.W - 1) / a.W) : 0u;
    }
    hipLaunchKernelGGL(conv_slab_kernel, dim3(a.mtiles * a.ntiles), dim3(NT), LDS_BYTES, st, a);
    return hipGetLastError();
}

AP_NS_END
