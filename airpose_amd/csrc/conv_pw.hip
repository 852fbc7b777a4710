// Pointwise (1 x 1, stride 1) convolution + BatchNorm (+ identity) + ReLU on block_img.hip's mainloop (16-bit storage, gfx950):
// the K >= 512 pointwise layers the image-resident blocks do not cover -- conv1 of layer4's bottlenecks and conv3 (+ identity) of
// its identity blocks (Bottleneck.forward, model_copenet.py:29-31 and :38-45, iterated by :64-65).
//
// Why a second pointwise kernel: on the 128 x 128-tile ring kernel these layers run at 670-750 TF/s (0.75 LDS operand fetches per
// MFMA, two workgroups per CU sharing the matrix pipe, three rounds of the chip with a short last one); block_img's conv1 phase
// does the same arithmetic at 0.32 fetches per MFMA.  This file is that phase as a kernel of its own:
//   * a workgroup = four waves, one per SIMD, 512 registers each; tile = 196 pixels (13 groups of 16: 4 images of 7 x 7, or one of
//     14 x 14 -- every layer3 / layer4 tensor is a whole number of tiles) x 256 output channels, 64 per wave: 52 accumulators of
//     16 x 16 in the accumulator half of the register file;
//   * weights never touch the LDS: a wave streams its 64 rows as MFMA A fragments from L2, packed at finalize time in consumption
//     order (one contiguous 4-KiB piece per K step of 32), through a four-piece register ring;
//   * x passes through a two-stage LDS ring in K chunks of 64 (register-staged two chunks ahead), 16-byte pieces at
//     piece ^ (pixel & 7): conflict-free ds_read_b128;
//   * the tiles of one pixel range (2 or 8 of them: Cout / 256) run on the same XCD at the same time, so x comes from HBM once;
//   * persistent over tiles; a workgroup keeps its output-channel range, so its weight stream simply wraps and the ring runs ahead
//     across tiles;
//   * identity: the staging loads of the last K iteration (which have no chunk left to fetch) fetch the first half of the identity
//     instead; the second half takes each register as the epilogue frees it.
// Three operand forms, one mainloop (template parameter SEG2):
//   0  x [M][Cin]                                             conv1 of layer4's bottlenecks; conv3 + identity (RES) only when forced;
//   1  [t2 [M][Cin] | x2 sampled at (ho stride, wo stride)]   conv3 + folded downsample of layer4.0 (model_copenet.py:41-42, 97-102);
//   2  nine taps of x [N][2 Ho][2 Wo][Cin], padding 1         the 3 x 3 / stride-2 conv2 of layer3.0 / layer4.0 (:32-34 with :18): a 64-channel
//                                                             chunk of ONE tap per staging step (im2col by address), out-of-image taps
//                                                             zeroed on their way into the LDS.
// K order per output element = the ring / lean / pair kernels' (K steps of 32 in order -- [tap][Cin] for the 3 x 3 -- one MFMA chain):
// bit-identical results, so the trunk may choose by problem size and by whether the pass shares the chip (ap_net_set_pw_conv;
// a one-wave-per-SIMD kernel keeps another pass's workgroups off its CUs: api.hip, trunk_chunk).
// Measured (512 images, in the pass): conv1 128 / 57 / 56 us (ring kernel 156 / 70 / 71), conv3 + downsample 171 (208), 3 x 3 / 2
// 148 / 131 (179 / 153); what bounds it is the latency budget of a staging load: vmcnt retires in order, so a load is waited for at
// the next ring wait behind it, four K steps after its issue, however deep it was requested.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int PW_STAGE = 224 * 128;                          // one staging buffer: 224 pixel slots x 64 channels
constexpr int PW_LDS = 2 * PW_STAGE;

#include "bi_helpers.inc"

// dst [N tile][wave][K step][fragment 4][lane 64][8 K values]: row (lane & 15) of fragment f = output channel
// ntile * 256 + wave * 64 + bi_row_channel(16 f + (lane & 15)), K columns 8 (lane >> 4) .. + 7 of the step's 32
__global__ void __launch_bounds__(256) pw_pack_kernel(const bf16_t* __restrict__ w, unsigned char* __restrict__ dst, int cin, int cout, int wld) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_wave = (size_t)(cin >> 5) * 256, per_nt = 4 * per_wave;
    if (idx >= (size_t)(cout >> 8) * per_nt) return;
    const int nt = (int)(idx / per_nt);
    const size_t r = idx - (size_t)nt * per_nt;
    const int wv = (int)(r / per_wave), p = (int)(r - (size_t)wv * per_wave);
    const int step = p >> 8, f = (p >> 6) & 3, lane = p & 63;
    const int ch = nt * 256 + wv * 64 + bi_row_channel(f * 16 + (lane & 15));
    *(u32x4*)(dst + idx * 16) = *(const u32x4*)(w + (size_t)ch * wld + step * 32 + (lane >> 4) * 8);
}

// SEG2: 0 plain | 1 a second K segment read at strided pixels | 2 a 3 x 3 / stride-2 convolution as nine pointwise taps
template <int NG, bool RES, int SEG2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) conv_pw_kernel(const PwArgs a) {
    static_assert(NG == 13, "196-pixel tiles");
    static_assert(!(RES && SEG2), "a stage's first block has no identity");
    constexpr bool K3 = SEG2 == 2;
    constexpr int TM = 196;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t rng = 0u;

    const int Cin = a.Cin, Cout = a.Cout, NN = Cout >> 8, NC1 = Cin >> 6;
    const int NC = K3 ? 9 * NC1 : SEG2 ? (Cin + a.Cin2) >> 6 : NC1, KS = 2 * NC;   // K = [x: Cin | x2: Cin2] / [tap][Cin], as the weight rows have it
    const int lcc = K3 ? 31 - __builtin_clz(NC1) : 0;       // (3 x 3: Cin / 64 is a power of two)
    const int MT8 = (a.M / TM + 7) & ~7;                     // pixel tiles, rounded up to whole groups of eight (one per XCD)
    const long T = (long)MT8 * NN;
    // tile t -> XCD t & 7 (= blockIdx & 7: the grid is a multiple of 8 NN), N tile (t >> 3) % NN, pixel tile ((t >> 3) / NN) * 8 + XCD:
    // the NN tiles of a pixel range are neighbours in dispatch order on ONE XCD, and t += gridDim keeps the N tile
    const int xcd = blockIdx.x & 7, ntile = (int)((blockIdx.x >> 3) % NN);
    const unsigned char* const xg = (const unsigned char*)a.x;
    const unsigned char* const rg = (const unsigned char*)a.res;
    const float* const bs = a.scale + ntile * 256 + wave * 64;
    const float* const bh = a.shift + ntile * 256 + wave * 64;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)a.y, 0, (int)((uint32_t)a.M * (uint32_t)Cout * 2u), 0x00020000);
    const unsigned char* const wsb = (const unsigned char*)a.wfrag + ((size_t)ntile * 4 + wave) * (size_t)KS * 4096;
    // every kernel-argument load completes here (scalar loads share lgkmcnt with the counted fragment reads)
    const unsigned char* const x2g = (const unsigned char*)a.x2;
    const int hw2 = SEG2 ? a.Ho * a.Wo : 1, wo2 = SEG2 ? a.Wo : 1, sw2 = SEG2 ? a.stride2 * a.W2 : 0, st2 = SEG2 ? a.stride2 : 0;
    const int cin2 = K3 ? a.Cin : SEG2 ? a.Cin2 : 0, w2row = K3 ? a.W2 : 0;
    const uint32_t img2 = SEG2 ? (uint32_t)a.H2 * a.W2 * cin2 * 2u : 0u;         // bytes per source image of the strided operand
    asm volatile("" ::"s"(xg), "s"(rg), "s"(bs), "s"(bh), "s"(wsb), "s"(Cin), "s"(Cout), "s"(KS), "s"(x2g), "s"(hw2), "s"(wo2), "s"(sw2), "s"(st2), "s"(cin2), "s"(img2), "s"(NC1), "s"(w2row), "s"(lcc));
    const uint32_t wlane = lane * 16;
    const unsigned char* wp = wsb;
    int wcnt = 0;
    u32x4 ar[4][4];
    auto refill = [&](auto SL, u32x4 (&r)[4][4]) __attribute__((always_inline)) {           // ring slot SL <- the next piece of the stream
        constexpr int sl = decltype(SL)::value;
        bi_gld<0>(r[sl][0], wlane, wp); bi_gld<1024>(r[sl][1], wlane, wp);
        bi_gld<2048>(r[sl][2], wlane, wp); bi_gld<3072>(r[sl][3], wlane, wp);
        wp += 4096;
        if (++wcnt == KS) { wcnt = 0; wp = wsb; }
    };
    sfor<0, 4>([&](auto S) __attribute__((always_inline)) { refill(S, ar); });

    // ---- x staging: thread = (slot column col, 16-byte piece pc) of slot rows xr0 + 2 j, j = 0 .. 6; slot = tile pixel (slots 196 ..
    // 223 fetch pixel 195 again: never used, always inside the tensor)
    const int xcol = (tid >> 3) & 15, xpc = tid & 7, xr0 = tid >> 7;
    const uint32_t xrow = (uint32_t)Cin * 2u;                // bytes per pixel row of x
    const uint32_t xoff = (uint32_t)(xr0 * 16 + xcol) * xrow + (uint32_t)xpc * 16u;
    const int s6 = (xr0 + 12) * 16 + xcol;
    const uint32_t xoff6 = (uint32_t)(s6 < TM ? s6 : TM - 1) * xrow + (uint32_t)xpc * 16u;
    const uint32_t xs_w = (uint32_t)((xr0 * 16 + xcol) * 128 + ((xpc ^ (xcol & 7)) << 4));
    // second K segment: the tile's pixels are whole output images (Ho * Wo | 196); slot -> byte offset of the strided pixel of x2
    // relative to the tile's first source image (the same for every tile)
    uint32_t xoff2[7];
    uint32_t topm = 0u, leftm = 0u;                          // 3 x 3: bit j = staging slot j lies in the top row / left column of its image
    if constexpr (SEG2 != 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int s = (xr0 + 2 * j) * 16 + xcol;
            s = s < TM ? s : TM - 1;
            const int im = s / hw2, r = s - im * hw2, ho = r / wo2, wo = r - ho * wo2;
            xoff2[j] = (uint32_t)im * img2 + (uint32_t)((ho * sw2 + wo * st2) * cin2) * 2u + (uint32_t)xpc * 16u;   // (3 x 3: the centre tap)
            topm |= (uint32_t)(ho == 0) << j;
            leftm |= (uint32_t)(wo == 0) << j;
        }
    }
    // 3 x 3: chunk c = 64 channels (c & (NC1 - 1)) of tap c >> lcc; the taps outside the image (dy = -1 in the top row, dx = -1 in the
    // left column: stride 2, even input sizes -- nothing falls off the bottom / right) read the centre pixel instead and are zeroed
    auto k3_tap = [&](int c, int& toff, uint32_t& inv) __attribute__((always_inline)) {
        const int tap = c >> lcc, dy = (tap * 11) >> 5, dx = tap - 3 * dy;
        toff = ((dy - 1) * w2row + (dx - 1)) * cin2 * 2;
        inv = (dy == 0 ? topm : 0u) | (dx == 0 ? leftm : 0u);
    };
    // B fragments out of a staging buffer: slot 16 g + li, piece 4 ks + kq at position piece ^ (slot & 7)
    const uint32_t xs_r = lds0 + (uint32_t)(li * 128 + (((li >> 2) & 1) << 6) + ((kq ^ (li & 3)) << 4));
    // identity / output: this lane's 8 channels (ntile * 256 + wave * 64 + q * 32 + 8 kq ..) of tile pixel 16 g + li
    const uint32_t orow = (uint32_t)Cout * 2u;
    const uint32_t ochan = (uint32_t)(ntile * 256 + wave * 64 + kq * 8) * 2u;
    const uint32_t idoff = (uint32_t)li * orow + ochan;
    const uint32_t idoff12 = (uint32_t)(li < 4 ? li : 3) * orow + ochan;                    // group 12: pixels 192 .. 195 + 12 junk lanes
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    f32x4 acc[4][14];
    u32x4 bf[8];

    for (long t = blockIdx.x; t < T; t += gridDim.x) {
        const int mtile = (int)((t >> 3) / NN) * 8 + xcd;
        const long m0 = (long)mtile * TM;
        if (m0 >= a.M) continue;                             // (uniform: the padding tiles of the last group of eight)
        const unsigned char* const xt = xg + (size_t)m0 * xrow;
        const unsigned char* const rt = RES ? rg + (size_t)m0 * orow : xg;
        const unsigned char* const x2t = SEG2 ? (K3 ? xg : x2g) + (size_t)(m0 / hw2) * img2 : xg;
        u32x4 xa[7], xb[7];                                  // staging registers: even / odd chunks, two chunks ahead
        // 7 loads, always: chunk c of x, or (c >= NC: nothing left to fetch) the same 16 bytes for every lane -- the queue stays uniform
        auto xload = [&](u32x4 (&r)[7], int c) __attribute__((always_inline)) {
            if constexpr (K3) {                              // (selects, not a branch: seven loads either way)
                const bool in = c < NC;
                int toff;
                uint32_t inv;
                k3_tap(in ? c : 0, toff, inv);
                const unsigned char* sp = in ? x2t + (c & (NC1 - 1)) * 128 : xg;
                sfor<0, 7>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value;
                    const uint32_t v = in ? xoff2[j] + (((inv >> j) & 1u) ? 0u : (uint32_t)toff) : 0u;
                    bi_gld<0>(r[j], v, sp);
                });
                return;
            }
            if constexpr (SEG2 == 1) {
                if (c >= NC1 && c < NC) {                    // (uniform branch: both sides issue exactly seven loads)
                    const unsigned char* sp2 = x2t + (c - NC1) * 128;
                    bi_gld<0>(r[0], xoff2[0], sp2); bi_gld<0>(r[1], xoff2[1], sp2); bi_gld<0>(r[2], xoff2[2], sp2); bi_gld<0>(r[3], xoff2[3], sp2);
                    bi_gld<0>(r[4], xoff2[4], sp2); bi_gld<0>(r[5], xoff2[5], sp2); bi_gld<0>(r[6], xoff2[6], sp2);
                    return;
                }
            }
            const bool real = c < NC1;
            const unsigned char* sp = real ? xt + c * 128 : xg;
            const uint32_t vo = real ? xoff : 0u, vo6 = real ? xoff6 : 0u;
            const uint32_t st = real ? 32u * xrow : 0u;
            bi_gld<0>(r[0], vo, sp); bi_gld<0>(r[1], vo, sp + st); bi_gld<0>(r[2], vo, sp + 2 * (size_t)st); bi_gld<0>(r[3], vo, sp + 3 * (size_t)st);
            bi_gld<0>(r[4], vo, sp + 4 * (size_t)st); bi_gld<0>(r[5], vo, sp + 5 * (size_t)st); bi_gld<0>(r[6], vo6, sp);   // (xoff6 carries its slot row itself)
        };
        // the last K iteration's staging slots fetch the identity of fragment half q = 0 instead: pixel groups GB .. GB + 6 (group 12 with
        // its junk lanes clamped, the non-existent group 13 = the dummy load)
        auto idload = [&](u32x4 (&r)[7], auto GB) __attribute__((always_inline)) {
            constexpr int gb = decltype(GB)::value;
            const unsigned char* sp = rt + (size_t)gb * 16 * orow;
            const size_t st = 16 * (size_t)orow;
            bi_gld<0>(r[0], idoff, sp); bi_gld<0>(r[1], idoff, sp + st); bi_gld<0>(r[2], idoff, sp + 2 * st); bi_gld<0>(r[3], idoff, sp + 3 * st);
            bi_gld<0>(r[4], idoff, sp + 4 * st);
            if constexpr (gb == 0) { bi_gld<0>(r[5], idoff, sp + 5 * st); bi_gld<0>(r[6], idoff, sp + 6 * st); }
            else { bi_gld<0>(r[5], idoff12, sp + 5 * st); bi_gld<0>(r[6], 0u, xg); }
        };
        auto xstore = [&](u32x4 (&r)[7], int stage, int c) __attribute__((always_inline)) {         // c: the chunk the registers hold
            int toff;
            uint32_t inv = 0u;
            if constexpr (K3) k3_tap(c, toff, inv);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                asm volatile("" : "+v"(r[j]));
                if (K3 && ((inv >> j) & 1u)) r[j] = u32x4{0u, 0u, 0u, 0u};
                bi_sts(smem, xs_w + stage * PW_STAGE + j * (32 * 128), r[j]);
            }
        };
        xload(xa, 0);
        xload(xb, 1);
        bi_wait_vm<7>();                                     // chunk 0 (and everything older: the ring, the last tile's stores)
        __syncthreads();                                     // (every wave is past its last fragment read of the previous tile)
        xstore(xa, 0, 0);
        xload(xa, 2);
        __syncthreads();
        // one chunk = two K steps of 32 out of staging buffer STAGE on ring slots SL0, SL0 + 1
        auto c1_pair = [&](auto STAGE, auto SL0, auto FIRST, u32x4 (&b)[8], u32x4 (&r)[4][4], f32x4 (&ac)[4][14]) __attribute__((always_inline)) {
            constexpr int stage = decltype(STAGE)::value, sl0 = decltype(SL0)::value;
            constexpr bool first = decltype(FIRST)::value != 0;
            const uint32_t ra = xs_r + stage * PW_STAGE, rb = ra ^ 64u;
            bi_pipe<2 * NG, 7, 8>(b,
                [&](auto I, u32x4& d) __attribute__((always_inline)) { constexpr int n = decltype(I)::value, ks = n / NG, g = n % NG; bi_ldsr<g * 2048>(d, ks ? rb : ra); },
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr int n = decltype(I)::value, ks = n / NG, g = n % NG, sl = sl0 + ks;
                    if constexpr (g == 0 && !first) bi_wait_vm<26>();       // the slot's piece: behind it 12 ring loads and 14 staging loads
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if constexpr (first && stage == 0 && ks == 0) bi_mma0(ac[f][g], r[sl][f], d);
                        else bi_mma(ac[f][g], r[sl][f], d);
                    }
                    if constexpr (g == NG - 1) refill(std::integral_constant<int, sl>{}, r);
                });
        };
        auto c1_iter = [&](int c, auto FIRST, auto LAST) __attribute__((always_inline)) {
            constexpr bool first = decltype(FIRST)::value != 0, last = decltype(LAST)::value != 0;
            // chunk c from stage 0; chunk c + 1 -> stage 1 (free since the barrier that ended chunk c - 1); then chunk c + 3 requested
            bi_wait_vm<(first ? 7 : 23)>();
            xstore(xb, 1, c + 1);
            if constexpr (last && RES) idload(xb, I0{}); else xload(xb, c + 3);
            c1_pair(I0{}, I0{}, FIRST, bf, ar, acc);
            __syncthreads();
            bi_wait_vm<(first ? 15 : 23)>();
            if constexpr (!last) xstore(xa, 0, c + 2);
            if constexpr (last && RES) idload(xa, std::integral_constant<int, 7>{}); else xload(xa, c + 4);
            c1_pair(I1{}, I2{}, I0{}, bf, ar, acc);
            __syncthreads();
        };
        c1_iter(0, I1{}, I0{});                              // (NC >= 4: the first iteration is never the last)
        for (int c = 2; c < NC - 2; c += 2) c1_iter(c, I0{}, I0{});
        c1_iter(NC - 2, I0{}, I1{});
        // the last iteration's staging loads (identity, or nothing) must have landed before their registers are read or reused:
        // behind them the ring loads of the last two K steps
        bi_wait_vm<8>();
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));
        bi_settle28(acc[0], acc[1]);                         // the last MFMA results settle before VALU reads them
        bi_settle28(acc[2], acc[3]);
        // ---- epilogue: BN (+ identity) + ReLU + 16-bit, 16 bytes per lane and (pixel group, fragment half)
        const uint32_t so = (uint32_t)((size_t)m0 * orow) + ochan;
        sfor<0, 2>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            const f32x4 s0 = *(const f32x4*)(bs + q * 32 + kq * 8), s1 = *(const f32x4*)(bs + q * 32 + kq * 8 + 4);
            const f32x4 h0 = *(const f32x4*)(bh + q * 32 + kq * 8), h1 = *(const f32x4*)(bh + q * 32 + kq * 8 + 4);
            // store offsets: ONE running register (the 26 tile-invariant offsets hipcc otherwise hoists out of the tile loop -- and
            // spills -- cost a scratch reload, i.e. a drained queue, per store)
            uint32_t vrun = so + (uint32_t)li * orow + q * 64;
            asm volatile("" : "+v"(vrun));
            sfor<0, NG>([&](auto G) __attribute__((always_inline)) {
                constexpr int g = decltype(G)::value;
                u32x4& idr = g < 7 ? xb[g] : xa[g - 7];
                if constexpr (RES && q == 1) {
                    bi_wait_vm<2 * NG - 2 - g>();            // behind identity load g of this half: NG - 1 - g (store, load) pairs, g stores
                    asm volatile("" : "+v"(idr));
                }
                const u32x4 o = bi_bn8(acc[2 * q][g], acc[2 * q + 1][g], s0, s1, h0, h1, RES ? &idr : nullptr, rng);
                const uint32_t vo = (g < NG - 1 || li < TM - 16 * (NG - 1)) ? vrun : 0xffffff00u;     // (junk lanes: dropped by the range check)
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, vo, 0, 0);
                vrun += 16u * orow;
                asm volatile("" : "+v"(vrun));
                if constexpr (RES && q == 0) {               // the register is consumed: the identity of half q = 1 takes it
                    __builtin_amdgcn_sched_barrier(0);
                    bi_gld<64>(idr, g == NG - 1 ? idoff12 : idoff, rt + (size_t)g * 16 * orow);
                }
            });
        });
    }
    bi_wait_vm<0>();                                         // (the ring ran ahead: nothing may land after the exit)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ar[j][0]), "+v"(ar[j][1]), "+v"(ar[j][2]), "+v"(ar[j][3]));
    ap_rng_flush(a.range_flag, rng);
}

}  // namespace

// 196 | M (whole tiles: every 14 x 14 tensor, and 7 x 7 tensors of 4 k images), K in whole pairs of 64-channel chunks, 256 | Cout
bool ap_conv_pw_supported(long M, int Cin, int Cout) {
    return M > 0 && M % 196 == 0 && Cin >= 256 && Cin % 128 == 0 && Cout % 256 == 0 && Cout >= 256 &&
           (size_t)M * Cout * 2 < 0xffffff00ull && (size_t)M * Cin * 2 < 0xffffff00ull;
}
// ... with a second K segment: [x: Cin | x2: Cin2 sampled with stride2], output images of Ho x Wo pixels with Ho * Wo | 196
bool ap_conv_pw_ds_supported(const PwArgs& a) {
    if (!a.x2 || a.res || a.Cin % 64 || a.Cin2 % 64 || a.Cin2 <= 0 || (a.Cin + a.Cin2) % 128 || a.Ho <= 0 || a.Wo <= 0 || 196 % (a.Ho * a.Wo)) return false;
    if (a.stride2 < 1 || (a.Ho - 1) * a.stride2 >= a.H2 || (a.Wo - 1) * a.stride2 >= a.W2) return false;
    if ((size_t)(a.M / (a.Ho * a.Wo)) * a.H2 * a.W2 * a.Cin2 * 2 >= 0xffffff00ull) return false;
    return ap_conv_pw_supported(a.M, a.Cin + a.Cin2, a.Cout) && (size_t)a.M * a.Cin * 2 < 0xffffff00ull;
}

// ... 3 x 3, stride 2, padding 1: x [N][H2][W2][Cin] with even H2, W2 onto Ho x Wo = H2 / 2 x W2 / 2 with Ho * Wo | 196; Cin / 64 a power of two
bool ap_conv_pw_k3_supported(const PwArgs& a) {
    if (!a.k3 || a.x2 || a.res || a.stride2 != 2 || a.Ho <= 0 || a.Wo <= 0 || a.H2 != 2 * a.Ho || a.W2 != 2 * a.Wo || 196 % (a.Ho * a.Wo)) return false;
    const int cc = a.Cin >> 6;
    if (a.Cin % 64 || cc < 2 || (cc & (cc - 1))) return false;
    if ((size_t)(a.M / (a.Ho * a.Wo)) * a.H2 * a.W2 * a.Cin * 2 >= 0xffffff00ull) return false;
    return ap_conv_pw_supported(a.M, 9 * a.Cin, a.Cout);
}

size_t ap_conv_pw_stream_bytes(int Cin, int Cout) { return (size_t)Cin * Cout * 2; }

// w [Cout][wld] K-contiguous 16-bit rows (wld >= Cin) as packed for the stand-alone kernels
hipError_t ap_launch_conv_pw_pack(const void* w, void* dst, int Cin, int Cout, int wld, hipStream_t st) {
    if (!w || !dst || Cin % 32 || Cout % 256 || wld < Cin) return hipErrorInvalidValue;
    const size_t pieces = (size_t)Cin * Cout / 8;
    hipLaunchKernelGGL(pw_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w, (unsigned char*)dst, Cin, Cout, wld);
    return hipGetLastError();
}

// grid: the largest multiple of 8 * (Cout / 256) that fits the device (one workgroup per CU), capped by the tile count
int ap_conv_pw_grid(long M, int Cout, int n_cu) {
    const int NN = Cout >> 8, unit = 8 * NN;
    const long T = (long)((M / 196 + 7) & ~7L) * NN;
    long g = (long)(n_cu / unit) * unit;
    if (g < unit) g = unit;
    return (int)(g < T ? g : T);
}

hipError_t ap_launch_conv_pw(const PwArgs& a, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    if (!a.x || !a.y || !a.wfrag || !a.scale || !a.shift || !a.relu) return hipErrorInvalidValue;
    if (a.k3 ? !ap_conv_pw_k3_supported(a) : a.x2 ? !ap_conv_pw_ds_supported(a) : !ap_conv_pw_supported(a.M, a.Cin, a.Cout)) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!n_cu_dev[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_pw_kernel<13, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_pw_kernel<13, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_pw_kernel<13, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_pw_kernel<13, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return e;
        n_cu_dev[dev] = n;
    }
    const int grid = ap_conv_pw_grid(a.M, a.Cout, n_cu_dev[dev]);
    if (a.k3) hipLaunchKernelGGL((conv_pw_kernel<13, false, 2>), dim3(grid), dim3(256), PW_LDS, st, a);
    else if (a.x2) hipLaunchKernelGGL((conv_pw_kernel<13, false, 1>), dim3(grid), dim3(256), PW_LDS, st, a);
    else if (a.res) hipLaunchKernelGGL((conv_pw_kernel<13, true, 0>), dim3(grid), dim3(256), PW_LDS, st, a);
    else hipLaunchKernelGGL((conv_pw_kernel<13, false, 0>), dim3(grid), dim3(256), PW_LDS, st, a);
    return hipGetLastError();
}

AP_NS_END
