// One identity bottleneck of layer3 (1024 -> 256 -> 256 -> 1024 at 14 x 14) as ONE image-resident kernel (16-bit storage, gfx950).
//
//   out = relu(bn3(conv3(relu(bn2(conv2_3x3(relu(bn1(conv1(x))))))) + x)          Bottleneck.forward, model_copenet.py:27-47
//
// What the launch-per-layer path cannot have: the 14 x 14 x 256 intermediates of a whole image (t1, then t2: 112 KB of 16-bit
// values) live in the LDS of one CU from the first convolution to the last.  A workgroup owns an image:
//   * four waves, one per SIMD, the whole 512-entry register file each (56 accumulators of 16 x 16 in the accumulator half);
//     a wave computes 64 (conv1 / conv2; 32 in conv3) output channels for ALL 224 pixel slots of the image, so an LDS operand
//     fragment feeds four MFMAs and a weight fragment fourteen: 18 operand fetches per 56 v_mfma_f32_16x16x32 (0.32 per MFMA;
//     the 128 x 128-tile kernels of this library: 0.75, the pair kernel: 1.0);
//   * pixels in 14 rows of 16 slots (columns 14, 15 hold zeros): a 3 x 3 tap is a constant slot shift, the zero columns are the
//     horizontal padding, one zero row above and below the vertical one -- no per-lane masks, no halo exchange, no slab;
//   * weights never touch the LDS: each wave streams ITS rows as MFMA A fragments straight from L2 (packed at finalize time in
//     fragment order and in consumption order: one contiguous 4-KiB piece per wave and K step of 32; 2.2 MB per block, the same
//     stream for every image, resident in the XCD's L2) through a four-piece register ring -- ordinary loads the compiler counts;
//   * x (401 KB per image) passes through a two-stage LDS ring in K chunks of 64 for conv1 (register-staged, two chunks ahead)
//     and is read a second time, as the identity, in the accumulator layout of conv3's epilogue;
//   * conv2 and conv3 run without a single barrier (their B operand is the resident image, their A operand is private).
// LDS: 258 slots x 512 B (16 zero slots + 1 above, 224 image slots, 16 + 1 below) = 129 KB; chunk c of slot u at position (c + 2 u) mod 16 of its 256-byte half so the
// 16 pixels of a ds_read_b128 lane group hit 16 different 16-byte bank slots for every tap shift.
// K order per output element = that of the kernels this one replaces (conv1 / conv3: K steps of 32 in order; conv2: the slab
// kernel's 64-channel chunk outer, tap inner): the same sums, bit for bit, so the trunk may choose between the two paths by
// problem size (an image per CU: only full rounds of the chip pay) without a pair's result depending on its batch.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int BI_HW = 14, BI_PIX = 196, BI_P = 256, BI_C = 1024;
constexpr int BI_SLOTS = 258;                                // physical slots: u = 16 (row + 1) + col + 1
constexpr int BI_T_BYTES = BI_SLOTS * 512;                   // 132 096
constexpr int BI_XS = 17 * 512;                              // x staging ring inside the image region (rows 0 ..), two stages
constexpr int BI_XS_STAGE = 224 * 128;                       // 28 672
constexpr int BI_STEPS12 = 32 + 72;                          // 4-KiB pieces of conv1 + conv2 per wave
constexpr int BI_SLOTS_TOTAL = BI_STEPS12 + 32;              // + conv3: 64 steps of 2 KiB
constexpr size_t BI_WAVE_BYTES = (size_t)BI_SLOTS_TOTAL * 4096;   // 557 056 per wave, 2 228 224 per block
static_assert(BI_XS + 2 * BI_XS_STAGE <= 241 * 512, "the staging ring stays clear of the zero rows");

#include "bi_helpers.inc"

// weight stream: wave w -> [conv1: 32 steps][conv2: 72 steps (64-channel chunk, tap, K half)] of 4 fragments (64 rows) + [conv3: 8 chunks x 8 steps] of
// 2 fragments (32 rows); fragment = [lane 64][8 K values]: row lane & 15, K columns 8 (lane >> 4) .. + 7 of the step's 32
__global__ void __launch_bounds__(256) blk_img_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                           const bf16_t* __restrict__ w3, unsigned char* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_wave = BI_WAVE_BYTES / 16;
    if (idx >= 4 * per_wave) return;
    const int w = (int)(idx / per_wave);
    const int p = (int)(idx - (size_t)w * per_wave);
    const bf16_t* src;
    if (p < BI_STEPS12 * 256) {
        const int step = p >> 8, f = (p >> 6) & 3, lane = p & 63;
        const int ch = w * 64 + bi_row_channel(f * 16 + (lane & 15));
        if (step < 32) src = w1 + (size_t)ch * BI_C + step * 32 + (lane >> 4) * 8;
        else {                                               // conv2 in the slab kernel's K order: 64-channel chunk, tap, K half
            const int s2 = step - 32, c64 = s2 / 18, tap = (s2 % 18) >> 1, ks = s2 & 1;
            src = w2 + (size_t)ch * (9 * BI_P) + tap * BI_P + c64 * 64 + ks * 32 + (lane >> 4) * 8;
        }
    } else {
        const int q = p - BI_STEPS12 * 256, step = q >> 7, f = (q >> 6) & 1, lane = q & 63;
        const int chunk = step >> 3, ks = step & 7;
        const int ch = chunk * 128 + w * 32 + bi_row_channel(f * 16 + (lane & 15));     // (f < 2: channels 0 .. 31 of the tile)
        src = w3 + (size_t)ch * BI_P + ks * 32 + (lane >> 4) * 8;
    }
    *(u32x4*)(dst + idx * 16) = *(const u32x4*)src;
}

// conv2, one pair of 64-channel chunks = 36 K steps = 480 groups: group n -> chunk parity, tap (dr, dc), K half, pixel row
struct BiP2 { int cpar, dr, dc, ks, g, gi, ng, step; };
constexpr BiP2 bi_p2(int n) {
    BiP2 r{};
    r.cpar = n / 240;
    const int m = n % 240;
    int q = 0;
    if (m < 78) { r.dr = -1; r.ng = 13; q = m; }
    else if (m < 162) { r.dr = 0; r.ng = 14; q = m - 78; }
    else { r.dr = 1; r.ng = 13; q = m - 162; }
    const int st = q / r.ng;
    r.gi = q % r.ng;
    r.dc = st / 2 - 1;
    r.ks = st % 2;
    r.g = (r.dr < 0 ? 1 : 0) + r.gi;                         // the row above / below the image contributes zeros: skipped
    r.step = r.cpar * 18 + (r.dr + 1) * 6 + st;
    return r;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) blk_img_kernel(const BlkImgArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t rng = 0u;

    // zero rows (slots 0 .. 16 and 241 .. 257): written once, never touched again (the staging ring lies between them)
    for (int idx = tid; idx < 34 * 32; idx += 256) {
        const int s = idx >> 5, c = idx & 31, u = s < 17 ? s : 241 + (s - 17);
        bi_sts(smem, u * 512 + c * 16, u32x4{0u, 0u, 0u, 0u});
    }

    const unsigned char* const xg = (const unsigned char*)a.x;
    const float *bs1 = a.s1, *bh1 = a.h1, *bs2 = a.s2, *bh2 = a.h2;
    const unsigned char* const bs3 = (const unsigned char*)a.s3 + wave * 128;
    const unsigned char* const bh3 = (const unsigned char*)a.h3 + wave * 128;
    const int nimg = a.N;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)a.y, 0, (int)((uint32_t)nimg * (BI_PIX * BI_C * 2u)), 0x00020000);
    // ---- weight stream of this wave: 4-KiB pieces in consumption order, four pieces ahead in registers
    const unsigned char* const wsb = (const unsigned char*)a.wfrag + (size_t)wave * BI_WAVE_BYTES;
    // every kernel-argument load completes here (scalar loads share lgkmcnt with the counted fragment reads)
    asm volatile("" ::"s"(xg), "s"(bs1), "s"(bh1), "s"(bs2), "s"(bh2), "s"(bs3), "s"(bh3), "s"(nimg), "s"(wsb));
    const uint32_t wlane = lane * 16;
    const unsigned char* wp = wsb;
    int wcnt = 0;
    u32x4 ar[4][4];
    auto refill = [&](auto SL, u32x4 (&r)[4][4]) __attribute__((always_inline)) {           // ring slot SL <- the next piece of the stream
        constexpr int sl = decltype(SL)::value;
        bi_gld<0>(r[sl][0], wlane, wp); bi_gld<1024>(r[sl][1], wlane, wp);
        bi_gld<2048>(r[sl][2], wlane, wp); bi_gld<3072>(r[sl][3], wlane, wp);
        wp += 4096;
        if (++wcnt == BI_SLOTS_TOTAL) { wcnt = 0; wp = wsb; }
    };
    sfor<0, 4>([&](auto S) __attribute__((always_inline)) { refill(S, ar); });

    // ---- x staging (conv1): thread = (column col, 16-byte piece pc) of rows xr0 + 2 j, j = 0 .. 6
    const int xcol = (tid >> 3) & 15, xpc = tid & 7, xr0 = tid >> 7;
    const uint32_t xoff = (uint32_t)(((xr0 * BI_HW + (xcol < BI_HW ? xcol : BI_HW - 1)) * BI_C + xpc * 8) * 2);   // (junk columns: a valid pixel, finite)
    const uint32_t xs_w = BI_XS + (uint32_t)((xr0 * 16 + xcol) * 128 + ((xpc ^ (xcol & 7)) << 4));
    // B fragments of conv1 out of a staging buffer: slot 16 g + li, piece 4 ks + kq at position piece ^ (slot & 7)
    const uint32_t xs_r = lds0 + BI_XS + (uint32_t)(li * 128 + (((li >> 2) & 1) << 6) + ((kq ^ (li & 3)) << 4));
    // B fragments out of the image region: physical slot u = 16 R + li + dc + 1 (R = row + 1), chunk c = 4 ks + kq at position
    // (c + 2 u) mod 16 of its 256-byte half: tpos(dc, ks & 3) + R * 8192 + (ks >> 2) * 256.  (Round 5 had c ^ (u & 15): conflict-free if
    // ds_read_b128 served 16 contiguous lanes per cycle -- its groups are {0-3, 12-15, 20-27}, .. and the XOR form put two lanes of
    // every group on the same banks, PMC: 31 % of this kernel's LDS cycles; the rotation has none: tools/probes/lds_groups.py.)
    auto tbase = [&](int dc) __attribute__((always_inline)) -> uint32_t { return lds0 + (uint32_t)((li + dc + 1) * 512); };
    auto tpos = [&](int dc, int j) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(((4 * j + kq + 2 * (li + dc + 1)) & 15) << 4); };
    // where this lane's 8 channels (chunk index cidx of 32) of pixel (row g, column li) go in the image region
    auto twr = [&](int g, int cidx) __attribute__((always_inline)) -> uint32_t {
        const int u = 16 * (g + 1) + li + 1;
        return (uint32_t)(u * 512 + (cidx & 16) * 16 + (((cidx + 2 * u) & 15) << 4));
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    f32x4 acc[4][14];
    u32x4 bf[16];
#ifdef AP_TRACE   // cycle stamps of wave 0 of workgroups 0 and 100, their SECOND image (24 slots each): tools/probes/blk_trace.py
    int img_no = 0;
#define BISTAMP(i) do { if (a.dbg && img_no == 1 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) { \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) a.dbg[(blockIdx.x ? 24 : 0) + (i)] = t_; } } while (0)
#else
#define BISTAMP(i) do { } while (0)
#endif

    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        const unsigned char* const ximg = xg + (size_t)img * (BI_PIX * BI_C * 2);
        // ================================================================ conv1: 16 chunks of 64 input channels
        BISTAMP(0);
        u32x4 xa[7], xb[7];                                  // staging registers: even / odd chunks, two chunks ahead
        auto xload = [&](u32x4 (&r)[7], int c) __attribute__((always_inline)) {             // 7 loads, always (c >= 16: the same 16 bytes for every lane)
            const unsigned char* sp = c < 16 ? ximg + c * 128 : xg;
            const uint32_t vo = c < 16 ? xoff : 0u;
            const uint32_t st = c < 16 ? 2 * BI_HW * BI_C * 2 : 0;
            bi_gld<0>(r[0], vo, sp); bi_gld<0>(r[1], vo, sp + st); bi_gld<0>(r[2], vo, sp + 2 * st); bi_gld<0>(r[3], vo, sp + 3 * st);
            bi_gld<0>(r[4], vo, sp + 4 * st); bi_gld<0>(r[5], vo, sp + 5 * st); bi_gld<0>(r[6], vo, sp + 6 * st);
        };
        auto xstore = [&](u32x4 (&r)[7], int stage) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                asm volatile("" : "+v"(r[j]));
                bi_sts(smem, xs_w + stage * BI_XS_STAGE + j * (32 * 128), r[j]);
            }
        };
        xload(xa, 0);
        xload(xb, 1);
        bi_wait_vm<7>();                                     // chunk 0 (and everything older: the ring, the last image's stores)
        __syncthreads();                                     // every wave is done with the image region (last image's conv3)
        xstore(xa, 0);
        xload(xa, 2);
        __syncthreads();
        // one K half-chunk pair = two K steps of 32 out of staging buffer STAGE on ring slots SL0, SL0 + 1
        auto c1_pair = [&](auto STAGE, auto SL0, auto FIRST, u32x4 (&b)[16], u32x4 (&r)[4][4], f32x4 (&ac)[4][14]) __attribute__((always_inline)) {
            constexpr int stage = decltype(STAGE)::value, sl0 = decltype(SL0)::value;
            constexpr bool first = decltype(FIRST)::value != 0;
            const uint32_t ra = xs_r + stage * BI_XS_STAGE, rb = ra ^ 64u;
            bi_pipe<28, 7>(b,
                [&](auto I, u32x4& d) __attribute__((always_inline)) { constexpr int n = decltype(I)::value, ks = n / 14, g = n % 14; bi_ldsr<g * 2048>(d, ks ? rb : ra); },
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr int n = decltype(I)::value, ks = n / 14, g = n % 14, sl = sl0 + ks;
                    if constexpr (g == 0 && !first) bi_wait_vm<26>();
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if constexpr (first && stage == 0 && ks == 0) bi_mma0(ac[f][g], r[sl][f], d);
                        else bi_mma(ac[f][g], r[sl][f], d);
                    }
                    if constexpr (g == 13) refill(std::integral_constant<int, sl>{}, r);
                });
        };
        auto c1_iter = [&](int c, auto FIRST) __attribute__((always_inline)) {
            constexpr bool first = decltype(FIRST)::value != 0;
            // chunk c from stage 0; chunk c + 1 -> stage 1 (free since the barrier that ended chunk c - 1); then chunk c + 3 requested
            bi_wait_vm<(first ? 7 : 23)>();
            xstore(xb, 1);
            xload(xb, c + 3);
            c1_pair(I0{}, I0{}, FIRST, bf, ar, acc);
            __syncthreads();
            bi_wait_vm<(first ? 15 : 23)>();
            if (c + 2 < 16) xstore(xa, 0);
            xload(xa, c + 4);
            c1_pair(I1{}, I2{}, I0{}, bf, ar, acc);
            __syncthreads();
        };
        BISTAMP(1);
        c1_iter(0, I1{});
        BISTAMP(2);
        for (int c = 2; c < 16; c += 2) c1_iter(c, I0{});
        // The staging loads of the last iterations fetch nothing that is used (chunks 16 .. 18 do not exist: they keep the queue
        // uniform).  Their destinations must nevertheless stay allocated until they have landed: a register the compiler
        // believes dead is handed to the next value, and the late load then overwrites it (seen: one output row wrong now and then).
        bi_wait_vm<8>();                                     // behind the last staging load: the ring loads of the last two K steps
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));
        BISTAMP(3);
        bi_settle28(acc[0], acc[1]);                         // the last MFMA results settle before VALU reads them
        bi_settle28(acc[2], acc[3]);
        // t1 = relu(bn1(.)) -> image region (every wave is past the last barrier: nobody reads the staging ring any more)
        auto to_lds = [&](const float* sc, const float* sh) __attribute__((always_inline)) {
            sfor<0, 2>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = Q;
                const int ch = wave * 64 + q * 32 + kq * 8;
                const f32x4 s0 = *(const f32x4*)(sc + ch), s1 = *(const f32x4*)(sc + ch + 4);
                const f32x4 h0 = *(const f32x4*)(sh + ch), h1 = *(const f32x4*)(sh + ch + 4);
#pragma unroll
                for (int g = 0; g < 14; ++g) {
                    u32x4 o = bi_bn8(acc[2 * q][g], acc[2 * q + 1][g], s0, s1, h0, h1, nullptr, rng);
                    if (li >= BI_HW) o = u32x4{0u, 0u, 0u, 0u};          // the zero columns = horizontal padding of conv2
                    bi_sts(smem, twr(g, wave * 8 + q * 4 + kq), o);
                }
            });
        };
        to_lds(bs1, bh1);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int g = 0; g < 14; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        bi_settle28(acc[0], acc[1]);                         // (v_accvgpr_write -> MFMA source: the same pad, generously)
        bi_settle28(acc[2], acc[3]);
        BISTAMP(4);

        // ================================================================ conv2: 4 channel chunks x 9 taps x 2 K halves, no barrier
        for (int cp = 0; cp < 2; ++cp) {                     // chunks 2 cp, 2 cp + 1: operand chunk 8 cp + 4 cpar + 2 ks' .. -> + cp * 256 bytes
            uint32_t tl[3][4], th[3][4];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const uint32_t tb = tbase(d - 1) + cp * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) { tl[d][j] = tb + tpos(d - 1, j); th[d][j] = tl[d][j] + 65536u; }
            }
            bi_pipe<480, 7>(bf,
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr BiP2 q = bi_p2(decltype(I)::value);
                    constexpr int R = q.g + q.dr + 1, j = 2 * q.cpar + q.ks;
                    bi_ldsr<(R & 7) * 8192>(d, R < 8 ? tl[q.dc + 1][j] : th[q.dc + 1][j]);
                },
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr BiP2 q = bi_p2(decltype(I)::value);
                    constexpr int sl = q.step & 3;
                    if constexpr (q.gi == 0) bi_wait_vm<12>();
#pragma unroll
                    for (int f = 0; f < 4; ++f) bi_mma(acc[f][q.g], ar[sl][f], d);
                    if constexpr (q.gi == q.ng - 1) refill(std::integral_constant<int, sl>{}, ar);
                });
        }
        BISTAMP(5);
        bi_settle28(acc[0], acc[1]);
        bi_settle28(acc[2], acc[3]);
        __syncthreads();                                     // every wave is done reading t1
        to_lds(bs2, bh2);
        __syncthreads();
        BISTAMP(6);

        // ================================================================ conv3: 8 chunks of 128 channels (32 per wave), no barrier
        {
            const uint32_t tb = tbase(0);
            uint32_t tl[4], th[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { tl[j] = tb + tpos(0, j); th[j] = tl[j] + 65536u; }
            // identity / output: this lane's 8 channels of pixel (row g, column li)
            const int pcol = li < BI_HW ? li : BI_HW - 1;
            const uint32_t idoff = (uint32_t)((pcol * BI_C + wave * 32 + kq * 8) * 2);
            const uint32_t img_off = (uint32_t)img * (BI_PIX * BI_C * 2u);
            u32x4 idr[14], bnr[4];
            auto chunk_fn = [&](int chunk, auto FIRSTC) __attribute__((always_inline)) {
                constexpr bool firstc = decltype(FIRSTC)::value != 0;
                const unsigned char* isb = ximg + chunk * 256;
                bi_gld<0>(idr[0], idoff, isb); bi_gld<0>(idr[1], idoff, isb + 28672); bi_gld<0>(idr[2], idoff, isb + 2 * 28672);
                bi_gld<0>(idr[3], idoff, isb + 3 * 28672); bi_gld<0>(idr[4], idoff, isb + 4 * 28672); bi_gld<0>(idr[5], idoff, isb + 5 * 28672);
                bi_gld<0>(idr[6], idoff, isb + 6 * 28672); bi_gld<0>(idr[7], idoff, isb + 7 * 28672); bi_gld<0>(idr[8], idoff, isb + 8 * 28672);
                bi_gld<0>(idr[9], idoff, isb + 9 * 28672); bi_gld<0>(idr[10], idoff, isb + 10 * 28672); bi_gld<0>(idr[11], idoff, isb + 11 * 28672);
                bi_gld<0>(idr[12], idoff, isb + 12 * 28672); bi_gld<0>(idr[13], idoff, isb + 13 * 28672);
                bi_gld<0>(bnr[0], (uint32_t)kq * 32u, bs3 + chunk * 512); bi_gld<16>(bnr[1], (uint32_t)kq * 32u, bs3 + chunk * 512);
                bi_gld<0>(bnr[2], (uint32_t)kq * 32u, bh3 + chunk * 512); bi_gld<16>(bnr[3], (uint32_t)kq * 32u, bh3 + chunk * 512);
#ifndef BI_LA3
#define BI_LA3 7
#endif
#ifndef BI_STAUX
#define BI_STAUX 0                                           // cache policy of the output stores (A/B): 2 = nt
#endif
                bi_pipe<112, BI_LA3>(bf,
                    [&](auto I, u32x4& d) __attribute__((always_inline)) {
                        constexpr int n = decltype(I)::value, ks = n / 14, g = n % 14, R = g + 1;
                        bi_ldsr<(R & 7) * 8192 + (ks >> 2) * 256>(d, R < 8 ? tl[ks & 3] : th[ks & 3]);
                    },
                    [&](auto I, u32x4& d) __attribute__((always_inline)) {
                        constexpr int n = decltype(I)::value, ks = n / 14, g = n % 14, sl = ks >> 1, fo = (ks & 1) * 2;
                        // slot sl was refilled one chunk ago: behind it 12 more ring loads, 14 stores, this chunk's 18 loads
                        if constexpr (g == 0 && (ks & 1) == 0) bi_wait_vm<(firstc ? 30 : 44)>();
                        if constexpr (ks == 0) { bi_mma0(acc[0][g], ar[sl][fo], d); bi_mma0(acc[1][g], ar[sl][fo + 1], d); }
                        else { bi_mma(acc[0][g], ar[sl][fo], d); bi_mma(acc[1][g], ar[sl][fo + 1], d); }
                        if constexpr (g == 13 && (ks & 1)) refill(std::integral_constant<int, sl>{}, ar);
                    });
                if (chunk == 1) BISTAMP(8);
                bi_wait_vm<16>();                            // identity + BatchNorm rows (behind them: this chunk's 16 ring loads)
                if (chunk == 1) BISTAMP(9);
#pragma unroll
                for (int g = 0; g < 14; ++g) asm volatile("" : "+v"(idr[g]));
                asm volatile("" : "+v"(bnr[0]), "+v"(bnr[1]), "+v"(bnr[2]), "+v"(bnr[3]));
                bi_settle28(acc[0], acc[1]);
                const f32x4 s0 = __builtin_bit_cast(f32x4, bnr[0]), s1 = __builtin_bit_cast(f32x4, bnr[1]);
                const f32x4 h0 = __builtin_bit_cast(f32x4, bnr[2]), h1 = __builtin_bit_cast(f32x4, bnr[3]);
                const uint32_t so = img_off + chunk * 256;
#pragma unroll
                for (int g = 0; g < 14; ++g) {
                    const u32x4 o = bi_bn8(acc[0][g], acc[1][g], s0, s1, h0, h1, &idr[g], rng);
                    const uint32_t vo = li < BI_HW ? so + g * 28672 + idoff : 0xffffff00u;      // (junk columns: dropped by the range check)
                    __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, vo, 0, BI_STAUX);
                }
            };
#ifndef BI_EXP
#define BI_EXP 0
#endif
            if (BI_EXP & 1) { __syncthreads(); asm volatile("s_sleep 30" ::: "memory"); __syncthreads(); }
            if (BI_EXP & 2) { bi_wait_vm<0>(); chunk_fn(0, I0{}); } else
            chunk_fn(0, I1{});
            BISTAMP(7);
            for (int c = 1; c < 8; ++c) { chunk_fn(c, I0{}); if (c == 1) BISTAMP(10); }
            BISTAMP(11);
        }
#ifdef AP_TRACE
        ++img_no;
#endif
    }
    bi_wait_vm<0>();                                         // (the ring ran ahead into the next image: nothing may land after the exit)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ar[j][0]), "+v"(ar[j][1]), "+v"(ar[j][2]), "+v"(ar[j][3]));
    ap_rng_flush(a.range_flag, rng);
}

}  // namespace

size_t ap_block_img_stream_bytes(void) { return 4 * BI_WAVE_BYTES; }

// w1 [256][1024], w2 [256][3][3][256], w3 [1024][256]: K-contiguous 16-bit rows as packed for the stand-alone kernels
hipError_t ap_launch_block_img_pack(const void* w1, const void* w2, const void* w3, void* dst, hipStream_t st) {
    if (!w1 || !w2 || !w3 || !dst) return hipErrorInvalidValue;
    const size_t pieces = 4 * BI_WAVE_BYTES / 16;
    hipLaunchKernelGGL(blk_img_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w1,
                       (const bf16_t*)w2, (const bf16_t*)w3, (unsigned char*)dst);
    return hipGetLastError();
}

hipError_t ap_launch_block_img(const BlkImgArgs& a, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    if (a.N <= 0 || !a.x || !a.y || !a.wfrag || !a.s1 || !a.h1 || !a.s2 || !a.h2 || !a.s3 || !a.h3) return hipErrorInvalidValue;
    if ((size_t)a.N * (BI_PIX * BI_C * 2) >= 0xffffff00ull) return hipErrorInvalidValue;                       // 32-bit offsets, out-of-range marker
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!n_cu_dev[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)blk_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BI_T_BYTES);
        if (e != hipSuccess) return e;
        n_cu_dev[dev] = n;
    }
    const int grid = a.N < n_cu_dev[dev] ? a.N : n_cu_dev[dev];
    hipLaunchKernelGGL(blk_img_kernel, dim3(grid), dim3(256), BI_T_BYTES, st, a);
    return hipGetLastError();
}

AP_NS_END
