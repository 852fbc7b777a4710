// Stride-2 3x3 convolution of layer2.0 (128 -> 128 channels, 56 x 56 -> 28 x 28) in POLYPHASE form, a quarter of an output image per
// workgroup (16-bit storage, gfx950): conv2 of the stage's first Bottleneck, model_copenet.py:32-34 with :18 (+ bn2 + ReLU).
//
// Output pixel (y, x) reads the input pixels (2 y + dr - 1, 2 x + dc - 1): the rows with dr = 1 are EVEN input rows, dr = 0 / 2 odd
// ones, the same for the columns -- so the nine taps fall on the four parity phases of the input, one tap on (even, even), two each on
// (even, odd) and (odd, even), four on (odd, odd), and inside a phase a tap is a constant shift of at most one row / column: the
// stride-1 machinery of conv_img3.hip (an operand image resident in LDS, a tap = a slot shift, no masks, no im2col) applies to each
// phase.  The ring kernel this replaces for the layer gathers its B operand pixel by pixel through the LDS ring at 655 TFLOP/s.
//   * a workgroup owns 14 x 14 output pixels of an image (a quarter); the phase sub-image it needs is 14 / 15 rows of 14 / 15 pixels
//     = one 15 x 16-slot region of 256-byte slots (60 KB, conv_img3's chunk rotation (c + 2 slot) mod 16); TWO regions: while the
//     four compute waves run the K steps of one phase, the other region receives the next phase;
//   * waves split by role, as in the persistent stem kernel (stem.hip): waves 0-3 compute (one per SIMD: 64 of the 128 output
//     channels x 7 of the 14 rows each: 28 accumulator tiles, an LDS operand fragment feeds four MFMAs, a weight fragment seven),
//     waves 4-7 issue the LDS-DMA of the next phase (global_load ... lds straight from the NHWC rows, swizzle and zero padding
//     applied on the source side) and wait for it -- in a queue of their own, so the DMA has a whole phase to land while the compute
//     waves' in-order vmcnt only ever counts their weight pieces and output stores (what kept conv_img3 / block_img from
//     overlapping their image DMA with their MFMA loops).  One workgroup barrier per phase;
//   * phase order (odd, odd) 16 K steps, (even, odd) 8, (odd, even) 8, (even, even) 4: the shortest phase is the last, so the next
//     quarter's first DMA also has the epilogue (BatchNorm + ReLU + 14 stores per lane) to hide under;
//   * weights: per-wave fragment streams from L2 in consumption order through a register ring, as in conv_img3.hip.
// K order per output element: phase-major (the taps in the order 0, 2, 6, 8 | 3, 5 | 1, 7 | 4, each over its four 32-channel K
// steps) -- NOT the ring kernel's tap order: the layer runs on this kernel at every batch size (ap_net_set_s2p), so a pair's result
// does not depend on its batch.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int SP_HI = 56, SP_HO = 28, SP_Q = 14, SP_P = 128;
constexpr int SP_REGION = (15 * 16 + 1) * 256;               // 61 696: 15 rows of 16 slots + the slot the junk lanes of the last row reach
constexpr int SP_BN = 2 * SP_REGION;                         // BatchNorm scale | shift (fp32) behind the two regions
constexpr int SP_LDS = SP_BN + 2 * SP_P * 4;                 // 124 416
constexpr int SP_STEPS = 36;
constexpr size_t SP_WAVE_BYTES = (size_t)SP_STEPS * 4096;    // per channel half; the layer: 294 912
#ifndef SP_RING
#define SP_RING 3                                            // weight pieces (K steps) in flight per compute wave
#endif
// Timing-only builds (results WRONG): -DSP_ABLATE=<bits>: 1 no image DMA after the first phase | 2 no output stores | 4 no MFMAs
#ifndef SP_ABLATE
#define SP_ABLATE 0
#endif

#include "bi_helpers.inc"

// K step -> phase, tap, 32-channel quarter.  Phase p: 0 (odd rows, odd cols) | 1 (even, odd) | 2 (odd, even) | 3 (even, even)
struct SpStep { int phase, dr, dc, j, rs, cs; };
constexpr SpStep sp_step(int s) {
    constexpr int taps[9] = {0, 2, 6, 8, 3, 5, 1, 7, 4};     // tap = dr * 3 + dc
    SpStep r{};
    const int t = taps[s >> 2];
    r.phase = s < 16 ? 0 : s < 24 ? 1 : s < 32 ? 2 : 3;
    r.dr = t / 3; r.dc = t % 3; r.j = s & 3;
    r.rs = r.dr == 2 ? 1 : 0; r.cs = r.dc == 2 ? 1 : 0;     // odd phases: row / column index (r0 - 1 + R): dr = 0 -> R = g, dr = 2 -> R = g + 1
    return r;
}
constexpr int sp_first(int phase) { return phase == 0 ? 0 : phase == 1 ? 16 : phase == 2 ? 24 : 32; }
constexpr int sp_count(int phase) { return phase == 0 ? 16 : phase == 3 ? 4 : 8; }

// weight stream: channel half cw -> [step] of 4 fragments (64 rows); fragment = [lane 64][8 K values]: row lane & 15, K columns
// 8 (lane >> 4) .. + 7 of the step's 32.  w2: [128][3][3][128] K-contiguous rows (as packed for the stand-alone kernels)
__global__ void __launch_bounds__(256) conv_s2p_pack_kernel(const bf16_t* __restrict__ w2, unsigned char* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_half = SP_WAVE_BYTES / 16;
    if (idx >= 2 * per_half) return;
    const int cw = (int)(idx / per_half);
    const int p = (int)(idx - (size_t)cw * per_half);
    const int step = p >> 8, f = (p >> 6) & 3, lane = p & 63;
    const int ch = cw * 64 + bi_row_channel(f * 16 + (lane & 15));
    const SpStep q = sp_step(step);
    const bf16_t* src = w2 + (size_t)ch * (9 * SP_P) + (q.dr * 3 + q.dc) * SP_P + q.j * 32 + (lane >> 4) * 8;
    *(u32x4*)(dst + idx * 16) = *(const u32x4*)src;
}

__device__ __forceinline__ void sp_settle14(f32x4 (&p)[7], f32x4 (&q)[7]) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(p[0]), "+a"(p[1]), "+a"(p[2]), "+a"(p[3]), "+a"(p[4]), "+a"(p[5]), "+a"(p[6]), "+a"(q[0]), "+a"(q[1]), "+a"(q[2]),
                   "+a"(q[3]), "+a"(q[4]), "+a"(q[5]), "+a"(q[6]));
    __builtin_amdgcn_sched_barrier(0);
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_s2p_kernel(const ConvS2pArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // both regions start as zeros (every slot a junk lane can reach holds a finite value from the first read on), BatchNorm rows to LDS
    for (int idx = tid; idx < 2 * SP_REGION / 16; idx += 512) bi_sts(smem, idx * 16, u32x4{0u, 0u, 0u, 0u});
    if (tid < 2 * SP_P) ((float*)(smem + SP_BN))[tid] = tid < SP_P ? a.scale[tid] : a.shift[tid - SP_P];

    const int nunits = a.nunits_pad;                         // quarter images, padded to whole groups of 32
    auto unit_of = [&](int u, int& img, int& qy, int& qx) __attribute__((always_inline)) {   // the four quarters of an image on one XCD (block b -> XCD b % 8)
        const int q = (u >> 3) & 3;
        qy = q >> 1; qx = q & 1;
        img = (u & 7) + 8 * (u >> 5);
    };
    const int u0 = blockIdx.x, ustep = gridDim.x;
    const int nmine = u0 < nunits ? (nunits - u0 + ustep - 1) / ustep : 0;   // units of this workgroup; 4 phases each

    if (wave >= 4) {
        // ------------------------------------------------------------ feeding waves: the LDS-DMA of phase P + 1 during phase P
        const int fw = wave - 4, dslot = lane >> 4, dp = lane & 15;
        const unsigned char* const xg = (const unsigned char*)a.x;
        const unsigned char* const zg = (const unsigned char*)a.zero;
        // phase `ph` of unit u into region `reg`: LDS position (row R, slot S, p) <- global chunk (p - 2 S) mod 16 of input pixel
        // (2 (r0 + R) - pr, 2 (c0 + S) - pc), or of the zero line where that pixel is outside the image / the slot is not needed
        auto dma = [&](int u, int ph, int reg) __attribute__((always_inline)) {
            int img, qy, qx;
            unit_of(u, img, qy, qx);
            const bool live = img < a.N;
            const unsigned char* const ximg = xg + (size_t)(live ? img : 0) * (SP_HI * SP_HI * SP_P * 2);
            const int pr = (ph == 0 || ph == 2) ? 1 : 0, pc = (ph == 0 || ph == 1) ? 1 : 0;
            const int nrow = 14 + pr, ncol = 14 + pc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int S = 4 * j + dslot, col = 2 * (qx * SP_Q + S) - pc;
                const bool cok = live && S < ncol && col >= 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = fw + 4 * i;                // rows fw, fw + 4, ..: 15 rows over 4 waves
                    if (R >= nrow) continue;                 // (wave-uniform)
                    const int row = 2 * (qy * SP_Q + R) - pr;
                    const bool ok = cok && row >= 0;
                    const unsigned char* src = ok ? ximg + ((size_t)(row * SP_HI + col) * (SP_P * 2) + (uint32_t)(((dp - 2 * S) & 15) << 4)) : zg + (dp << 4);
                    const uint32_t m0v = lds0 + (uint32_t)(reg * SP_REGION + (16 * R + 4 * j) * 256);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
                }
            }
        };
        __syncthreads();                                     // (the zeroing above is done before the first DMA lands)
        if (nmine > 0) dma(u0, 0, 0);
        bi_wait_vm<0>();
        __syncthreads();
        for (int k = 0; k < nmine; ++k) {
            const int u = u0 + k * ustep;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                if (ph < 3) { if (!(SP_ABLATE & 1)) dma(u, ph + 1, (ph + 1) & 1); }
                else if (k + 1 < nmine && !(SP_ABLATE & 1)) dma(u + ustep, 0, 0);
                bi_wait_vm<0>();
                __syncthreads();
            }
        }
        return;
    }

    // ---------------------------------------------------------------- compute waves
    const int cw = wave & 1, rh = wave >> 1;                 // this wave's 64 output channels, its 7 rows
    const int li = lane & 15, kq = lane >> 4;
    uint32_t rng = 0u;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)a.y, 0, (int)((uint32_t)a.N * (SP_HO * SP_HO * SP_P * 2u) + (a.y_tiled ? 4096u : 0u)), 0x00020000);
    const unsigned char* const wsb = (const unsigned char*)a.wfrag + (size_t)cw * SP_WAVE_BYTES;
    asm volatile("" ::"s"(wsb), "s"(nunits));
    const uint32_t wlane = lane * 16;
    const unsigned char* wp = wsb;
    int wcnt = 0;
    u32x4 ar[SP_RING][4];
    auto refill = [&](auto SL, u32x4 (&r)[SP_RING][4]) __attribute__((always_inline)) {
        constexpr int sl = decltype(SL)::value;
        bi_gld<0>(r[sl][0], wlane, wp); bi_gld<1024>(r[sl][1], wlane, wp);
        bi_gld<2048>(r[sl][2], wlane, wp); bi_gld<3072>(r[sl][3], wlane, wp);
        wp += 4096;
        if (++wcnt == SP_STEPS) { wcnt = 0; wp = wsb; }
    };
    sfor<0, SP_RING>([&](auto S) __attribute__((always_inline)) { refill(S, ar); });
    // B fragments: region row R = 7 rh + g + rs, slot m = li + cs, chunk c = 4 j + kq at position (c + 2 m) mod 16 (conv_img3.hip)
    uint32_t tl[2][2][4];                                    // [region][cs][j]
#pragma unroll
    for (int reg = 0; reg < 2; ++reg)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tl[reg][cs][j] = lds0 + (uint32_t)(reg * SP_REGION + rh * 7 * 4096 + (li + cs) * 256 + (((4 * j + kq + 2 * (li + cs)) & 15) << 4));
    f32x4 acc[4][7];
    u32x4 bf[8];
    __syncthreads();
    bi_wait_vm<0>();                                         // the first ring pieces
    __syncthreads();
    for (int k = 0; k < nmine; ++k) {
        const int u = u0 + k * ustep;
        int img, qy, qx;
        unit_of(u, img, qy, qx);
        sfor<0, 4>([&](auto PH) __attribute__((always_inline)) {
            constexpr int ph = decltype(PH)::value, s0 = sp_first(ph), ns = sp_count(ph);
            bi_pipe<ns * 7, 7, 8>(bf,
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr int n = decltype(I)::value, g = n % 7;
                    constexpr SpStep q = sp_step(s0 + n / 7);
                    bi_ldsr<(g + q.rs) * 4096>(d, tl[ph & 1][q.cs][q.j]);
                },
                [&](auto I, u32x4& d) __attribute__((always_inline)) {
                    constexpr int n = decltype(I)::value, step = s0 + n / 7, g = n % 7, sl = step % SP_RING;
                    // the pieces of steps 0 .. SP_RING - 1 were requested during the previous quarter and are older than its stores: the
                    // wait behind those covered them; later: the piece of this step, SP_RING - 1 younger pieces behind it
                    if constexpr (g == 0 && step >= SP_RING) bi_wait_vm<4 * (SP_RING - 1)>();
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if constexpr ((SP_ABLATE & 4) != 0 && step != 0) asm volatile("" : "+a"(acc[f][g]) : "v"(ar[sl][f]), "v"(d));
                        else if constexpr (step == 0) bi_mma0(acc[f][g], ar[sl][f], d);
                        else bi_mma(acc[f][g], ar[sl][f], d);
                    }
                    if constexpr (g == 6) refill(std::integral_constant<int, sl>{}, ar);
                });
            if constexpr (ph < 3) __syncthreads();           // this phase's region may be refilled; the next one has landed
        });
        sp_settle14(acc[0], acc[1]);
        sp_settle14(acc[2], acc[3]);
        // ================================================================ bn2 + ReLU + 16-bit, 14 stores of 16 bytes per lane
        const bool live = img < a.N && li < SP_Q;
        const uint32_t col = (uint32_t)(qx * SP_Q + li);
        const float* const bn = (const float*)(smem + SP_BN);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = cw * 64 + q * 32 + kq * 8;
            const f32x4 s0v = *(const f32x4*)(bn + ch), s1v = *(const f32x4*)(bn + ch + 4);
            const f32x4 h0v = *(const f32x4*)(bn + SP_P + ch), h1v = *(const f32x4*)(bn + SP_P + ch + 4);
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const u32x4 o = bi_bn8(acc[2 * q][g], acc[2 * q + 1][g], s0v, s1v, h0v, h1v, nullptr, rng);
                const uint32_t m = ((uint32_t)img * SP_HO + (uint32_t)(qy * SP_Q + rh * 7 + g)) * SP_HO + col;     // linear pixel index
                const uint32_t off = a.y_tiled ? (uint32_t)(((m >> 4) * (SP_P >> 3) + (uint32_t)(ch >> 3)) * 256u + (m & 15u) * 16u)
                                               : (uint32_t)(m * (SP_P * 2u) + (uint32_t)ch * 2u);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, (live && !(SP_ABLATE & 2)) ? off : 0xffffff00u, 0, 0);
            }
        }
        // everything older than this wave's 14 stores has landed (the first ring pieces of the next quarter among it)
        bi_wait_vm<14>();
        __syncthreads();                                     // the fourth barrier of the quarter (the next one's first phase has landed)
    }
    bi_wait_vm<0>();                                         // (the ring ran ahead: nothing may land after the exit)
#pragma unroll
    for (int j = 0; j < SP_RING; ++j) asm volatile("" : "+v"(ar[j][0]), "+v"(ar[j][1]), "+v"(ar[j][2]), "+v"(ar[j][3]));
    ap_rng_flush(a.range_flag, rng);
}

}  // namespace

size_t ap_conv_s2p_stream_bytes(void) { return 2 * SP_WAVE_BYTES; }

bool ap_conv_s2p_supported(int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    return H == SP_HI && W == SP_HI && Cin == SP_P && Cout == SP_P && k == 3 && stride == 2 && pad == 1;
}

// w2: [128][3][3][128] K-contiguous 16-bit rows as packed for the stand-alone kernels
hipError_t ap_launch_conv_s2p_pack(const void* w2, void* dst, hipStream_t st) {
    if (!w2 || !dst) return hipErrorInvalidValue;
    const size_t pieces = 2 * SP_WAVE_BYTES / 16;
    hipLaunchKernelGGL(conv_s2p_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w2, (unsigned char*)dst);
    return hipGetLastError();
}

hipError_t ap_launch_conv_s2p(const ConvS2pArgs& a, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    if (a.N <= 0 || !a.x || !a.y || !a.wfrag || !a.scale || !a.shift || !a.zero) return hipErrorInvalidValue;
    if ((size_t)a.N * (SP_HI * SP_HI * SP_P * 2) >= 0xffff0000ull) return hipErrorInvalidValue;                   // 32-bit offsets, out-of-range marker
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!n_cu_dev[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_s2p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
        if (e != hipSuccess) return e;
        n_cu_dev[dev] = n;
    }
    // quarter images in groups of 32 (8 images: the four quarters of an image on one XCD); grid: one workgroup per CU, a multiple of 32
    const int nu32 = (4 * a.N + 31) / 32 * 32;
    int grid = n_cu_dev[dev] / 32 * 32;
    if (grid < 32) grid = 32;
    if (grid > nu32) grid = nu32;
    ConvS2pArgs b = a;
    b.nunits_pad = nu32;
    hipLaunchKernelGGL(conv_s2p_kernel, dim3(grid), dim3(512), SP_LDS, st, b);
    return hipGetLastError();
}

AP_NS_END
