// Stride-1 3x3 convolution of layer2 (128 -> 128 channels at 28 x 28) with HALF AN IMAGE resident in the LDS of one CU (16-bit
// storage, gfx950): conv2 of a Bottleneck, model_copenet.py:32-34 with :18 (+ bn2 + ReLU, :33-34).
//
// This is the conv2 phase of block_img.hip (an image-resident 3x3 at 17.8 cycles per MFMA: 90 % of the matrix pipe's issue rate, 0.32
// operand fetches per MFMA) as a kernel of its own, for the stage whose image does not fit a CU: 28 x 28 x 128 16-bit values are
// 200 KB.  The image is split by COLUMNS: a workgroup owns 14 of the 28 columns over all 28 rows, plus one halo column on either
// side -- 16 slots per row, exactly the row of block_img's layout:
//   * pixels in 28 rows of 16 slots: slot S of a row = image column c0 - 1 + S (c0 = 0 or 14); the slot outside the image (S = 0 of
//     the left half, S = 15 of the right one) holds zeros = the horizontal padding, the other outer slot is the neighbour half's
//     first column (the halo: fetched, never produced); a zero row above and below: a 3x3 tap is a constant slot shift, no masks;
//   * 482 slots x 256 B = 120.5 KB, 16-byte chunk c of slot u at position (c + 2 u) mod 16 (conflict-free ds_read_b128 under every tap
//     shift for the instruction's real lane groups: see tbase below);
//   * four waves, one per SIMD, the whole register file each: a wave computes 64 of the 128 output channels for 14 of the 28 rows
//     (56 accumulators of 16 x 16 in the accumulator half): an LDS operand fragment feeds four MFMAs, a weight fragment fourteen;
//   * weights never touch the LDS: each wave streams the rows of ITS 64 channels from L2 as MFMA A fragments packed in register and
//     consumption order (4 KiB per K step of 32; 288 KB per layer, the same for every half image) through a four-piece register ring;
//   * the half image arrives by LDS-DMA straight from the NHWC rows (global_load ... lds, 112 x 1 KiB, the chunk swizzle and the zero
//     slots applied on the source side), requested right behind the last operand read of the previous half image -- BEFORE that
//     one's BatchNorm / ReLU epilogue and its 28 output stores, so the epilogue and the store issue run under the DMA and the counted
//     wait behind them ("at most the 28 stores are younger") does not drain the stores.
// K order per output element = conv_slab.hip's (64-channel chunk outer, tap inner, K half inner): the same sums, bit for bit, so the
// trunk may take either kernel by problem size without a pair's result depending on its batch.
#include <type_traits>

#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int CI_H = 28, CI_HALF = 14, CI_P = 128;
constexpr int CI_SLOTS = 30 * 16 + 2;                        // u = 16 (row + 1) + S; two slots past the end for the shifted reads of the junk lanes
constexpr int CI_LDS = CI_SLOTS * 256;                       // 123 392
constexpr int CI_STEPS = 36;                                 // K steps of 32: 2 channel chunks x 9 taps x 2 K halves
constexpr size_t CI_WAVE_BYTES = (size_t)CI_STEPS * 4096;    // per channel half: 147 456; the layer: 294 912

#include "bi_helpers.inc"

// Timing-only builds (results WRONG, times valid): -DCI_ABLATE=<bits>: 1 no image DMA after the first | 2 no output stores | 4 no MFMAs
#ifndef CI_ABLATE
#define CI_ABLATE 0
#endif
#ifndef CI_RING
#define CI_RING 4                                            // weight pieces (K steps) in flight per wave
#endif

// weight stream: channel half cw -> [step (64-channel chunk, tap, K half)] of 4 fragments (64 rows); fragment = [lane 64][8 K values]:
// row lane & 15, K columns 8 (lane >> 4) .. + 7 of the step's 32.  w2: [128][3][3][128] K-contiguous rows (as packed for conv_slab)
__global__ void __launch_bounds__(256) conv_img3_pack_kernel(const bf16_t* __restrict__ w2, unsigned char* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_half = CI_WAVE_BYTES / 16;
    if (idx >= 2 * per_half) return;
    const int cw = (int)(idx / per_half);
    const int p = (int)(idx - (size_t)cw * per_half);
    const int step = p >> 8, f = (p >> 6) & 3, lane = p & 63;
    const int ch = cw * 64 + bi_row_channel(f * 16 + (lane & 15));
    const int c64 = step / 18, tap = (step % 18) >> 1, ks = step & 1;
    const bf16_t* src = w2 + (size_t)ch * (9 * CI_P) + tap * CI_P + c64 * 64 + ks * 32 + (lane >> 4) * 8;
    *(u32x4*)(dst + idx * 16) = *(const u32x4*)src;
}

// K step -> channel chunk, tap, K half
struct CiStep { int c64, dr, dc, ks; };
constexpr CiStep ci_step(int s) {
    CiStep r{};
    r.c64 = s / 18;
    const int tap = (s % 18) >> 1;
    r.dr = tap / 3 - 1;
    r.dc = tap % 3 - 1;
    r.ks = s & 1;
    return r;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) conv_img3_kernel(const ConvImg3Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, rh = wave >> 1;                 // this wave's 64 output channels, its 14 rows
    const int li = lane & 15, kq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t rng = 0u;

    // zero rows (slots 0 .. 15 and 464 .. 481): written once, never touched again
    for (int idx = tid; idx < 34 * 16; idx += 256) {
        const int s = idx >> 4, c = idx & 15, u = s < 16 ? s : 464 + (s - 16);
        bi_sts(smem, u * 256 + c * 16, u32x4{0u, 0u, 0u, 0u});
    }

    const unsigned char* const xg = (const unsigned char*)a.x;
    const unsigned char* const zg = (const unsigned char*)a.zero;
    const int nhalf = a.nhalf_pad;                           // half images, padded to whole groups of 16 (a.N need not be a multiple of 8)
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)a.y, 0, (int)((uint32_t)a.N * (CI_H * CI_H * CI_P * 2u) + (a.y_tiled ? 4096u : 0u)), 0x00020000);
    // ---- weight stream of this wave's channel half: 4-KiB pieces in consumption order, four pieces ahead in registers
    const unsigned char* const wsb = (const unsigned char*)a.wfrag + (size_t)cw * CI_WAVE_BYTES;
    // BatchNorm rows of the lane's channels (pair q: channels 64 cw + 32 q + 8 kq .. + 7), once
    f32x4 bs[2][2], bh[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ch = cw * 64 + q * 32 + kq * 8;
        bs[q][0] = *(const f32x4*)(a.scale + ch); bs[q][1] = *(const f32x4*)(a.scale + ch + 4);
        bh[q][0] = *(const f32x4*)(a.shift + ch); bh[q][1] = *(const f32x4*)(a.shift + ch + 4);
    }
    // every kernel-argument load completes here (scalar loads share lgkmcnt with the counted fragment reads)
    asm volatile("" ::"s"(xg), "s"(zg), "s"(nhalf), "s"(wsb));
    const uint32_t wlane = lane * 16;
    const unsigned char* wp = wsb;
    int wcnt = 0;
    u32x4 ar[CI_RING][4];
    auto refill = [&](auto SL, u32x4 (&r)[CI_RING][4]) __attribute__((always_inline)) {           // ring slot SL <- the next piece of the stream
        constexpr int sl = decltype(SL)::value;
        bi_gld<0>(r[sl][0], wlane, wp); bi_gld<1024>(r[sl][1], wlane, wp);
        bi_gld<2048>(r[sl][2], wlane, wp); bi_gld<3072>(r[sl][3], wlane, wp);
        wp += 4096;
        if (++wcnt == CI_STEPS) { wcnt = 0; wp = wsb; }
    };
    sfor<0, CI_RING>([&](auto S) __attribute__((always_inline)) { refill(S, ar); });

    // ---- image DMA: wave w fetches rows w, w + 4, ..; per row four 1-KiB pieces of four slots; lane = (slot of the piece, chunk position p):
    // LDS position (slot S, p) <- global chunk (p - 2 S) mod 16 of pixel (row, c0 - 1 + S), or of the zero line where that column is outside the image
    const int dslot = lane >> 4, dp = lane & 15;
    auto half_of = [&](int hi, int& img, int& half) __attribute__((always_inline)) {       // both halves of an image on one XCD (block b -> XCD b % 8)
        half = (hi >> 3) & 1;
        img = (hi & 7) + 8 * (hi >> 4);
    };
    auto dma_half = [&](int hi) __attribute__((always_inline)) {
        int img, half;
        half_of(hi, img, half);
        const bool live = img < a.N;
        const unsigned char* const ximg = xg + (size_t)(live ? img : 0) * (CI_H * CI_H * CI_P * 2);
        const int c0 = half * CI_HALF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int S = 4 * j + dslot, col = c0 - 1 + S;
            const bool ok = live && col >= 0 && col < CI_H;
            const unsigned char* src = ok ? ximg + ((size_t)col * CI_P * 2 + (uint32_t)(((dp - 2 * S) & 15) << 4)) : zg + (dp << 4);
            const uint32_t rstep = ok ? CI_H * CI_P * 2 : 0u;
            src += (size_t)wave * rstep;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int row = wave + 4 * i;
                const uint32_t m0v = lds0 + (uint32_t)((16 * (row + 1) + 4 * j) * 256);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
                src += 4 * (size_t)rstep;
            }
        }
    };

    // B fragments out of the image: physical slot u = 16 R + m, m = li + dc + 1 (R = row + 1); chunk c = 8 c64 + 4 ks + kq at position
    // (c + 2 m) mod 16 of its slot.  (Rounds 5-6 had c ^ (m & 15), conflict-free if ds_read_b128 served 16 CONTIGUOUS lanes per
    // cycle; its groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, .. (MI355X_MICROARCH.md), i.e. eight pixels of one K quarter
    // with eight of the next, and the XOR form puts two of them on the same banks under every tap shift: PMC, 25 % of this kernel's
    // LDS cycles.  The rotation by 2 m is conflict-free for all four groups and the three shifts: tools/probes/lds_groups.py.)
    auto tbase = [&](int dc) __attribute__((always_inline)) -> uint32_t {
        const int m = li + dc + 1;
        return lds0 + (uint32_t)(rh * CI_HALF * 4096 + m * 256);
    };
    uint32_t tl[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const uint32_t tb = tbase(d - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[d][j] = tb + (uint32_t)(((4 * j + kq + 2 * (li + d)) & 15) << 4);
    }
    using I0 = std::integral_constant<int, 0>;

    f32x4 acc[4][14];
    u32x4 bf[16];

    int hi = blockIdx.x;
    if (hi < nhalf) dma_half(hi);
    bi_wait_vm<0>();
    __syncthreads();
    for (; hi < nhalf; hi += gridDim.x) {
        int img, half;
        half_of(hi, img, half);
        // ================================================================ 36 K steps x 14 pixel rows, no barrier
        bi_pipe<CI_STEPS * 14, 7>(bf,
            [&](auto I, u32x4& d) __attribute__((always_inline)) {
                constexpr int n = decltype(I)::value, g = n % 14;
                constexpr CiStep q = ci_step(n / 14);
                bi_ldsr<(g + q.dr + 1) * 4096>(d, tl[q.dc + 1][2 * q.c64 + q.ks]);
            },
            [&](auto I, u32x4& d) __attribute__((always_inline)) {
                constexpr int n = decltype(I)::value, step = n / 14, g = n % 14, sl = step % CI_RING;
                // the ring pieces of steps 0 .. 3 were requested during the previous half image and are older than its DMA and stores: the
                // wait behind those covered them; from step 4 on: the piece of this step, three younger pieces behind it
                if constexpr (g == 0 && step >= CI_RING) bi_wait_vm<4 * (CI_RING - 1)>();
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if constexpr ((CI_ABLATE & 4) != 0 && step != 0) asm volatile("" : "+a"(acc[f][g]) : "v"(ar[sl][f]), "v"(d));
                    else if constexpr (step == 0) bi_mma0(acc[f][g], ar[sl][f], d);
                    else bi_mma(acc[f][g], ar[sl][f], d);
                }
                if constexpr (g == 13) refill(std::integral_constant<int, sl>{}, ar);
            });
        bi_settle28(acc[0], acc[1]);
        bi_settle28(acc[2], acc[3]);
        __syncthreads();                                     // every wave is done reading this half image
        const int hn = hi + (int)gridDim.x;
        if (hn < nhalf && !(CI_ABLATE & 1)) dma_half(hn);    // the next one: in flight under the epilogue
        // ================================================================ bn2 + ReLU + 16-bit, 28 stores of 16 bytes per lane
        const bool live = img < a.N && li < CI_HALF;
        const uint32_t col = (uint32_t)(half * CI_HALF + li);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ch = cw * 64 + q * 32 + kq * 8;
#pragma unroll
            for (int g = 0; g < 14; ++g) {
                const u32x4 o = bi_bn8(acc[2 * q][g], acc[2 * q + 1][g], bs[q][0], bs[q][1], bh[q][0], bh[q][1], nullptr, rng);
                const uint32_t m = ((uint32_t)img * CI_H + (uint32_t)(rh * CI_HALF + g)) * CI_H + col;     // linear pixel index
                const uint32_t off = a.y_tiled ? (uint32_t)(((m >> 4) * (CI_P >> 3) + (uint32_t)(ch >> 3)) * 256u + (m & 15u) * 16u)
                                               : (uint32_t)(m * (CI_P * 2u) + (uint32_t)ch * 2u);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, (live && !(CI_ABLATE & 2)) ? off : 0xffffff00u, 0, 0);
            }
        }
        // the next half image has landed: vector-memory operations retire in order and only this wave's 28 stores are younger than its DMA
        bi_wait_vm<28>();
        __syncthreads();
    }
    bi_wait_vm<0>();                                         // (the ring ran ahead: nothing may land after the exit)
#pragma unroll
    for (int j = 0; j < CI_RING; ++j) asm volatile("" : "+v"(ar[j][0]), "+v"(ar[j][1]), "+v"(ar[j][2]), "+v"(ar[j][3]));
    ap_rng_flush(a.range_flag, rng);
}

}  // namespace

size_t ap_conv_img3_stream_bytes(void) { return 2 * CI_WAVE_BYTES; }

bool ap_conv_img3_supported(int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    return H == CI_H && W == CI_H && Cin == CI_P && Cout == CI_P && k == 3 && stride == 1 && pad == 1;
}

// w2: [128][3][3][128] K-contiguous 16-bit rows as packed for the stand-alone kernels
hipError_t ap_launch_conv_img3_pack(const void* w2, void* dst, hipStream_t st) {
    if (!w2 || !dst) return hipErrorInvalidValue;
    const size_t pieces = 2 * CI_WAVE_BYTES / 16;
    hipLaunchKernelGGL(conv_img3_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w2, (unsigned char*)dst);
    return hipGetLastError();
}

hipError_t ap_launch_conv_img3(const ConvImg3Args& a, hipStream_t st) {
    static int n_cu_dev[AP_MAX_DEVICES] = {};
    if (a.N <= 0 || !a.x || !a.y || !a.wfrag || !a.scale || !a.shift || !a.zero) return hipErrorInvalidValue;
    if ((size_t)a.N * (CI_H * CI_H * CI_P * 2) >= 0xffff0000ull) return hipErrorInvalidValue;                     // 32-bit offsets, out-of-range marker
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!n_cu_dev[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_img3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CI_LDS);
        if (e != hipSuccess) return e;
        n_cu_dev[dev] = n;
    }
    // half images in groups of 16 (8 images: both halves of an image on one XCD); grid: one workgroup per CU, a multiple of 16
    const int nhalf16 = (2 * a.N + 15) / 16 * 16;
    int grid = n_cu_dev[dev] / 16 * 16;
    if (grid < 16) grid = 16;
    if (grid > nhalf16) grid = nhalf16;
    ConvImg3Args b = a;
    b.nhalf_pad = nhalf16;
    hipLaunchKernelGGL(conv_img3_kernel, dim3(grid), dim3(256), CI_LDS, st, b);
    return hipGetLastError();
}

AP_NS_END
