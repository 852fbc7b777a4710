// Pointwise (1x1, stride 1) convolution on a LEAN workgroup: three workgroups per CU instead of two (bf16, gfx950).
//
// Same 128x128 tile, 8 waves (64x32 per wave), same MFMA sequence per output element as conv_pipe.hip's <bf16,128,128,2,4,2>
// (bit-identical results; see there for the reference lines it replaces).  What the probes of round 2 say about the
// pointwise layers: a K step costs the round trip of its operand tile, the round trip is about twice the MFMA
// time that should cover it, a deeper ring does not pay for its LDS, and what does help is more independent workgroups per
// CU.  At 122 registers and 68 KiB per workgroup the ring kernel fits two.  This kernel is cut to fit three:
//   * K steps of 32 channels (one MFMA per accumulator): an operand tile is 16 KiB, a 3-slot ring 48 KiB;
//   * no fragment double-buffering (6 fragment registers x 4 instead of 12 x 4), 32-bit DMA offsets: <= 80 VGPRs,
//     six waves per SIMD; the fragment-read latency a wave no longer hides itself is covered by the other five;
//   * epilogue staged through LDS in two halves of 64 channels (35 KiB instead of 68).
// LDS image of a tile: 16 one-KiB pieces (8 of activations, 8 of weights), a piece = 16 rows (pixels / channels) x 64 B;
// the 16-byte chunk c of row p sits at slot 4p + (c ^ h[p >> 2]), h = {0, 3, 2, 1}: a ds_read_b128 group (16 lanes = chunk a of
// rows 0-3 and 12-15 plus chunk a^1 of rows 4-11) then covers the 16 bank quads exactly once.  The swizzle lives on the DMA
// SOURCE address (LDS-DMA writes lane-linearly).
//
// POOL variant (the last convolution of the trunk, layer4.2 conv3: model_copenet.py:38-47 followed by AvgPool2d(7) + view,
// :173-175): the 231 MB block output is never written.  A workgroup owns a SUPER-TILE of 5 images = 245 pixel rows x 128 channels,
// computed as two 128-row sub-tiles through the same K loop; the epilogue (BatchNorm, + identity, ReLU, rounding to the 16-bit
// storage type -- the values the stand-alone path would have stored) reduces the pixels of every image per channel IN THE ORDER OF
// avgpool_kernel (stem.hip): eight partitions p, p + 8, ... summed serially, then the partitions in order, / 49 -- so the features
// are bit-identical to conv + avgpool_kernel, for every batch size and position of the image in the batch.  The image that
// straddles the two sub-tiles (rows 98..146) carries its eight partial sums in LDS from the first sub-tile to the second.
#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

constexpr int BM = 128, BN = 128, NT = 512, FM = 4, FN = 2, WAVES_N = 4;
constexpr int SLOTS = 3, TILE_BYTES = 16384, A_BYTES = 8192;
constexpr int HALF = 64, CLD = HALF + 4;                   // epilogue stage: 128 x 68 floats per half
constexpr int LDS_BYTES = SLOTS * TILE_BYTES;               // 49152 (>= 128 * 68 * 4 = 34816)
static_assert(BM * CLD * 4 <= LDS_BYTES, "the epilogue stage reuses the ring");
// POOL variant: 5 images (49 pixels each) per super-tile; partial sums of a sub-tile's <= 3 image segments behind the stage
// (inside the ring), the straddling image's carried partials behind the ring
constexpr int POOL_IMGS = 5, POOL_PIX = 49, POOL_ROWS = POOL_IMGS * POOL_PIX;       // 245 of 2 x 128 rows
constexpr int PSUM_OFF = BM * CLD * 4, PSUM_BYTES = 3 * 8 * HALF * 4;                // [seg 3][part 8][64] floats
constexpr int CARRY_BYTES = 2 * 8 * HALF * 4;                                        // [half 2][part 8][64] floats
static_assert(PSUM_OFF + PSUM_BYTES <= LDS_BYTES, "partial sums fit behind the stage");
static_assert(3 * (LDS_BYTES + CARRY_BYTES) <= 160 * 1024, "three workgroups per CU");

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
__device__ __forceinline__ void mfma_acc(f32x4& c, const u32x4& w, const u32x4& x) {
    asm volatile(AP_MFMA16_ASM " %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}

template <bool POOL>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6))) conv_lean_kernel(const ConvArgs p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile % p.ntiles;      // POOL: bm = super-tile (5 images)
    const int KT = p.Cin / 32;
    const unsigned char* zg = (const unsigned char*)p.zero;
    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    asm volatile("" ::"s"(zg), "s"(xg), "s"(wg));           // kernel-argument loads complete here (see conv_slab.hip)

    constexpr int NSUB = POOL ? 2 : 1;
    const int row0 = POOL ? bm * POOL_ROWS : bm * BM;         // first pixel row of the (super-)tile
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    for (int sub = 0; sub < NSUB; ++sub) {
    if (POOL && sub) __syncthreads();                        // the previous sub-tile's stage / partial sums are read: the ring is free
    // per-lane state is derived from an opaque copy of the thread id inside the sub-tile loop: it is recomputed for the second
    // sub-tile instead of being kept alive across the first one's epilogue (80 registers: three workgroups per CU)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    // ---------------------------------------------------------------- DMA: wave w carries activation piece w (pixels
    // 16w .. 16w+15 of the tile) and weight piece w (channels 16w .. 16w+15) of every K step; lane l fills slot l of the piece
    const int prow = lane >> 2;                              // row of the piece this lane's slot belongs to
    const int pchunk = (lane & 3) ^ ((0x1230 >> ((prow >> 2) * 4)) & 3);   // h = {0, 3, 2, 1}
    // operand rows of this sub-tile (POOL: rows past the 245th of the super-tile are zero rows)
    const int rl_dma = sub * BM + wave * 16 + prow, m = row0 + rl_dma;
    const bool a_ok = m < p.M && (!POOL || rl_dma < POOL_ROWS);
    uint32_t aoff = a_ok ? (uint32_t)(((size_t)m * p.ldx + pchunk * 8) * sizeof(T)) : 0u;
    uint32_t woff = (uint32_t)(((size_t)(bn * BN + wave * 16 + prow) * p.wld + pchunk * 8) * sizeof(T));   // rows padded to 128: valid
    auto issue_tile = [&](int slot) {
        const unsigned char* as = a_ok ? xg + aoff : zg;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)as,
                                         (__attribute__((address_space(3))) void*)(smem + slot * TILE_BYTES + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + woff),
                                         (__attribute__((address_space(3))) void*)(smem + slot * TILE_BYTES + A_BYTES + wave * 1024),
                                         16, 0, 0);
        aoff += 64u; woff += 64u;
        asm volatile("" : "+v"(aoff), "+v"(woff));           // keep the offsets 32-bit (uniform base + VGPR offset)
    };

    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g4 = lane >> 4;
    const uint32_t fslot = (uint32_t)(lr * 4 + (g4 ^ ((0x1230 >> ((lr >> 2) * 4)) & 3))) * 16u;
    const uint32_t xa = lds0 + wm * 4096 + fslot;            // fragment fm at + fm * 1024
    const uint32_t wa = lds0 + A_BYTES + wn * 2048 + fslot;  // fragment fn at + fn * 1024
    f32x4 acc[FM][FN];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---------------------------------------------------------------- K loop: one barrier per 32-deep step, two tiles of
    // look-ahead.  Step k: fragments of tile k (slot k % 3) -> tile k+2 into the slot tile k-1 has left -> 8 MFMAs ->
    // wait for tile k+1 (tile k+2 stays in flight) -> barrier.
    issue_tile(0);
    if (KT > 1) issue_tile(1);
    if (KT > 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int k = 0; k < KT; ++k) {
        const uint32_t so = slot * TILE_BYTES;
        u32x4 xf[FM], wf[FN];
        xf[0] = lds_read_b128<0>(xa + so);
        xf[1] = lds_read_b128<1024>(xa + so);
        xf[2] = lds_read_b128<2048>(xa + so);
        xf[3] = lds_read_b128<3072>(xa + so);
        wf[0] = lds_read_b128<0>(wa + so);
        wf[1] = lds_read_b128<1024>(wa + so);
        if (k + 2 < KT) issue_tile(slot == 0 ? 2 : slot - 1);   // (k + 2) % 3
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkmcnt<0>();
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) mfma_acc(acc[fm][fn], wf[fn], xf[fm]);
        if (k + 1 < KT) {
            if (k + 2 < KT) wait_vmcnt<2>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMA results settle before the epilogue's VALU reads them

    // ---------------------------------------------------------------- epilogue in two halves of 64 channels (as
    // conv_pipe.hip: fp32 stage in LDS, BatchNorm + residual + ReLU, 16-byte coalesced stores)
    float* ct = (float*)smem;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    constexpr int CPR = HALF / 8, NIT = BM * CPR / NT;       // 8 chunks of 8 channels per row and half, 2 per thread
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the bf16 set
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                     // ring reads (h = 0) / previous half's stage reads (h = 1) are done
        if ((wn >> 1) == h) {
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int chl = (wn & 1) * 32 + fn * 16 + g4 * 4;        // within the half
                const int ch = bn * BN + h * HALF + chl;
                const float4 sc = *(const float4*)(p.scale + ch);
                const float4 sh = *(const float4*)(p.shift + ch);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const int px = wm * 64 + fm * 16 + lr;
                    float4 v;
                    v.x = acc[fm][fn][0] * sc.x + sh.x;
                    v.y = acc[fm][fn][1] * sc.y + sh.y;
                    v.z = acc[fm][fn][2] * sc.z + sh.z;
                    v.w = acc[fm][fn][3] * sc.w + sh.w;
                    *(float4*)(ct + px * CLD + chl) = v;
                }
            }
        }
        __syncthreads();
        u32x4 rv[NIT];
        if (rg) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * NT, px = q / CPR, cc = q - px * CPR;
                const int rl = sub * BM + px, mm = row0 + rl, ch = bn * BN + h * HALF + cc * 8;
                const bool ok = mm < p.M && ch < p.Cout && (!POOL || rl < POOL_ROWS);
                rv[it] = *(const u32x4*)(ok ? rg + (size_t)mm * p.ldr + ch : (const T*)p.zero);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = tid + it * NT, px = q / CPR, cc = q - px * CPR;
            const int rl = sub * BM + px, mm = row0 + rl, ch = bn * BN + h * HALF + cc * 8;
            if (!POOL && (mm >= p.M || ch >= p.Cout)) continue;
            float* sp = ct + px * CLD + cc * 8;
            float4 a = *(const float4*)sp, b = *(const float4*)(sp + 4);
            if (rg) {
                float lo, hi;
#ifdef AP_F16
                (void)lo; (void)hi;
                ap_res_add2(a.x, a.y, rv[it][0]); ap_res_add2(a.z, a.w, rv[it][1]);
                ap_res_add2(b.x, b.y, rv[it][2]); ap_res_add2(b.z, b.w, rv[it][3]);
#else
                unpack_bf16x2(rv[it][0], lo, hi); a.x += lo; a.y += hi;
                unpack_bf16x2(rv[it][1], lo, hi); a.z += lo; a.w += hi;
                unpack_bf16x2(rv[it][2], lo, hi); b.x += lo; b.y += hi;
                unpack_bf16x2(rv[it][3], lo, hi); b.z += lo; b.w += hi;
#endif
            }
            if (p.relu) {
                a.x = ap_relu(a.x); a.y = ap_relu(a.y); a.z = ap_relu(a.z); a.w = ap_relu(a.w);
                b.x = ap_relu(b.x); b.y = ap_relu(b.y); b.z = ap_relu(b.z); b.w = ap_relu(b.w);
            }
            u32x4 o;
            o[0] = pack_bf16x2(a.x, a.y); o[1] = pack_bf16x2(a.z, a.w);
            o[2] = pack_bf16x2(b.x, b.y); o[3] = pack_bf16x2(b.z, b.w);
            ap_rng_note4(rng, o[0], o[1], o[2], o[3], p.relu != 0);
            if constexpr (!POOL) {
                *(u32x4*)(yg + (size_t)mm * p.ldy + ch) = o;
            } else {
                // the values the stand-alone path would have stored (rounded to the 16-bit type), back into this thread's own
                // chunk of the stage: the pooling pass below reads them per channel
                unpack_bf16x2(o[0], a.x, a.y); unpack_bf16x2(o[1], a.z, a.w);
                unpack_bf16x2(o[2], b.x, b.y); unpack_bf16x2(o[3], b.z, b.w);
                *(float4*)sp = a; *(float4*)(sp + 4) = b;
            }
        }
        if constexpr (POOL) {
            // ---- pooling of this half: thread = (partition k, channel c); the sub-tile's image segments in turn.
            // super-tile rows: image i = rows 49 i .. 49 i + 48.  sub 0 holds images 0, 1 and pixels 0..29 of image 2;
            // sub 1 holds pixels 30..48 of image 2 (local rows 0..18), image 3 (19..67), image 4 (68..116)
            __syncthreads();
            float* psum = (float*)(smem + PSUM_OFF);                     // [seg][k][c]
            float* carry = (float*)(smem + LDS_BYTES) + h * 8 * HALF;    // [k][c]: image 2's partials, sub 0 -> sub 1
            const int k = tid >> 6, c = tid & 63;
#pragma unroll
            for (int seg = 0; seg < 3; ++seg) {
                // local row of the segment's first pixel, first / last pixel of the image it holds
                const int r0 = sub == 0 ? seg * POOL_PIX : (seg == 0 ? 0 : 19 + (seg - 1) * POOL_PIX);
                const int p0 = (sub == 1 && seg == 0) ? 30 : 0;
                const int p1 = (sub == 0 && seg == 2) ? 30 : POOL_PIX;
                float sacc = (sub == 1 && seg == 0) ? carry[k * HALF + c] : 0.f;
                // pixels k, k + 8, ... of the image, ascending: the order of avgpool_kernel's partition k
                int px0 = k + ((p0 - k + 7) & ~7);                       // first pixel >= p0 congruent to k (p0 = 0: k itself)
                if (p0 == 0) px0 = k;
                for (int px = px0; px < p1; px += 8) sacc += ct[(r0 + px - p0) * CLD + c];
                if (sub == 0 && seg == 2) carry[k * HALF + c] = sacc;
                else psum[(seg * 8 + k) * HALF + c] = sacc;
            }
            __syncthreads();
            if (tid < 3 * HALF) {
                const int seg = tid >> 6;
                const int img_l = sub == 0 ? seg : seg + 2;              // sub 0: images 0, 1 (seg 2 is carried); sub 1: 2, 3, 4
                const int img = bm * POOL_IMGS + img_l, ch = bn * BN + h * HALF + c;
                if (!(sub == 0 && seg == 2) && img < p.N && ch < p.Cout) {
                    float t = psum[(seg * 8) * HALF + c];
#pragma unroll
                    for (int kk = 1; kk < 8; ++kk) t += psum[(seg * 8 + kk) * HALF + c];
                    p.pool_out[(size_t)img * p.Cout + ch] = t / 49.0f;
                }
            }
        }
    }
    ap_rng_flush(p.range_flag, rng);
    }   // sub-tile
}

}  // namespace

// pointwise, stride 1, bf16, no second K segment, tensors addressable with 32-bit byte offsets
bool ap_conv_lean_supported(const ConvArgs& a, int kind) {
    return kind == K_BF16 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.x2 && a.Cin % 32 == 0 && a.Cin >= 32 &&
           (long long)a.M * a.ldx * 2 < 0xffffffffll && (long long)a.wld * 2 * ((a.Cout + 127) / 128 * 128) < 0xffffffffll;
}

hipError_t ap_launch_conv_lean(ConvArgs a, hipStream_t st) {
    if (!a.zero || !ap_conv_lean_supported(a, K_BF16)) return hipErrorInvalidValue;
    a.ntiles = (a.Cout + BN - 1) / BN;
    if (a.pool_out) {
        // conv + BatchNorm + identity + ReLU + AvgPool2d(7) + view: 7 x 7 images, every pixel of an image in one super-tile
        if (a.Ho * a.Wo != POOL_PIX || a.M != a.N * POOL_PIX || !a.res || !a.relu) return hipErrorInvalidValue;
        static bool attr_set[AP_MAX_DEVICES] = {};
        int dev = 0;
        hipError_t e = ap_current_device(&dev);
        if (e != hipSuccess) return e;
        if (!attr_set[dev]) {
            e = hipFuncSetAttribute((const void*)conv_lean_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + CARRY_BYTES);
            if (e != hipSuccess) return e;
            attr_set[dev] = true;
        }
        a.mtiles = (a.N + POOL_IMGS - 1) / POOL_IMGS;
        hipLaunchKernelGGL(conv_lean_kernel<true>, dim3(a.mtiles * a.ntiles), dim3(NT), LDS_BYTES + CARRY_BYTES, st, a);
        return hipGetLastError();
    }
    a.mtiles = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL(conv_lean_kernel<false>, dim3(a.mtiles * a.ntiles), dim3(NT), LDS_BYTES, st, a);
    return hipGetLastError();
}

AP_NS_END
