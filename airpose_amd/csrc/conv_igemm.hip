// Implicit-GEMM convolution / GEMM for gfx950 (CDNA4), NHWC activations, K-contiguous weights.
//
// Replaces every nn.Conv2d + BatchNorm2d (+ residual) + ReLU instance of the ResNet-50 trunk
// (reference: copenet/src/copenet/models/model_copenet.py:27-47 Bottleneck.forward, :161-176
// forward_feat_ext) and, in its fp32 instantiation, the dense contractions of the regressor
// (model_copenet.py:185-202) and of the SMPL-X blend shapes.
//
//   y[m][co] = act( (sum_{r,s,ci} x[n, ho*st-pad+r, wo*st-pad+s, ci] * w[co][r][s][ci]) * scale[co]
//                   + shift[co] (+ res[m][co]) )
//
// Tiling: one workgroup = 256 threads = 4 wave64 computing a BM x BN (pixels x channels) tile; the
// K loop walks (tap, channel-chunk) steps of 128 bytes per row (64 bf16 / 32 fp32).  Both operand
// tiles are staged global -> registers -> LDS (predicated, so the 3x3 halo and ragged M are plain
// zero fills), double buffered, one barrier per K step, the next tile's global loads in flight under
// the current tile's MFMAs.  LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by
// (row & 7): conflict-free for the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups.
// MFMA: weights are the A operand, activations the B operand, so a lane ends up with 4 consecutive
// output channels of one pixel (C/D layout: col = lane & 15, row = 4*(lane >> 4) + reg).
//   bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//   fp32: v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), four per 16-byte chunk.
// Epilogue: scale/shift in fp32, tile goes through LDS as fp32 so the residual add, ReLU and the
// single rounding to the storage type happen on whole 16-byte row segments (coalesced load/store).
#include "ap_common.h"
#include "kernels.h"

AP_NS_BEGIN

namespace {

template <typename T> using Elem = ElemKind<T>;

template <typename T, int FM, int FN>
__device__ __forceinline__ void mma_chunk(const u32x4 (&xf)[FM], const u32x4 (&wf)[FN], f32x4 (&acc)[FM][FN]) {
    if constexpr (Elem<T>::KIND == K_BF16) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fm][fn] = ap_mfma16(
                    __builtin_bit_cast(bf16x8, wf[fn]), __builtin_bit_cast(bf16x8, xf[fm]), acc[fm][fn]);
    } else {
        // lane group g holds k = 4g..4g+3 of a 16-deep slab; MFMA t contracts {4g'+t : g'=0..3}
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

// split-bf16 (planar): hi and lo fragments of the same 8 K elements per lane; w_hi*x_hi, w_hi*x_lo, w_lo*x_hi, each pass over
// all accumulators before the next so that the MFMAs of one accumulator are FM*FN instructions apart
template <int FM, int FN>
__device__ __forceinline__ void mma_split3(const u32x4 (&xh)[FM], const u32x4 (&xl)[FM], const u32x4 (&wh)[FN],
                                           const u32x4 (&wl)[FN], f32x4 (&acc)[FM][FN]) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fm][fn] = ap_mfma16(
                    __builtin_bit_cast(bf16x8, pass == 2 ? wl[fn] : wh[fn]),
                    __builtin_bit_cast(bf16x8, pass == 1 ? xl[fm] : xh[fm]), acc[fm][fn]);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int FM = BM / WAVES_M / 16, FN = BN / WAVES_N / 16;
    constexpr int XR = BM / 32, WR = BN / 32;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int CLD = BN + 4;                 // fp32 epilogue tile row stride (floats)
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int bm = tile / p.ntiles, bn = tile % p.ntiles;

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wg = (const T*)p.w;

    // ---------------------------------------------------------------- loader state
    const int lc = tid & 7, lrow0 = tid >> 3;
    const T* xptr[XR];
    int hi0[XR], wi0[XR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = bm * BM + lrow0 + 32 * i;
        if (m < p.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            xptr[i] = xg + ((size_t)n * p.H * p.W + (ptrdiff_t)hi0[i] * p.W + wi0[i]) * p.ldx + lc * EPC;
        } else {
            hi0[i] = -0x40000000;               // never passes the bounds test
            wi0[i] = 0;
            xptr[i] = xg;
        }
    }
    const T* wptr[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) wptr[j] = wg + (size_t)(bn * BN + lrow0 + 32 * j) * p.wld + lc * EPC;
    // second K segment (downsample branch): row pointers into x2 at the strided pixel
    const T* x2ptr[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = bm * BM + lrow0 + 32 * i;
        x2ptr[i] = nullptr;
        if (p.x2 && m < p.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            x2ptr[i] = (const T*)p.x2 + (((size_t)n * p.H2 + (size_t)ho * p.stride2) * p.W2 + (size_t)wo * p.stride2) * p.ldx2 + lc * EPC;
        }
    }

    u32x4 xs[XR], ws[WR];
    const int cpb = p.Cin / BK;                 // channel chunks per tap
    const int cpb2 = p.x2 ? p.Cin2 / BK : 0;    // chunks of the second K segment
    const int KT = p.KH * p.KW * cpb + cpb2;
    int r = 0, s = 0, cb = 0;                   // tap / channel-chunk of the tile being LOADED

    auto load_tile = [&](int kt) {
        if (r < p.KH) {
            const ptrdiff_t xoff = ((ptrdiff_t)r * p.W + s) * p.ldx + cb * BK;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                // unconditional load from a safe address + select: no branch, no scratch
                const bool ok = (unsigned)(hi0[i] + r) < (unsigned)p.H && (unsigned)(wi0[i] + s) < (unsigned)p.W;
                const u32x4 v = *(const u32x4*)(ok ? xptr[i] + xoff : xg);
                xs[i].x = ok ? v.x : 0u; xs[i].y = ok ? v.y : 0u; xs[i].z = ok ? v.z : 0u; xs[i].w = ok ? v.w : 0u;
            }
        } else {                                // second segment: 1x1 on x2
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const bool ok = x2ptr[i] != nullptr;
                const u32x4 v = *(const u32x4*)(ok ? x2ptr[i] + cb * BK : xg);
                xs[i].x = ok ? v.x : 0u; xs[i].y = ok ? v.y : 0u; xs[i].z = ok ? v.z : 0u; xs[i].w = ok ? v.w : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < WR; ++j) ws[j] = *(const u32x4*)(wptr[j] + (size_t)kt * BK);
        if (++cb == (r < p.KH ? cpb : cpb2)) { cb = 0; if (r < p.KH && ++s == p.KW) { s = 0; ++r; } }
    };
    const int st_off = lrow0 * 128 + ((lc ^ (lrow0 & 7)) << 4);
    auto store_tile = [&](int buf) {
        unsigned char* xb = smem + buf * STAGE + st_off;
        unsigned char* wb = xb + BM * 128;
#pragma unroll
        for (int i = 0; i < XR; ++i) *(u32x4*)(xb + i * 32 * 128) = xs[i];
#pragma unroll
        for (int j = 0; j < WR; ++j) *(u32x4*)(wb + j * 32 * 128) = ws[j];
    };

    // ---------------------------------------------------------------- MFMA state
    const int lr = lane & 15, g = lane >> 4;
    // chunk of a 128-byte row read by lane group g: bf16 / fp32: g then 4 + g (two 32- / 16-deep halves);
    // split-bf16: 2g (hi parts of elements 8g .. 8g+7) and 2g + 1 (their lo parts)
    constexpr bool SPLIT = Elem<T>::KIND == K_SPLIT;
    const int c0 = SPLIT ? 2 * g : g, c1 = SPLIT ? 2 * g + 1 : 4 + g;
    const int sw0 = ((c0 ^ (lr & 7)) << 4), sw1 = ((c1 ^ (lr & 7)) << 4);
    const int xfrag = (wm * (BM / WAVES_M) + lr) * 128;
    const int wfrag = BM * 128 + (wn * (BN / WAVES_N) + lr) * 128;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);
        const unsigned char* sb = smem + (kt & 1) * STAGE;
        u32x4 xf[FM], wf[FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) xf[fm] = *(const u32x4*)(sb + xfrag + fm * 16 * 128 + sw0);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) wf[fn] = *(const u32x4*)(sb + wfrag + fn * 16 * 128 + sw0);
        if constexpr (SPLIT) {
            u32x4 xl[FM], wl[FN];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xl[fm] = *(const u32x4*)(sb + xfrag + fm * 16 * 128 + sw1);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) wl[fn] = *(const u32x4*)(sb + wfrag + fn * 16 * 128 + sw1);
            mma_split3<FM, FN>(xf, xl, wf, wl, acc);
        } else {
            mma_chunk<T, FM, FN>(xf, wf, acc);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = *(const u32x4*)(sb + xfrag + fm * 16 * 128 + sw1);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) wf[fn] = *(const u32x4*)(sb + wfrag + fn * 16 * 128 + sw1);
            mma_chunk<T, FM, FN>(xf, wf, acc);
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    float* ct = (float*)smem;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int chl = wn * (BN / WAVES_N) + fn * 16 + g * 4;      // channel within the tile
        const int ch = bn * BN + chl;
        const float4 sc = *(const float4*)(p.scale + ch);
        const float4 sh = *(const float4*)(p.shift + ch);
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
    __syncthreads();
    // epilogue items: 8 channels for bf16 (16 B) and split-bf16 (32 B = 8 hi | 8 lo), 4 channels for fp32 (16 B)
    constexpr int KIND = Elem<T>::KIND;
    constexpr int EPO = KIND == K_F32 ? 4 : 8;
    constexpr int CPR = BN / EPO;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    uint32_t rng = 0u;                                       // fp16 range sentinel (ap_common.h); nothing in the other kinds
    const uint32_t smask = p.relu ? 0xffffffffu : 0x7fff7fffu;
    for (int q = tid; q < BM * CPR; q += 256) {
        int px, cc;
        ap_epi_item(q, CPR, KIND == K_BF16 && p.y_tiled != 0, px, cc);
        const int m = bm * BM + px, ch = bn * BN + cc * EPO;
        if (m >= p.M || ch >= p.Cout) continue;
        const float* src = ct + px * CLD + cc * EPO;
        if constexpr (KIND == K_BF16) {
            float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
            if (rg) {
                const u32x4 rv = *(const u32x4*)(rg + (size_t)m * p.ldr + ch);
                float lo, hi;
                unpack_bf16x2(rv.x, lo, hi); a.x += lo; a.y += hi;
                unpack_bf16x2(rv.y, lo, hi); a.z += lo; a.w += hi;
                unpack_bf16x2(rv.z, lo, hi); b.x += lo; b.y += hi;
                unpack_bf16x2(rv.w, lo, hi); b.z += lo; b.w += hi;
            }
            if (p.relu) {
                a.x = ap_relu(a.x); a.y = ap_relu(a.y); a.z = ap_relu(a.z); a.w = ap_relu(a.w);
                b.x = ap_relu(b.x); b.y = ap_relu(b.y); b.z = ap_relu(b.z); b.w = ap_relu(b.w);
            }
            u32x4 o;
            o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w);
            o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
            ap_rng_note(rng, o.x & smask); ap_rng_note(rng, o.y & smask); ap_rng_note(rng, o.z & smask); ap_rng_note(rng, o.w & smask);
            *(u32x4*)(yg + (p.y_tiled ? ap_tiled_off((size_t)m, ch, p.Cout) : (size_t)m * p.ldy + ch)) = o;
        } else if constexpr (KIND == K_SPLIT) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[e];
            const bool second = ch + 4 < p.Cout;             // (fp32 output: Cout may end on a multiple of 4)
            if (rg) {
                if (p.out_f32) {
                    const float* rp = (const float*)rg + (size_t)m * p.ldr + ch;
                    const float4 r0 = *(const float4*)rp, r1 = second ? *(const float4*)(rp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                } else {
                    const u32x4* rp = (const u32x4*)(rg + (size_t)m * p.ldr + ch);
                    float r[8];
                    split8_unpack(rp[0], rp[1], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
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
        } else {
            float4 a = *(const float4*)src;
            if (rg) {
                const float4 rv = *(const float4*)(rg + (size_t)m * p.ldr + ch);
                a.x += rv.x; a.y += rv.y; a.z += rv.z; a.w += rv.w;
            }
            if (p.relu) { a.x = ap_relu(a.x); a.y = ap_relu(a.y); a.z = ap_relu(a.z); a.w = ap_relu(a.w); }
            *(float4*)(yg + (size_t)m * p.ldy + ch) = a;
        }
    }
    if constexpr (KIND == K_BF16) ap_rng_flush(p.range_flag, rng);
}

template <int BM, int BN>
constexpr int lds_bytes() {
    constexpr int stage = 2 * (BM + BN) * 128, epi = BM * (BN + 4) * 4;
    return stage > epi ? stage : epi;
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t launch_cfg(ConvArgs a, hipStream_t st) {
    static bool attr_set[AP_MAX_DEVICES] = {};
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN>;
    constexpr int lds = lds_bytes<BM, BN>();
    int dev = 0;
    hipError_t e = ap_current_device(&dev);
    if (e != hipSuccess) return e;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_T(const ConvArgs& a, hipStream_t st) {
    if (a.Cout <= 64) {
        if (a.M <= 64 * 192) return launch_cfg<T, 64, 64, 2, 2>(a, st);
        return launch_cfg<T, 128, 64, 2, 2>(a, st);
    }
    // small-M problems (regressor, late layers at tiny batch): smaller tiles fill more CUs
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 256) return launch_cfg<T, 64, 64, 2, 2>(a, st);
    return launch_cfg<T, 128, 128, 2, 2>(a, st);
}

}  // namespace

// Rows of the packed weight matrix must be padded (zero rows) to this multiple.
int ap_conv_cout_pad(void) { return 128; }

hipError_t ap_launch_conv(const ConvArgs& a, int kind, hipStream_t st) {
    if (kind == K_BF16) return launch_T<bf16_t>(a, st);
#ifndef AP_F16                                               // (the fp16 set carries the 16-bit kind only)
    if (kind == K_SPLIT) return launch_T<bsplit_t>(a, st);
    if (kind == K_F32) return launch_T<float>(a, st);
#endif
    return hipErrorInvalidValue;
}

AP_NS_END
