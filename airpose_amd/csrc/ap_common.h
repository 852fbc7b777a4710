// Shared device/host helpers for the AirPose gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit storage type of the throughput kernels is a property of the TRANSLATION UNIT: every kernel source that touches 16-bit
// activations / weights (conv_*.hip, bottleneck2.hip, stem.hip) is compiled twice into the ONE library -- plainly (bf16 storage,
// v_mfma_f32_16x16x32_bf16, namespace k_bf16) and with -DAP_F16 (IEEE fp16 storage, v_mfma_f32_16x16x32_f16, namespace k_f16: same
// MFMA rate, 11 instead of 8 significand bits, 5 exponent bits) -- and api.hip picks the set by the handle's precision
// (AP_PREC_BF16 / AP_PREC_F16).  Two translation units instead of one template parameter on purpose: the hand-counted kernels sit
// at their register budgets, and co-compiled instantiations of one template perturb each other's register allocation.
// In a kernel source, "h16" / bf16_t / bf16x8 / K_BF16 mean "the 16-bit type of this translation unit".
#ifdef AP_F16
#define AP_NS k_f16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
#define AP_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#define AP_CVTPK_ASM "v_cvt_pk_f16_f32"
#else
#define AP_NS k_bf16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define AP_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#define AP_CVTPK_ASM "v_cvt_pk_bf16_f32"
#endif
#define AP_NS_BEGIN namespace AP_NS {
#define AP_NS_END }
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef uint16_t bf16_t;   // storage type of a 16-bit activation / weight (bf16, or fp16 in the -DAP_F16 translation units)

// D = A (16 x 32) B (32 x 16) + C on the matrix pipe, operands in the 16-bit type of this translation unit
__device__ __forceinline__ f32x4 ap_mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef AP_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

#define AP_WAVE 64

#ifdef AP_F16
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round-to-nearest-even
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
#endif
// ReLU that KEEPS a (positive-signed) NaN and +inf: max on the bit pattern as a signed integer -- every value with the sign bit
// set (negative numbers, -0, -inf) becomes +0, everything else is untouched.  One v_max_i32, like fmaxf's one v_max_f32, and the
// same result for every finite input (up to the sign of zero); but fmaxf(NaN, 0) = 0 would swallow the NaN an overflowed (+inf)
// fp16 activation turns into in the next convolution -- and with it the evidence the fp16 range sentinel looks for at the end of
// the trunk (ap_net_set_range_check).  The register epilogues of conv_pair.hip / bottleneck2.hip do the same on packed pairs
// (v_pk_max_i16).
__device__ __forceinline__ float ap_relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// fp16 range sentinel (AP_PREC_F16 only; compiled out of the bf16 translation units).  A stored activation can only leave the fp16
// range where an fp32 result is converted for storage, and every stored activation of the trunk is post-ReLU, i.e. an overflow is
// born as +inf = 0x7c00.  Every epilogue therefore folds the packed dwords it stores into a per-thread running maximum over signed
// 16-bit halves (ONE v_pk_max_i16 per two stored values: +inf is the largest positive pattern below the NaNs, negative patterns
// never win) and, once per thread at the end of the kernel, sets the handle's host-mapped flag if either half reached 0x7c00.
// (Checking only the pooled features at the end of the trunk is not enough: the NaNs an inf turns into downstream carry a set
// sign bit on this hardware and every ReLU clears them -- measured, tests/test_gpu_parity.py::test_f16_activation_overflow_is_reported.)
// Use: `uint32_t rng = 0u;` per thread, ap_rng_note(rng, packed_dword) at every store of packed values, ap_rng_flush(flag, rng)
// at the end of the kernel.  All three compile to nothing in the bf16 translation units.
#ifdef AP_F16
__device__ __forceinline__ void ap_rng_note(uint32_t& m, uint32_t packed) {
    asm("v_pk_max_i16 %0, %0, %1" : "+v"(m) : "v"(packed));
}
// epilogues WITHOUT a ReLU (stand-alone operator, unfused downsample branch) can also overflow to -inf: sign bits masked first
__device__ __forceinline__ void ap_rng_note_signed(uint32_t& m, uint32_t packed) { ap_rng_note(m, packed & 0x7fff7fffu); }
// two packed dwords of NON-NEGATIVE values (post-ReLU) per instruction: gfx950's three-input packed fp16 maximum (for values >= +0
// the same order as the integer maximum above; a NaN propagates and is >= 0x7c00 as well)
__device__ __forceinline__ void ap_rng_note2(uint32_t& m, uint32_t a, uint32_t b) {
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b));
}
// the four dwords of a 16-byte store; relu: wave-uniform
__device__ __forceinline__ void ap_rng_note4(uint32_t& m, uint32_t a, uint32_t b, uint32_t c, uint32_t d, bool relu) {
    if (relu) { ap_rng_note2(m, a, b); ap_rng_note2(m, c, d); }
    else { ap_rng_note_signed(m, a); ap_rng_note_signed(m, b); ap_rng_note_signed(m, c); ap_rng_note_signed(m, d); }
}
__device__ __forceinline__ void ap_rng_flush(int* flag, uint32_t m) {
    if (flag && ((m & 0xffffu) >= 0x7c00u || (m >> 16) >= 0x7c00u))
        __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#else
__device__ __forceinline__ void ap_rng_note(uint32_t&, uint32_t) {}
__device__ __forceinline__ void ap_rng_note_signed(uint32_t&, uint32_t) {}
__device__ __forceinline__ void ap_rng_note2(uint32_t&, uint32_t, uint32_t) {}
__device__ __forceinline__ void ap_rng_note4(uint32_t&, uint32_t, uint32_t, uint32_t, uint32_t, bool) {}
__device__ __forceinline__ void ap_rng_flush(int*, uint32_t) {}
#endif

// Fragment-tiled activation layout [M/16][C/8][16 pixels][8 channels] of 16-bit elements (the t2 / identity / block-output tensors
// between a convolution and the fused pair kernel, conv_pair.hip): offset, in elements, of the 8-channel group ch..ch+7 of pixel m.
// The pair kernel's lane (pixel lr, channel group g4) pieces of a wave instruction then form one contiguous KiB.
__device__ __forceinline__ size_t ap_tiled_off(size_t m, int ch, int C) {
    return ((m >> 4) * (size_t)(C >> 3) + (size_t)(ch >> 3)) * 128 + (m & 15) * 8;
}
// item q of an LDS-staged epilogue (CPR 8-channel chunks per tile row) -> (pixel row px, chunk cc).  NHWC output: consecutive
// threads walk the chunks of a row (contiguous 16-byte pieces); tiled output: consecutive threads walk 16 pixels of one chunk
// (contiguous 16-byte pieces of a 256-byte micro-tile)
__device__ __forceinline__ void ap_epi_item(int q, int cpr, bool tiled, int& px, int& cc) {
    if (tiled) { px = (q & 15) | ((q / (16 * cpr)) << 4); cc = (q >> 4) % cpr; }
    else { px = q / cpr; cc = q - px * cpr; }
}

// two fp32 -> one dword of two bf16, round to nearest even.  One v_cvt_pk_bf16_f32 for the PAIR: written as two (__bf16) casts
// and an or, hipcc emits the same instruction once per VALUE (second source a dummy) plus a shift / or to merge them -- three
// VALU instructions per pair in every epilogue of the trunk instead of one (identical results: it is the same conversion).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifdef AP_PACK_CAST                                          // A/B build: the two-cast form
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#endif
    uint32_t r;
    asm(AP_CVTPK_ASM " %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
#ifdef AP_F16
    lo = bf16_to_f32((bf16_t)(u & 0xffffu));
    hi = bf16_to_f32((bf16_t)(u >> 16));
#else
    lo = __builtin_bit_cast(float, u << 16);
    hi = __builtin_bit_cast(float, u & 0xffff0000u);
#endif
}

// a += lo half, b += hi half of a packed 16-bit pair (residual / identity adds of the epilogues).  fp16 storage: ONE mixed-precision
// FMA per value (v_fma_mix_f32 d = f16(src0.half) * 1.0 + d: the product is exact, so this is the conversion followed by the
// fp32 add, bit for bit) instead of v_cvt_f32_f16 (+ a shift for the high half) and an add: 2 instead of 5 VALU per pair.
#ifdef AP_F16
__device__ __forceinline__ void ap_res_add2(float& a, float& b, uint32_t u) {
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(u));
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b) : "v"(u));
}
#endif

// Split-bf16 storage ("bf16x2" precision): one value as two bf16, hi = rne(x) and lo = rne(x - hi), value = hi + lo
// (16 mantissa bits, the bytes of an fp32).  Layout: PLANAR in groups of 8 consecutive elements of the innermost
// (channel / K) dimension -- 32 bytes = [8 x bf16 hi | 8 x bf16 lo] -- so that the 16-byte chunk a lane feeds to
// v_mfma_f32_16x16x32_bf16 is 8 hi parts or 8 lo parts of the same 8 K elements, and a product of two such values is
//   mfma(w_hi, x_hi) + mfma(w_hi, x_lo) + mfma(w_lo, x_hi)          (lo*lo, 2^-18 relative, is dropped)
// three MFMAs per 8 K elements, fp32 accumulate: 5.3x the rate of the exact-fp32 v_mfma_f32_16x16x4_f32 path at ~2^-17
// relative operand error.  An element is addressed as a 4-byte unit (bsplit_t) and only ever touched in aligned groups of 8.
struct bsplit_t { uint32_t u; };
__device__ __forceinline__ void split8_pack(const float (&v)[8], u32x4& hi, u32x4& lo) {
    bf16_t h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = f32_to_bf16(v[i]);
        l[i] = f32_to_bf16(v[i] - bf16_to_f32(h[i]));
    }
    hi.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); hi.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    hi.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); hi.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
    lo.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16); lo.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
    lo.z = (uint32_t)l[4] | ((uint32_t)l[5] << 16); lo.w = (uint32_t)l[6] | ((uint32_t)l[7] << 16);
}
__device__ __forceinline__ void split8_unpack(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
    const uint32_t hh[4] = {hi.x, hi.y, hi.z, hi.w}, ll[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float h0, h1, l0, l1;
        unpack_bf16x2(hh[i], h0, h1);
        unpack_bf16x2(ll[i], l0, l1);
        v[2 * i] = h0 + l0;
        v[2 * i + 1] = h1 + l1;
    }
}
// bf16 positions (in units of 2 bytes from the start of a row of 4-byte elements) of element i
__device__ __forceinline__ int split_hi_pos(int i) { return (i >> 3) * 16 + (i & 7); }
__device__ __forceinline__ int split_lo_pos(int i) { return (i >> 3) * 16 + 8 + (i & 7); }
// storage kinds (= the AP_PREC_* values of include/airpose_hip.h)
constexpr int K_F32 = 0, K_BF16 = 1, K_SPLIT = 2;
template <typename T> struct ElemKind;
template <> struct ElemKind<float>    { static constexpr int KIND = K_F32,   EPC = 4; };
template <> struct ElemKind<bf16_t>   { static constexpr int KIND = K_BF16,  EPC = 8; };
template <> struct ElemKind<bsplit_t> { static constexpr int KIND = K_SPLIT, EPC = 4; };
// one stored element <-> float
template <typename T> __device__ __forceinline__ float elem_load(const T* p);
template <> __device__ __forceinline__ float elem_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float elem_load<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void elem_store(T* p, float v);
template <> __device__ __forceinline__ void elem_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void elem_store<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// Bijective XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): give every XCD a
// contiguous run of logical tile ids so neighbouring tiles (which share an operand panel) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + i;
}

// hipFuncSetAttribute and the CU count are per DEVICE: launch-side caches are keyed by the current device id
// (two nets of one process on two GPUs, e.g. copenet_sep, each get their >64 KiB dynamic-LDS opt-in).
#define AP_MAX_DEVICES 16
#define AP_STEM_WLD 240                                      // row stride (elements) of the stem's packed MFMA weights (api.hip packs, stem.hip reads)
static inline hipError_t ap_current_device(int* dev) {
    hipError_t e = hipGetDevice(dev);
    if (e != hipSuccess) return e;
    return (*dev < 0 || *dev >= AP_MAX_DEVICES) ? hipErrorInvalidDevice : hipSuccess;
}

// host-side 16-bit helpers (weight packing; api.hip packs for either storage type)
static inline float host_bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
static inline uint16_t host_f32_to_bf16(float f) {                                                           // RNE
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// fp16 (RNE); a finite value that leaves the fp16 range sets *overflow (ap_net_finalize refuses such a checkpoint in AP_PREC_F16)
static inline uint16_t host_f32_to_f16(float f, bool* overflow) {
    const uint16_t h = __builtin_bit_cast(uint16_t, (_Float16)f);
    if ((h & 0x7fffu) == 0x7c00u && f == f && f - f == 0.f) *overflow = true;
    return h;
}
static inline float host_f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
static inline void host_split_parts(float f, uint16_t* hi, uint16_t* lo) {
    *hi = host_f32_to_bf16(f);
    *lo = host_f32_to_bf16(f - host_bf16_to_f32(*hi));
}
// n floats (n % 8 == 0, rows are multiples of 8) -> planar split-bf16: dst holds 2n uint16, group g at [16g, 16g + 16)
static inline void host_split_pack_planar(const float* src, size_t n, uint16_t* dst) {
    for (size_t i = 0; i < n; ++i)
        host_split_parts(src[i], &dst[(i >> 3) * 16 + (i & 7)], &dst[(i >> 3) * 16 + 8 + (i & 7)]);
}
