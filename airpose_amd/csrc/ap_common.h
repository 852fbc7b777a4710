// Shared device/host helpers for the AirPose gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef uint16_t bf16_t;   // storage type of a bf16 activation / weight

#define AP_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round-to-nearest-even
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, u << 16);
    hi = __builtin_bit_cast(float, u & 0xffff0000u);
}

// Split-bf16 storage ("bf16x2" precision): one value as two bf16, hi = rne(x) in the low half of a 32-bit word and
// lo = rne(x - hi) in the high half, value = hi + lo (16 mantissa bits, same bytes as fp32).  Products of two such
// values run on the bf16 matrix pipe: the 8 bf16 positions of a 16-byte fragment chunk are 4 (hi, lo) pairs, so
//   mfma(w, x)          contracts  hi*hi + lo*lo   per pair
//   mfma(w, rot16(x))   contracts  hi*lo + lo*hi
// i.e. the full product (hi + lo)(hi' + lo') in two v_mfma_f32_16x16x32_bf16 per 4 K elements, fp32 accumulate:
// 4x the rate of the exact-fp32 v_mfma_f32_16x16x4_f32 path at ~2^-17 relative operand error.
struct bsplit_t { uint32_t u; };
__device__ __forceinline__ uint32_t split_pack(float f) {
    const bf16_t hi = f32_to_bf16(f);
    const bf16_t lo = f32_to_bf16(f - bf16_to_f32(hi));
    return (uint32_t)hi | ((uint32_t)lo << 16);
}
__device__ __forceinline__ float split_unpack(uint32_t u) {
    return __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xffff0000u);
}
__device__ __forceinline__ u32x4 split_rot16(const u32x4& v) {
    u32x4 r;
    r.x = (v.x >> 16) | (v.x << 16); r.y = (v.y >> 16) | (v.y << 16);
    r.z = (v.z >> 16) | (v.z << 16); r.w = (v.w >> 16) | (v.w << 16);
    return r;
}
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
template <> __device__ __forceinline__ float elem_load<bsplit_t>(const bsplit_t* p) { return split_unpack(p->u); }
template <typename T> __device__ __forceinline__ void elem_store(T* p, float v);
template <> __device__ __forceinline__ void elem_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void elem_store<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void elem_store<bsplit_t>(bsplit_t* p, float v) { p->u = split_pack(v); }

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
static inline hipError_t ap_current_device(int* dev) {
    hipError_t e = hipGetDevice(dev);
    if (e != hipSuccess) return e;
    return (*dev < 0 || *dev >= AP_MAX_DEVICES) ? hipErrorInvalidDevice : hipSuccess;
}

// host-side bf16 helpers (weights packing)
static inline uint16_t host_f32_to_bf16(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline uint32_t host_split_pack(float f) {
    const uint16_t hi = host_f32_to_bf16(f);
    const float hf = __builtin_bit_cast(float, (uint32_t)hi << 16);
    const uint16_t lo = host_f32_to_bf16(f - hf);
    return (uint32_t)hi | ((uint32_t)lo << 16);
}
