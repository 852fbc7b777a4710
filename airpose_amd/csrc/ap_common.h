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
