// Shared device/host helpers for the frameino_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/frameino_hip.h"


typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define FINO_LDS __attribute__((address_space(3)))
#define FINO_GLB __attribute__((address_space(1)))

// ---- dtype tags -----------------------------------------------------------------------------
struct BF16 {
    typedef bf16x8_t vec8;
    typedef __bf16 scalar;
    static constexpr int kId = FINO_BF16;
    static __device__ __forceinline__ float to_f32(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
        return __builtin_bit_cast(uint16_t, b);
    }
    static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
struct F16 {
    typedef f16x8_t vec8;
    typedef _Float16 scalar;
    static constexpr int kId = FINO_F16;
    static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = T::to_f32((uint16_t)(w[i] & 0xffffu));
        f[2 * i + 1] = T::to_f32((uint16_t)(w[i] >> 16));
    }
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)T::from_f32(f[2 * i]) | ((uint32_t)T::from_f32(f[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename T>
__device__ __forceinline__ float round_to(float f) { return T::to_f32(T::from_f32(f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- host side ------------------------------------------------------------------------------
void fino_set_error(const char* fmt, ...);
#define FINO_CHECK(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            fino_set_error(__VA_ARGS__);       \
            return (code);                     \
        }                                      \
    } while (0)
#define FINO_LAUNCH_CHECK()                                                        \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            fino_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return FINO_ERR_LAUNCH;                                                \
        }                                                                          \
    } while (0)

// Per-device one-time setup (a process may drive several GPUs, possibly from several host threads; hipFuncSetAttribute
// applies to the current device only).  The guarded action is idempotent, so a benign double initialisation is harmless.
constexpr int kFinoMaxDevices = 64;
static inline int fino_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kFinoMaxDevices) dev = 0;
    return dev;
}
struct FinoPerDeviceOnce {
    std::atomic<int> done[kFinoMaxDevices];
};
static inline int fino_max_smem_once(FinoPerDeviceOnce& once, const void* fn, int bytes, const char* who) {
    const int dev = fino_current_device();
    if (once.done[dev].load(std::memory_order_acquire)) return FINO_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        fino_set_error("%s: hipFuncSetAttribute failed: %s", who, hipGetErrorString(e));
        return FINO_ERR_LAUNCH;
    }
    once.done[dev].store(1, std::memory_order_release);
    return FINO_OK;
}

static inline bool fino_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
