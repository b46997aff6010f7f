// Shared definitions of the MFMA GEMM / implicit-GEMM conv kernels (fino_gemm.hip).
#pragma once
#include "fino_common.h"

namespace fino_gemm_ns {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kTileBytes = BM * BK * 2;        // 32 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;    // A + W
constexpr int kCsStride = BN * 2 + 16;         // padded epilogue row (bytes)
constexpr int kSmemBytes = (2 * kStageBytes > BM * kCsStride) ? 2 * kStageBytes : BM * kCsStride;

struct GemmParams {
    const uint16_t* a;
    const uint16_t* w;
    const uint16_t* bias;
    uint16_t* c;
    const uint16_t* r;
    const float* gate;
    const int32_t* sel;
    int64_t m, n, k, lda, ldw, ldc, ldr, mod_stride;
    int tiles_m, tiles_n;
    // implicit-GEMM convolution (CONV variant): A is a channels-last activation [T_in, H_in, W_in, lda]; row m of the
    // GEMM is output position (t, h, w); K runs tap-major, channel-minor (cin_chunks x 64 channels per tap).
    int to, ho, wo, ti, hi, wi;      // output / input extents
    int kt, kh, kw, st, sh, sw, pt, ph, pw, up, cin_chunks;
    const uint16_t* zero_page;       // >= 128 B of zeros: source of out-of-range taps (LDS-DMA cannot zero-fill)
};

__device__ __forceinline__ float gelu_tanh_f32(float x) {
    // 0.5*x*(1+tanh(u)) == x*sigmoid(2u),  u = sqrt(2/pi)*(x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}

// swizzle of the 16-byte chunk index inside a 128-byte tile row
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

}  // namespace fino_gemm_ns
