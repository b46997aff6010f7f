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
    // = x / (1 + 2^(-x*(c1 + c3*x^2))),  c1 = 2*sqrt(2/pi)*log2(e), c3 = 0.044715*c1: 7 VALU ops with v_exp_f32 and
    // v_rcp_f32 (1 ulp) instead of an IEEE division (~10 more ops) -- the result is rounded to bf16/fp16 right after.
    const float c1 = 2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c3 = 0.044715f * c1;
    const float e = __builtin_amdgcn_exp2f(-x * (c3 * x * x + c1));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// swizzle of the 16-byte chunk index inside a 128-byte tile row
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

}  // namespace fino_gemm_ns
