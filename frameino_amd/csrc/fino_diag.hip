// Diagnostics (never on the product path): what the chip sustains when NOTHING but the matrix pipe works.
// fino_diag_mfma_peak: every wave issues `iters` x 16 independent v_mfma_f32_32x32x16_bf16 (or 16x16x32) from registers,
// 1 or 2 waves per SIMD on every CU -- the dense MFMA rate under the board's power cap, i.e. the ceiling any
// MFMA-bound kernel of this library can be compared with (tools/mfma_peak.py).
#include "fino_common.h"

namespace {

template <int KIND>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    // operands from the caller's buffer (zeros, small integers or gaussian noise: the power drawn -- and under the cap
    // the clock -- depends on how many operand bits toggle)
    bf16x8_t a, b;
    {
        const uint4 ua = reinterpret_cast<const uint4*>(out)[16 + threadIdx.x];
        const uint4 ub = reinterpret_cast<const uint4*>(out)[16 + 256 + threadIdx.x];
        a = __builtin_bit_cast(bf16x8_t, ua);
        b = __builtin_bit_cast(bf16x8_t, ub);
    }
    float s = 0.f;
    if constexpr (KIND == 2) {
        // fp8 (e4m3) operands of the block-scaled MFMA, all scales 1.0 (e8m0 127): 32 bytes per lane and operand, behind the bf16 ones
        typedef int i32x8_t __attribute__((ext_vector_type(8)));
        const uint4* o8 = reinterpret_cast<const uint4*>(out) + 16 + 512;
        const uint4 a0 = o8[2 * threadIdx.x], a1 = o8[2 * threadIdx.x + 1];
        const uint4 b0 = o8[512 + 2 * threadIdx.x], b1 = o8[512 + 2 * threadIdx.x + 1];
        const i32x8_t a8 = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
        const i32x8_t b8 = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        f32x16_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[t][j];
    } else if constexpr (KIND == 3) {
        // the same loop on fp16 operands (the caller fills the SAME 16 bytes per lane with fp16 values): what the matrix pipe
        // sustains under the cap at the dtype the reference app loads the DiT in -- 10 mantissa bits toggle instead of 7
        const f16x8_t ah = __builtin_bit_cast(f16x8_t, a), bh = __builtin_bit_cast(f16x8_t, b);
        f32x16_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[t][j];
    } else if constexpr (KIND == 0) {
        f32x16_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[t][j];
    } else {
        f32x4_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[t][j];
    }
    if (s == 12345.678f) out[0] = s;          // keep the accumulators alive
}

}  // namespace

// kind 0: 32x32x16 (16 per iteration), kind 1: 16x16x32 (32 per iteration); both = 524288 FLOP per wave-iteration.
// kind 2: 16 x v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 operands (scales 1.0) = 2097152 FLOP per wave-iteration.
// kind 3: kind 0's loop with v_mfma_f32_32x32x16_f16 (the operand bytes are read as fp16).
// scratch: >= 256 B + 2 x 256 x 16 B; bytes 256.. hold the A and B operands of the 256 lanes (caller-filled); kind 2 reads its
// 2 x 256 x 32 B of fp8 operands behind them (scratch >= 256 + 8192 + 16384 B).
// blocks of 256 threads (one wave per SIMD); waves_per_simd in {1, 2} -> blocks = CUs * waves_per_simd.  Returns the FLOPs
// launched in *flops.
extern "C" int fino_diag_mfma_peak(int kind, int waves_per_simd, int iters, void* scratch, double* flops, void* stream) {
    FINO_CHECK(kind >= 0 && kind <= 3 && (waves_per_simd == 1 || waves_per_simd == 2) && iters > 0 && scratch && flops,
               FINO_ERR_ARG, "fino_diag_mfma_peak: bad arguments");
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, fino_current_device()) != hipSuccess || cus <= 0)
        cus = 256;
    const int blocks = cus * waves_per_simd;
    if (kind == 0)
        mfma_peak_kernel<0><<<blocks, 256, 0, (hipStream_t)stream>>>((float*)scratch, iters);
    else if (kind == 1)
        mfma_peak_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>((float*)scratch, iters);
    else if (kind == 3)
        mfma_peak_kernel<3><<<blocks, 256, 0, (hipStream_t)stream>>>((float*)scratch, iters);
    else
        mfma_peak_kernel<2><<<blocks, 256, 0, (hipStream_t)stream>>>((float*)scratch, iters);
    FINO_LAUNCH_CHECK();
    *flops = (double)blocks * 4.0 * (double)iters * (kind == 2 ? 2097152.0 : 524288.0);
    return FINO_OK;
}
