// HBM-bound kernels of the DiT forward: LayerNorm+AdaLN modulate, gated residual, RMSNorm(all heads)+RoPE,
// per-head LayerNorm+RoPE (CogVideoX), patchify/unpatchify gathers, sampler glue.
//
// Design (MI355X): one 64-lane wave owns one token row; a lane moves 16 B (8 elements) per access, the row stays in
// registers between the statistics pass and the apply pass, so every tensor is read once and written once
// (algorithmic bytes = 2*rows*dim*2 B per pass).  4 waves per workgroup, rows*... >> 256 workgroups.
// Built with -ffp-contract=off: the rounding points are the reference's (mul, mul, sub -- not fma).
#include <type_traits>

#include "fino_common.h"
#include "fino_gemm_common.h"      // mx_quant8 / mx_scale_index: the MXFP8 activation layout fino_gemm_mxfp8 consumes

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxPasses = 8;  // dim <= 8 * 512 = 4096

template <typename T, int NP>
struct RowRegs {
    float v[NP][8];
};

#ifndef FINO_EW_NT
#define FINO_EW_NT 0          /* A/B build knob, ln_modulate_kernel: bit 0 non-temporal row loads, bit 1 non-temporal row stores.
                                 Measured: stores alone 73 -> 51 us in isolation, and NOTHING on the step -- the
                                 consumer GEMM then reads its A operand from HBM instead of the Infinity Cache.  Off. */
#endif
#define EW_STORE_NT(PTR_, VAL_)                                                                \
    {                                                                                          \
        const uint4 v__ = (VAL_);                                                              \
        __builtin_nontemporal_store(u32x4_t{v__.x, v__.y, v__.z, v__.w}, reinterpret_cast<u32x4_t*>(PTR_)); \
    }
#define EW_STORE(PTR_, VAL_) (*reinterpret_cast<uint4*>(PTR_) = (VAL_))

template <typename T, int NP, bool NT = false>
__device__ __forceinline__ void load_row(const uint16_t* __restrict__ p, int dim, int lane, float (&v)[NP][8]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < dim) {
            uint4 u;
            if constexpr (NT) {
                const u32x4_t t4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p + c));
                u = make_uint4(t4[0], t4[1], t4[2], t4[3]);
            } else {
                u = *reinterpret_cast<const uint4*>(p + c);
            }
            unpack8<T>(u, v[i]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
}

template <int NP>
__device__ __forceinline__ void ln_stats(const float (&v)[NP][8], int dim, int lane, float eps, float& mean,
                                         float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
    mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < dim) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / (float)dim;
    rstd = 1.0f / sqrtf(var + eps);
}

// ---------------------------------------------------------------------------------------------------------
// MODE 0: y = T(LN(x)*(1+scale)+shift)                        (Wan AdaLN-zero: fp32 math, one rounding)
// MODE 1: y = T(LN(x)*w+b)                                    (affine LayerNorm; w/b may be null)
// MODE 2: y = T(T(T(LN(x)*w+b) * T(1+scale)) + shift)         (CogVideoXLayerNormZero / AdaLayerNorm executed in T:
//                                                              the reference rounds after every tensor op)
// QOUT: instead of the T row, its MXFP8 quantisation (e4m3 bytes to q [rows, dim] + one e8m0 scale per 32 channels in the
// fino_quantize_mxfp8 layout): the bytes fino_quantize_mxfp8 would produce from the T-rounded y, without writing y.
template <typename T, int NP, int MODE, bool QOUT = false>
__global__ __launch_bounds__(kWavesPerBlock * 64) void ln_modulate_kernel(
    const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
    const float* __restrict__ p_w, const float* __restrict__ p_b, const float* __restrict__ p_shift,
    const float* __restrict__ p_scale, int64_t mod_stride, const int32_t* __restrict__ sel, float eps,
    uint8_t* __restrict__ q = nullptr, uint8_t* __restrict__ qs = nullptr, int64_t rows_pad = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NP][8];
    load_row<T, NP, (FINO_EW_NT & 1) != 0>(x + row * ldx, dim, lane, v);
    float mean, rstd;
    ln_stats<NP>(v, dim, lane, eps, mean, rstd);
    const int64_t moff = (MODE != 1 && sel) ? (int64_t)sel[row] * mod_stride : 0;
    auto ld8 = [](const float* p, float (&o)[8]) {
        const float4 a0 = *reinterpret_cast<const float4*>(p);
        const float4 a1 = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a0.x; o[1] = a0.y; o[2] = a0.z; o[3] = a0.w; o[4] = a1.x; o[5] = a1.y; o[6] = a1.z; o[7] = a1.w;
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c >= dim) continue;
        float o[8], w[8], b[8], sc[8], sh[8];
        if (MODE != 0 && p_w) ld8(p_w + c, w);
        if (MODE != 0 && p_b) ld8(p_b + c, b);
        if (MODE != 1) { ld8(p_scale + moff + c, sc); ld8(p_shift + moff + c, sh); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = (v[i][j] - mean) * rstd;
            if (MODE == 0) {
                o[j] = n * (1.0f + sc[j]) + sh[j];
            } else {
                float t = n;
                if (p_w) t = t * w[j];
                if (p_b) t = t + b[j];
                if (MODE == 2) t = round_to<T>(round_to<T>(t) * round_to<T>(1.0f + sc[j])) + sh[j];
                o[j] = t;
            }
        }
        if constexpr (QOUT) {
            float r8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] = round_to<T>(o[j]);
            int e;
            const uint2 qv = fino_gemm_ns::mx_quant8(r8, e);
            *reinterpret_cast<uint2*>(q + row * dim + c) = qv;
            if ((c & 31) == 0) qs[fino_gemm_ns::mx_scale_index(row, c, rows_pad)] = (uint8_t)(e + 127);
        } else if constexpr ((FINO_EW_NT & 2) != 0) EW_STORE_NT(y + row * ldy + c, pack8<T>(o))
        else EW_STORE(y + row * ldy + c, pack8<T>(o));
    }
}

// out = T(float(x) + float(y)*gate) or T(x + y)
template <typename T>
__global__ __launch_bounds__(256) void gated_residual_kernel(const uint16_t* __restrict__ x,
                                                             const uint16_t* __restrict__ y,
                                                             uint16_t* __restrict__ out, int64_t rows, int dim,
                                                             int64_t ldx, int64_t ldy, int64_t ldo,
                                                             const float* __restrict__ gate, int64_t mod_stride,
                                                             const int32_t* __restrict__ sel, int staged) {
    const int chunks = dim >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunks;
        const int c = (int)(i - row * chunks) * 8;
        float a[8], b[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + row * ldx + c), a);
        unpack8<T>(*reinterpret_cast<const uint4*>(y + row * ldy + c), b);
        if (gate) {
            const float* g = gate + (sel ? (int64_t)sel[row] * mod_stride : 0) + c;
            const float4 g0 = *reinterpret_cast<const float4*>(g);
            const float4 g1 = *reinterpret_cast<const float4*>(g + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = a[j] + (staged ? round_to<T>(b[j] * gg[j]) : b[j] * gg[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = a[j] + b[j];
        }
        *reinterpret_cast<uint4*>(out + row * ldo + c) = pack8<T>(o);
    }
}

// RMSNorm over the full row + weight (T) + RoPE, in place.  grid.y selects nothing; one wave per row.
// One row of fino_rmsnorm_rope*: RMSNorm over the row's `dim` channels (weight in T, diffusers' rounding points), RoPE on
// adjacent channel pairs, out_scale, then stored in place or scattered by head (out != nullptr).
template <typename T, int NP>
__device__ __forceinline__ void rmsnorm_rope_row(uint16_t* __restrict__ xrow, int64_t row, int dim,
                                                 const uint16_t* __restrict__ w, float eps,
                                                 const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                 int head_dim, float out_scale, uint16_t* __restrict__ out,
                                                 const int64_t* __restrict__ head_off,
                                                 const int64_t* __restrict__ head_ld, int lane) {
    float v[NP][8];
    load_row<T, NP>(xrow, dim, lane, v);
    float rs = 1.f;
    if (w) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) q += v[i][j] * v[i][j];
        const float var = wave_sum(q) / (float)dim;
        rs = 1.0f / sqrtf(var + eps);
    }
    const int half = head_dim >> 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c >= dim) continue;
        float o[8];
        if (w) {
            float ww[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(w + c), ww);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = round_to<T>(round_to<T>(v[i][j] * rs) * ww[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[i][j];
        }
        if (cos_t) {
            const int pidx = (c % head_dim) >> 1;  // first of 4 pairs
            const float4 cs = *reinterpret_cast<const float4*>(cos_t + row * half + pidx);
            const float4 sn = *reinterpret_cast<const float4*>(sin_t + row * half + pidx);
            const float cc[4] = {cs.x, cs.y, cs.z, cs.w};
            const float ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x1 = o[2 * j], x2 = o[2 * j + 1];
                o[2 * j] = x1 * cc[j] - x2 * ss[j];
                o[2 * j + 1] = x1 * ss[j] + x2 * cc[j];
            }
        }
        // fino_rmsnorm_rope_scaled: the fp32 result times out_scale, rounded ONCE (1.0f: the plain op, bit for bit)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= out_scale;
        if (out) {     // scatter: head hd of the row lands at out + head_off[hd] + row * head_ld[hd]
            const int hd = c / head_dim;
            EW_STORE(out + head_off[hd] + row * head_ld[hd] + (c - hd * head_dim), pack8<T>(o));
        } else {
            EW_STORE(xrow + c, pack8<T>(o));
        }
    }
}

// Up to three column segments of `dim` channels of the same rows in ONE launch (blockIdx.y = segment): q | k | v of a fused
// projection.  A segment without weight and without RoPE is a copy (meaningful only with a scatter destination).
struct RmsRopeParts {
    const uint16_t* w[3];
    float eps[3];
    float out_scale[3];
    int rope[3];
};
template <typename T, int NP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void rmsnorm_rope_kernel(uint16_t* __restrict__ x, int64_t rows,
                                                                           int dim, int64_t ldx, const RmsRopeParts pp,
                                                                           const float* __restrict__ cos_t,
                                                                           const float* __restrict__ sin_t,
                                                                           int head_dim, uint16_t* __restrict__ out,
                                                                           const int64_t* __restrict__ head_off,
                                                                           const int64_t* __restrict__ head_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int s = blockIdx.y;
    const bool rope = pp.rope[s] != 0;
    rmsnorm_rope_row<T, NP>(x + row * ldx + (int64_t)s * dim, row, dim, pp.w[s], pp.eps[s], rope ? cos_t : nullptr,
                            rope ? sin_t : nullptr, head_dim, pp.out_scale[s], out,
                            head_off ? head_off + (int64_t)s * (dim / head_dim) : nullptr, head_ld, lane);
}

// The statistic of rmsnorm_rope_row alone: rrms[row] = 1 / sqrt(mean(x^2) + eps), the same loads, the same order of summation,
// the same bits -- for a consumer that applies the normalisation itself (fino_attn_probs normalises q while it loads it).
template <typename T, int NP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void row_rrms_kernel(const uint16_t* __restrict__ x, int64_t rows, int dim,
                                                                       int64_t ldx, float eps, float* __restrict__ rrms) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NP][8];
    load_row<T, NP>(x + row * ldx, dim, lane, v);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) q += v[i][j] * v[i][j];
    const float var = wave_sum(q) / (float)dim;
    if (lane == 0) rrms[row] = 1.0f / sqrtf(var + eps);
}

// CogVideoX: per-head LayerNorm(head_dim) (affine, T params; statistics fp32, output rounded to T) then RoPE
// out = T(float(x)*cos + float(rot(x))*sin) on rows >= rope_row0.  One lane owns 8 channels; a head spans
// head_dim/8 consecutive lanes (8 for 64, 16 for 128) -> xor-shuffle reduction inside the group.
template <typename T>
__global__ __launch_bounds__(256) void headnorm_rope_kernel(uint16_t* __restrict__ x, int batch, int64_t rows,
                                                            int heads, int head_dim, int64_t ldx,
                                                            int64_t batch_stride, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ b, float eps,
                                                            const float* __restrict__ cos_t,
                                                            const float* __restrict__ sin_t, int64_t rope_row0,
                                                            float out_scale) {
    const int lanes_per_head = head_dim >> 3;
    const int64_t chunks_per_row = (int64_t)heads * lanes_per_head;
    const int64_t total = (int64_t)batch * rows * chunks_per_row;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < total;
    const int64_t ii = active ? i : total - 1;
    const int64_t br = ii / chunks_per_row;
    const int cw = (int)(ii - br * chunks_per_row);
    const int64_t bi = br / rows;
    const int64_t row = br - bi * rows;
    const int cin = (cw % lanes_per_head) * 8;  // channel inside the head
    uint16_t* p = x + bi * batch_stride + row * ldx + (int64_t)cw * 8;
    float v[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(p), v);
    if (w) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        for (int off = lanes_per_head >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)head_dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[j] - mean;
            q += d * d;
        }
        for (int off = lanes_per_head >> 1; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float rstd = 1.0f / sqrtf(q / (float)head_dim + eps);
        float ww[8], bb[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(w + cin), ww);
        unpack8<T>(*reinterpret_cast<const uint4*>(b + cin), bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = round_to<T>((v[j] - mean) * rstd * ww[j] + bb[j]);
    }
    if (cos_t && row >= rope_row0) {
        const float* cp = cos_t + (row - rope_row0) * head_dim + cin;
        const float* sp = sin_t + (row - rope_row0) * head_dim + cin;
        float cc[8], ss[8];
        *reinterpret_cast<float4*>(cc) = *reinterpret_cast<const float4*>(cp);
        *reinterpret_cast<float4*>(cc + 4) = *reinterpret_cast<const float4*>(cp + 4);
        *reinterpret_cast<float4*>(ss) = *reinterpret_cast<const float4*>(sp);
        *reinterpret_cast<float4*>(ss + 4) = *reinterpret_cast<const float4*>(sp + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xr = v[2 * j], xi = v[2 * j + 1];
            o[2 * j] = xr * cc[2 * j] + (-xi) * ss[2 * j];
            o[2 * j + 1] = xi * cc[2 * j + 1] + xr * ss[2 * j + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = o[j];
    }
    // fino_headnorm_rope_scaled (1.0f: the plain op, bit for bit): q for FINO_ATTN_SCALE_FOLDED
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= out_scale;
    if (active) *reinterpret_cast<uint4*>(p) = pack8<T>(v);
}

// ---------------------------------------------------------------------------------------------------------
// patchify: a[token][((c*pt+dt)*ph+dh)*pw+dw] = x[c][f*pt+dt][i*ph+dh][j*pw+dw]; thread = one (token, c, dt, dh)
// run of pw elements.  Tokens vary fastest across threads so the global reads walk W contiguously.
__global__ __launch_bounds__(256) void patchify_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ a,
                                                       int C, int F, int H, int W, int pt, int ph, int pw,
                                                       int64_t lda) {
    const int ppf = F / pt, pph = H / ph, ppw = W / pw;
    const int64_t L = (int64_t)ppf * pph * ppw;
    const int64_t runs = (int64_t)C * pt * ph;
    const int64_t total = L * runs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tok = i % L;
        const int64_t run = i / L;  // (c*pt+dt)*ph+dh
        const int dh = (int)(run % ph);
        const int dt = (int)((run / ph) % pt);
        const int c = (int)(run / ((int64_t)ph * pt));
        const int j = (int)(tok % ppw);
        const int ii = (int)((tok / ppw) % pph);
        const int f = (int)(tok / ((int64_t)ppw * pph));
        const uint16_t* src = x + (((int64_t)c * F + (f * pt + dt)) * H + (ii * ph + dh)) * W + (int64_t)j * pw;
        uint16_t* dst = a + tok * lda + run * pw;
        for (int d = 0; d < pw; ++d) dst[d] = src[d];
    }
}

// unpatchify: out[c][f*pt+dt][i*ph+dh][j*pw+dw] = y[token][((dt*ph+dh)*pw+dw)*Cout + c]
__global__ __launch_bounds__(256) void unpatchify_kernel(const uint16_t* __restrict__ y, uint16_t* __restrict__ out,
                                                         int Cout, int F, int H, int W, int pt, int ph, int pw,
                                                         int64_t ldy) {
    const int pph = H / ph, ppw = W / pw;
    const int64_t total = (int64_t)Cout * F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int f = (int)((i / ((int64_t)W * H)) % F);
        const int c = (int)(i / ((int64_t)W * H * F));
        const int64_t tok = ((int64_t)(f / pt) * pph + h / ph) * ppw + w / pw;
        const int col = (((f % pt) * ph + (h % ph)) * pw + (w % pw)) * Cout + c;
        out[i] = y[tok * ldy + col];
    }
}

// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void wan_model_input_kernel(const float* __restrict__ lat,
                                                              const float* __restrict__ cond,
                                                              const float* __restrict__ idl,
                                                              const float* __restrict__ traj,
                                                              uint16_t* __restrict__ out, int C, int Fg, int Fid,
                                                              int HW) {
    const int Ft = Fg + Fid;
    const int64_t total = (int64_t)2 * C * Ft * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % HW);
        const int f = (int)((i / HW) % Ft);
        const int c = (int)(i / ((int64_t)HW * Ft));
        float v;
        if (c < C) {
            if (f < Fg) {
                // (1-m)*cond + m*lat with m in {0,1}: evaluates exactly like the reference's fp32 blend
                const float m = f == 0 ? 0.f : 1.f;
                v = (1.f - m) * cond[(int64_t)c * HW + s] + m * lat[((int64_t)c * Fg + f) * HW + s];
            } else {
                v = idl[((int64_t)c * Fid + (f - Fg)) * HW + s];
            }
        } else {
            v = traj[((int64_t)(c - C) * Ft + f) * HW + s];
        }
        out[i] = T::from_f32(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cfg_euler_kernel(const uint16_t* __restrict__ pc,
                                                        const uint16_t* __restrict__ pu, float* __restrict__ lat,
                                                        int C, int Fg, int Ft, int HW, float g,
                                                        const float* __restrict__ dt_dev, int round_out) {
    const float dt = *dt_dev;
    const int64_t total = (int64_t)C * Fg * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % HW);
        const int f = (int)((i / HW) % Fg);
        const int c = (int)(i / ((int64_t)HW * Fg));
        const int64_t pi = ((int64_t)c * Ft + f) * HW + s;
        float n = T::to_f32(pc[pi]);
        if (pu) {
            const float u = T::to_f32(pu[pi]);
            n = round_to<T>(u + round_to<T>(g * round_to<T>(n - u)));
        }
        float v = lat[i] + dt * n;
        if (round_out) v = round_to<T>(v);
        lat[i] = v;
    }
}

// CogVideoX sampler step (pipeline_cogvideox_i2v_motion_FrameINO.py:893-927 with a v-prediction DDIM step):
//   v = u + g*(c - u) in fp32 (the reference upcasts noise_pred, :896);  x0 = T(sa*x) - sb*v;  x' = T(T(ca*x) + cb*x0)
//   (x is the T-typed latent; products of x with the scheduler's 0-dim coefficients stay in T under torch promotion).
// coef = {sa, sb, ca, cb, g} on the device.  pred: [2, Ft, C*HW] (uncond, cond), lat: [Fg, C*HW] in place.
template <typename T>
__global__ __launch_bounds__(256) void cfg_vpred_step_kernel(const uint16_t* __restrict__ pred,
                                                             uint16_t* __restrict__ lat, int64_t n_lat,
                                                             int64_t batch_stride, const float* __restrict__ coef,
                                                             int has_uncond) {
    const float sa = coef[0], sb = coef[1], ca = coef[2], cb = coef[3], g = coef[4];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lat; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (has_uncond) {
            const float u = T::to_f32(pred[i]);
            const float c = T::to_f32(pred[batch_stride + i]);
            v = u + g * (c - u);
        } else {
            v = T::to_f32(pred[i]);
        }
        const float x = T::to_f32(lat[i]);
        const float x0 = round_to<T>(sa * x) - sb * v;
        lat[i] = T::from_f32(round_to<T>(ca * x) + cb * x0);
    }
}

// CogVideoXDPMScheduler.step (diffusers, third-party; the sampler the CogVideoX-5B-I2V repo ships and the reference's
// training-time validation builds, train_code/train_cogvideox_motion_FrameINO.py:692; pipeline call site :915-926):
// SDE-DPM-Solver++ (2M) on v-prediction, fused with CFG.  Linear in {x, v, x0_old, noise}; the host folds the step's
// scalars into coef = {sa, sb, m1, m2, m3, m4, mn, g, use_old}:
//   v  = u + g*(c-u) (fp32: the pipeline takes noise_pred.float(), :897);  x0 = T(sa*x) - sb*v            (fp32)
//   d  = use_old ? m3*x0 - m4*x0_old : x0;   x' = T( T(m1*x) - m2*d + T(mn*noise) );   x0_old <- x0
// (products of the T-typed latent / noise with the scheduler's 0-dim coefficients stay in T under torch promotion.)
template <typename T>
__global__ __launch_bounds__(256) void cfg_dpm_step_kernel(const uint16_t* __restrict__ pred, uint16_t* __restrict__ lat,
                                                           float* __restrict__ x0_old, const uint16_t* __restrict__ noise,
                                                           int64_t n_lat, int64_t batch_stride,
                                                           const float* __restrict__ coef, int has_uncond) {
    const float sa = coef[0], sb = coef[1], m1 = coef[2], m2 = coef[3], m3 = coef[4], m4 = coef[5], mn = coef[6];
    const float g = coef[7];
    const bool use_old = coef[8] != 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lat; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (has_uncond) {
            const float u = T::to_f32(pred[i]);
            const float c = T::to_f32(pred[batch_stride + i]);
            v = u + g * (c - u);
        } else {
            v = T::to_f32(pred[i]);
        }
        const float x = T::to_f32(lat[i]);
        const float x0 = round_to<T>(sa * x) - sb * v;
        const float d = use_old ? m3 * x0 - m4 * x0_old[i] : x0;
        const float nz = round_to<T>(mn * T::to_f32(noise[i]));
        lat[i] = T::from_f32(round_to<T>(m1 * x) - m2 * d + nz);
        x0_old[i] = x0;
    }
}

// UniPC (bh2, predict-x0, flow sigmas) multistep update fused with CFG, one pass over the latents
// (diffusers UniPCMultistepScheduler.step as Wan2.2-TI2V-5B-Diffusers configures it; pipeline :882-891).
// Every tensor op of the published algorithm is linear in {x, last_sample, m0, m1, v}; the host folds the step's
// scalars into coef = {g, sigma, use_corr, Cx, C0, C1, Ct, Px, P0, P1}:
//   v   = T(u + T(g*T(c-u)));  m_t = x - T(sigma*v)                         (convert_model_output, flow prediction)
//   x_c = use_corr ? Cx*last + C0*m0 + C1*m1 + Ct*m_t : x                    (multistep_uni_c_bh_update)
//   x'  = Px*x_c + P0*m_t + P1*m0                                            (multistep_uni_p_bh_update)
//   last <- x_c;  m1 <- m0;  m0 <- m_t;  x <- x'
template <typename T>
__global__ __launch_bounds__(256) void cfg_unipc_kernel(const uint16_t* __restrict__ pc, const uint16_t* __restrict__ pu,
                                                        float* __restrict__ x, float* __restrict__ last,
                                                        float* __restrict__ m0, float* __restrict__ m1, int C, int Fg,
                                                        int Ft, int HW, const float* __restrict__ coef) {
    const float g = coef[0], sigma = coef[1], use_corr = coef[2];
    const float Cx = coef[3], C0 = coef[4], C1 = coef[5], Ct = coef[6], Px = coef[7], P0 = coef[8], P1 = coef[9];
    const int64_t total = (int64_t)C * Fg * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % HW);
        const int f = (int)((i / HW) % Fg);
        const int c = (int)(i / ((int64_t)HW * Fg));
        const int64_t pi = ((int64_t)c * Ft + f) * HW + s;
        float v = T::to_f32(pc[pi]);
        if (pu) {
            const float u = T::to_f32(pu[pi]);
            v = round_to<T>(u + round_to<T>(g * round_to<T>(v - u)));
        }
        const float xv = x[i], a0 = m0[i];
        const float mt = xv - round_to<T>(sigma * v);
        const float xc = use_corr != 0.f ? Cx * last[i] + C0 * a0 + C1 * m1[i] + Ct * mt : xv;
        x[i] = Px * xc + P0 * mt + P1 * a0;
        last[i] = xc;
        m1[i] = a0;
        m0[i] = mt;
    }
}

template <int NPmax, typename F>
inline bool dispatch_np(int dim, F&& f) {
    const int np = (dim + 511) / 512;
    switch (np) {
        case 1: f(std::integral_constant<int, 1>{}); return true;
        case 2: f(std::integral_constant<int, 2>{}); return true;
        case 3: f(std::integral_constant<int, 3>{}); return true;
        case 4: f(std::integral_constant<int, 4>{}); return true;
        case 5: f(std::integral_constant<int, 5>{}); return true;
        case 6: f(std::integral_constant<int, 6>{}); return true;
        case 7: f(std::integral_constant<int, 7>{}); return true;
        case 8: f(std::integral_constant<int, 8>{}); return true;
        default: return false;
    }
}

inline int grid_1d(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 8;  // ~8 blocks per CU, grid-stride the rest
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <int MODE>
int launch_ln_q(const void* x, void* q, void* qs, int64_t rows, int dim, int64_t ldx, const float* w, const float* b,
                const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel, float eps, int dtype,
                hipStream_t st) {
    const dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
    const int64_t rows_pad = (rows + 255) / 256 * 256;
    const bool ok = dispatch_np<kMaxPasses>(dim, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        if (dtype == FINO_BF16)
            ln_modulate_kernel<BF16, NP, MODE, true><<<grid, block, 0, st>>>((const uint16_t*)x, nullptr, rows, dim, ldx, 0, w,
                                                                           b, shift, scale, mod_stride, sel, eps,
                                                                           (uint8_t*)q, (uint8_t*)qs, rows_pad);
        else
            ln_modulate_kernel<F16, NP, MODE, true><<<grid, block, 0, st>>>((const uint16_t*)x, nullptr, rows, dim, ldx, 0, w,
                                                                          b, shift, scale, mod_stride, sel, eps,
                                                                          (uint8_t*)q, (uint8_t*)qs, rows_pad);
    });
    FINO_CHECK(ok, FINO_ERR_UNSUPPORTED, "layernorm: dim %d > %d unsupported", dim, kMaxPasses * 512);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

template <int MODE>
int launch_ln(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy, const float* w, const float* b,
              const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel, float eps, int dtype,
              hipStream_t st) {
    const dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
    const bool ok = dispatch_np<kMaxPasses>(dim, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        if (dtype == FINO_BF16)
            ln_modulate_kernel<BF16, NP, MODE><<<grid, block, 0, st>>>((const uint16_t*)x, (uint16_t*)y, rows, dim,
                                                                     ldx, ldy, w, b, shift, scale, mod_stride, sel, eps);
        else
            ln_modulate_kernel<F16, NP, MODE><<<grid, block, 0, st>>>((const uint16_t*)x, (uint16_t*)y, rows, dim,
                                                                    ldx, ldy, w, b, shift, scale, mod_stride, sel, eps);
    });
    FINO_CHECK(ok, FINO_ERR_UNSUPPORTED, "layernorm: dim %d > %d unsupported", dim, kMaxPasses * 512);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

}  // namespace

#define CHECK_ROWS_DIM(fn)                                                                                    \
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, fn ": dtype %d", dtype);                \
    FINO_CHECK(rows >= 0 && dim > 0 && dim % 8 == 0, FINO_ERR_ARG, fn ": rows=%lld dim=%d (dim %% 8 != 0)",   \
               (long long)rows, dim);                                                                         \
    if (rows == 0) return FINO_OK;

extern "C" int fino_adaln_modulate(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                                   const float* shift, const float* scale, int64_t mod_stride, const int32_t* sel,
                                   float eps, int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_adaln_modulate");
    FINO_CHECK(x && y && shift && scale, FINO_ERR_ARG, "fino_adaln_modulate: null pointer");
    FINO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && mod_stride % 4 == 0 && fino_aligned16(x) && fino_aligned16(y) &&
                   fino_aligned16(shift) && fino_aligned16(scale),
               FINO_ERR_ARG, "fino_adaln_modulate: 16-byte alignment required");
    return launch_ln<0>(x, y, rows, dim, ldx, ldy, nullptr, nullptr, shift, scale, mod_stride, sel, eps, dtype,
                        (hipStream_t)stream);
}

// mode 0 = fino_adaln_modulate, 1 = fino_layernorm (w / b may be NULL), 2 = fino_layernorm_zero; see include/frameino_hip.h
extern "C" int fino_ln_mxfp8(int mode, const void* x, void* q, void* scales, int64_t rows, int dim, int64_t ldx,
                             const float* w, const float* b, const float* shift, const float* scale, int64_t mod_stride,
                             const int32_t* sel, float eps, int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_ln_mxfp8");
    FINO_CHECK(mode >= 0 && mode <= 2 && x && q && scales, FINO_ERR_ARG, "fino_ln_mxfp8: mode %d / null pointer", mode);
    FINO_CHECK(mode == 1 || (shift && scale), FINO_ERR_ARG, "fino_ln_mxfp8: modulation rows missing");
    FINO_CHECK(dim % 128 == 0, FINO_ERR_ARG, "fino_ln_mxfp8: dim=%d must be a multiple of 128", dim);
    FINO_CHECK(ldx % 8 == 0 && mod_stride % 4 == 0 && fino_aligned16(x) && fino_aligned16(shift) && fino_aligned16(scale) &&
                   fino_aligned16(w) && fino_aligned16(b) && ((uintptr_t)q & 7) == 0,
               FINO_ERR_ARG, "fino_ln_mxfp8: alignment");
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) return launch_ln_q<0>(x, q, scales, rows, dim, ldx, nullptr, nullptr, shift, scale, mod_stride, sel, eps, dtype, st);
    if (mode == 1) return launch_ln_q<1>(x, q, scales, rows, dim, ldx, w, b, nullptr, nullptr, 0, nullptr, eps, dtype, st);
    return launch_ln_q<2>(x, q, scales, rows, dim, ldx, w, b, shift, scale, mod_stride, sel, eps, dtype, st);
}

extern "C" int fino_layernorm(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                              const float* w, const float* b, float eps, int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_layernorm");
    FINO_CHECK(x && y, FINO_ERR_ARG, "fino_layernorm: null pointer");
    FINO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && fino_aligned16(x) && fino_aligned16(y) && fino_aligned16(w) &&
                   fino_aligned16(b),
               FINO_ERR_ARG, "fino_layernorm: 16-byte alignment required");
    return launch_ln<1>(x, y, rows, dim, ldx, ldy, w, b, nullptr, nullptr, 0, nullptr, eps, dtype, (hipStream_t)stream);
}

extern "C" int fino_layernorm_zero(const void* x, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                                   const float* w, const float* b, const float* shift, const float* scale,
                                   int64_t mod_stride, const int32_t* sel, float eps, int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_layernorm_zero");
    FINO_CHECK(x && y && shift && scale, FINO_ERR_ARG, "fino_layernorm_zero: null pointer");
    FINO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && mod_stride % 4 == 0 && fino_aligned16(x) && fino_aligned16(y) &&
                   fino_aligned16(shift) && fino_aligned16(scale) && fino_aligned16(w) && fino_aligned16(b),
               FINO_ERR_ARG, "fino_layernorm_zero: 16-byte alignment required");
    return launch_ln<2>(x, y, rows, dim, ldx, ldy, w, b, shift, scale, mod_stride, sel, eps, dtype, (hipStream_t)stream);
}

static int gated_residual_impl(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                               int64_t ldo, const float* gate, int64_t mod_stride, const int32_t* sel, int staged,
                               int dtype, void* stream);

extern "C" int fino_gated_residual(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx,
                                   int64_t ldy, int64_t ldo, const float* gate, int64_t mod_stride,
                                   const int32_t* sel, int dtype, void* stream) {
    return gated_residual_impl(x, y, out, rows, dim, ldx, ldy, ldo, gate, mod_stride, sel, 0, dtype, stream);
}

extern "C" int fino_gated_residual_staged(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx,
                                          int64_t ldy, int64_t ldo, const float* gate, int64_t mod_stride,
                                          const int32_t* sel, int dtype, void* stream) {
    return gated_residual_impl(x, y, out, rows, dim, ldx, ldy, ldo, gate, mod_stride, sel, 1, dtype, stream);
}

static int gated_residual_impl(const void* x, const void* y, void* out, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                               int64_t ldo, const float* gate, int64_t mod_stride, const int32_t* sel, int staged,
                               int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_gated_residual");
    FINO_CHECK(x && y && out, FINO_ERR_ARG, "fino_gated_residual: null pointer");
    FINO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && mod_stride % 4 == 0 && fino_aligned16(x) &&
                   fino_aligned16(y) && fino_aligned16(out) && fino_aligned16(gate),
               FINO_ERR_ARG, "fino_gated_residual: 16-byte alignment required");
    const int grid = grid_1d(rows * (dim / 8));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        gated_residual_kernel<BF16><<<grid, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)y, (uint16_t*)out,
                                                          rows, dim, ldx, ldy, ldo, gate, mod_stride, sel, staged);
    else
        gated_residual_kernel<F16><<<grid, 256, 0, st>>>((const uint16_t*)x, (const uint16_t*)y, (uint16_t*)out,
                                                         rows, dim, ldx, ldy, ldo, gate, mod_stride, sel, staged);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

static int rmsnorm_rope_impl(void* x, int64_t rows, int dim, int64_t ldx, int parts, const RmsRopeParts& pp,
                             const float* cos_t, const float* sin_t, int head_dim, void* out, const int64_t* head_off,
                             const int64_t* head_ld, int dtype, void* stream) {
    CHECK_ROWS_DIM("fino_rmsnorm_rope");
    FINO_CHECK(x, FINO_ERR_ARG, "fino_rmsnorm_rope: null pointer");
    FINO_CHECK(parts >= 1 && parts <= 3, FINO_ERR_ARG, "fino_rmsnorm_rope: parts %d", parts);
    bool any_rope = false;
    for (int i = 0; i < parts; ++i) {
        FINO_CHECK(pp.out_scale[i] > 0.f, FINO_ERR_ARG, "fino_rmsnorm_rope_scaled: out_scale must be > 0");
        FINO_CHECK(fino_aligned16(pp.w[i]), FINO_ERR_ARG, "fino_rmsnorm_rope: 16-byte alignment required");
        any_rope = any_rope || pp.rope[i];
    }
    FINO_CHECK((cos_t == nullptr) == (sin_t == nullptr), FINO_ERR_ARG, "fino_rmsnorm_rope: cos/sin must both be set");
    FINO_CHECK(!any_rope || cos_t, FINO_ERR_ARG, "fino_rmsnorm_rope: a segment asks for RoPE without tables");
    FINO_CHECK((!cos_t && !out) || (head_dim > 0 && head_dim % 8 == 0 && dim % head_dim == 0), FINO_ERR_ARG,
               "fino_rmsnorm_rope: head_dim=%d must divide dim=%d and be a multiple of 8", head_dim, dim);
    FINO_CHECK(ldx % 8 == 0 && fino_aligned16(x) && fino_aligned16(cos_t) && fino_aligned16(sin_t) && fino_aligned16(out),
               FINO_ERR_ARG, "fino_rmsnorm_rope: 16-byte alignment required");
    FINO_CHECK(parts == 1 || dim % 8 == 0, FINO_ERR_ARG, "fino_rmsnorm_rope: segments need dim % 8 == 0");
    const dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock), (unsigned)parts), block(kWavesPerBlock * 64);
    hipStream_t st = (hipStream_t)stream;
    const bool ok = dispatch_np<kMaxPasses>(dim, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        if (dtype == FINO_BF16)
            rmsnorm_rope_kernel<BF16, NP><<<grid, block, 0, st>>>((uint16_t*)x, rows, dim, ldx, pp, cos_t, sin_t, head_dim,
                                                                  (uint16_t*)out, head_off, head_ld);
        else
            rmsnorm_rope_kernel<F16, NP><<<grid, block, 0, st>>>((uint16_t*)x, rows, dim, ldx, pp, cos_t, sin_t, head_dim,
                                                                 (uint16_t*)out, head_off, head_ld);
    });
    FINO_CHECK(ok, FINO_ERR_UNSUPPORTED, "fino_rmsnorm_rope: dim %d > %d unsupported", dim, kMaxPasses * 512);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_row_rrms(const void* x, int64_t rows, int dim, int64_t ldx, float eps, float* rrms, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_row_rrms: dtype %d", dtype);
    FINO_CHECK(x && rrms && rows >= 0 && dim > 0 && dim % 8 == 0 && ldx % 8 == 0 && fino_aligned16(x), FINO_ERR_ARG,
               "fino_row_rrms: bad arguments");
    if (rows == 0) return FINO_OK;
    const dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
    hipStream_t st = (hipStream_t)stream;
    const bool ok = dispatch_np<kMaxPasses>(dim, [&](auto np) {
        constexpr int NP = decltype(np)::value;
        if (dtype == FINO_BF16) row_rrms_kernel<BF16, NP><<<grid, block, 0, st>>>((const uint16_t*)x, rows, dim, ldx, eps, rrms);
        else row_rrms_kernel<F16, NP><<<grid, block, 0, st>>>((const uint16_t*)x, rows, dim, ldx, eps, rrms);
    });
    FINO_CHECK(ok, FINO_ERR_UNSUPPORTED, "fino_row_rrms: dim %d > %d unsupported", dim, kMaxPasses * 512);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

static RmsRopeParts one_part(const void* weight, float eps, float out_scale, bool rope) {
    RmsRopeParts pp = {};
    pp.w[0] = (const uint16_t*)weight; pp.eps[0] = eps; pp.out_scale[0] = out_scale; pp.rope[0] = rope ? 1 : 0;
    return pp;
}

extern "C" int fino_rmsnorm_rope_scaled(void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                                        const float* cos_t, const float* sin_t, int head_dim, float out_scale, int dtype,
                                        void* stream) {
    return rmsnorm_rope_impl(x, rows, dim, ldx, 1, one_part(weight, eps, out_scale, cos_t != nullptr), cos_t, sin_t,
                             head_dim, nullptr, nullptr, nullptr, dtype, stream);
}

extern "C" int fino_rmsnorm_rope_scatter(const void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                                         const float* cos_t, const float* sin_t, int head_dim, float out_scale, void* out,
                                         const int64_t* head_off, const int64_t* head_ld, int dtype, void* stream) {
    FINO_CHECK(out && head_off && head_ld, FINO_ERR_ARG, "fino_rmsnorm_rope_scatter: null destination / table");
    return rmsnorm_rope_impl(const_cast<void*>(x), rows, dim, ldx, 1, one_part(weight, eps, out_scale, cos_t != nullptr),
                             cos_t, sin_t, head_dim, out, head_off, head_ld, dtype, stream);
}

extern "C" int fino_qkv_rmsnorm_rope(void* qkv, int64_t rows, int dim, int64_t ldx, const void* q_weight, float q_eps,
                                     const void* k_weight, float k_eps, const float* cos_t, const float* sin_t,
                                     int head_dim, float q_out_scale, void* out, const int64_t* head_off,
                                     const int64_t* head_ld, int dtype, void* stream) {
    FINO_CHECK((out == nullptr) == (head_off == nullptr) && (out == nullptr) == (head_ld == nullptr), FINO_ERR_ARG,
               "fino_qkv_rmsnorm_rope: out, head_off and head_ld go together");
    RmsRopeParts pp = {};
    pp.w[0] = (const uint16_t*)q_weight; pp.eps[0] = q_eps; pp.out_scale[0] = q_out_scale; pp.rope[0] = cos_t ? 1 : 0;
    pp.w[1] = (const uint16_t*)k_weight; pp.eps[1] = k_eps; pp.out_scale[1] = 1.0f; pp.rope[1] = cos_t ? 1 : 0;
    pp.w[2] = nullptr; pp.eps[2] = 0.f; pp.out_scale[2] = 1.0f; pp.rope[2] = 0;
    // in place: q and k only (v is left alone); scattered: v travels too, as a copy
    return rmsnorm_rope_impl(qkv, rows, dim, ldx, out ? 3 : 2, pp, cos_t, sin_t, head_dim, out, head_off, head_ld, dtype,
                             stream);
}

extern "C" int fino_rmsnorm_rope(void* x, int64_t rows, int dim, int64_t ldx, const void* weight, float eps,
                                 const float* cos_t, const float* sin_t, int head_dim, int dtype, void* stream) {
    return fino_rmsnorm_rope_scaled(x, rows, dim, ldx, weight, eps, cos_t, sin_t, head_dim, 1.0f, dtype, stream);
}

extern "C" int fino_headnorm_rope_scaled(void* x, int batch, int64_t rows, int heads, int head_dim, int64_t ldx,
                                         int64_t batch_stride, const void* w, const void* b, float eps,
                                         const float* cos_t, const float* sin_t, int64_t rope_row0, float out_scale,
                                         int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_headnorm_rope: dtype %d", dtype);
    FINO_CHECK(x && batch >= 0 && rows >= 0 && heads > 0 && out_scale > 0.f, FINO_ERR_ARG, "fino_headnorm_rope: bad shape");
    FINO_CHECK(head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 128, FINO_ERR_UNSUPPORTED,
               "fino_headnorm_rope: head_dim %d not in {16,32,64,128}", head_dim);
    FINO_CHECK((w == nullptr) == (b == nullptr) && (cos_t == nullptr) == (sin_t == nullptr), FINO_ERR_ARG,
               "fino_headnorm_rope: w/b and cos/sin come in pairs");
    FINO_CHECK(ldx % 8 == 0 && batch_stride % 8 == 0 && fino_aligned16(x) && fino_aligned16(w) && fino_aligned16(b) &&
                   fino_aligned16(cos_t) && fino_aligned16(sin_t),
               FINO_ERR_ARG, "fino_headnorm_rope: 16-byte alignment required");
    const int64_t total = (int64_t)batch * rows * heads * (head_dim / 8);
    if (total == 0) return FINO_OK;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        headnorm_rope_kernel<BF16><<<grid, 256, 0, st>>>((uint16_t*)x, batch, rows, heads, head_dim, ldx, batch_stride,
                                                         (const uint16_t*)w, (const uint16_t*)b, eps, cos_t, sin_t,
                                                         rope_row0, out_scale);
    else
        headnorm_rope_kernel<F16><<<grid, 256, 0, st>>>((uint16_t*)x, batch, rows, heads, head_dim, ldx, batch_stride,
                                                        (const uint16_t*)w, (const uint16_t*)b, eps, cos_t, sin_t,
                                                        rope_row0, out_scale);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_headnorm_rope(void* x, int batch, int64_t rows, int heads, int head_dim, int64_t ldx,
                                  int64_t batch_stride, const void* w, const void* b, float eps, const float* cos_t,
                                  const float* sin_t, int64_t rope_row0, int dtype, void* stream) {
    return fino_headnorm_rope_scaled(x, batch, rows, heads, head_dim, ldx, batch_stride, w, b, eps, cos_t, sin_t, rope_row0,
                                     1.0f, dtype, stream);
}

extern "C" int fino_patchify(const void* x, void* a, int channels, int frames, int height, int width, int pt, int ph,
                             int pw, int64_t lda, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_patchify: dtype %d", dtype);
    FINO_CHECK(x && a && channels > 0 && pt > 0 && ph > 0 && pw > 0, FINO_ERR_ARG, "fino_patchify: bad arguments");
    FINO_CHECK(frames % pt == 0 && height % ph == 0 && width % pw == 0, FINO_ERR_ARG,
               "fino_patchify: (%d,%d,%d) not divisible by patch (%d,%d,%d)", frames, height, width, pt, ph, pw);
    FINO_CHECK(lda >= (int64_t)channels * pt * ph * pw, FINO_ERR_ARG, "fino_patchify: lda too small");
    const int64_t total = (int64_t)channels * frames * height * width / pw;
    if (total == 0) return FINO_OK;
    patchify_kernel<<<grid_1d(total), 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, (uint16_t*)a, channels,
                                                                     frames, height, width, pt, ph, pw, lda);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_unpatchify(const void* y, void* out, int cout, int frames, int height, int width, int pt, int ph,
                               int pw, int64_t ldy, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_unpatchify: dtype %d", dtype);
    FINO_CHECK(y && out && cout > 0 && pt > 0 && ph > 0 && pw > 0, FINO_ERR_ARG, "fino_unpatchify: bad arguments");
    FINO_CHECK(frames % pt == 0 && height % ph == 0 && width % pw == 0, FINO_ERR_ARG,
               "fino_unpatchify: (%d,%d,%d) not divisible by patch (%d,%d,%d)", frames, height, width, pt, ph, pw);
    FINO_CHECK(ldy >= (int64_t)cout * pt * ph * pw, FINO_ERR_ARG, "fino_unpatchify: ldy too small");
    const int64_t total = (int64_t)cout * frames * height * width;
    if (total == 0) return FINO_OK;
    unpatchify_kernel<<<grid_1d(total), 256, 0, (hipStream_t)stream>>>((const uint16_t*)y, (uint16_t*)out, cout,
                                                                       frames, height, width, pt, ph, pw, ldy);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_wan_model_input(const float* lat, const float* cond, const float* id_lat, const float* traj,
                                    void* out, int channels, int gen_frames, int id_frames, int height, int width,
                                    int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_wan_model_input: dtype %d", dtype);
    FINO_CHECK(lat && cond && traj && out && (id_frames == 0 || id_lat), FINO_ERR_ARG,
               "fino_wan_model_input: null pointer");
    FINO_CHECK(channels > 0 && gen_frames > 0 && id_frames >= 0 && height > 0 && width > 0, FINO_ERR_ARG,
               "fino_wan_model_input: bad shape");
    const int64_t total = (int64_t)2 * channels * (gen_frames + id_frames) * height * width;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        wan_model_input_kernel<BF16><<<grid_1d(total), 256, 0, st>>>(lat, cond, id_lat, traj, (uint16_t*)out, channels,
                                                                     gen_frames, id_frames, height * width);
    else
        wan_model_input_kernel<F16><<<grid_1d(total), 256, 0, st>>>(lat, cond, id_lat, traj, (uint16_t*)out, channels,
                                                                    gen_frames, id_frames, height * width);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_cfg_vpred_step(const void* pred, void* lat, int64_t n_lat, int64_t batch_stride, const float* coef_dev,
                                   int has_uncond, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_cfg_vpred_step: dtype %d", dtype);
    FINO_CHECK(pred && lat && coef_dev && n_lat > 0 && (!has_uncond || batch_stride >= n_lat), FINO_ERR_ARG,
               "fino_cfg_vpred_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        cfg_vpred_step_kernel<BF16><<<grid_1d(n_lat), 256, 0, st>>>((const uint16_t*)pred, (uint16_t*)lat, n_lat,
                                                                   batch_stride, coef_dev, has_uncond);
    else
        cfg_vpred_step_kernel<F16><<<grid_1d(n_lat), 256, 0, st>>>((const uint16_t*)pred, (uint16_t*)lat, n_lat,
                                                                  batch_stride, coef_dev, has_uncond);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_cfg_dpm_step(const void* pred, void* lat, float* x0_old, const void* noise, int64_t n_lat,
                                 int64_t batch_stride, const float* coef_dev, int has_uncond, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_cfg_dpm_step: dtype %d", dtype);
    FINO_CHECK(pred && lat && x0_old && noise && coef_dev && n_lat > 0 && (!has_uncond || batch_stride >= n_lat),
               FINO_ERR_ARG, "fino_cfg_dpm_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        cfg_dpm_step_kernel<BF16><<<grid_1d(n_lat), 256, 0, st>>>((const uint16_t*)pred, (uint16_t*)lat, x0_old,
                                                                 (const uint16_t*)noise, n_lat, batch_stride, coef_dev,
                                                                 has_uncond);
    else
        cfg_dpm_step_kernel<F16><<<grid_1d(n_lat), 256, 0, st>>>((const uint16_t*)pred, (uint16_t*)lat, x0_old,
                                                                (const uint16_t*)noise, n_lat, batch_stride, coef_dev,
                                                                has_uncond);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_cfg_unipc_step(const void* cond_pred, const void* uncond_pred, float* x, float* last, float* m0,
                                   float* m1, int channels, int gen_frames, int total_frames, int height, int width,
                                   const float* coef_dev, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_cfg_unipc_step: dtype %d", dtype);
    FINO_CHECK(cond_pred && x && last && m0 && m1 && coef_dev, FINO_ERR_ARG, "fino_cfg_unipc_step: null pointer");
    FINO_CHECK(channels > 0 && gen_frames > 0 && total_frames >= gen_frames && height > 0 && width > 0, FINO_ERR_ARG,
               "fino_cfg_unipc_step: bad shape");
    const int64_t total = (int64_t)channels * gen_frames * height * width;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        cfg_unipc_kernel<BF16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)cond_pred, (const uint16_t*)uncond_pred,
                                                               x, last, m0, m1, channels, gen_frames, total_frames,
                                                               height * width, coef_dev);
    else
        cfg_unipc_kernel<F16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)cond_pred, (const uint16_t*)uncond_pred,
                                                              x, last, m0, m1, channels, gen_frames, total_frames,
                                                              height * width, coef_dev);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_cfg_euler_step(const void* cond_pred, const void* uncond_pred, float* lat, int channels,
                                   int gen_frames, int total_frames, int height, int width, float guidance,
                                   const float* dt_dev, int round_out, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_cfg_euler_step: dtype %d", dtype);
    FINO_CHECK(cond_pred && lat && dt_dev, FINO_ERR_ARG, "fino_cfg_euler_step: null pointer");
    FINO_CHECK(channels > 0 && gen_frames > 0 && total_frames >= gen_frames && height > 0 && width > 0, FINO_ERR_ARG,
               "fino_cfg_euler_step: bad shape");
    const int64_t total = (int64_t)channels * gen_frames * height * width;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        cfg_euler_kernel<BF16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)cond_pred,
                                                               (const uint16_t*)uncond_pred, lat, channels, gen_frames,
                                                               total_frames, height * width, guidance, dt_dev,
                                                               round_out);
    else
        cfg_euler_kernel<F16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)cond_pred, (const uint16_t*)uncond_pred,
                                                              lat, channels, gen_frames, total_frames, height * width,
                                                              guidance, dt_dev, round_out);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
