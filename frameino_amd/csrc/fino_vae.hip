// HBM-bound kernels of the Wan 3D causal VAE (reference: architecture/autoencoder_kl_wan.py).  All activations are
// channels-last [T, H, W, Cpad] (Cpad = channels padded with zeros to a multiple of 64 so that every conv is an
// implicit GEMM over 128-byte channel chunks, see fino_gemm.hip CONV).  Padding channels stay exactly zero through
// every kernel here (gamma/bias pads are zero).
#include "fino_common.h"

namespace {

// ---- storage of an activation element: bf16 / fp16 (16 bits) or -- FINO_F32, the VAE computing like the fp32 the reference
// app runs it in (app.py:157) -- fp32.  The kernels below are written once against this trait; the index arithmetic of the
// reference's rearrangements exists in one place.
struct F32 {};
template <typename T> struct Elem {
    typedef uint16_t type;
    static __device__ __forceinline__ float ld(const uint16_t* p) { return T::to_f32(*p); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = T::from_f32(v); }
    static __device__ __forceinline__ void ld8(const uint16_t* p, float (&v)[8]) { unpack8<T>(*reinterpret_cast<const uint4*>(p), v); }
    static __device__ __forceinline__ void st8(uint16_t* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack8<T>(v); }
};
template <> struct Elem<F32> {
    typedef float type;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// ---- split-bf16 planes of an fp32 value: x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (each
// difference exact in fp32; 3 x 8 significant bits).  Products of the expansions truncated to the terms >= 2^-16 relative --
// (hi,hi) (hi,mid) (hi,lo) (mid,hi) (mid,mid) (lo,hi), or (hi,hi) (hi,lo) (lo,hi) with two planes -- are exact in the MFMA's
// fp32 accumulator: the matrix pipe computes an fp32-faithful product at 1/6 (1/3) of its bf16 rate, against 1/16 for the
// fp32-input MFMA (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at the vector rate).
__device__ __forceinline__ void split3(float x, uint16_t (&pl)[3]) {
    pl[0] = BF16::from_f32(x);
    const float r1 = x - BF16::to_f32(pl[0]);
    pl[1] = BF16::from_f32(r1);
    pl[2] = BF16::from_f32(r1 - BF16::to_f32(pl[1]));
}

// out[row][s * cols + c] = plane seg_plane(s) of x[row][c], s < nseg: `planes_packed` holds seg_plane(s) in 2 bits each.
//   A operand of fino_conv3d_split: planes side by side, nseg = planes, packed = 0b100100 (0, 1, 2)
//   A operand of a plain GEMM (fino_gemm + FINO_EPI_F32): one plane per product, (0,0,0,1,1,2) or (0,0,1)
//   W operand: the partner plane of each product, (0,1,2,0,1,0) or (0,1,0)
// One thread = 8 consecutive columns of one row: 32 bytes read, nseg x 16 bytes written.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int64_t rows,
                                                         int cols, int64_t ldx, int64_t ldo, int nseg, uint32_t planes_packed) {
    const int chunks = cols >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % chunks) * 8;
        const int64_t row = i / chunks;
        float v[8];
        Elem<F32>::ld8(x + row * ldx + c0, v);
        uint32_t w[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint16_t a[3], b[3];
            split3(v[2 * j], a);
            split3(v[2 * j + 1], b);
#pragma unroll
            for (int q = 0; q < 3; ++q) w[q][j] = (uint32_t)a[q] | ((uint32_t)b[q] << 16);
        }
        for (int s = 0; s < nseg; ++s) {
            const int q = (planes_packed >> (2 * s)) & 3;
            const uint4 o = q == 0 ? make_uint4(w[0][0], w[0][1], w[0][2], w[0][3])
                          : (q == 1 ? make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]) : make_uint4(w[2][0], w[2][1], w[2][2], w[2][3]));
            *reinterpret_cast<uint4*>(out + row * ldo + (int64_t)s * cols + c0) = o;
        }
    }
}

// WanRMS_norm (:201-202) + optional SiLU on fp32 activations, all of it in fp32 as the reference's fp32 VAE computes it
// (F.normalize(x, dim=1) * scale * gamma: the quotient first, then the two products; SiLU = t * sigmoid(t) by an IEEE division);
// the result leaves as fp32 (nseg = 0) or directly as the split planes the next convolution reads (split_bf16_kernel's layouts).
// One wave per position (rows of <= 1024 channels stay in registers between the statistic and the apply).
__global__ __launch_bounds__(256) void rmsnorm_silu_cl_f32_kernel(const float* __restrict__ x, float* __restrict__ y32,
                                                                  uint16_t* __restrict__ ysp, int64_t rows, int cpad,
                                                                  float sqrt_c, const float* __restrict__ gamma, int silu,
                                                                  int nseg, uint32_t planes_packed) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* px = x + row * cpad;
    float ss = 0.f;
    for (int c = lane * 4; c < cpad; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(px + c);
        ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
    }
    ss = wave_sum(ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane * 4; c < cpad; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(px + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float in[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = in[j] / den * sqrt_c * gg[j];
            o[j] = silu ? t / (1.0f + expf(-t)) : t;
        }
        if (nseg == 0) {
            *reinterpret_cast<float4*>(y32 + row * cpad + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            uint16_t a[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(o[j], a[j]);
            for (int s = 0; s < nseg; ++s) {
                const int q = (planes_packed >> (2 * s)) & 3;
                uint16_t e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = q == 0 ? a[j][0] : (q == 1 ? a[j][1] : a[j][2]);      // (no dynamic register index)
                const uint32_t w0 = (uint32_t)e[0] | ((uint32_t)e[1] << 16), w1 = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
                *reinterpret_cast<uint2*>(ysp + row * ((int64_t)nseg * cpad) + (int64_t)s * cpad + c) = make_uint2(w0, w1);
            }
        }
    }
}

// WanRMS_norm (:201-202) + optional SiLU: y = act( x / max(||x||_2, 1e-12) * sqrt(C) * gamma ).  A position's row of
// `cpad` channels is LPR = min(64, cpad/8) lanes of 16 bytes (looped when cpad > 512), so a wave handles 64 / LPR
// positions at once (2 at 256 channels -- the widths of the decoder's largest stages -- 8 at 64) and the sum of
// squares is reduced inside each LPR-lane group.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void rmsnorm_silu_cl_kernel(const uint16_t* __restrict__ x,
                                                              uint16_t* __restrict__ y, int64_t rows, int cpad,
                                                              float sqrt_c, const float* __restrict__ gamma,
                                                              int silu) {
    constexpr int kRowsPerWave = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kRowsPerWave + lane / LPR;
    const bool live = row < rows;
    const uint16_t* px = x + (live ? row : 0) * cpad;
    float ss = 0.f;
    if (live)
        for (int c = sub * 8; c < cpad; c += LPR * 8) {
            float v[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(px + c), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
        }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (!live) return;
    const float inv = sqrt_c / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = sub * 8; c < cpad; c += LPR * 8) {
        float v[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(px + c), v);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c);
        const float4 g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[j] * inv * g[j];
            o[j] = silu ? t / (1.0f + __expf(-t)) : t;
        }
        *reinterpret_cast<uint4*>(y + row * cpad + c) = pack8<T>(o);
    }
}

// in-place row softmax of scale*s (fp32 math), rows of n valid columns with leading dimension ld
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(typename Elem<T>::type* __restrict__ s, int64_t rows, int n, int64_t ld,
                                                           float scale) {
    typedef Elem<T> E;
    constexpr bool kF32 = sizeof(typename E::type) == 4;       // fp32 rows: accurate expf and a division, as torch's fp32 softmax
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    typename E::type* p = s + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, E::ld(p + c));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) sum += kF32 ? expf((E::ld(p + c) - mx) * scale) : __expf((E::ld(p + c) - mx) * scale);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < n; c += 64) {
        if (kF32) E::st(p + c, expf((E::ld(p + c) - mx) * scale) / sum);
        else E::st(p + c, __expf((E::ld(p + c) - mx) * scale) * inv);
    }
}

// out = main + DupUp3D(x) (:90-131, whole-sequence: frame 0 keeps its last temporal copy).  channels-last.
//   out[to, ho, wo, c] = main[...] + x[t, ho/fs, wo/fs, (((c*ft + a)*fs + b)*fs + d) / rep]
// One thread = 8 consecutive output channels (16-byte main load / store); their 8 source channels k/rep are one 16-byte
// load when they are consecutive too (rep == ft*fs*fs: equal widths), scalar gathers from a 16..32-byte window otherwise.
template <typename T>
__global__ __launch_bounds__(256) void dup_up3d_add_kernel(const typename Elem<T>::type* __restrict__ mainp,
                                                           const typename Elem<T>::type* __restrict__ x,
                                                           typename Elem<T>::type* __restrict__ out,
                                                           int t_out, int h_out, int w_out, int c_out, int c_out_pad,
                                                           int h_in, int w_in, int c_in_pad, int ft, int fs, int rep) {
    typedef Elem<T> E;
    const int chunks = c_out_pad >> 3;
    const int64_t total = (int64_t)t_out * h_out * w_out * chunks;
    const int factor = ft * fs * fs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % chunks) * 8;
        const int64_t pos = i / chunks;
        const int wo = (int)(pos % w_out);
        const int ho = (int)((pos / w_out) % h_out);
        const int to = (int)(pos / ((int64_t)w_out * h_out));
        float v[8];
        E::ld8(mainp + pos * c_out_pad + c0, v);
        int t, a;
        if (to == 0) { t = 0; a = ft - 1; } else { t = 1 + (to - 1) / ft; a = (to - 1) % ft; }
        const typename E::type* xr = x + (((int64_t)t * h_in + ho / fs) * w_in + wo / fs) * c_in_pad;
        const int sub = (a * fs + (ho % fs)) * fs + (wo % fs);            // k = c * factor + sub
        if (rep == factor && c0 + 8 <= c_out) {
            float s[8];
            E::ld8(xr + c0, s);                                           // k / rep = c
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += s[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < c_out) v[j] += E::ld(xr + ((c0 + j) * factor + sub) / rep);
        }
        E::st8(out + pos * c_out_pad + c0, v);
    }
}

// out = main + AvgDown3D(x) (:37-87, whole-sequence: one zero frame in front when T is odd).  channels-last.
template <typename T>
__global__ __launch_bounds__(256) void avg_down3d_add_kernel(const typename Elem<T>::type* __restrict__ mainp,
                                                             const typename Elem<T>::type* __restrict__ x,
                                                             typename Elem<T>::type* __restrict__ out, int t_out, int h_out, int w_out,
                                                             int c_out, int c_out_pad, int t_in, int h_in, int w_in,
                                                             int c_in_pad, int ft, int fs, int group, int pad_t) {
    const int64_t total = (int64_t)t_out * h_out * w_out * c_out_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c_out_pad);
        const int64_t pos = i / c_out_pad;
        const int wo = (int)(pos % w_out);
        const int ho = (int)((pos / w_out) % h_out);
        const int to = (int)(pos / ((int64_t)w_out * h_out));
        float v = Elem<T>::ld(mainp + i);
        if (c < c_out) {
            float s = 0.f;
            for (int g = 0; g < group; ++g) {
                const int r = c * group + g;              // rearranged channel (cin, a, b, d)
                const int d = r % fs;
                const int b = (r / fs) % fs;
                const int a = (r / (fs * fs)) % ft;
                const int cin = r / (fs * fs * ft);
                const int t = to * ft + a - pad_t;
                if (t >= 0)
                    s += Elem<T>::ld(x + (((int64_t)t * h_in + ho * fs + b) * w_in + wo * fs + d) * c_in_pad + cin);
            }
            v += s / (float)group;
        }
        Elem<T>::st(out + i, v);
    }
}

// decoder tail: unpatchify (:935-952) + clamp(-1,1):  y [T, H, W, cpad] (c*p*p valid) -> out fp32 [C, T, H*p, W*p]
//   channel index = (c*p + q)*p + r  ->  out[c, t, h*p + r, w*p + q]
template <typename T>
__global__ __launch_bounds__(256) void vae_unpatchify_clamp_kernel(const typename Elem<T>::type* __restrict__ y,
                                                                   float* __restrict__ out, int t, int h, int w,
                                                                   int cpad, int c, int ps) {
    const int hp = h * ps, wp = w * ps;
    const int64_t total = (int64_t)c * t * hp * wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % wp);
        const int yy = (int)((i / wp) % hp);
        const int tt = (int)((i / ((int64_t)wp * hp)) % t);
        const int cc = (int)(i / ((int64_t)wp * hp * t));
        const int ch = (cc * ps + (x % ps)) * ps + (yy % ps);
        const float v = Elem<T>::ld(y + (((int64_t)tt * h + yy / ps) * w + x / ps) * cpad + ch);
        out[i] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

// encoder head: patchify (:912-932): x fp32 [C, T, H*p, W*p] -> y [T, H, W, cpad] (zero pad channels)
template <typename T>
__global__ __launch_bounds__(256) void vae_patchify_kernel(const float* __restrict__ x, typename Elem<T>::type* __restrict__ y, int t,
                                                           int h, int w, int cpad, int c, int ps) {
    const int hp = h * ps, wp = w * ps;
    const int64_t total = (int64_t)t * h * w * cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpad);
        const int64_t pos = i / cpad;
        const int ww = (int)(pos % w);
        const int hh = (int)((pos / w) % h);
        const int tt = (int)(pos / ((int64_t)w * h));
        float v = 0.f;
        if (ch < c * ps * ps) {
            const int r = ch % ps, q = (ch / ps) % ps, cc = ch / (ps * ps);
            v = x[(((int64_t)cc * t + tt) * hp + hh * ps + r) * wp + ww * ps + q];
        }
        Elem<T>::st(y + i, v);
    }
}

inline int grid_1d(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// PTR(type, p): the pointer argument cast to the storage type of the dispatched dtype (uint16_t or float)
#define VAE_DISPATCH3(KERNEL, GRID, ARGS_BF16, ARGS_F16, ARGS_F32)                              \
    do {                                                                                        \
        hipStream_t st_ = (hipStream_t)stream;                                                  \
        if (dtype == FINO_BF16) KERNEL<BF16><<<(GRID), 256, 0, st_>>> ARGS_BF16;                \
        else if (dtype == FINO_F16) KERNEL<F16><<<(GRID), 256, 0, st_>>> ARGS_F16;              \
        else KERNEL<F32><<<(GRID), 256, 0, st_>>> ARGS_F32;                                     \
        FINO_LAUNCH_CHECK();                                                                    \
    } while (0)
#define VAE_CHECK_DT(fn) FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, fn ": dtype %d", dtype)
#define VAE_CHECK_DT3(fn) \
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16 || dtype == FINO_F32, FINO_ERR_ARG, fn ": dtype %d", dtype)

extern "C" int fino_rmsnorm_silu_cl(const void* x, void* y, int64_t rows, int c_valid, int c_pad, const float* gamma,
                                    int silu, int dtype, void* stream) {
    VAE_CHECK_DT("fino_rmsnorm_silu_cl");
    FINO_CHECK(x && y && gamma && rows >= 0 && c_valid > 0 && c_pad >= c_valid && c_pad % 8 == 0, FINO_ERR_ARG,
               "fino_rmsnorm_silu_cl: bad arguments (c_valid=%d c_pad=%d)", c_valid, c_pad);
    FINO_CHECK(fino_aligned16(x) && fino_aligned16(y) && fino_aligned16(gamma), FINO_ERR_ARG,
               "fino_rmsnorm_silu_cl: 16-byte alignment required");
    if (rows == 0) return FINO_OK;
    hipStream_t st_ = (hipStream_t)stream;
    const int lpr = c_pad / 8 >= 64 ? 64 : c_pad / 8;
    const float sc = sqrtf((float)c_valid);
#define RMS_LAUNCH(L_)                                                                                              \
    {                                                                                                               \
        const unsigned grid = (unsigned)((rows + 4 * (64 / L_) - 1) / (4 * (64 / L_)));                             \
        if (dtype == FINO_BF16)                                                                                     \
            rmsnorm_silu_cl_kernel<BF16, L_><<<grid, 256, 0, st_>>>((const uint16_t*)x, (uint16_t*)y, rows, c_pad, sc, gamma, silu); \
        else                                                                                                        \
            rmsnorm_silu_cl_kernel<F16, L_><<<grid, 256, 0, st_>>>((const uint16_t*)x, (uint16_t*)y, rows, c_pad, sc, gamma, silu);  \
    }
    if (lpr == 64) RMS_LAUNCH(64)
    else if (lpr == 32) RMS_LAUNCH(32)
    else if (lpr == 16) RMS_LAUNCH(16)
    else if (lpr == 8) RMS_LAUNCH(8)
    else RMS_LAUNCH(64)                    /* rows that are not a power-of-two number of chunks: one row per wave */
#undef RMS_LAUNCH
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_softmax_rows(void* s, int64_t rows, int n, int64_t ld, float scale, int dtype, void* stream) {
    VAE_CHECK_DT3("fino_softmax_rows");
    FINO_CHECK(s && rows >= 0 && n > 0 && ld >= n, FINO_ERR_ARG, "fino_softmax_rows: bad arguments");
    if (rows == 0) return FINO_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    VAE_DISPATCH3(softmax_rows_kernel, grid, ((uint16_t*)s, rows, n, ld, scale), ((uint16_t*)s, rows, n, ld, scale),
                  ((float*)s, rows, n, ld, scale));
    return FINO_OK;
}

extern "C" int fino_dup_up3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in, int c_in,
                                 int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s, int dtype,
                                 void* stream) {
    VAE_CHECK_DT3("fino_dup_up3d_add");
    FINO_CHECK(main_in && x && out && t_in > 0 && factor_t >= 1 && factor_s >= 1, FINO_ERR_ARG,
               "fino_dup_up3d_add: bad arguments");
    const int factor = factor_t * factor_s * factor_s;
    FINO_CHECK((c_out * factor) % c_in == 0, FINO_ERR_ARG, "fino_dup_up3d_add: out_channels*factor %% in_channels");
    const int t_out = 1 + (t_in - 1) * factor_t;
    FINO_CHECK(c_out_pad % 8 == 0 && c_in_pad % 8 == 0, FINO_ERR_ARG, "fino_dup_up3d_add: padded widths %% 8");
    const int64_t total = (int64_t)t_out * h_in * factor_s * w_in * factor_s * (c_out_pad / 8);
#define DUP_ARGS(TY) ((const TY*)main_in, (const TY*)x, (TY*)out, t_out, h_in * factor_s, w_in * factor_s, c_out, c_out_pad, h_in, \
                      w_in, c_in_pad, factor_t, factor_s, c_out * factor / c_in)
    VAE_DISPATCH3(dup_up3d_add_kernel, grid_1d(total), DUP_ARGS(uint16_t), DUP_ARGS(uint16_t), DUP_ARGS(float));
#undef DUP_ARGS
    return FINO_OK;
}

extern "C" int fino_avg_down3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in,
                                   int c_in, int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s,
                                   int dtype, void* stream) {
    VAE_CHECK_DT3("fino_avg_down3d_add");
    FINO_CHECK(main_in && x && out && t_in > 0 && factor_t >= 1 && factor_s >= 1, FINO_ERR_ARG,
               "fino_avg_down3d_add: bad arguments");
    const int factor = factor_t * factor_s * factor_s;
    FINO_CHECK((c_in * factor) % c_out == 0 && h_in % factor_s == 0 && w_in % factor_s == 0, FINO_ERR_ARG,
               "fino_avg_down3d_add: shape not divisible");
    const int pad_t = (factor_t - t_in % factor_t) % factor_t;
    const int t_out = (t_in + pad_t) / factor_t;
    const int64_t total = (int64_t)t_out * (h_in / factor_s) * (w_in / factor_s) * c_out_pad;
#define AVG_ARGS(TY) ((const TY*)main_in, (const TY*)x, (TY*)out, t_out, h_in / factor_s, w_in / factor_s, c_out, c_out_pad, t_in, \
                      h_in, w_in, c_in_pad, factor_t, factor_s, c_in * factor / c_out, pad_t)
    VAE_DISPATCH3(avg_down3d_add_kernel, grid_1d(total), AVG_ARGS(uint16_t), AVG_ARGS(uint16_t), AVG_ARGS(float));
#undef AVG_ARGS
    return FINO_OK;
}

extern "C" int fino_vae_unpatchify_clamp(const void* y, float* out, int t, int h, int w, int c_pad, int channels,
                                         int patch, int dtype, void* stream) {
    VAE_CHECK_DT3("fino_vae_unpatchify_clamp");
    FINO_CHECK(y && out && t > 0 && h > 0 && w > 0 && patch >= 1 && channels * patch * patch <= c_pad, FINO_ERR_ARG,
               "fino_vae_unpatchify_clamp: bad arguments");
    const int64_t total = (int64_t)channels * t * h * patch * w * patch;
    VAE_DISPATCH3(vae_unpatchify_clamp_kernel, grid_1d(total), ((const uint16_t*)y, out, t, h, w, c_pad, channels, patch),
                  ((const uint16_t*)y, out, t, h, w, c_pad, channels, patch), ((const float*)y, out, t, h, w, c_pad, channels, patch));
    return FINO_OK;
}

extern "C" int fino_vae_patchify(const float* x, void* y, int t, int h, int w, int c_pad, int channels, int patch,
                                 int dtype, void* stream) {
    VAE_CHECK_DT3("fino_vae_patchify");
    FINO_CHECK(x && y && t > 0 && h > 0 && w > 0 && patch >= 1 && channels * patch * patch <= c_pad, FINO_ERR_ARG,
               "fino_vae_patchify: bad arguments");
    const int64_t total = (int64_t)t * h * w * c_pad;
    VAE_DISPATCH3(vae_patchify_kernel, grid_1d(total), (x, (uint16_t*)y, t, h, w, c_pad, channels, patch),
                  (x, (uint16_t*)y, t, h, w, c_pad, channels, patch), (x, (float*)y, t, h, w, c_pad, channels, patch));
    return FINO_OK;
}

extern "C" int fino_split_bf16(const float* x, void* out, int64_t rows, int cols, int64_t ldx, int64_t ldo, int nseg,
                               unsigned planes_packed, void* stream) {
    FINO_CHECK(x && out && rows >= 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldx % 4 == 0, FINO_ERR_ARG,
               "fino_split_bf16: bad arguments (cols=%d must be a multiple of 8)", cols);
    FINO_CHECK(nseg >= 1 && nseg <= 6 && ldo >= (int64_t)nseg * cols && ldo % 8 == 0, FINO_ERR_ARG,
               "fino_split_bf16: nseg=%d (1 .. 6), ldo must cover nseg * cols", nseg);
    for (int s_ = 0; s_ < nseg; ++s_)
        FINO_CHECK(((planes_packed >> (2 * s_)) & 3u) <= 2u, FINO_ERR_ARG, "fino_split_bf16: plane index 3 in segment %d", s_);
    FINO_CHECK(fino_aligned16(x) && fino_aligned16(out), FINO_ERR_ARG, "fino_split_bf16: 16-byte alignment required");
    if (rows == 0) return FINO_OK;
    split_bf16_kernel<<<grid_1d(rows * (cols / 8)), 256, 0, (hipStream_t)stream>>>(x, (uint16_t*)out, rows, cols, ldx, ldo, nseg,
                                                                                   planes_packed);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_rmsnorm_silu_cl_f32(const float* x, void* y, int64_t rows, int c_valid, int c_pad, const float* gamma,
                                        int silu, int nseg, unsigned planes_packed, void* stream) {
    FINO_CHECK(x && y && gamma && rows >= 0 && c_valid > 0 && c_pad >= c_valid && c_pad % 8 == 0, FINO_ERR_ARG,
               "fino_rmsnorm_silu_cl_f32: bad arguments (c_valid=%d c_pad=%d)", c_valid, c_pad);
    FINO_CHECK(nseg >= 0 && nseg <= 6, FINO_ERR_ARG, "fino_rmsnorm_silu_cl_f32: nseg=%d (0 = fp32 output, 1 .. 6 planes)", nseg);
    FINO_CHECK(fino_aligned16(x) && fino_aligned16(y) && fino_aligned16(gamma), FINO_ERR_ARG,
               "fino_rmsnorm_silu_cl_f32: 16-byte alignment required");
    if (rows == 0) return FINO_OK;
    rmsnorm_silu_cl_f32_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        x, (float*)y, (uint16_t*)y, rows, c_pad, sqrtf((float)c_valid), gamma, silu, nseg, planes_packed);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
