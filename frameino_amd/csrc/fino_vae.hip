// HBM-bound kernels of the Wan 3D causal VAE (reference: architecture/autoencoder_kl_wan.py).  All activations are
// channels-last [T, H, W, Cpad] (Cpad = channels padded with zeros to a multiple of 64 so that every conv is an
// implicit GEMM over 128-byte channel chunks, see fino_gemm.hip CONV).  Padding channels stay exactly zero through
// every kernel here (gamma/bias pads are zero).
#include "fino_common.h"

namespace {

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
__global__ __launch_bounds__(256) void softmax_rows_kernel(uint16_t* __restrict__ s, int64_t rows, int n, int64_t ld,
                                                           float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    uint16_t* p = s + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, T::to_f32(p[c]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) sum += __expf((T::to_f32(p[c]) - mx) * scale);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < n; c += 64) p[c] = T::from_f32(__expf((T::to_f32(p[c]) - mx) * scale) * inv);
}

// out = main + DupUp3D(x) (:90-131, whole-sequence: frame 0 keeps its last temporal copy).  channels-last.
//   out[to, ho, wo, c] = main[...] + x[t, ho/fs, wo/fs, (((c*ft + a)*fs + b)*fs + d) / rep]
// One thread = 8 consecutive output channels (16-byte main load / store); their 8 source channels k/rep are one 16-byte
// load when they are consecutive too (rep == ft*fs*fs: equal widths), scalar gathers from a 16..32-byte window otherwise.
template <typename T>
__global__ __launch_bounds__(256) void dup_up3d_add_kernel(const uint16_t* __restrict__ mainp,
                                                           const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                           int t_out, int h_out, int w_out, int c_out, int c_out_pad,
                                                           int h_in, int w_in, int c_in_pad, int ft, int fs, int rep) {
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
        unpack8<T>(*reinterpret_cast<const uint4*>(mainp + pos * c_out_pad + c0), v);
        int t, a;
        if (to == 0) { t = 0; a = ft - 1; } else { t = 1 + (to - 1) / ft; a = (to - 1) % ft; }
        const uint16_t* xr = x + (((int64_t)t * h_in + ho / fs) * w_in + wo / fs) * c_in_pad;
        const int sub = (a * fs + (ho % fs)) * fs + (wo % fs);            // k = c * factor + sub
        if (rep == factor && c0 + 8 <= c_out) {
            float s[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(xr + c0), s);      // k / rep = c
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += s[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < c_out) v[j] += T::to_f32(xr[((c0 + j) * factor + sub) / rep]);
        }
        *reinterpret_cast<uint4*>(out + pos * c_out_pad + c0) = pack8<T>(v);
    }
}

// out = main + AvgDown3D(x) (:37-87, whole-sequence: one zero frame in front when T is odd).  channels-last.
template <typename T>
__global__ __launch_bounds__(256) void avg_down3d_add_kernel(const uint16_t* __restrict__ mainp,
                                                             const uint16_t* __restrict__ x,
                                                             uint16_t* __restrict__ out, int t_out, int h_out, int w_out,
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
        float v = T::to_f32(mainp[i]);
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
                    s += T::to_f32(x[(((int64_t)t * h_in + ho * fs + b) * w_in + wo * fs + d) * c_in_pad + cin]);
            }
            v += s / (float)group;
        }
        out[i] = T::from_f32(v);
    }
}

// decoder tail: unpatchify (:935-952) + clamp(-1,1):  y [T, H, W, cpad] (c*p*p valid) -> out fp32 [C, T, H*p, W*p]
//   channel index = (c*p + q)*p + r  ->  out[c, t, h*p + r, w*p + q]
template <typename T>
__global__ __launch_bounds__(256) void vae_unpatchify_clamp_kernel(const uint16_t* __restrict__ y,
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
        const float v = T::to_f32(y[(((int64_t)tt * h + yy / ps) * w + x / ps) * cpad + ch]);
        out[i] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

// encoder head: patchify (:912-932): x fp32 [C, T, H*p, W*p] -> y [T, H, W, cpad] (zero pad channels)
template <typename T>
__global__ __launch_bounds__(256) void vae_patchify_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int t,
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
        y[i] = T::from_f32(v);
    }
}

inline int grid_1d(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define VAE_DISPATCH(KERNEL, GRID, ...)                                                         \
    do {                                                                                        \
        hipStream_t st_ = (hipStream_t)stream;                                                  \
        if (dtype == FINO_BF16) KERNEL<BF16><<<(GRID), 256, 0, st_>>>(__VA_ARGS__);             \
        else KERNEL<F16><<<(GRID), 256, 0, st_>>>(__VA_ARGS__);                                 \
        FINO_LAUNCH_CHECK();                                                                    \
    } while (0)
#define VAE_CHECK_DT(fn) FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, fn ": dtype %d", dtype)

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
    VAE_CHECK_DT("fino_softmax_rows");
    FINO_CHECK(s && rows >= 0 && n > 0 && ld >= n, FINO_ERR_ARG, "fino_softmax_rows: bad arguments");
    if (rows == 0) return FINO_OK;
    VAE_DISPATCH(softmax_rows_kernel, (unsigned)((rows + 3) / 4), (uint16_t*)s, rows, n, ld, scale);
    return FINO_OK;
}

extern "C" int fino_dup_up3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in, int c_in,
                                 int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s, int dtype,
                                 void* stream) {
    VAE_CHECK_DT("fino_dup_up3d_add");
    FINO_CHECK(main_in && x && out && t_in > 0 && factor_t >= 1 && factor_s >= 1, FINO_ERR_ARG,
               "fino_dup_up3d_add: bad arguments");
    const int factor = factor_t * factor_s * factor_s;
    FINO_CHECK((c_out * factor) % c_in == 0, FINO_ERR_ARG, "fino_dup_up3d_add: out_channels*factor %% in_channels");
    const int t_out = 1 + (t_in - 1) * factor_t;
    FINO_CHECK(c_out_pad % 8 == 0 && c_in_pad % 8 == 0, FINO_ERR_ARG, "fino_dup_up3d_add: padded widths %% 8");
    const int64_t total = (int64_t)t_out * h_in * factor_s * w_in * factor_s * (c_out_pad / 8);
    VAE_DISPATCH(dup_up3d_add_kernel, grid_1d(total), (const uint16_t*)main_in, (const uint16_t*)x, (uint16_t*)out,
                 t_out, h_in * factor_s, w_in * factor_s, c_out, c_out_pad, h_in, w_in, c_in_pad, factor_t, factor_s,
                 c_out * factor / c_in);
    return FINO_OK;
}

extern "C" int fino_avg_down3d_add(const void* main_in, const void* x, void* out, int t_in, int h_in, int w_in,
                                   int c_in, int c_in_pad, int c_out, int c_out_pad, int factor_t, int factor_s,
                                   int dtype, void* stream) {
    VAE_CHECK_DT("fino_avg_down3d_add");
    FINO_CHECK(main_in && x && out && t_in > 0 && factor_t >= 1 && factor_s >= 1, FINO_ERR_ARG,
               "fino_avg_down3d_add: bad arguments");
    const int factor = factor_t * factor_s * factor_s;
    FINO_CHECK((c_in * factor) % c_out == 0 && h_in % factor_s == 0 && w_in % factor_s == 0, FINO_ERR_ARG,
               "fino_avg_down3d_add: shape not divisible");
    const int pad_t = (factor_t - t_in % factor_t) % factor_t;
    const int t_out = (t_in + pad_t) / factor_t;
    const int64_t total = (int64_t)t_out * (h_in / factor_s) * (w_in / factor_s) * c_out_pad;
    VAE_DISPATCH(avg_down3d_add_kernel, grid_1d(total), (const uint16_t*)main_in, (const uint16_t*)x, (uint16_t*)out,
                 t_out, h_in / factor_s, w_in / factor_s, c_out, c_out_pad, t_in, h_in, w_in, c_in_pad, factor_t,
                 factor_s, c_in * factor / c_out, pad_t);
    return FINO_OK;
}

extern "C" int fino_vae_unpatchify_clamp(const void* y, float* out, int t, int h, int w, int c_pad, int channels,
                                         int patch, int dtype, void* stream) {
    VAE_CHECK_DT("fino_vae_unpatchify_clamp");
    FINO_CHECK(y && out && t > 0 && h > 0 && w > 0 && patch >= 1 && channels * patch * patch <= c_pad, FINO_ERR_ARG,
               "fino_vae_unpatchify_clamp: bad arguments");
    const int64_t total = (int64_t)channels * t * h * patch * w * patch;
    VAE_DISPATCH(vae_unpatchify_clamp_kernel, grid_1d(total), (const uint16_t*)y, out, t, h, w, c_pad, channels, patch);
    return FINO_OK;
}

extern "C" int fino_vae_patchify(const float* x, void* y, int t, int h, int w, int c_pad, int channels, int patch,
                                 int dtype, void* stream) {
    VAE_CHECK_DT("fino_vae_patchify");
    FINO_CHECK(x && y && t > 0 && h > 0 && w > 0 && patch >= 1 && channels * patch * patch <= c_pad, FINO_ERR_ARG,
               "fino_vae_patchify: bad arguments");
    const int64_t total = (int64_t)t * h * w * c_pad;
    VAE_DISPATCH(vae_patchify_kernel, grid_1d(total), x, (uint16_t*)y, t, h, w, c_pad, channels, patch);
    return FINO_OK;
}
