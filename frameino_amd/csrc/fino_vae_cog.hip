// HBM-bound kernels of the CogVideoX 3D causal VAE (diffusers `AutoencoderKLCogVideoX`, third-party: the VAE the
// reference's CogVideoX pipelines call, pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:380-396, :426-431, :809-826).
// Activations are channels-last [T, H, W, Cpad] like the Wan VAE's (fino_vae.hip); the convolutions are fino_conv3d.
//
//   GroupNorm(32, C, eps 1e-6) over one frame batch: statistics per group over (C/G channels x T x H x W).
//     pass 1  gn_partial_kernel : per-workgroup partial per-CHANNEL sum / sum of squares (fp32, fixed row partition ->
//                                 deterministic), one 16-byte chunk of 8 channels per lane
//     pass 2  gn_finalize_kernel: partials -> per-group mean / rstd (fp64) -> per-channel affine (a_c, b_c)
//     pass 3  gn_apply_kernel   : y = T(a_c x + b_c)  [ -> T(y * Y[z]) -> T(. + B[z]) ]  [ -> T(silu(.)) ]
//   where Y / B are the conv_y / conv_b outputs of CogVideoXSpatialNorm3D evaluated ONCE at latent resolution and read
//   through the nearest-neighbour index map of F.interpolate (first frame of an odd-length batch mapped separately),
//   instead of interpolating the latent to full resolution and running two 1x1x1 convolutions there.
#include "fino_common.h"

namespace {

constexpr int kGnBlocks = 1024;      // partial rows of the statistics workspace

// x [rows, cpad]; partial [kGnBlocks, 2, cpad] fp32.  blockDim = 256 = (cpad/8 chunk lanes) x (256 / (cpad/8) row lanes)
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* __restrict__ x, float* __restrict__ partial,
                                                         int64_t rows, int cpad) {
    __shared__ float red[2][256][8];
    const int chunks = cpad >> 3;
    const int rl = 256 / chunks;                    // row lanes per block (cpad <= 2048, power-of-two multiple of 64)
    const int ch = threadIdx.x % chunks, rr = threadIdx.x / chunks;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (rr < rl)
        for (int64_t r = r0 + rr; r < r1; r += rl) {
            float v[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(x + r * cpad + ch * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = s[j]; red[1][threadIdx.x][j] = q[j]; }
    __syncthreads();
    if (rr == 0) {
        for (int k = 1; k < rl; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += red[0][k * chunks + ch][j]; q[j] += red[1][k * chunks + ch][j]; }
        float* p = partial + (int64_t)blockIdx.x * 2 * cpad + ch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p[j] = s[j]; p[cpad + j] = q[j]; }
    }
}

// one block; thread c = channel.  ab [2, cpad]: a_c = rstd_g * gamma_c, b_c = beta_c - mean_g * a_c (0 for pad channels)
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ ab,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int nblocks, int cpad, int channels, int groups, double count,
                                                           float eps) {
    __shared__ double cs[2048], cq[2048];
    for (int c = threadIdx.x; c < cpad; c += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            s += (double)partial[(int64_t)b * 2 * cpad + c];
            q += (double)partial[(int64_t)b * 2 * cpad + cpad + c];
        }
        cs[c] = s; cq[c] = q;
    }
    __syncthreads();
    const int cg = channels / groups;
    for (int c = threadIdx.x; c < cpad; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        if (c < channels) {
            const int g = c / cg;
            double s = 0.0, q = 0.0;
            for (int k = g * cg; k < (g + 1) * cg; ++k) { s += cs[k]; q += cq[k]; }
            const double n = count * cg;
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            a = rstd * gamma[c];
            b = beta[c] - (float)mean * a;
        }
        ab[c] = a;
        ab[cpad + c] = b;
    }
}

// nearest-neighbour source index of F.interpolate(mode="nearest"): floor(dst * in / out), clamped
__device__ __forceinline__ int nearest_src(int dst, int n_in, int n_out) {
    const int s = (int)floorf((float)dst * ((float)n_in / (float)n_out));
    return s < n_in - 1 ? s : n_in - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       const float* __restrict__ ab, const uint16_t* __restrict__ my,
                                                       const uint16_t* __restrict__ mb, int t, int h, int w, int cpad,
                                                       int tz, int hz, int wz, int silu) {
    const int chunks = cpad >> 3;
    const int64_t total = (int64_t)t * h * w * chunks;
    const bool split = t > 1 && (t & 1);          // first frame of an odd-length batch is mapped on its own
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const int64_t row = i / chunks;
        float v[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + row * cpad + ch * 8), v);
        const float4 a0 = *reinterpret_cast<const float4*>(ab + ch * 8), a1 = *reinterpret_cast<const float4*>(ab + ch * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ab + cpad + ch * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(ab + cpad + ch * 8 + 4);
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = round_to<T>(a[j] * v[j] + b[j]);
        if (my) {
            const int xw = (int)(row % w);
            const int yh = (int)((row / w) % h);
            const int ft = (int)(row / ((int64_t)w * h));
            int zt;
            if (split) zt = ft == 0 ? 0 : 1 + nearest_src(ft - 1, tz - 1, t - 1);
            else zt = nearest_src(ft, tz, t);
            const int64_t zrow = ((int64_t)zt * hz + nearest_src(yh, hz, h)) * wz + nearest_src(xw, wz, w);
            float yy[8], bb[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(my + zrow * cpad + ch * 8), yy);
            unpack8<T>(*reinterpret_cast<const uint4*>(mb + zrow * cpad + ch * 8), bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = round_to<T>(round_to<T>(o[j] * yy[j]) + bb[j]);
        }
        if (silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] / (1.0f + __expf(-o[j]));
        }
        *reinterpret_cast<uint4*>(y + row * cpad + ch * 8) = pack8<T>(o);
    }
}

// CogVideoXDownsample3D's temporal compression: frame pairs averaged; the first frame of an odd-length batch kept
template <typename T>
__global__ __launch_bounds__(256) void avg_pool_time2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                             int t_in, int64_t frame_chunks) {
    const int odd = t_in & 1;
    const int t_out = odd + (t_in - odd) / 2;
    const int64_t total = (int64_t)t_out * frame_chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % frame_chunks;
        const int f = (int)(i / frame_chunks);
        uint4 out;
        if (odd && f == 0) {
            out = *reinterpret_cast<const uint4*>(x + c * 8);
        } else {
            const int s = odd ? 2 * f - 1 : 2 * f;
            float a[8], b[8], o[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(x + ((int64_t)s * frame_chunks + c) * 8), a);
            unpack8<T>(*reinterpret_cast<const uint4*>(x + ((int64_t)(s + 1) * frame_chunks + c) * 8), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (a[j] + b[j]) * 0.5f;
            out = pack8<T>(o);
        }
        *reinterpret_cast<uint4*>(y + i * 8) = out;
    }
}

// diffusers' blend_v / blend_h of overlapping VAE tiles (AutoencoderKLCogVideoX.tiled_encode / tiled_decode; the same loops as
// the in-tree architecture/autoencoder_kl_wan.py:1254-1268), in place on tile b from the last `extent` rows (axis 0) / columns
// (axis 1) of its upper / left neighbour a, channels-last tiles of one frame count and -- along the other axis -- one extent:
//     b[.., y, ..] = T( T(a[.., n_a - extent + y, ..] * (1 - y / extent)) + T(b[.., y, ..] * (y / extent)) )
// with the rounding points of the reference's T-typed tensor arithmetic (tensor x Python float -> T, then the sum -> T).
template <typename T>
__global__ __launch_bounds__(256) void blend_tiles_kernel(const uint16_t* __restrict__ a, uint16_t* __restrict__ b, int t,
                                                          int ha, int wa, int hb, int wb, int chunks, int extent, int axis) {
    // rows of b touched: axis 0 -> y < extent (all x), axis 1 -> x < extent (all y)
    const int nh = axis == 0 ? extent : hb, nw = axis == 0 ? wb : extent;
    const int64_t total = (int64_t)t * nh * nw * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        int64_t r = i / chunks;
        const int x = (int)(r % nw); r /= nw;
        const int y = (int)(r % nh);
        const int f = (int)(r / nh);
        const int k = axis == 0 ? y : x;
        const float wb_ = (float)((double)k / (double)extent), wa_ = (float)(1.0 - (double)k / (double)extent);
        const int ya = axis == 0 ? ha - extent + y : y, xa = axis == 0 ? x : wa - extent + x;
        float va[8], vb[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(a + ((((int64_t)f * ha + ya) * wa + xa) * chunks + c) * 8), va);
        uint16_t* pb = b + ((((int64_t)f * hb + y) * wb + x) * chunks + c) * 8;
        unpack8<T>(*reinterpret_cast<const uint4*>(pb), vb);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = round_to<T>(va[j] * wa_) + round_to<T>(vb[j] * wb_);
        *reinterpret_cast<uint4*>(pb) = pack8<T>(o);
    }
}

inline unsigned grid_1d(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g > 262144 ? 262144 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int64_t fino_groupnorm_workspace_bytes(int c_pad) {
    return c_pad > 0 ? ((int64_t)kGnBlocks * 2 + 2) * c_pad * 4 : 0;
}

extern "C" int fino_groupnorm_cl(const void* x, void* y, int t, int h, int w, int channels, int c_pad, int groups,
                                 const float* gamma, const float* beta, float eps, const void* mod_scale,
                                 const void* mod_shift, int tz, int hz, int wz, int silu, void* workspace,
                                 int64_t workspace_bytes, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_groupnorm_cl: dtype %d", dtype);
    FINO_CHECK(x && y && gamma && beta && workspace, FINO_ERR_ARG, "fino_groupnorm_cl: null pointer");
    FINO_CHECK(t > 0 && h > 0 && w > 0 && channels > 0 && c_pad >= channels && c_pad % 64 == 0 && c_pad <= 2048 &&
                   (c_pad & (c_pad - 1)) == 0 && groups > 0 && channels % groups == 0,
               FINO_ERR_ARG, "fino_groupnorm_cl: need c_pad a power of two in [64, 2048] and groups | channels");
    FINO_CHECK((mod_scale == nullptr) == (mod_shift == nullptr), FINO_ERR_ARG,
               "fino_groupnorm_cl: mod_scale and mod_shift come together");
    FINO_CHECK(!mod_scale || (tz > 0 && hz > 0 && wz > 0 && tz <= t && (t == 1 || !(t & 1) || tz > 1)), FINO_ERR_ARG,
               "fino_groupnorm_cl: bad modulation grid %d x %d x %d for %d frames", tz, hz, wz, t);
    FINO_CHECK(workspace_bytes >= fino_groupnorm_workspace_bytes(c_pad) && fino_aligned16(workspace), FINO_ERR_ARG,
               "fino_groupnorm_cl: workspace of %lld bytes needed", (long long)fino_groupnorm_workspace_bytes(c_pad));
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)t * h * w;
    float* partial = (float*)workspace;
    float* ab = partial + (int64_t)kGnBlocks * 2 * c_pad;
    const int nblocks = rows < kGnBlocks ? (int)rows : kGnBlocks;
    const int64_t total = rows * (c_pad / 8);
    if (dtype == FINO_BF16) {
        gn_partial_kernel<BF16><<<nblocks, 256, 0, st>>>((const uint16_t*)x, partial, rows, c_pad);
        FINO_LAUNCH_CHECK();
        gn_finalize_kernel<<<1, 1024, 0, st>>>(partial, ab, gamma, beta, nblocks, c_pad, channels, groups, (double)rows, eps);
        FINO_LAUNCH_CHECK();
        gn_apply_kernel<BF16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)x, (uint16_t*)y, ab,
                                                               (const uint16_t*)mod_scale, (const uint16_t*)mod_shift, t, h,
                                                               w, c_pad, tz, hz, wz, silu);
    } else {
        gn_partial_kernel<F16><<<nblocks, 256, 0, st>>>((const uint16_t*)x, partial, rows, c_pad);
        FINO_LAUNCH_CHECK();
        gn_finalize_kernel<<<1, 1024, 0, st>>>(partial, ab, gamma, beta, nblocks, c_pad, channels, groups, (double)rows, eps);
        FINO_LAUNCH_CHECK();
        gn_apply_kernel<F16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)x, (uint16_t*)y, ab,
                                                              (const uint16_t*)mod_scale, (const uint16_t*)mod_shift, t, h,
                                                              w, c_pad, tz, hz, wz, silu);
    }
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_avg_pool_time2(const void* x, void* y, int t_in, int h, int w, int c_pad, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_avg_pool_time2: dtype %d", dtype);
    FINO_CHECK(x && y && t_in > 1 && h > 0 && w > 0 && c_pad > 0 && c_pad % 8 == 0, FINO_ERR_ARG,
               "fino_avg_pool_time2: bad arguments (t_in must be > 1)");
    const int64_t fc = (int64_t)h * w * (c_pad / 8);
    const int t_out = (t_in & 1) + (t_in - (t_in & 1)) / 2;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        avg_pool_time2_kernel<BF16><<<grid_1d(t_out * fc), 256, 0, st>>>((const uint16_t*)x, (uint16_t*)y, t_in, fc);
    else
        avg_pool_time2_kernel<F16><<<grid_1d(t_out * fc), 256, 0, st>>>((const uint16_t*)x, (uint16_t*)y, t_in, fc);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_vae_blend_tiles(const void* a, void* b, int t, int h_a, int w_a, int h_b, int w_b, int c_pad, int extent,
                                    int axis, int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_vae_blend_tiles: dtype %d", dtype);
    FINO_CHECK(a && b && t > 0 && h_a > 0 && w_a > 0 && h_b > 0 && w_b > 0 && c_pad > 0 && c_pad % 8 == 0 &&
                   (axis == 0 || axis == 1) && fino_aligned16(a) && fino_aligned16(b),
               FINO_ERR_ARG, "fino_vae_blend_tiles: bad arguments");
    // diffusers clamps the extent to both tiles (blend_extent = min(a.shape, b.shape, blend_extent)); the other axis must agree
    const int lim = axis == 0 ? (h_a < h_b ? h_a : h_b) : (w_a < w_b ? w_a : w_b);
    const int ext = extent < lim ? extent : lim;
    FINO_CHECK(axis == 0 ? w_a == w_b : h_a == h_b, FINO_ERR_ARG,
               "fino_vae_blend_tiles: the tiles must agree along the axis that is not blended (%d x %d against %d x %d)", h_a, w_a,
               h_b, w_b);
    if (ext <= 0) return FINO_OK;
    const int64_t total = (int64_t)t * (axis == 0 ? ext : h_b) * (axis == 0 ? w_b : ext) * (c_pad / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        blend_tiles_kernel<BF16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)a, (uint16_t*)b, t, h_a, w_a, h_b, w_b,
                                                                 c_pad / 8, ext, axis);
    else
        blend_tiles_kernel<F16><<<grid_1d(total), 256, 0, st>>>((const uint16_t*)a, (uint16_t*)b, t, h_a, w_a, h_b, w_b, c_pad / 8,
                                                                ext, axis);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
