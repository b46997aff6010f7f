// Condition builders that sit in front of the denoising path: the trajectory video of
// data_loader/video_dataset_motion.py:120-206 (`prepare_traj_tensor`, called by app.py:616-620) built on the device
// instead of numpy + OpenCV on the host: coloured squares on a white canvas per frame, a 45x45 isotropic Gaussian blur
// (cv2.filter2D, BORDER_REFLECT_101), truncation to uint8 and the x/255*2-1 transform.  HBM-bound, trivially small.
#include "fino_common.h"

namespace {

inline unsigned grid_1d(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g > 262144 ? 262144 : g);
}

// canvas[f, c, y, x] (fp32, 0..255): white, then every point of frame f in order paints the square
// [y-r, y+r) x [x-r, x+r) clipped to the canvas (:153-160); points outside the canvas are skipped (:149-150).
// points: int32 [n, 4] = (x, y, r, g, b packed as r | g<<8 | b<<16) -> stored as (x, y, rgb, unused)
__global__ __launch_bounds__(256) void traj_paint_kernel(const int32_t* __restrict__ pts,
                                                         const int32_t* __restrict__ frame_off, float* __restrict__ out,
                                                         int frames, int height, int width, int radius) {
    const int64_t total = (int64_t)frames * height * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % width);
        const int y = (int)((i / width) % height);
        const int f = (int)(i / ((int64_t)width * height));
        int rgb = 0xffffff;
        for (int p = frame_off[f]; p < frame_off[f + 1]; ++p) {
            const int px = pts[4 * p], py = pts[4 * p + 1];
            if (px < 0 || px >= width || py < 0 || py >= height) continue;
            const int y0 = min(height, max(0, py - radius)), y1 = min(height, max(0, py + radius));
            const int x0 = min(width, max(0, px - radius)), x1 = min(width, max(0, px + radius));
            if (y >= y0 && y < y1 && x >= x0 && x < x1) rgb = pts[4 * p + 2];
        }
        const int64_t plane = (int64_t)height * width;
        float* o = out + (int64_t)f * 3 * plane + (int64_t)y * width + x;
        o[0] = (float)(rgb & 0xff);
        o[plane] = (float)((rgb >> 8) & 0xff);
        o[2 * plane] = (float)((rgb >> 16) & 0xff);
    }
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// one separable pass of the blur along x (AXIS 0) or y (AXIS 1); FINAL: truncate to uint8 (numpy astype) and map to
// [-1, 1] (:172, :39-42)
template <int AXIS, bool FINAL>
__global__ __launch_bounds__(256) void blur_pass_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ w, int taps, int planes, int height,
                                                        int width) {
    const int64_t total = (int64_t)planes * height * width;
    const int half = taps / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % width);
        const int y = (int)((i / width) % height);
        const float* base = in + (i - (int64_t)y * width - x);
        // blur the "ink" 255 - v (first pass converts, second pass converts back): the weights sum to 1, so this is
        // the same filter, but a white neighbourhood gives exactly 255 instead of 255 * (sum of rounded weights) -- which
        // the uint8 truncation would turn into 254 or 255 by the luck of the rounding
        float acc = 0.f;
        for (int k = 0; k < taps; ++k) {
            const int xx = AXIS == 0 ? reflect101(x + k - half, width) : x;
            const int yy = AXIS == 1 ? reflect101(y + k - half, height) : y;
            const float v = base[(int64_t)yy * width + xx];
            acc += w[k] * (AXIS == 0 ? 255.0f - v : v);
        }
        if (FINAL) {
            const float q = (float)(unsigned char)fminf(fmaxf(255.0f - acc, 0.f), 255.f);      // truncation toward zero
            acc = q / 255.0f * 2.0f - 1.0f;
        }
        out[i] = acc;
    }
}

// Pixel-area resampling (cv2.INTER_AREA's area relation) of an HWC uint8 image into the rectangle
// [off_y, off_y+rh) x [off_x, off_x+rw) of an HWC uint8 canvas; everything outside the rectangle is `fill`.
// app.py:303 (first frame -> resized_height x resized_width), :322-326 (pasted into the black inference canvas),
// :662-681 (ID reference: scaled, centred, zero-padded to the canvas).  Destination pixel (y, x) of the rectangle covers
// the source interval [y*sy, (y+1)*sy) x [x*sx, (x+1)*sx) (sy = src_h / rh, sx = src_w / rw); the value is the
// overlap-weighted mean of the source pixels, accumulated in fp32 and rounded half-to-even (cv::saturate_cast).
__global__ __launch_bounds__(256) void resize_area_pad_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                 int src_h, int src_w, int rh, int rw, int out_h,
                                                                 int out_w, int off_y, int off_x, int fill) {
    const int64_t total = (int64_t)out_h * out_w;
    const float sy = (float)src_h / (float)rh, sx = (float)src_w / (float)rw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_w) - off_x, y = (int)(i / out_w) - off_y;
        uint8_t* o = dst + i * 3;
        if (x < 0 || x >= rw || y < 0 || y >= rh) {
            o[0] = o[1] = o[2] = (uint8_t)fill;
            continue;
        }
        const float fy0 = y * sy, fy1 = fminf((y + 1) * sy, (float)src_h);
        const float fx0 = x * sx, fx1 = fminf((x + 1) * sx, (float)src_w);
        const int y0 = (int)floorf(fy0), y1 = min(src_h, (int)ceilf(fy1));
        const int x0 = (int)floorf(fx0), x1 = min(src_w, (int)ceilf(fx1));
        float acc[3] = {0.f, 0.f, 0.f}, area = 0.f;
        for (int yy = y0; yy < y1; ++yy) {
            const float wy = fminf(fy1, (float)(yy + 1)) - fmaxf(fy0, (float)yy);
            if (wy <= 0.f) continue;
            for (int xx = x0; xx < x1; ++xx) {
                const float wx = fminf(fx1, (float)(xx + 1)) - fmaxf(fx0, (float)xx);
                if (wx <= 0.f) continue;
                const uint8_t* p = src + ((int64_t)yy * src_w + xx) * 3;
                const float w = wy * wx;
                acc[0] += w * p[0]; acc[1] += w * p[1]; acc[2] += w * p[2];
                area += w;
            }
        }
        const float inv = area > 0.f ? 1.0f / area : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int q = __float2int_rn(acc[c] * inv);
            o[c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    }
}

// HWC uint8 -> CHW fp32 in [-1, 1] (app.py:109-113 train_transforms + :692 permute; also the pipeline's image input)
__global__ __launch_bounds__(256) void u8_hwc_to_chw_unit_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                 int h, int w) {
    const int64_t plane = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c * plane + i] = (float)src[i * 3 + c] / 255.0f * 2.0f - 1.0f;
    }
}

}  // namespace

extern "C" int fino_resize_area_pad_u8(const void* src, void* dst, int src_h, int src_w, int region_h, int region_w,
                                       int out_h, int out_w, int off_y, int off_x, int fill, void* stream) {
    FINO_CHECK(src && dst, FINO_ERR_ARG, "fino_resize_area_pad_u8: null pointer");
    FINO_CHECK(src_h > 0 && src_w > 0 && region_h > 0 && region_w > 0 && out_h > 0 && out_w > 0 && off_y >= 0 &&
                   off_x >= 0 && off_y + region_h <= out_h && off_x + region_w <= out_w && fill >= 0 && fill <= 255,
               FINO_ERR_ARG, "fino_resize_area_pad_u8: the resized region must lie inside the canvas");
    const int64_t total = (int64_t)out_h * out_w;
    resize_area_pad_u8_kernel<<<grid_1d(total), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)src, (uint8_t*)dst, src_h, src_w, region_h, region_w, out_h, out_w, off_y, off_x, fill);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_u8_hwc_to_chw_unit(const void* src, float* dst, int height, int width, void* stream) {
    FINO_CHECK(src && dst && height > 0 && width > 0, FINO_ERR_ARG, "fino_u8_hwc_to_chw_unit: bad arguments");
    u8_hwc_to_chw_unit_kernel<<<grid_1d((int64_t)height * width), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)src, dst, height, width);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_traj_paint(const int32_t* points, const int32_t* frame_offsets, float* canvas, int frames, int height,
                               int width, int radius, void* stream) {
    FINO_CHECK(points && frame_offsets && canvas, FINO_ERR_ARG, "fino_traj_paint: null pointer");
    FINO_CHECK(frames > 0 && height > 0 && width > 0 && radius >= 0, FINO_ERR_ARG, "fino_traj_paint: bad shape");
    const int64_t total = (int64_t)frames * height * width;
    traj_paint_kernel<<<grid_1d(total), 256, 0, (hipStream_t)stream>>>(points, frame_offsets, canvas, frames, height,
                                                                        width, radius);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_traj_blur_quantize(const float* canvas, float* scratch, float* out, const float* taps_dev, int taps,
                                       int planes, int height, int width, void* stream) {
    FINO_CHECK(canvas && scratch && out && taps_dev, FINO_ERR_ARG, "fino_traj_blur_quantize: null pointer");
    FINO_CHECK(taps > 0 && (taps & 1) && planes > 0 && height > 0 && width > 0, FINO_ERR_ARG,
               "fino_traj_blur_quantize: taps must be odd, shapes positive");
    const int64_t total = (int64_t)planes * height * width;
    hipStream_t st = (hipStream_t)stream;
    blur_pass_kernel<0, false><<<grid_1d(total), 256, 0, st>>>(canvas, scratch, taps_dev, taps, planes, height, width);
    FINO_LAUNCH_CHECK();
    blur_pass_kernel<1, true><<<grid_1d(total), 256, 0, st>>>(scratch, out, taps_dev, taps, planes, height, width);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
