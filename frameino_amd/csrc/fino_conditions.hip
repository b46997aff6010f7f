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

}  // namespace

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
