// Shared definitions of the MFMA GEMM / implicit-GEMM conv kernels (fino_gemm.hip).
#pragma once
#include "fino_common.h"

#ifdef FINO_GEMM_STAMP
// diagnostic build only (make stamp): per-wave s_memtime deltas of workgroup 17: [wave][0..3] main-loop segments, [4] K-tiles,
// [5] prologue + loop, [6] epilogue until the staged tile is in LDS (residual loads issued, conversion, LDS writes, barrier),
// [7] epilogue store loop (LDS reads, residual / gate arithmetic, global stores issued)
__device__ unsigned long long fino_gemm_dbg[8 * 8];
#define STAMP(V_) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V_) :: "memory"); }
#endif

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
    int group_m;                     // tile rows per raster group (tile_coords); 0 = default
    // column split of the output (fino_gemm_split_n): tile columns at n0 >= n_split go to c2 (leading dimension ldc2,
    // column n_split = its column 0); n_split is a multiple of BN, 0 = one output
    uint16_t* c2;
    int64_t ldc2, n_split;
    // K-blocked A (fino_gemm_blocked_a; the ABLK kernels): columns [j * 64 a_tpb, (j + 1) * 64 a_tpb) of a row live at
    // a + j * a_blk_elems + row * lda -- the layout the heads all-to-all returns the attention output in ([peer][token]
    // [heads of that peer]); a_inv = ceil(65536 / a_tpb) turns the K-tile index into its block by a multiply and a shift
    int a_tpb, a_inv;
    int64_t a_blk_elems;
    // ... and two-level: K block b = j * a_groups + g lives at a + g * a_grp_elems + j * a_blk_elems (the heads travel in
    // a_groups groups, each group's all-to-all returning its own [peer][token][heads] buffer); a_ginv = ceil(65536 / a_groups)
    int a_groups, a_ginv;
    int64_t a_grp_elems;
    // implicit-GEMM convolution (CONV variant): A is a channels-last activation [T_in, H_in, W_in, lda]; row m of the
    // GEMM is output position (t, h, w); K runs tap-major, channel-minor (cin_chunks x 64 channels per tap).
    int to, ho, wo, ti, hi, wi;      // output / input extents
    int kt, kh, kw, st, sh, sw, pt, ph, pw, up, cin_chunks;
    const uint16_t* zero_page;       // >= 128 B of zeros: source of out-of-range taps (LDS-DMA cannot zero-fill)
    // split-bf16 convolution (fino_conv3d_split: the Wan VAE computing like fp32, reference app.py:157): A holds a_nplanes
    // bf16 planes of an fp32 activation side by side ([hi | lo] or [hi | mid | lo], a_cc 64-channel chunks each, lda =
    // a_nplanes * 64 * a_cc), W one weight plane per PRODUCT of the truncated expansion -- (hi,hi) (hi,lo) (lo,hi), or
    // (hi,hi) (hi,mid) (hi,lo) (mid,hi) (mid,mid) (lo,hi) -- and a tap's K walk visits the products in that order: the
    // A plane of product s is {0,0,1}[s] / {0,0,0,1,1,2}[s].  a_cc = 0: plain convolution (cin_chunks chunks per tap).
    int a_cc, a_nplanes;
    // MXFP8 output (QOUT epilogues of the fp8 kernel): e4m3 bytes [m, n] + scales in the fino_quantize_mxfp8 layout
    uint8_t* cq;
    uint8_t* cs;
    int64_t cs_rows_pad;
};

// ---- MXFP8 quantisation of 8 consecutive values held by one lane; 4 adjacent lanes form the 32-element block ----
// e = exponent with amax / 2^e in (224, 448]; returns the 8 e4m3 bytes (lo, hi words) and e through `e_out`.
__device__ __forceinline__ uint2 mx_quant8(const float (&v)[8], int& e_out) {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    int e = -127;
    if (amax > 0.f) {
        int ex;
        (void)frexpf(amax * (1.0f / 448.0f), &ex);      // amax/448 = f * 2^ex, f in [0.5, 1)
        e = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
    }
    const float inv = __builtin_amdgcn_ldexpf(1.0f, -e);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, w1, true);
    e_out = e;
    return make_uint2((uint32_t)w0, (uint32_t)w1);
}

// index of the e8m0 scale of (row, 32-column block starting at col) in the [cols/128][rows_pad/256][1024] layout
__device__ __forceinline__ int64_t mx_scale_index(int64_t row, int64_t col, int64_t rows_pad) {
    const int rr = (int)(row & 255);
    return ((col >> 7) * (rows_pad >> 8) + (row >> 8)) * 1024 + ((col >> 5) & 3) * 256 + (rr & 15) * 16 + (rr >> 4);
}

__device__ __forceinline__ float gelu_tanh_f32(float x) {
    // 0.5*x*(1+tanh(u)) == x*sigmoid(2u),  u = sqrt(2/pi)*(x + 0.044715 x^3)
    // = x / (1 + 2^(-x*(c1 + c3*x^2))),  c1 = 2*sqrt(2/pi)*log2(e), c3 = 0.044715*c1: 7 VALU ops with v_exp_f32 and
    // v_rcp_f32 (1 ulp) instead of an IEEE division (~10 more ops) -- the result is rounded to bf16/fp16 right after.
    const float c1 = 2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c3 = 0.044715f * c1;
    const float e = __builtin_amdgcn_exp2f(-x * (c3 * x * x + c1));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Raster group height of the plain GEMM (tile_coords), from the measured A/B of profiles/r02_gemm_raster.md: an XCD's 32
// concurrent tiles form a G-row x 32/G-column window.  L2-fill traffic is lowest at G = 4 / 8 (12 operand panels per 32
// tiles) but costs little (2.5x the traffic = +8 % time), and what wins per shape is the panel LENGTH: long-K GEMMs
// prefer G = 1 (a window is whole output rows: one A panel and every W panel stream once), wide-N ones G = 2, and
// N = 3072 (12 tile columns) G = 16.
inline int gemm_default_group_m(int tiles_n, int64_t k) {
    if (k >= 8192) return 1;
    if (tiles_n >= 48) return 2;
    if (tiles_n <= 12) return 16;
    return 4;
}

// swizzle of the 16-byte chunk index inside a 128-byte tile row
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// ================= device pieces shared by the bf16/fp16 kernels (fino_gemm.hip) and the MXFP8 kernel =================
// ---- tile id -> (tm, tn): contiguous id range per XCD, 4-tile-high groups inside (operand reuse in that XCD's L2) ----
// blocks b, b + 8, ... share an XCD: give each XCD a contiguous range of the `nwg` ids (bijective for any nwg)
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
}
// linear raster id -> (tm, tn): groups of GROUP_M tile rows, column-major inside a group
__device__ __forceinline__ void tile_raster(const GemmParams& p, int id, int& tm, int& tn) {
    const int GROUP_M = (p.group_m & 0xff) > 0 ? (p.group_m & 0xff) : 4;
    const int group = id / (GROUP_M * p.tiles_n);
    const int first_m = group * GROUP_M;
    const int gsz = (p.tiles_m - first_m) < GROUP_M ? (p.tiles_m - first_m) : GROUP_M;
    const int in_group = id - group * GROUP_M * p.tiles_n;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
    if (p.group_m & 0x100) tm = p.tiles_m - 1 - tm;      // tile rows last to first (FINO_TUNE_GEMM_RASTER = 1: A/B knob)
}
__device__ __forceinline__ void tile_coords(const GemmParams& p, int& tm, int& tn) {
    tile_raster(p, xcd_remap((int)blockIdx.x, p.tiles_m * p.tiles_n), tm, tn);
}

// Shared by both main-loop variants.  Every wave must be past its last LDS operand read (barrier) before the call.
// MI = 16-row fragments per wave: the tile is 32 * MI rows high (256 by default; fino_gemm.hip picks lower tiles for row
// counts that 256-row tiles would spread badly over the CUs)
// The bias values of a lane's 4 x 4 output columns (fp32).  Issued FIRST in an epilogue: vector-memory results return in
// issue order, so whatever is loaded AFTER a long transfer (the residual tile: 128 KB per CU, ~14k cycles at the HBM rate)
// cannot be used before all of it has arrived, and the accumulator conversion needs the bias at once.  One branch per TILE
// (interior column block, 8-byte aligned bias): four 8-byte loads back to back and one wait; the per-element path for the
// ragged edge.
template <typename T>
__device__ __forceinline__ void gemm_load_bias(const GemmParams& p, int64_t n0, int lane, int wn, float (&bv)[4][4]) {
    const bool fast = p.bias && n0 + BN <= p.n && (reinterpret_cast<uintptr_t>(p.bias) & 7) == 0;      // tile-uniform
    if (fast) {
        uint2 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b4[j] = *reinterpret_cast<const uint2*>(p.bias + n0 + wn * 64 + j * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bv[j][0] = T::to_f32((uint16_t)(b4[j].x & 0xffffu)); bv[j][1] = T::to_f32((uint16_t)(b4[j].x >> 16));
            bv[j][2] = T::to_f32((uint16_t)(b4[j].y & 0xffffu)); bv[j][3] = T::to_f32((uint16_t)(b4[j].y >> 16));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int64_t gn = n0 + wn * 64 + j * 16 + (lane >> 4) * 4 + e;
                if (gn >= p.n) gn = p.n - 1;
                bv[j][e] = p.bias ? T::to_f32(p.bias[gn]) : 0.f;
            }
        }
    }
}

// fp32 output (FINO_EPI_F32 / FINO_EPI_F32_RESIDUAL): C = acc + bias [+ R] with bias, R and C in fp32 (GemmParams::bias / r / c
// hold float pointers, ldc / ldr count floats).  No LDS staging -- a 256-row fp32 tile is twice the LDS -- and none needed: a
// lane owns 4 consecutive columns of 16 rows per (i, j), one 16-byte access each (a row's 64 bytes come from 4 lanes).  These
// epilogues close the split-bf16 GEMMs, whose main loop is 3 or 6 times a bf16 one: the epilogue's share is a third / a sixth.
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_f32(f32x4_t (&acc)[MI][4], const GemmParams& p, int64_t m0, int64_t n0, int lane,
                                                  int wm, int wn) {
    const float* bias = reinterpret_cast<const float*>(p.bias);
    const float* r = reinterpret_cast<const float*>(p.r);
    float* c = reinterpret_cast<float*>(p.c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t gn = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        if (gn >= p.n) continue;                                  // (N is a multiple of 4: a lane's 4 columns are in or out together)
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = *reinterpret_cast<const float4*>(bias + gn);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t gm = m0 + wm * (16 * MI) + i * 16 + (lane & 15);
            if (gm >= p.m) continue;
            float4 y = make_float4(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
            if (EPI == FINO_EPI_F32_RESIDUAL) {
                const float4 rv = *reinterpret_cast<const float4*>(r + gm * p.ldr + gn);
                y.x += rv.x; y.y += rv.y; y.z += rv.z; y.w += rv.w;
            }
            *reinterpret_cast<float4*>(c + gm * p.ldc + gn) = y;
        }
    }
}

template <typename T, int EPI, bool QOUT, int MI>
__device__ __forceinline__ void gemm_epilogue_t(f32x4_t (&acc)[MI][4], const GemmParams& p, char* smem, int64_t m0,
                                                int64_t n0, int tid, int lane, int wm, int wn) {
    // ---- epilogue: y = T(acc + bias) [-> gelu] -> LDS tile -> whole-row global stores ----
    // lane holds n = wn*64 + j*16 + (lane>>4)*4 + e (e = 0..3), m = wm*16*MI + i*16 + (lane&15)
    constexpr bool kHasRes =
        EPI == FINO_EPI_RESIDUAL || EPI == FINO_EPI_GATED_RESIDUAL || EPI == FINO_EPI_GATED_RESIDUAL_STAGED;
    constexpr bool kGated = EPI == FINO_EPI_GATED_RESIDUAL || EPI == FINO_EPI_GATED_RESIDUAL_STAGED;
    constexpr int kIters = (32 * MI * BN / 8) / kThreads;      // 2 * MI row-chunks of 16 bytes per thread (16 at 256 rows)
#ifdef FINO_GEMM_STAMP
    unsigned long long te0, te1, te2;
    __builtin_amdgcn_sched_barrier(0); STAMP(te0) __builtin_amdgcn_sched_barrier(0);
#endif
    // ORDER OF ISSUE (round 4; stamps in profiles/r04_gemm_tile_stamps_{before,after}.txt).  Vector-memory results return in
    // issue order and a __syncthreads() waits for all of them, so with the residual tile (128 KB per CU, every CU at once:
    // ~14k cycles at the HBM rate) requested first, the accumulator conversion of a residual epilogue started 14.6k cycles
    // after the epilogue did (7.5k for the bias-only one) and its store loop ran on scratch spills.  Now: bias first; then
    // HALF of the residual rows (they fly under the conversion); the conversion and its LDS writes; a RAW barrier (LDS only);
    // the other half of the residual rows, once the 128 accumulator registers are dead; the store loop.  Same-box A/B
    // (profiles/r04_gemm_epilogue_ab.txt, M = 24640): out-projection 475 -> 438 us, FFN-down 1724 -> 1681 us, the six block
    // GEMMs 5735 -> 5635 us, the denoise step 286.8 -> 284.3 ms; no kernel of the family spills any more.
    float bv[4][4];
    gemm_load_bias<T>(p, n0, lane, wn, bv);
    __builtin_amdgcn_sched_barrier(0);
    uint4 rres[kHasRes ? kIters : 1];
    int rsel[kGated ? kIters : 1];
    constexpr int kFirst = MI == 8 ? kIters / 2 : kIters;      // lower tiles have the registers for all of it up front
    auto load_residual = [&](int it0, int it1) {
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            if (it < it0 || it >= it1) continue;
            const int idx = it * kThreads + tid;
            int64_t gm = m0 + (idx >> 5), gn = n0 + (idx & 31) * 8;
            gm = gm < p.m ? gm : p.m - 1;
            gn = gn < p.n ? gn : p.n - 8;
            rres[kHasRes ? it : 0] = *reinterpret_cast<const uint4*>(p.r + gm * p.ldr + gn);
            if (kGated) rsel[kGated ? it : 0] = p.sel ? p.sel[gm] : 0;
        }
    };
    if (kHasRes) load_residual(0, kFirst);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * (16 * MI) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = acc[i][j][e] + bv[j][e];
                if (EPI == FINO_EPI_GELU_TANH) y[e] = gelu_tanh_f32(round_to<T>(y[e]));
            }
            const uint32_t w0 = (uint32_t)T::from_f32(y[0]) | ((uint32_t)T::from_f32(y[1]) << 16);
            const uint32_t w1 = (uint32_t)T::from_f32(y[2]) | ((uint32_t)T::from_f32(y[3]) << 16);
            const int col = wn * 64 + j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<uint2*>(smem + row * kCsStride + col * 2) = make_uint2(w0, w1);
        }
    }
    // (a raw barrier behind the LDS writes: __syncthreads() would also wait for every residual load in flight)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kHasRes && kFirst < kIters) load_residual(kFirst, kIters);
    __builtin_amdgcn_sched_barrier(0);
#ifdef FINO_GEMM_STAMP
    __builtin_amdgcn_sched_barrier(0); STAMP(te1) __builtin_amdgcn_sched_barrier(0);
#endif
    int gsel = -1;
    float gg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int idx = it * kThreads + tid;
        const int row = idx >> 5;
        const int ch = idx & 31;
        const int64_t gm = m0 + row, gn = n0 + ch * 8;
        if (gm >= p.m || gn >= p.n) continue;
        uint4 yv = *reinterpret_cast<const uint4*>(smem + row * kCsStride + ch * 16);
        if (kHasRes) {
            float y[8], rv[8], o[8];
            unpack8<T>(yv, y);
            unpack8<T>(rres[kHasRes ? it : 0], rv);
            if (kGated) {
                // a thread keeps one 8-column chunk for all its rows, and the gate row changes with the timestep row
                // of the token (2 distinct rows in FrameINO): reload the 8 fp32 gates only when the selector changes
                // (was: 2 x float4 per row -- 256 KB of L1/L2 reads per tile, more than the residual itself)
                if (rsel[kGated ? it : 0] != gsel) {
                    gsel = rsel[kGated ? it : 0];
                    const float* g = p.gate + (int64_t)gsel * p.mod_stride + gn;
                    const float4 g0 = *reinterpret_cast<const float4*>(g);
                    const float4 g1 = *reinterpret_cast<const float4*>(g + 4);
                    gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w;
                    gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = rv[e] + (EPI == FINO_EPI_GATED_RESIDUAL_STAGED ? round_to<T>(y[e] * gg[e]) : y[e] * gg[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rv[e] + y[e];
            }
            yv = pack8<T>(o);
        }
        if constexpr (QOUT) {
            // the T-rounded result quantised in place of a separate pass: same bytes as fino_quantize_mxfp8 on C
            float y[8];
            unpack8<T>(yv, y);
            int e;
            const uint2 qv = mx_quant8(y, e);
            *reinterpret_cast<uint2*>(p.cq + gm * p.n + gn) = qv;
            if ((ch & 3) == 0) p.cs[mx_scale_index(gm, gn, p.cs_rows_pad)] = (uint8_t)(e + 127);
        } else if (p.n_split > 0 && n0 >= p.n_split) {
            *reinterpret_cast<uint4*>(p.c2 + gm * p.ldc2 + (gn - p.n_split)) = yv;
        } else {
            *reinterpret_cast<uint4*>(p.c + gm * p.ldc + gn) = yv;
        }
    }
#ifdef FINO_GEMM_STAMP
    __builtin_amdgcn_sched_barrier(0); STAMP(te2) __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 17 && (tid & 63) == 0) {
        fino_gemm_dbg[(tid >> 6) * 8 + 6] = te1 - te0;
        fino_gemm_dbg[(tid >> 6) * 8 + 7] = te2 - te1;
    }
#endif
}

template <typename T, int EPI, bool QOUT = false, int MI = 8>
__device__ __forceinline__ void gemm_epilogue(f32x4_t (&acc)[MI][4], const GemmParams& p, char* smem, int64_t m0,
                                              int64_t n0, int tid, int lane, int wm, int wn) {
    if constexpr (EPI == FINO_EPI_F32 || EPI == FINO_EPI_F32_RESIDUAL)
        gemm_epilogue_f32<EPI, MI>(acc, p, m0, n0, lane, wm, wn);
    else
        gemm_epilogue_t<T, EPI, QOUT, MI>(acc, p, smem, m0, n0, tid, lane, wm, wn);
}

}  // namespace fino_gemm_ns
