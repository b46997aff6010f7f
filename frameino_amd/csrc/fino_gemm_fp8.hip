// MXFP8 (OCP e4m3 + one e8m0 scale per 32 elements along K) linear layers for BASELINE config 5 ("fp8 MFMA path";
// there is no reference counterpart -- SURVEY F11 -- so parity is against this library's own bf16 path):
//   fino_quantize_mxfp8 : [rows, cols] bf16|fp16 -> e4m3 bytes + e8m0 block scales
//   fino_gemm_mxfp8     : C = epilogue(dequant(A) . dequant(W)^T + bias), fp32 accumulate, same fused epilogues
//
// The GEMM is the ping-pong kernel of fino_gemm.hip with ONE change in the matrix phase: the LDS image of a K-tile is
// byte-identical in shape (256 rows x 128 B, same swizzle, same LDS-DMA pieces), but 128 B of a row are now 128
// K-elements, and v_mfma_scale_f32_16x16x128_f8f6f4 consumes, per lane, exactly the two 16-byte chunks the bf16
// kernel reads for its two k-steps (probed on hardware, tools/fp8/mfma_scale_probe.hip: lane (row r, group g) holds
// k = 16g..16g+15 and 64+16g..64+16g+15; the scale operand of lane group g is the scale of K-block g).  One scaled MFMA
// replaces two bf16 MFMAs at the same 32 cycles: twice the K per tile at the same bytes, LDS traffic and MFMA time.
// Scales ride along as two extra 1-KiB DMA pieces per K-tile ([K/128][rows_pad][4] layout makes them contiguous).
#include <stdlib.h>

#include "fino_gemm_common.h"

using namespace fino_gemm_ns;

namespace {

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int kScaleBytes = 2048;                         // A scales 1 KiB + W scales 1 KiB per stage
constexpr int kStageF8 = kStageBytes + kScaleBytes;       // 66 KiB
constexpr int kSmemF8 = (2 * kStageF8 > BM * kCsStride) ? 2 * kStageF8 : BM * kCsStride;

struct Fp8Params {
    GemmParams g;                  // a / w = e4m3 bytes; lda / ldw in BYTES (= elements); k in elements
    const uint8_t* sa;             // [k/128][m_pad/256][1024]: inside a KiB [K-block g][row & 15][row >> 4]
    const uint8_t* sw;             // [k/128][n_pad/256][1024]
    int64_t m_pad, n_pad;
};

// ---------------------------------------------------------------------------------------------------- quantize
// lane: 8 consecutive elements; 4 lanes = one 32-element block; e = exponent with amax / 2^e in [224, 448]
template <typename T>
__global__ __launch_bounds__(256) void mxfp8_quantize_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                             uint8_t* __restrict__ scales, int64_t rows, int64_t cols,
                                                             int64_t ldx, int64_t rows_pad) {
    const int64_t chunks = cols >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / chunks;
        const int64_t c = (i - row * chunks) * 8;
        float v[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + row * ldx + c), v);
        int e;
        const uint2 qv = mx_quant8(v, e);
        *reinterpret_cast<uint2*>(q + row * cols + c) = qv;
        // scales of one (K-tile of 128, 256-row tile): 1 KiB ordered [K-block g][row & 15][row >> 4], so that the
        // 8 (A) / 4 (W) fragment scales a GEMM lane needs are one ds_read_b64 / ds_read_b32
        if ((c & 31) == 0) scales[mx_scale_index(row, c, rows_pad)] = (uint8_t)(e + 127);
    }
}

// ---------------------------------------------------------------------------------------------------- GEMM
template <typename T, int EPI, bool QOUT>
__global__ __launch_bounds__(kThreads, 2) void gemm_mxfp8_kernel(const Fp8Params fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams& p = fp.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;  // group
    const int wn = wave & 3;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // ---- LDS-DMA pieces (bytes; same piece map as gemm_pp_kernel) ----
    const int r8 = lane >> 3;
    const int skb = ((lane & 7) ^ ((4 * (wn & 1) + (lane >> 4)) & 7)) * 16;
    const uint8_t* a8 = reinterpret_cast<const uint8_t*>(p.a);
    const uint8_t* w8 = reinterpret_cast<const uint8_t*>(p.w);
    const int nk = (int)(p.k / 128);
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a8, 0, (int)((p.m - 1) * p.lda + p.k), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)w8, 0, (int)((p.n - 1) * p.ldw + p.k), 0x00020000);
    const __amdgpu_buffer_rsrc_t sa_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)fp.sa, 0, (int)(nk * fp.m_pad * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sw_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)fp.sw, 0, (int)(nk * fp.n_pad * 4), 0x00020000);
    uint32_t a_off[4], w_off[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int64_t gm = m0 + wm * 128 + q * 32 + wn * 8 + r8;
        gm = gm < p.m ? gm : p.m - 1;
        a_off[q] = (uint32_t)(gm * p.lda + skb);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int64_t gn = n0 + q * 32 + wn * 8 + r8;
        gn = gn < p.n ? gn : p.n - 1;
        w_off[q] = (uint32_t)(gn * p.ldw + skb);
    }
    const uint32_t sa_off = (uint32_t)(m0 * 4 + lane * 16), sw_off = (uint32_t)(n0 * 4 + lane * 16);
    const int sa_tile = (int)(fp.m_pad * 4), sw_tile = (int)(fp.n_pad * 4);
#define F8_DMA_A(STAGE_, KT_, Q_)                                                                                 \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                     \
        a_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageF8 + (wm * 128 + (Q_) * 32 + wn * 8) * 128), 16,         \
        a_off[Q_], (KT_) * 128, 0, 0);
#define F8_DMA_W(STAGE_, KT_, Q_)                                                                                 \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                     \
        w_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageF8 + kTileBytes + ((Q_) * 32 + wn * 8) * 128), 16,       \
        w_off[Q_], (KT_) * 128, 0, 0);
    // the K-tile's scales: wave 0 of group 0 brings A's 1 KiB, wave 1 of group 0 brings W's
#define F8_DMA_S(STAGE_, KT_)                                                                                     \
    {                                                                                                             \
        if (wave == 0)                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sa_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageF8 + kStageBytes), \
                                                     16, sa_off, (KT_) * sa_tile, 0, 0);                          \
        if (wave == 1)                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                             \
                sw_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageF8 + kStageBytes + 1024), 16, sw_off,           \
                (KT_) * sw_tile, 0, 0);                                                                           \
    }

    const int frow = lane & 15;
    const int g4 = lane >> 4;
    const int pch0 = g4 ^ (frow >> 1);
    const int a_base = (wm * 128 + frow) * 128;
    const int w_base = kTileBytes + (wn * 64 + frow) * 128;
    const int sa_base = kStageBytes + g4 * 256 + frow * 16 + wm * 8;            // 8 bytes: fragments i = 0..7
    const int sw_base = kStageBytes + 1024 + g4 * 256 + frow * 16 + wn * 4;     // 4 bytes: fragments j = 0..3

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0 and 1 whole
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        F8_DMA_A(0, 0, q)
        if (wm == 0) F8_DMA_W(0, 0, q) else F8_DMA_W(0, 0, 4 + q)
    }
    F8_DMA_S(0, 0)
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            F8_DMA_A(1, 1, q)
            if (wm == 0) F8_DMA_W(1, 1, q) else F8_DMA_W(1, 1, 4 + q)
        }
        F8_DMA_S(1, 1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();

    u32x4_t af[2][8], wf[2][4];
    uint32_t sa_pk[2], sw_pk;
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const char* sb = smem + cur * kStageF8;
        // ---------------- LOAD(t) ----------------
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pch = (pch0 ^ (kk << 2)) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[kk][j] = *reinterpret_cast<const u32x4_t*>(sb + w_base + j * 2048 + pch);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[kk][i] = *reinterpret_cast<const u32x4_t*>(sb + a_base + i * 2048 + pch);
        }
        {   // e8m0 scales of my (row, K-block g4): byte b of a register serves fragment 4*reg + b (op_sel)
            const uint2 sa2 = *reinterpret_cast<const uint2*>(sb + sa_base);
            sa_pk[0] = sa2.x;
            sa_pk[1] = sa2.y;
            sw_pk = *reinterpret_cast<const uint32_t*>(sb + sw_base);
        }
        if (t >= 1 && t + 1 < nk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) F8_DMA_A(cur ^ 1, t + 1, q)
            if (wm == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) F8_DMA_W(cur ^ 1, t + 1, q)
                F8_DMA_S(cur ^ 1, t + 1)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE(t): 32 scaled MFMAs (K = 128 each) from registers ----------------
#define F8_MMA(I_, J_)                                                                                            \
    {                                                                                                             \
        const i32x8_t wv_ = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, wf[0][J_]),                       \
                                                    __builtin_bit_cast(i32x4_t, wf[1][J_]), 0, 1, 2, 3, 4, 5, 6, 7); \
        const i32x8_t av_ = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, af[0][I_]),                       \
                                                    __builtin_bit_cast(i32x4_t, af[1][I_]), 0, 1, 2, 3, 4, 5, 6, 7); \
        acc[I_][J_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv_, av_, acc[I_][J_], 0, 0, (J_), sw_pk,  \
                                                                       (I_) & 3, sa_pk[(I_) >> 2]);               \
    }
        // serpentine over the columns: exactly one operand register set changes between consecutive MFMAs (as the bf16 loop,
        // fino_gemm.hip GP_ORDER 3: fewer operand switches draw less power)
#define F8_ROW(I_) F8_MMA(I_, 0) F8_MMA(I_, 1) F8_MMA(I_, 2) F8_MMA(I_, 3)
#define F8_WOR(I_) F8_MMA(I_, 3) F8_MMA(I_, 2) F8_MMA(I_, 1) F8_MMA(I_, 0)
        F8_ROW(0) F8_WOR(1) F8_ROW(2) F8_WOR(3) F8_ROW(4) F8_WOR(5) F8_ROW(6) F8_WOR(7)
#undef F8_ROW
#undef F8_WOR
#undef F8_MMA
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef F8_DMA_A
#undef F8_DMA_W
#undef F8_DMA_S
    gemm_epilogue<T, EPI, QOUT>(acc, p, smem, m0, n0, tid, lane, wm, wn);
}

template <typename T, int EPI, bool QOUT = false>
int launch_mxfp8(const Fp8Params& fp, hipStream_t st) {
    static FinoPerDeviceOnce once;
    if (int rc = fino_max_smem_once(once, reinterpret_cast<const void*>(&gemm_mxfp8_kernel<T, EPI, QOUT>), kSmemF8, "fino_gemm_mxfp8")) return rc;
    gemm_mxfp8_kernel<T, EPI, QOUT><<<dim3((unsigned)(fp.g.tiles_m * fp.g.tiles_n)), kThreads, kSmemF8, st>>>(fp);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

template <typename T>
int launch_mxfp8_e(const Fp8Params& fp, int epi, hipStream_t st) {
    switch (epi) {
        case FINO_EPI_NONE: return launch_mxfp8<T, FINO_EPI_NONE>(fp, st);
        case FINO_EPI_GELU_TANH: return launch_mxfp8<T, FINO_EPI_GELU_TANH>(fp, st);
        case FINO_EPI_RESIDUAL: return launch_mxfp8<T, FINO_EPI_RESIDUAL>(fp, st);
        case FINO_EPI_GATED_RESIDUAL_STAGED: return launch_mxfp8<T, FINO_EPI_GATED_RESIDUAL_STAGED>(fp, st);
        default: return launch_mxfp8<T, FINO_EPI_GATED_RESIDUAL>(fp, st);
    }
}

}  // namespace

extern "C" int64_t fino_mxfp8_scale_bytes(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0 || cols % 128) return 0;
    return (cols / 128) * ((rows + 255) / 256 * 256) * 4;
}

extern "C" int fino_quantize_mxfp8(const void* x, void* q, void* scales, int64_t rows, int64_t cols, int64_t ldx,
                                   int dtype, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_quantize_mxfp8: dtype %d", dtype);
    FINO_CHECK(x && q && scales, FINO_ERR_ARG, "fino_quantize_mxfp8: null pointer");
    FINO_CHECK(rows > 0 && cols > 0 && cols % 128 == 0 && ldx % 8 == 0 && ldx >= cols, FINO_ERR_ARG,
               "fino_quantize_mxfp8: cols must be a multiple of 128 (rows=%lld cols=%lld)", (long long)rows,
               (long long)cols);
    FINO_CHECK(fino_aligned16(x) && ((uintptr_t)q & 7) == 0, FINO_ERR_ARG, "fino_quantize_mxfp8: alignment");
    const int64_t rows_pad = (rows + 255) / 256 * 256;
    const int64_t total = rows * (cols / 8);
    int64_t grid = (total + 255) / 256;
    if (grid > 262144) grid = 262144;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16)
        mxfp8_quantize_kernel<BF16><<<(unsigned)grid, 256, 0, st>>>((const uint16_t*)x, (uint8_t*)q, (uint8_t*)scales,
                                                                    rows, cols, ldx, rows_pad);
    else
        mxfp8_quantize_kernel<F16><<<(unsigned)grid, 256, 0, st>>>((const uint16_t*)x, (uint8_t*)q, (uint8_t*)scales,
                                                                   rows, cols, ldx, rows_pad);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

static int mxfp8_fill(Fp8Params& fp, const void* aq, const void* a_scales, const void* wq, const void* w_scales,
                      const void* bias, int64_t m, int64_t n, int64_t k) {
    GemmParams& p = fp.g;
    p.a = (const uint16_t*)aq; p.w = (const uint16_t*)wq; p.bias = (const uint16_t*)bias;
    p.m = m; p.n = n; p.k = k; p.lda = k; p.ldw = k;
    p.tiles_m = (int)((m + BM - 1) / BM);
    p.tiles_n = (int)((n + BN - 1) / BN);
    p.group_m = fino_tune_get(FINO_TUNE_GEMM_GROUP_M);
    fp.sa = (const uint8_t*)a_scales; fp.sw = (const uint8_t*)w_scales;
    fp.m_pad = (m + 255) / 256 * 256; fp.n_pad = (n + 255) / 256 * 256;
    return FINO_OK;
}

extern "C" int fino_gemm_mxfp8_q(const void* aq, const void* a_scales, const void* wq, const void* w_scales,
                                 const void* bias, void* cq, void* c_scales, int64_t m, int64_t n, int64_t k,
                                 int epilogue, int bias_dtype, void* stream) {
    FINO_CHECK(bias_dtype == FINO_BF16 || bias_dtype == FINO_F16, FINO_ERR_ARG, "fino_gemm_mxfp8_q: dtype %d", bias_dtype);
    FINO_CHECK(aq && a_scales && wq && w_scales && cq && c_scales, FINO_ERR_ARG, "fino_gemm_mxfp8_q: null pointer");
    FINO_CHECK(m >= 0 && n > 0 && k > 0 && k % 128 == 0 && n % 128 == 0, FINO_ERR_ARG,
               "fino_gemm_mxfp8_q: K=%lld and N=%lld must be multiples of 128", (long long)k, (long long)n);
    FINO_CHECK(epilogue == FINO_EPI_NONE || epilogue == FINO_EPI_GELU_TANH, FINO_ERR_ARG,
               "fino_gemm_mxfp8_q: epilogue %d (NONE or GELU_TANH)", epilogue);
    FINO_CHECK(fino_aligned16(aq) && fino_aligned16(wq) && fino_aligned16(a_scales) && fino_aligned16(w_scales) &&
                   ((uintptr_t)cq & 7) == 0,
               FINO_ERR_ARG, "fino_gemm_mxfp8_q: alignment");
    FINO_CHECK(m * k < (1ll << 31) && n * k < (1ll << 31), FINO_ERR_UNSUPPORTED, "fino_gemm_mxfp8_q: operand > 2 GiB");
    if (m == 0) return FINO_OK;
    Fp8Params fp = {};
    mxfp8_fill(fp, aq, a_scales, wq, w_scales, bias, m, n, k);
    fp.g.cq = (uint8_t*)cq; fp.g.cs = (uint8_t*)c_scales; fp.g.cs_rows_pad = fp.m_pad;
    hipStream_t st = (hipStream_t)stream;
    if (bias_dtype == FINO_BF16)
        return epilogue == FINO_EPI_NONE ? launch_mxfp8<BF16, FINO_EPI_NONE, true>(fp, st)
                                         : launch_mxfp8<BF16, FINO_EPI_GELU_TANH, true>(fp, st);
    return epilogue == FINO_EPI_NONE ? launch_mxfp8<F16, FINO_EPI_NONE, true>(fp, st)
                                     : launch_mxfp8<F16, FINO_EPI_GELU_TANH, true>(fp, st);
}

extern "C" int fino_gemm_mxfp8(const void* aq, const void* a_scales, const void* wq, const void* w_scales,
                               const void* bias, void* c, int64_t m, int64_t n, int64_t k, int64_t ldc, int epilogue,
                               const void* r, int64_t ldr, const float* gate, int64_t mod_stride, const int32_t* sel,
                               int out_dtype, void* stream) {
    FINO_CHECK(out_dtype == FINO_BF16 || out_dtype == FINO_F16, FINO_ERR_ARG, "fino_gemm_mxfp8: out dtype %d", out_dtype);
    FINO_CHECK(aq && a_scales && wq && w_scales && c, FINO_ERR_ARG, "fino_gemm_mxfp8: null pointer");
    FINO_CHECK(m >= 0 && n > 0 && k > 0 && k % 128 == 0 && n % 8 == 0, FINO_ERR_ARG,
               "fino_gemm_mxfp8: K=%lld must be a multiple of 128, N=%lld of 8", (long long)k, (long long)n);
    FINO_CHECK(ldc % 8 == 0 && ldc >= n && fino_aligned16(aq) && fino_aligned16(wq) && fino_aligned16(c) &&
                   fino_aligned16(a_scales) && fino_aligned16(w_scales),
               FINO_ERR_ARG, "fino_gemm_mxfp8: alignment / leading dimension");
    FINO_CHECK(epilogue >= FINO_EPI_NONE && epilogue <= FINO_EPI_GATED_RESIDUAL_STAGED, FINO_ERR_ARG,
               "fino_gemm_mxfp8: epilogue %d", epilogue);
    if (epilogue >= FINO_EPI_RESIDUAL)
        FINO_CHECK(r && ldr % 8 == 0 && ldr >= n && fino_aligned16(r), FINO_ERR_ARG, "fino_gemm_mxfp8: residual operand");
    if (epilogue == FINO_EPI_GATED_RESIDUAL || epilogue == FINO_EPI_GATED_RESIDUAL_STAGED)
        FINO_CHECK(gate && fino_aligned16(gate) && mod_stride % 4 == 0, FINO_ERR_ARG, "fino_gemm_mxfp8: gate operand");
    FINO_CHECK(m * k < (1ll << 31) && n * k < (1ll << 31), FINO_ERR_UNSUPPORTED, "fino_gemm_mxfp8: operand > 2 GiB");
    if (m == 0) return FINO_OK;
    Fp8Params fp = {};
    GemmParams& p = fp.g;
    p.a = (const uint16_t*)aq; p.w = (const uint16_t*)wq; p.bias = (const uint16_t*)bias; p.c = (uint16_t*)c;
    p.r = (const uint16_t*)r; p.gate = gate; p.sel = sel;
    p.m = m; p.n = n; p.k = k; p.lda = k; p.ldw = k; p.ldc = ldc; p.ldr = ldr; p.mod_stride = mod_stride;
    p.tiles_m = (int)((m + BM - 1) / BM);
    p.tiles_n = (int)((n + BN - 1) / BN);
    p.group_m = fino_tune_get(FINO_TUNE_GEMM_GROUP_M);
    fp.sa = (const uint8_t*)a_scales; fp.sw = (const uint8_t*)w_scales;
    fp.m_pad = (m + 255) / 256 * 256; fp.n_pad = (n + 255) / 256 * 256;
    hipStream_t st = (hipStream_t)stream;
    return out_dtype == FINO_BF16 ? launch_mxfp8_e<BF16>(fp, epilogue, st) : launch_mxfp8_e<F16>(fp, epilogue, st);
}
