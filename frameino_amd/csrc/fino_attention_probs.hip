// Attention PROBABILITIES over a short key sequence: P = softmax(q.K^T) written out per head, for the text cross-attention of the
// Wan DiT once the zero-padded tail of the prompt is one key (fino_attn_fwd_tail): with 9 .. ~130 keys per head the
// product that follows, (P.V) W_o^T (architecture/transformer_wan.py:108 then :117), is cheaper re-associated as P.(V W_o^T) --
// V_h W_o,h^T is a per-prompt constant the model caches next to the text K / V, and the out-projection becomes a GEMM over
// K = heads x keys (1728 or 384) instead of 3072.  So this kernel stops at P: no online softmax (every key of a row is in
// registers at once), no P.V.
//
// One workgroup = 4 waves x 32 query rows of one (batch, head); the head's K rows (<= 128 x 256 B) are staged once into an
// XOR-swizzled LDS image (chunk c of row k at c ^ (k & 15): the 16 lanes a ds_read_b128 serves read 16 different rows, 16
// different bank groups).  S^T = K.Q^T on v_mfma_f32_32x32x16 with K as the A operand (the orientation of fino_attention.hip: the
// query sits on the lane, register j of lane (r, h) is key (j & 3) + 8 (j >> 2) + 4 h of a 32-key block), per-sample key counts
// and the logit offset of the last key as in attn_ppw_kernel<T, true>, softmax in fp32, P rounded ONCE to T.  Output layout
// [batch][row][head][kp] (kp = key columns per head, a multiple of 8; columns from a sample's key count on are written as zeros):
// the [rows, heads x kp] matrix the GEMM reads as its A operand.
#include <stdlib.h>

#include "fino_attention_common.h"
using namespace fino_attn_ns;

namespace {

struct ProbsParams {
    const uint16_t* q;
    const uint16_t* k;
    uint16_t* p;
    int batch, heads, lq, lk;          // lk: key rows allocated per sample (<= 32 NKB)
    int64_t q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, p_bs, p_rs;
    int kp;
    // q_rrms != nullptr: q is the RAW projection; row i of sample b is normalised while it is loaded, T(T(q rrms[b q_rrms_bs + i]) w)
    // with w = q_weight [heads x 128] -- the arithmetic and rounding points of fino_rmsnorm_rope (diffusers' RMSNorm)
    const float* q_rrms;
    const uint16_t* q_weight;
    int64_t q_rrms_bs;
    float scale_log2;
    int lk_b[4];
    float log2_mult[4];
};

template <typename T, int NKB>
__global__ __launch_bounds__(256) void attn_probs_kernel(const ProbsParams p) {
    typedef typename T::vec8 vec8;
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192];      // K image (NKB x 8 KiB), then 8 KiB of output rows per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int hb = blockIdx.y;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;

    // ---- the head's K rows -> LDS: only the 32-key blocks this sample has keys in (rows past its count inside the last block are
    //      whatever the caller left there -- finite -- and only ever meet the -inf mask) ----
    const int lkc = p.lk_b[bi < 4 ? bi : 3];
    const float lm = p.log2_mult[bi < 4 ? bi : 3];
    const int nkb = (lkc + 31) >> 5;                     // uniform over the workgroup
    const uint16_t* kp_ = p.k + bi * p.k_bs + head * p.k_hs;
    for (int i = 0; i < nkb * 2; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 4, ch = c & 15;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (row < p.lk) u = *reinterpret_cast<const uint4*>(kp_ + (int64_t)row * p.k_rs + ch * 8);
        *reinterpret_cast<uint4*>(smem + row * 256 + ((ch ^ (row & 15)) << 4)) = u;
    }
    // ---- Q fragments: lane (r, h) holds Q[row][16 ks + 8 h .. + 7] ----
    const int qrow = blockIdx.x * 128 + wave * 32 + r;
    const int qrc = qrow < p.lq ? qrow : p.lq - 1;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs + (int64_t)qrc * p.q_rs + 8 * h;
    vec8 qf[8];
    if (p.q_rrms) {
        const float rs = p.q_rrms[bi * p.q_rrms_bs + qrc];
        const uint16_t* wp = p.q_weight + head * 128 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float f[8], ww[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(qp + 16 * ks), f);
            unpack8<T>(*reinterpret_cast<const uint4*>(wp + 16 * ks), ww);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = round_to<T>(round_to<T>(f[j] * rs) * ww[j]);
            qf[ks] = __builtin_bit_cast(vec8, pack8<T>(f));
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(vec8, *reinterpret_cast<const uint4*>(qp + 16 * ks));
    }
    __syncthreads();

    f32x16_t acc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[kb][j] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb < nkb) {                                  // (blocks without keys: their accumulators stay 0 and are masked below)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int row = kb * 32 + r;
                const uint4 a = *reinterpret_cast<const uint4*>(smem + row * 256 + (((2 * ks + h) ^ (row & 15)) << 4));
                acc[kb] = T::mfma32(__builtin_bit_cast(vec8, a), qf[ks], acc[kb]);
            }
        }
    }

    // ---- softmax over the sample's keys (the last one stands for a run of identical keys: + log2 of its multiplicity) ----
    const float c2 = p.scale_log2;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = kb * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
            float x = acc[kb][j] * c2;
            x = key == lkc - 1 ? x + lm : x;
            x = key >= lkc ? -INFINITY : x;
            acc[kb][j] = x;
            m = fmaxf(m, x);
        }
    }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float e = __builtin_amdgcn_exp2f(acc[kb][j] - m);
            acc[kb][j] = e;
            l += e;
        }
    }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l;
    // ---- P rows leave as 16-byte pieces of whole rows: the wave's 32 rows x kp columns go through its own 8 KiB of the (now idle)
    //      K image -- the direct form is 8-byte stores that touch 32 cache lines per instruction ----
    __syncthreads();                                     // every wave has read its K fragments
    char* stage = smem + wave * 8192;                    // [32 rows][256 B]
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int key0 = kb * 32 + 8 * g4 + 4 * h;
            if (key0 < p.kp) {
                const uint32_t w0 = (uint32_t)T::from_f32(acc[kb][4 * g4 + 0] * inv) | ((uint32_t)T::from_f32(acc[kb][4 * g4 + 1] * inv) << 16);
                const uint32_t w1 = (uint32_t)T::from_f32(acc[kb][4 * g4 + 2] * inv) | ((uint32_t)T::from_f32(acc[kb][4 * g4 + 3] * inv) << 16);
                // 16-byte piece (key0 >> 3) of row r, XOR-swizzled by the row so that the 32 rows of a write spread over the banks
                *reinterpret_cast<uint2*>(stage + r * 256 + ((((key0 >> 3)) ^ (r & 15)) << 4) + ((key0 & 4) << 1)) = make_uint2(w0, w1);
            }
        }
    }
    __syncthreads();
    const int pieces = p.kp >> 3;                        // 16-byte pieces per row
    const int total = 32 * pieces;
    const int row_base = blockIdx.x * 128 + wave * 32;
    uint16_t* pbase = p.p + bi * p.p_bs + head * p.kp;
    for (int i = lane; i < total; i += 64) {
        const int rr = i / pieces, pc = i - rr * pieces;
        const uint4 u = *reinterpret_cast<const uint4*>(stage + rr * 256 + ((pc ^ (rr & 15)) << 4));
        if (row_base + rr < p.lq) *reinterpret_cast<uint4*>(pbase + (int64_t)(row_base + rr) * p.p_rs + pc * 8) = u;
    }
}

}  // namespace

// P = softmax(scale q.K^T) per head over a short key sequence (head_dim 128, lk <= 128 allocated key rows per sample,
// batch <= 4; q_rrms / q_weight: see ProbsParams -- NULL for a q that is already normalised): p [batch][lq][heads][kp] (strides p_bs / p_rs in elements, kp a multiple of 8 with max lk_b <= kp <= lk rounded up
// to 8; columns from lk_b[b] on are zeros).  lk_b / tail_mult as fino_attn_fwd_tail.
extern "C" int fino_attn_probs_supported(int batch, int heads, int64_t lq, int64_t lk, int head_dim) {
    return head_dim == 128 && batch >= 1 && batch <= 4 && heads > 0 && lq > 0 && lk >= 1 && lk <= 128 &&
           (int64_t)batch * heads <= 65535;
}

extern "C" int fino_attn_probs(const void* q, const void* k, void* p, int batch, int heads, int64_t lq, int64_t lk,
                               int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                               int kp, int64_t p_bs, int64_t p_rs, float scale, int dtype, const int* lk_b,
                               const float* tail_mult, const float* q_rrms, int64_t q_rrms_bs, const void* q_weight,
                               void* stream) {
    FINO_CHECK((q_rrms == nullptr) == (q_weight == nullptr) && fino_aligned16(q_weight), FINO_ERR_ARG,
               "fino_attn_probs: q_rrms and q_weight come together (q_weight 16-byte aligned)");
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_probs: dtype %d", dtype);
    FINO_CHECK(q && k && p && lk_b && tail_mult, FINO_ERR_ARG, "fino_attn_probs: null pointer");
    FINO_CHECK(fino_attn_probs_supported(batch, heads, lq, lk, head_dim), FINO_ERR_UNSUPPORTED,
               "fino_attn_probs: needs head_dim 128, batch <= 4, lk <= 128 (got head_dim %d, batch %d, lk %lld)", head_dim, batch,
               (long long)lk);
    FINO_CHECK(fino_aligned16(q) && fino_aligned16(k) && fino_aligned16(p) && q_rs % 8 == 0 && k_rs % 8 == 0 && q_hs % 8 == 0 &&
                   k_hs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && p_bs % 8 == 0 && p_rs % 8 == 0,
               FINO_ERR_ARG, "fino_attn_probs: pointers and strides must be 16-byte aligned");
    FINO_CHECK(kp > 0 && kp % 8 == 0 && kp <= 128 && p_rs >= (int64_t)heads * kp, FINO_ERR_ARG,
               "fino_attn_probs: kp %d (a multiple of 8, <= 128), row stride %lld", kp, (long long)p_rs);
    FINO_CHECK(scale > 0.f || scale == FINO_ATTN_SCALE_FOLDED, FINO_ERR_ARG, "fino_attn_probs: scale");
    if (lq == 0) return FINO_OK;
    ProbsParams pp;
    pp.q = (const uint16_t*)q; pp.k = (const uint16_t*)k; pp.p = (uint16_t*)p;
    pp.batch = batch; pp.heads = heads; pp.lq = (int)lq; pp.lk = (int)lk;
    pp.q_bs = q_bs; pp.q_rs = q_rs; pp.q_hs = q_hs; pp.k_bs = k_bs; pp.k_rs = k_rs; pp.k_hs = k_hs; pp.p_bs = p_bs; pp.p_rs = p_rs;
    pp.kp = kp;
    pp.q_rrms = q_rrms; pp.q_weight = (const uint16_t*)q_weight; pp.q_rrms_bs = q_rrms_bs;
    pp.scale_log2 = scale == FINO_ATTN_SCALE_FOLDED ? 1.0f : scale * 1.4426950408889634f;
    for (int b = 0; b < 4; ++b) {
        const int bb = b < batch ? b : batch - 1;
        FINO_CHECK(lk_b[bb] >= 1 && lk_b[bb] <= lk && lk_b[bb] <= kp && tail_mult[bb] >= 1.0f, FINO_ERR_ARG,
                   "fino_attn_probs: lk_b[%d] = %d (of %lld, kp %d), tail_mult %g", bb, lk_b[bb], (long long)lk, kp,
                   (double)tail_mult[bb]);
        pp.lk_b[b] = lk_b[bb];
        pp.log2_mult[b] = log2f(tail_mult[bb]);
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((lq + 127) / 128), (unsigned)(batch * heads));
    int max_lk = 0;
    for (int b = 0; b < batch; ++b) max_lk = lk_b[b] > max_lk ? lk_b[b] : max_lk;
    // as many 32-key blocks of accumulators as the keys and the output columns need: 72 / ~100 / 130 registers
    const int need = max_lk > kp ? max_lk : kp;
    const int nkb = need <= 64 ? 2 : (need <= 96 ? 3 : 4);
#define PROBS_LAUNCH(T_)                                                                                      \
    {                                                                                                         \
        if (nkb == 2) attn_probs_kernel<T_, 2><<<grid, 256, 0, st>>>(pp);                                     \
        else if (nkb == 3) attn_probs_kernel<T_, 3><<<grid, 256, 0, st>>>(pp);                                \
        else attn_probs_kernel<T_, 4><<<grid, 256, 0, st>>>(pp);                                              \
    }
    if (dtype == FINO_BF16) PROBS_LAUNCH(BF16) else PROBS_LAUNCH(F16)
#undef PROBS_LAUNCH
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
