// Shared by the attention kernels (fino_attention.hip: 8-wave ping-pong / one-barrier loops, partials, merge;
// fino_attention_w4.hip: 4-wave, one wave per SIMD).
#pragma once
#include "fino_common.h"

#define FINO_ATTN_WAVES 8      // waves per workgroup (4 waves x 2 workgroups per CU measured slower: 853 vs 935 TFLOP/s)

__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

namespace fino_attn_ns {

struct AttnParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* v;
    uint16_t* o;
    int batch, heads;
    int lq, lk;
    int64_t q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_rs, o_hs;
    float scale_log2;
    int nqb;  // q blocks per head
    // XCD balance when batch*heads is not a multiple of 8 (see attn_map_block): every head is cut into `vsplit` virtual
    // heads of `nqb_v` q-blocks; vsplit = 1, nqb_v = nqb otherwise
    int vsplit, nqb_v;
    // tail split (see plan_split): per XCD the first `full_x` blocks run whole; the key tiles of the remaining `rem_x`
    // blocks form one stream that `nwg` workgroups cut into ranges of `per` tiles, leaving (O, m, l) partials in `ws`.
    int full_x, rem_x, nwg, per;
    float* ws;
    // all_partial (fino_attn_partial): EVERY block leaves its (O, m, l) in ws[(head-batch * nqb + q-block)] instead of
    // storing O -- attention over one key range of several, merged by fino_attn_merge
    int all_partial;
    // fino_attn_fwd_tail (the walking kernel only): batch element b attends to its first tail_lk[b] key rows, the last of which
    // stands for a run of identical keys -- its logit gets tail_bias[b] = log2(multiplicity) / scale_log2 (raw q.k units) --
    // and rows from tail_lk[b] on are ignored.  tail_n = 0: off.
    int tail_n;
    int tail_lk[4];
    float tail_bias[4];
};

// (XCD, slot in that XCD's list of blocks) -> (head-batch, q-block); false: an empty slot.  The q-blocks of one head run on
// ONE XCD (hb = xcd + 8 g), so its K/V stream through that XCD's L2 only -- as long as batch*heads is a multiple of 8.
// Otherwise (3, 6 or 12 heads per rank after the multi-GPU heads exchange) whole XCDs would idle: then every head is cut
// into vsplit = 8 / gcd(batch*heads, 8) virtual heads of nqb_v = ceil(nqb / vsplit) consecutive q-blocks, whose count IS a
// multiple of 8, and the same list walk spreads them over all eight XCDs.
#if defined(__HIPCC__)
__device__ __forceinline__ bool attn_map_block(const AttnParams& p, int xcd, int bx, int& hb, int& qb) {
    const int hv = xcd + 8 * (bx / p.nqb_v);
    hb = hv / p.vsplit;
    qb = (hv - hb * p.vsplit) * p.nqb_v + bx % p.nqb_v;
    return hb < p.batch * p.heads && qb < p.nqb;
}
#endif
// host side of the same rule: virtual heads per launch and q-blocks per virtual head
inline void attn_virtual_heads(int batch, int heads, int nqb, int& vsplit, int& nqb_v) {
    const int hb = batch * heads;
    int g = hb % 8 == 0 ? 8 : (hb % 4 == 0 ? 4 : (hb % 2 == 0 ? 2 : 1));      // gcd(hb, 8)
    vsplit = 8 / g;
    if (vsplit > nqb) vsplit = 1;                    // fewer q-blocks than pieces: nothing to spread
    nqb_v = (nqb + vsplit - 1) / vsplit;
}

// floats per partial: O^T accumulators in thread order + per-thread m and l
template <int D>
constexpr int64_t partial_floats() { return (int64_t)(D / 32) * 16 * (FINO_ATTN_WAVES * 64) + 2 * (FINO_ATTN_WAVES * 64); }

constexpr int kQRowsPerWave = 32;
constexpr int kWaves = FINO_ATTN_WAVES;   // waves (32 query rows each) per workgroup
constexpr int kQBlock = kQRowsPerWave * kWaves;  // 256
constexpr int kKV = 64;
// Deferred rescale: the running maximum m moves only when a row maximum exceeds it by more than this (log2 units), so
// between rescales P = exp2(s - m) <= 2^thr.  P is rounded to the operand type for the P.V product: fp16 tops out at
// 65504 = 2^16 - 32, bf16 has fp32's exponent; l and O are fp32 sums of at most Lk terms (2^14 * 2^40 << 2^127).  Every
// quantity of a row scales by the same 2^(stale m), so the result does not depend on the threshold; 8 (r01) made peaky
// logits rescale every few tiles, which costs the 4-wave kernel an AGPR round trip of O.
template <typename T>
constexpr float rescale_thr() { return T::kId == FINO_BF16 ? 40.0f : 14.0f; }

// Byte offset of 16-byte chunk `ch` of row `row` inside a [kKV][D] tile.
//  D=128 (256-B rows): ch ^ (((row&3)<<2) | ((row>>2)&3))
//  D=64  (128-B rows): ch ^ (((row&3)<<1) | ((row>>2)&1))       (8 chunks per row)
template <int D>
__device__ __forceinline__ int lds_off(int row, int ch) {
    if constexpr (D == 128) {
        return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    } else {
        return row * 128 + ((ch ^ (((row & 3) << 1) | ((row >> 2) & 1))) << 4);
    }
}


// Byte offset of 16-byte chunk `ch` of row `row` inside a K tile.  head_dim 128: the shared lds_off image (no conflicts
// measured).  head_dim 64 (128-byte rows): lds_off's swizzle was built for the transposed V reads and lets rows r and
// r + 8 of a 32-row K fragment read share banks (SQ_LDS_BANK_CONFLICT 276 M cycles per launch at the CogVideoX shape,
// profiles/r02_attn_d64_before_summary.json); a ds_read_b128 group of 16 lanes covers 16 rows x one chunk = two rows
// per 256-byte bank line, so the 8 rows of a parity need 8 distinct chunk slots: swizzle = (row >> 1) & 7.
template <int D>
__device__ __forceinline__ int k_lds_off(int row, int ch) {
    if constexpr (D == 128) return lds_off<D>(row, ch);
    else return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
}


}  // namespace fino_attn_ns

#if defined(__HIPCC__)
// One wave's O^T accumulators (32 query rows x D channels, scaled by `inv`) -> T rows, through 32 x 2 D bytes of LDS at
// `lds_base`: the lane that holds 4 consecutive channels of a row writes their 8 bytes into an XOR-swizzled [32 rows][2 D B]
// image; then D / 8 lanes read one whole row as 16-byte pieces, rows[k] = piece (lane % (D / 8)) of row
// k * (512 / D) + lane / (D / 8).  A store of rows[k] covers 4 (head_dim 128) or 8 (64) complete rows = 8 whole cache lines; the
// direct form -- 16 (8) stores of 8 bytes per lane -- touches 32 lines per instruction, and cost the short-key kernel 4.7 us per
// q-block (profiles/r04_attn_ppw.txt).  All LDS traffic is inline asm (the compiler would drain vmcnt in front of LDS accesses
// it can see while LDS-DMA is in flight); the wave reads only what it wrote.
template <typename T, int D>
__device__ __forceinline__ void attn_rows_through_lds(const f32x16_t (&o)[D / 32], float inv, uint32_t lds_base, int r, int h,
                                                      int lane, u32x4_t (&rows)[D / 16]) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    constexpr int kC = D / 8;                         // 16-byte pieces per row
    const uint32_t wbase = lds_base + (uint32_t)(r * (2 * D) + (h << 3));
    const uint32_t rsw = (uint32_t)(r & (kC - 1));
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w0 = (uint32_t)T::from_f32(o[dt][4 * g + 0] * inv) | ((uint32_t)T::from_f32(o[dt][4 * g + 1] * inv) << 16);
            const uint32_t w1 = (uint32_t)T::from_f32(o[dt][4 * g + 2] * inv) | ((uint32_t)T::from_f32(o[dt][4 * g + 3] * inv) << 16);
            const uint32_t a = wbase + ((((uint32_t)(4 * dt + g)) ^ rsw) << 4);
            asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(u32x2_t{w0, w1}) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t row0 = (uint32_t)(lane / kC), c = (uint32_t)(lane % kC);
#pragma unroll
    for (int k = 0; k < D / 16; ++k) {
        const uint32_t row = (uint32_t)(k * (64 / kC)) + row0;
        const uint32_t a = lds_base + row * (2 * D) + ((c ^ (row & (kC - 1))) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(rows[k]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < D / 16; ++k) asm volatile("" : "+v"(rows[k]));
}
// the global side of it: rows[k] -> row (first_row + k * (512 / D) + lane / (D / 8)) of o, 16-byte piece lane % (D / 8)
template <int D>
__device__ __forceinline__ void attn_store_rows(const u32x4_t (&rows)[D / 16], uint16_t* op, int64_t o_rs, int first_row, int lq,
                                                int lane) {
    constexpr int kC = D / 8;
    const int row0 = first_row + lane / kC;
#pragma unroll
    for (int k = 0; k < D / 16; ++k) {
        const int qr = row0 + k * (64 / kC);
        if (qr < lq) *reinterpret_cast<u32x4_t*>(op + (int64_t)qr * o_rs + (lane % kC) * 8) = rows[k];
    }
}
#endif

// the tail split's plan and the merge of its (O, m, l) partials, for kernels outside fino_attention.hip that write the same
// partial layout (fino_attention_fp8.hip): defined in fino_attention.hip
void fino_attn_plan_split(int batch, int heads, int nqb, int nt, int& full_x, int& rem_x, int& nwg, int& per);
int fino_attn_launch_combine(const fino_attn_ns::AttnParams& p, int dtype, int head_dim, hipStream_t st);

// 4-wave kernel: defined in fino_attention_w4.hip (head_dim 128; head_dim 64 with the folded softmax scale only)
bool fino_attn_w4_supports(int head_dim, float scale_log2);
int fino_attn_launch_w4(const fino_attn_ns::AttnParams& p, int dtype, int head_dim, hipStream_t st);
