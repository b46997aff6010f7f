// Non-causal flash attention forward, head_dim 128: 4 waves per workgroup, ONE wave per SIMD with the whole
// 512-register file (the "4-wave, one-wave-per-SIMD" structure of the gfx950 playbook, written for this path).
// Same operator as fino_attention.hip (F.scaled_dot_product_attention at architecture/transformer_wan.py:108 of the
// reference); same S^T = K.Q^T / O^T = V^T.P^T register layouts, same swizzled LDS tile image.
//
//   * workgroup = 4 waves = 256 query rows of one (batch, head); a wave owns 64 rows = two 32-row sub-blocks, so every
//     K / V^T fragment read from LDS feeds two MFMAs (half the LDS traffic per FLOP of the 8-wave kernel).
//   * K/V tiles of 64 keys arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, swizzle applied on the
//     source chunk, rows past the last key zero-filled by the buffer range check) into 2 + 2 ring slots; ONE barrier per
//     tile.
//   * The wave software-pipelines itself: per tile two phases of 32 MFMAs,
//       phase 1: S(t+1) = K(t+1).Q^T   ||  exp2 of the second sub-block of S(t), bf16 packing of P(t)
//       phase 2: O^T += V(t)^T.P(t)^T  ||  row max of S(t+1), exp2 of its first sub-block
//     with the vector work cut into slices that follow each MFMA in program order (a wave issues in order: the
//     placement is the schedule).
#include "fino_attention_common.h"
using namespace fino_attn_ns;
#include "fino_attention_w4_regs.h"

#ifdef FINO_ATTN_STAMP
__device__ unsigned long long fino_attn_w4_dbg[64];
extern "C" int fino_attn_w4_debug_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fino_attn_w4_dbg), sizeof(unsigned long long) * 64);
}
#define W4_STAMP(V_) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V_) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define W4_STAMP(V_)
#endif

namespace {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kW4Waves = 4;
constexpr int kW4Threads = kW4Waves * 64;
// slice schedules of the fold loop (phase 1 has S1 = 4 + 4 * D/16 MFMA slices, phase 2 S2 = 8 * D/32):
constexpr int w4_s1(int d) { return 4 + 4 * (d / 16); }
constexpr int w4_s2(int d) { return 8 * (d / 32 + (d == 64 ? 1 : 0)); }   // (head_dim 64: + the ones tile of the row sums)
// 32 items dealt over `n` slices in order, item e in slice e * n / 32: slice s holds items [w4_lo(s, n), w4_lo(s + 1, n))
constexpr int w4_lo(int s, int n) { return s <= 0 ? 0 : (s >= n ? 32 : (32 * s + n - 1) / n); }

#define W4_FENCE __builtin_amdgcn_sched_barrier(0);
// Between two inline-asm statements of which the second reads a register the first one writes, the hazard recogniser
// (which counts inline asm as zero wait states) pads with an s_nop.  W4_SEP puts a real, ordered, instruction there.
#ifndef W4_SEP
#define W4_SEP        /* measured: s_setprio 0 there 1266 TFLOP/s, the recogniser's s_nop 1277 -- left to the s_nop */
#endif
#define W4_LDS_PTR(TYPE_, ADDR_) ((FINO_LDS TYPE_*)(uintptr_t)(uint32_t)(ADDR_))

// The MFMAs go through inline asm so that the register FILE of every operand is this file's decision: O^T (128
// registers) and the Q fragments (64) live in AGPRs for the whole kernel, S / P / the LDS fragments in VGPRs.  Left to
// the allocator (builtins), the 450 live registers are shuffled between the two files in bulk (1300 v_accvgpr moves
// per two tiles, spills).  The asm hides the MFMA from the hazard recogniser: consumers of S sit a phase behind its
// MFMAs by construction, and the places where O is touched by vector code are padded with explicit s_nop.
// This file is compiled with -fno-honor-nans (Makefile): fmaxf then needs no canonicalising v_max of its inputs and
// fuses to v_max3_f32 by itself (the asm helpers of the 8-wave kernel make the compiler pad every one with an s_nop).
__device__ __forceinline__ float fmx(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmx3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <typename T>
__device__ __forceinline__ uint32_t w4_pack2(float a, float b) {
    typedef typename T::scalar sc2 __attribute__((ext_vector_type(2)));
    const sc2 v = {(typename T::scalar)a, (typename T::scalar)b};
    return __builtin_bit_cast(uint32_t, v);
}
template <typename T>
__device__ __forceinline__ void w4_mfma_qk0(f32x16_t& d, const u32x4_t& a, const u32x4_t& bq) {
    if constexpr (T::kId == FINO_BF16)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(bq));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(bq));
}
// d = a . b (both operands in VGPRs): the "ones x (-m)" product that opens every S accumulator
template <typename T>
__device__ __forceinline__ void w4_mfma_vv0(f32x16_t& d, const u32x4_t& a, const u32x4_t& b) {
    if constexpr (T::kId == FINO_BF16)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <typename T>
__device__ __forceinline__ void w4_mfma_qk(f32x16_t& d, const u32x4_t& a, const u32x4_t& bq) {
    if constexpr (T::kId == FINO_BF16)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(bq));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(bq));
}


// D = 128 (Wan) or 64 (CogVideoX; FOLD only).
// FOLD: q arrives pre-multiplied by softmax_scale * log2(e) (p.scale_log2 == 1): -m is folded into the S accumulators by
// one more MFMA product and the softmax is a bare exp2 (see the prologue).
template <typename T, int D, int VAR, bool FOLD>
__global__ __attribute__((amdgpu_flat_work_group_size(kW4Threads, kW4Threads), amdgpu_waves_per_eu(1, 1)))
void attn_w4_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kKS = D / 16;   // k-steps of QK^T (8 / 4)
    constexpr int kDT = D / 32;   // d-tiles of O^T (4 / 2)
    constexpr int kW4TileBytes = kKV * D * 2;            // 16 / 8 KiB; LDS: K ring (2) + V ring (2)
    constexpr int kNPW = kW4TileBytes / 1024 / kW4Waves;   // 1-KiB LDS-DMA pieces per wave and tile (4 / 2)
    static_assert(FOLD || D == 128, "head_dim 64 is built for the folded-scale path only");
    // head_dim 64 is bound by the softmax's vector work (per MFMA twice the exp2 / add / pack of head_dim 128): there the
    // row sums l move to the matrix pipe -- one more "d-tile" of P.V whose V^T rows are all ones (8 MFMAs per tile into the
    // spare accumulator tuples 2 / 6; l is rescaled with O for free) instead of 64 v_add per lane and tile.
    constexpr bool kMS = (D == 64);
    constexpr int kPT = kDT + (kMS ? 1 : 0);              // P.V products per key step and sub-block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // XCD-aware block -> (head-batch, q-block): all q-blocks of a head share blockIdx % 8.  Whole blocks first; then the
    // XCD's last rem_x blocks as one stream of rem_x * nt key tiles cut into nwg equal ranges (tail split, same plan
    // and same partial layout as the 8-wave kernel: attn_combine_kernel / attn_merge_kernel finish either).
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int ntall = (p.lk + kKV - 1) / kKV;
    int npieces = 1, first_b = 0;
    int64_t g0 = 0, g1 = 0;
    if (slot >= p.full_x) {
        g0 = (int64_t)(slot - p.full_x) * p.per;
        g1 = g0 + p.per < (int64_t)p.rem_x * ntall ? g0 + p.per : (int64_t)p.rem_x * ntall;
        first_b = (int)(g0 / ntall);
        npieces = (int)((g1 - 1) / ntall) - first_b + 1;
    }
  for (int piece = 0; piece < npieces; ++piece) {
    if (piece > 0) __syncthreads();                 // every wave is done reading the previous piece's LDS tiles
    int bx = slot, part = -1, t_begin = 0, t_end = ntall;
    if (slot >= p.full_x) {
        const int tb = first_b + piece;
        const int64_t b0 = (int64_t)tb * ntall;
        t_begin = g0 > b0 ? (int)(g0 - b0) : 0;
        t_end = g1 - b0 < ntall ? (int)(g1 - b0) : ntall;
        bx = p.full_x + tb;
        if (t_begin != 0 || t_end != ntall) part = ((xcd * p.nwg) + (slot - p.full_x)) * 2 + piece;
    }
    // (wave-uniform by construction; said explicitly, or the piece loop makes the compiler treat the K/V resources
    //  and DMA offsets as divergent and wrap every LDS-DMA in a readfirstlane loop)
    int hb_, qb_;
    const bool mapped = attn_map_block(p, xcd, bx, hb_, qb_);
    const int hb = __builtin_amdgcn_readfirstlane(hb_);
    const int qb = __builtin_amdgcn_readfirstlane(qb_);
    t_begin = __builtin_amdgcn_readfirstlane(t_begin);
    t_end = __builtin_amdgcn_readfirstlane(t_end);
    part = __builtin_amdgcn_readfirstlane(part);
    if (!mapped) continue;
    if (p.all_partial) part = hb * p.nqb + qb;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    const uint16_t* kp = p.k + bi * p.k_bs + head * p.k_hs + (int64_t)t_begin * kKV * p.k_rs;
    const uint16_t* vp = p.v + bi * p.v_bs + head * p.v_hs + (int64_t)t_begin * kKV * p.v_rs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    // keys of this workgroup's range, re-based to 0 (a multiple of kKV precedes it, so tail masks are unchanged)
    const int lk = (t_end * kKV < p.lk ? t_end * kKV : p.lk) - t_begin * kKV;
    const int nt = (lk + kKV - 1) / kKV;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[row][16*ks + 8h .. +7] of both sub-blocks ----
    int qrow[2];
    u32x4_t qf[2][kKS];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        qrow[qs] = qb * kQBlock + wave * 64 + qs * 32 + r;
        const int qc = qrow[qs] < p.lq ? qrow[qs] : p.lq - 1;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            uint4 u = *reinterpret_cast<const uint4*>(qp + (int64_t)qc * p.q_rs + 16 * ks + 8 * h);
            if (qrow[qs] >= p.lq) u = make_uint4(0, 0, 0, 0);   // rows past Lq are never stored: zero operands draw the least power
            qf[qs][ks] = u32x4_t{u.x, u.y, u.z, u.w};
            asm volatile("" : "+a"(qf[qs][ks]));        // lives in AGPRs from here on (else: copied there before every MFMA)
        }
    }

    // ---- LDS-DMA pieces: wave w moves rows 16w .. 16w+15 of a tile as 4 pieces of 4 rows (1 KiB each) ----
    // lane l of piece P = 4w + j lands at row 4P + (l >> 4), physical chunk l & 15; it fetches the logical chunk
    // (l & 15) ^ swz(row), swz(row) = ((row & 3) << 2) | ((row >> 2) & 3) = (((l >> 4) & 3) << 2) | (j & 3).
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (int)((((int64_t)lk - 1) * p.k_rs + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)vp, 0, (int)((((int64_t)lk - 1) * p.v_rs + D) * 2), 0x00020000);
    // a piece = 1 KiB of the LDS image = 4 rows x 16 chunks (D = 128) or 8 rows x 8 chunks (D = 64); lane l lands at row
    // l / chunks-per-row, physical chunk l % chunks-per-row and fetches the logical chunk phys ^ swizzle(row)
    uint32_t kvo[4], vvo[4];       // (kNPW used; a template-dependent array size here silently drops the HOST instantiation)
#pragma unroll
    for (int j = 0; j < kNPW; ++j) {
        constexpr int cpr = D / 8, rpp = 1024 / (D * 2);
        const int row = 16 * wave + rpp * j + lane / cpr;
        const int chk = ((k_lds_off<D>(row, lane % cpr) - row * (D * 2)) >> 4);     // = (lane % cpr) ^ K swizzle(row)
        const int chv = ((lds_off<D>(row, lane % cpr) - row * (D * 2)) >> 4);      // = (lane % cpr) ^ V swizzle(row)
        kvo[j] = (uint32_t)((row * p.k_rs + chk * 8) * 2);
        vvo[j] = (uint32_t)((row * p.v_rs + chv * 8) * 2);
    }
    const int k_tile_bytes = (int)(kKV * p.k_rs * 2), v_tile_bytes = (int)(kKV * p.v_rs * 2);
    // K(t) -> slot t & 1 (bytes 0 / 16 K), V(t) -> slot 2 + (t & 1)
#define W4_DMA_K(T_)                                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < kNPW; ++j_) __builtin_amdgcn_raw_ptr_buffer_load_lds(              \
        k_rsrc, (FINO_LDS void*)(smem + ((T_) & 1) * kW4TileBytes + (kNPW * wave + j_) * 1024), 16, kvo[j_],   \
        (T_) * k_tile_bytes, 0, 0);
#define W4_DMA_K1(T_, SLOT_, J_)                                                                               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                  \
        k_rsrc, (FINO_LDS void*)(smem + (SLOT_) * kW4TileBytes + (kNPW * wave + (J_)) * 1024), 16, kvo[J_],    \
        (T_) * k_tile_bytes, 0, 0);
#define W4_DMA_V1(T_, SLOT_, J_)                                                                               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                  \
        v_rsrc, (FINO_LDS void*)(smem + (2 + (SLOT_)) * kW4TileBytes + (kNPW * wave + (J_)) * 1024), 16, vvo[J_], \
        (T_) * v_tile_bytes, 0, 0);
#define W4_DMA_V(T_)                                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < kNPW; ++j_) __builtin_amdgcn_raw_ptr_buffer_load_lds(              \
        v_rsrc, (FINO_LDS void*)(smem + (2 + ((T_) & 1)) * kW4TileBytes + (kNPW * wave + j_) * 1024), 16, vvo[j_], \
        (T_) * v_tile_bytes, 0, 0);

    // ---- per-lane LDS fragment addresses (raw: the dynamic segment is the kernel's only LDS and starts at 0) ----
    if ((uint32_t)(uintptr_t)(FINO_LDS char*)smem != 0u) __builtin_trap();
    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    const int g1 = (lane >> 4) & 1;
    const uint32_t ka_base = k_lds_off<D>(r, h);
    const uint32_t vl_base = 2 * kW4TileBytes + lds_off<D>(4 * h + tq, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
    const uint32_t vh_base = 2 * kW4TileBytes + lds_off<D>(4 * h + tq + 8, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);

    w4_o_zero();                   // O^T: a128..a255, touched only through fino_attention_w4_regs.h
    float m_run[2] = {-INFINITY, -INFINITY};
    const float c2 = p.scale_log2;           // FOLD: 1 (unused)
    float l_run[2] = {0.f, 0.f};

    // row max of 8 accumulator registers
#define W4_MAX8(S_, O_) fmx(fmx3(fmx3(S_[O_], S_[O_ + 1], S_[O_ + 2]), fmx3(S_[O_ + 3], S_[O_ + 4], S_[O_ + 5]), \
                                S_[O_ + 6]), S_[O_ + 7])
#define W4_SWAPMAX(MX_, OUT_)                                                                                  \
    {                                                                                                          \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = fmx(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                        \
    }
    // keys past lk (zero K rows of a ragged last tile) leave the row max and get p = exp2(-inf) = 0
#define W4_MASK(S_, T_)                                                                                        \
    if ((T_) == nt - 1 && (lk & (kKV - 1))) {                                                                  \
        int rem_ = lk - (T_) * kKV - 4 * h;            /* keys left from this lane's first one */               \
        asm volatile("" : "+v"(rem_));                 /* opaque: or 32 loop-invariant lane masks are hoisted into SGPRs */ \
        _Pragma("unroll") for (int qs_ = 0; qs_ < 2; ++qs_) _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) { \
            const int key_ = (j_ & 3) + 8 * (j_ >> 2);                                                         \
            if (key_ >= rem_) S_[qs_][0][j_] = -INFINITY;                                                      \
            if (key_ + 32 >= rem_) S_[qs_][1][j_] = -INFINITY;                                                 \
        }                                                                                                      \
    }

    // ---- prologue: K(0), V(0), K(1); S(0) = K(0).Q^T unpipelined; its max; first sub-block's exp2 ----
    W4_DMA_K(0)
    W4_DMA_V(0)
    if (nt > 1) { W4_DMA_K(1) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16_t sa[2][2], sb[2][2];   // S of the tile in flight / of the next one: [sub-block][key half]
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
        const uint32_t a = ka_base ^ (ks << 5);
        const u32x4_t a0 = *W4_LDS_PTR(const u32x4_t, a);
        const u32x4_t a1 = *W4_LDS_PTR(const u32x4_t, a + 32 * D * 2);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            if (ks == 0) {
                w4_mfma_qk0<T>(sa[qs][0], a0, qf[qs][ks]);
                w4_mfma_qk0<T>(sa[qs][1], a1, qf[qs][ks]);
            } else {
                w4_mfma_qk<T>(sa[qs][0], a0, qf[qs][ks]);
                w4_mfma_qk<T>(sa[qs][1], a1, qf[qs][ks]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // MFMA results -> vector reads
    W4_MASK(sa, 0)
    float psum0 = 0.f;             // row sum of sub-block 0's P of the tile in flight (added to l_run in phase 1)
    // The running maximum is kept as a value of the operand type T: -m then rides into every later S accumulator as
    // one more product ("ones" x (-m), exact in the fp32 accumulate), and the softmax is exp2(S) with no per-element
    // scale-and-subtract.  ones: A[key][k = 0] = 1;  mneg[qs]: B[k = 0][q] = -m[q]  (k = 0 lives in the h = 0 lanes).
    u32x4_t ones = {h == 0 ? (uint32_t)T::from_f32(1.0f) : 0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(ones));
    u32x4_t mneg[2];
    const uint32_t one2 = (uint32_t)T::from_f32(1.0f) * 0x10001u;
    u32x4_t ones_v = {one2, one2, one2, one2};                // a V^T fragment of ones (kMS)
    // opaque from here on: a known constant is re-materialised (v_mov) right in front of the asm MFMA that reads it, which
    // is a VALU-write -> MFMA-read hazard the recogniser cannot see
    asm volatile("" : "+v"(ones_v));
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const float mx = fmx(fmx3(W4_MAX8(sa[qs][0], 0), W4_MAX8(sa[qs][0], 8), W4_MAX8(sa[qs][1], 0)),
                             W4_MAX8(sa[qs][1], 8));
        float mxx;
        W4_SWAPMAX(mx, mxx)
        if constexpr (FOLD) {
            m_run[qs] = T::to_f32(T::from_f32(mxx));
            mneg[qs] = u32x4_t{h == 0 ? (uint32_t)T::from_f32(-m_run[qs]) : 0u, 0u, 0u, 0u};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int j = 0; j < 16; ++j) sa[qs][kh][j] -= m_run[qs];
        } else {
            m_run[qs] = mxx * c2;
            mneg[qs] = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sa[0][kh][j] = FOLD ? __builtin_amdgcn_exp2f(sa[0][kh][j]) : __builtin_amdgcn_exp2f(sa[0][kh][j] * c2 - m_run[0]);
            psum0 += sa[0][kh][j];
        }

    u32x4_t pb[2][4];              // P(t) packed bf16: pb[sub-block][2 * key half + (j >> 3)]

    // The fragment reads of the tile loop are inline asm with hand-counted lgkmcnt waits.  As compiler-visible loads
    // every ds_read issued after an LDS-DMA of the same tile would get an s_waitcnt vmcnt(0) in front of it (the DMA
    // writes LDS and nothing tells the compiler that it is another ring slot): the wave would sit out the HBM latency
    // of its own prefetch in every tile.  LDS returns data in order, so "lgkmcnt(n)" = all but the last n reads landed.
#define W4_LD128(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_));
#define W4_LDTR(DST_, ADDR_, OFF_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_));
#define W4_WAIT_LGKM(N_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" ::: "memory");
    // loop-invariant fragment addresses: ring slots and the +32-row / +16-key-row steps are instruction offsets
    uint32_t ka_addr[kKS], vl_addr[kDT], vh_addr[kDT];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) ka_addr[ks] = ka_base ^ (ks << 5);
#pragma unroll
    for (int dt = 0; dt < kDT; ++dt) {
        vl_addr[dt] = (vl_base - 2 * kW4TileBytes) ^ (dt << 6);
        vh_addr[dt] = (vh_base - 2 * kW4TileBytes) ^ (dt << 6);
    }

    // ---- FOLD = false (any softmax scale): scale-and-subtract per element ----
    // softmax pieces.  W4_FMA: y = c.s - m of element E_ (e = 16 * key half + j), one slice AHEAD of its exp2 so that the
    // transcendental never waits for its operand; W4_EXPY: x = exp2(y), and the row sum takes the PREVIOUS x (a
    // transcendental's result is not ready for the next issue).  The empty asm pins each piece to its slice: pure
    // arithmetic otherwise sinks to its use, a phase later.
#define W4_FMA(S_, QS_, E_, M_, Y_)                                                                            \
    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(Y_) : "v"(S_[QS_][(E_) >> 4][(E_) & 15]), "s"(c2), "v"(M_));
#define W4_EXPY(S_, QS_, E_, Y_, SUM_, PEND_)                                                                  \
    {                                                                                                          \
        float x_;                                                                                              \
        W4_SEP                                                                                                 \
        asm volatile("v_exp_f32 %0, %2\n\tv_add_f32 %1, %1, %3" : "=&v"(x_), "+v"(SUM_) : "v"(Y_), "v"(PEND_));  \
        S_[QS_][(E_) >> 4][(E_) & 15] = x_;                                                                    \
        PEND_ = x_;                                                                                            \
    }
    // ---- FOLD = true ----
    // one softmax element (e = 16 * key half + j): x = exp2(S), and the row sum takes the PREVIOUS x (a transcendental's
    // result is not ready for the next issue).  asm: the placement in its slice is the schedule.
#define W4_EXP1(S_, QS_, E_, SUM_, PEND_)                                                                      \
    {                                                                                                          \
        float x_;                                                                                              \
        asm volatile("v_exp_f32 %0, %2\n\tv_add_f32 %1, %1, %3" : "=&v"(x_), "+v"(SUM_)                        \
                     : "v"(S_[QS_][(E_) >> 4][(E_) & 15]), "v"(PEND_));                                        \
        S_[QS_][(E_) >> 4][(E_) & 15] = x_;                                                                    \
        PEND_ = x_;                                                                                            \
    }
#define W4_EL(S_, QS_, E_) S_[QS_][(E_) >> 4][(E_) & 15]
    // pack elements (E_, E_ + 1) of sub-block QS_ (E_ even) into P's B-operand registers
#define W4_PACK2(S_, QS_, E_)                                                                                  \
    {                                                                                                          \
        uint32_t w_;                                                                                           \
        if constexpr (T::kId == FINO_BF16)                                                                     \
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w_)                                             \
                         : "v"(S_[QS_][(E_) >> 4][(E_) & 15]), "v"(S_[QS_][(E_) >> 4][((E_) & 15) + 1]));      \
        else                                                                                                   \
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w_)                                              \
                         : "v"(S_[QS_][(E_) >> 4][(E_) & 15]), "v"(S_[QS_][(E_) >> 4][((E_) & 15) + 1]));      \
        pb[QS_][(E_) >> 3][((E_) & 7) >> 1] = w_;                                                              \
    }
    // V^T fragment pair n (= kDT * key step + d-tile) of the V slot at byte VS_ -> ring entry n & 3 (two 64-bit halves)
#define W4_LOADV(N_, VS_)                                                                                      \
    {                                                                                                          \
        W4_LDTR(vlo_[(N_) & 3], vl_addr[(N_) % kDT], (VS_) + ((N_) / kDT) * 16 * D * 2)                        \
        W4_LDTR(vhi_[(N_) & 3], vh_addr[(N_) % kDT], (VS_) + ((N_) / kDT) * 16 * D * 2)                        \
    }

    // exp2 of elements [LO_, LO_ + N_) of a sub-block and the row-sum adds of the elements BEFORE each of them (PREV_ =
    // element LO_ - 1, or zero), as ONE asm statement (between two statements the hazard recogniser pads with s_nop).
    // LO_, N_ are constant expressions (the slices are spelled out by literal index below).
#define W4_EXPADD(S_, QS_, LO_, N_, PREV_, SUM_)                                                               \
    {                                                                                                          \
        if constexpr (kMS) {                    /* row sums on the matrix pipe: exp2 only */                    \
            if constexpr ((N_) >= 1) asm volatile("v_exp_f32 %0, %0" : "+v"(W4_EL(S_, QS_, LO_)));             \
            if constexpr ((N_) >= 2) asm volatile("v_exp_f32 %0, %0" : "+v"(W4_EL(S_, QS_, (LO_) + 1)));       \
            if constexpr ((N_) >= 3) asm volatile("v_exp_f32 %0, %0" : "+v"(W4_EL(S_, QS_, (LO_) + 2)));       \
        } else if constexpr ((N_) == 1)                                                                        \
            asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(W4_EL(S_, QS_, LO_)), "+v"(SUM_) : "v"(PREV_)); \
        else if constexpr ((N_) == 2)                                                                          \
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %2, %2, %0"  \
                         : "+v"(W4_EL(S_, QS_, LO_)), "+v"(W4_EL(S_, QS_, (LO_) + 1)), "+v"(SUM_) : "v"(PREV_)); \
        else if constexpr ((N_) == 3)                                                                          \
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_add_f32 %3, %3, %4\n\t"   \
                         "v_add_f32 %3, %3, %0\n\tv_add_f32 %3, %3, %1"                                        \
                         : "+v"(W4_EL(S_, QS_, LO_)), "+v"(W4_EL(S_, QS_, (LO_) + 1)), "+v"(W4_EL(S_, QS_, (LO_) + 2)), \
                           "+v"(SUM_) : "v"(PREV_));                                                           \
    }
    // element E_ - 1 as the PREV_ operand above (zero before element 0)
#define W4_PREV(S_, QS_, E_) ((E_) >= 1 ? W4_EL(S_, QS_, (E_) >= 1 ? (E_) - 1 : 0) : zero_f)
    const float zero_f = 0.f;

    // ---- phase 1, slice S_ (literal; S1 = 4 + 4 kKS slices): MFMA S_ of "S(t+1) = ones.(-m) + K(t+1).Q~^T" (4 opening
    //      products, which also cover the first K fragment's LDS latency, then kKS k-steps x 4) || its share of the 32
    //      exp2 of sub-block 1 of S(t) and of the 32 packed pairs (sub-block 0 first: exp2'ed a phase ago; a pair of
    //      sub-block 1 one slice after its second exp2), dealt by w4_lo; the 2 kNPW LDS-DMA pieces one per k-step ----
    // packs of phase-1 slice (lo_, n_, plo_ in scope): k-th element of this slice -> its sub-block-0 pair; k-th element of
    // the PREVIOUS slice, if odd -> the sub-block-1 pair it completes
#define W4_TILE_BARRIER __builtin_amdgcn_s_barrier();
    constexpr bool kDoDma = true;
    constexpr bool kNoMax = false;
#define W4_P1_PACK(SC_, K_)                                                                                    \
    if constexpr ((K_) < n_ && lo_ + (K_) < 16) { W4_PACK2(SC_, 0, 2 * (lo_ + (K_))) }                         \
    if constexpr (s_ >= 1 && plo_ + (K_) < lo_ && ((plo_ + (K_)) & 1)) { W4_PACK2(SC_, 1, plo_ + (K_) - 1) }
#define W4_P1(SC_, SN_, PAR_, SL_)                                                                             \
    if constexpr ((SL_) < w4_s1(D)) {                                                                          \
        constexpr int s_ = (SL_), ks_ = (s_ - 4) >> 2, i_ = s_ & 3, qs_ = i_ & 1, kh_ = i_ >> 1;              \
        if constexpr (s_ >= 4 && i_ == 0) {                                                                    \
            if constexpr (has_next_ && ks_ + 1 < kKS) {                                                        \
                W4_LD128(ka_[(ks_ + 1) & 1][0], ka_addr[ks_ + 1], ks_off_)                                     \
                W4_LD128(ka_[(ks_ + 1) & 1][1], ka_addr[ks_ + 1], ks_off_ + 32 * D * 2)                        \
            }                                                                                                  \
            if constexpr (ks_ == kKS - 2) { W4_LOADV(0, vs_off_) }                                             \
            if constexpr (ks_ == kKS - 1) { W4_LOADV(1, vs_off_) }                                             \
            /* K(ks) landed: behind it are K(ks+1) (2 reads), in the last two k-steps also V pairs 0, 1 */      \
            if constexpr (has_next_) {                                                                         \
                if constexpr (ks_ < kKS - 2) { W4_WAIT_LGKM(2) } else { W4_WAIT_LGKM(4) }                      \
            }                                                                                                  \
        }                                                                                                      \
        if constexpr (has_next_) {                                                                             \
            if constexpr (s_ < 4) w4_mfma_vv0<T>(SN_[qs_][kh_], ones, mneg[qs_]);                              \
            else w4_mfma_qk<T>(SN_[qs_][kh_], ka_[ks_ & 1][kh_], qf[qs_][ks_ < 0 ? 0 : ks_]);                  \
        }                                                                                                      \
        if constexpr (s_ >= 4 && i_ == 1) {                                                                    \
            if constexpr (ks_ < kNPW) { if (dma_k_) { W4_DMA_K1(t_ + 2, PAR_, ks_ < 0 ? 0 : ks_) } }           \
            else if constexpr (ks_ < 2 * kNPW) { if (has_next_ && kDoDma) { W4_DMA_V1(t_ + 1, 1 - (PAR_), ks_ - kNPW) } } \
        }                                                                                                      \
        {                                                                                                      \
            constexpr int lo_ = w4_lo(s_, w4_s1(D)), n_ = w4_lo(s_ + 1, w4_s1(D)) - lo_;                        \
            W4_EXPADD(SC_, 1, lo_, n_, W4_PREV(SC_, 1, lo_), psum1_)                                           \
            constexpr int plo_ = w4_lo(s_ - 1, w4_s1(D));                 /* the previous slice's elements */  \
            W4_P1_PACK(SC_, 0) W4_P1_PACK(SC_, 1) W4_P1_PACK(SC_, 2)                                           \
        }                                                                                                      \
        W4_FENCE                                                                                               \
    }
    // ---- phase 2, key step ST_ (0..3), product PI_ (0..kPT-1: d-tile, or the ones tile of the row sums when PI_ == kDT),
    //      sub-block QS_ (all literal): slice s = (ST_ * kPT + PI_) * 2 + QS_ of S2 = 8 kPT.  Slices 0..3 a quarter of both
    //      sub-blocks' row maxima each, slice 4 the rescale decision, the 32 exp2 of sub-block 0 of S(t+1) from slice 5 on.
    //      V^T fragments: real pair n = kDT * ST_ + PI_ is read two pairs ahead into a ring of 4 ----
#define W4_P2(SN_, ST_, PI_, QS_)                                                                              \
    if constexpr ((PI_) < kPT) {                                                                               \
        constexpr int st_ = (ST_), pi_ = (PI_), qs_ = (QS_), s_ = (st_ * kPT + pi_) * 2 + qs_;                 \
        constexpr int n_ = kDT * st_ + (pi_ < kDT ? pi_ : 0);          /* real V pair (unused for the ones tile) */ \
        if constexpr (qs_ == 0 && pi_ < kDT) {                                                                 \
            if constexpr (n_ + 2 < 4 * kDT) { W4_LOADV(n_ + 2, vs_off_) }                                      \
            /* pair n landed: behind it are pairs n+1, n+2 (2 reads each) */                                   \
            if constexpr (n_ + 2 < 4 * kDT) { W4_WAIT_LGKM(4) } else if constexpr (n_ + 1 < 4 * kDT) { W4_WAIT_LGKM(2) } \
            else { W4_WAIT_LGKM(0) }                                                                           \
            va_ = u32x4_t{vlo_[n_ & 3][0], vlo_[n_ & 3][1], vhi_[n_ & 3][0], vhi_[n_ & 3][1]};                 \
        }                                                                                                      \
        if constexpr (pi_ < kDT) w4_o_mfma<T>(4 * qs_ + pi_, va_, pb[qs_][st_]);                               \
        else w4_o_mfma<T>(4 * qs_ + 2, ones_v, pb[qs_][st_]);          /* l^T += ones . P^T (tuples 2 / 6) */   \
        if constexpr (has_next_) {                                                                             \
            if constexpr (kNoMax) {                                                                            \
            } else if constexpr (s_ < 4) {      /* a quarter of both sub-blocks' row maxima: two independent chains */ \
                constexpr int kh_ = s_ >> 1, o_ = 8 * (s_ & 1);                                                \
                mxp_[0] = fmx(mxp_[0], W4_MAX8(SN_[0][kh_], o_));                                              \
                mxp_[1] = fmx(mxp_[1], W4_MAX8(SN_[1][kh_], o_));                                              \
                asm volatile("" : "+v"(mxp_[0]), "+v"(mxp_[1]));                                               \
            } else if constexpr (s_ == 4) {                                                                    \
                /* S(t+1) is relative to m_run already: its row max IS the excess over the running maximum.     \
                   Deferred rescale: only past rescale_thr<T>() (then all lanes move, each to its own maximum). */ \
                float mx0_, mx1_;                                                                              \
                W4_SWAPMAX(mxp_[0], mx0_)                                                                      \
                W4_SWAPMAX(mxp_[1], mx1_)                                                                      \
                resc_ = __any(fmx(mx0_, mx1_) > rescale_thr<T>());                                             \
                if (__builtin_expect(resc_, 0)) {                                                              \
                    const float mx_[2] = {mx0_, mx1_};                                                         \
                    _Pragma("unroll") for (int q2_ = 0; q2_ < 2; ++q2_) {                                      \
                        const float mn_ = T::to_f32(T::from_f32(m_run[q2_] + fmx(mx_[q2_], 0.f)));             \
                        dm_[q2_] = mn_ - m_run[q2_];                                                           \
                        m_run[q2_] = mn_;                                                                      \
                        mneg[q2_][0] = h == 0 ? (uint32_t)T::from_f32(-mn_) : 0u;                              \
                        _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                    \
                            _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) SN_[q2_][kh_][j_] -= dm_[q2_];   \
                    }                                                                                          \
                }                                                                                              \
            } else {                                                                                           \
                constexpr int lo_ = w4_lo(s_ - 5, w4_s2(D) - 5), cnt_ = w4_lo(s_ - 4, w4_s2(D) - 5) - lo_;      \
                W4_EXPADD(SN_, 0, lo_, cnt_, W4_PREV(SN_, 0, lo_), psn_)                                       \
            }                                                                                                  \
            if constexpr (kNoMax && s_ >= 5) {                                                                 \
                constexpr int lo_ = w4_lo(s_ - 5, w4_s2(D) - 5), cnt_ = w4_lo(s_ - 4, w4_s2(D) - 5) - lo_;      \
                W4_EXPADD(SN_, 0, lo_, cnt_, W4_PREV(SN_, 0, lo_), psn_)                                       \
            }                                                                                                  \
        }                                                                                                      \
        W4_FENCE                                                                                               \
    }
#define W4_P1x4(SC_, SN_, PAR_, B_) W4_P1(SC_, SN_, PAR_, (B_)) W4_P1(SC_, SN_, PAR_, (B_) + 1) W4_P1(SC_, SN_, PAR_, (B_) + 2) W4_P1(SC_, SN_, PAR_, (B_) + 3)
#define W4_P2x2(SN_, ST_, PI_) W4_P2(SN_, ST_, PI_, 0) W4_P2(SN_, ST_, PI_, 1)
#define W4_P2STEP(SN_, ST_) W4_P2x2(SN_, ST_, 0) W4_P2x2(SN_, ST_, 1) W4_P2x2(SN_, ST_, 2) W4_P2x2(SN_, ST_, 3) W4_P2x2(SN_, ST_, 4)

    // ---- one key tile t of parity PAR_ (literal): SC_ = S(t) (sub-block 0 already exp2'ed, its row sum in psum0),
    //      SN_ = S(t+1).  Ring slots: K(t+1) is read from K slot 1 - PAR_, V(t) from V slot PAR_; the DMA of K(t+2)
    //      goes to K slot PAR_ and that of V(t+1) to V slot 1 - PAR_ (both free since the barrier). ----
#define W4_TILE_F(SC_, SN_, TT_, PAR_, HN_)                                                                    \
    {                                                                                                          \
        const int t_ = (TT_);                                                                                  \
        constexpr bool has_next_ = (HN_);                                                                      \
        constexpr int ks_off_ = (1 - (PAR_)) * kW4TileBytes;                  /* K slot read in phase 1 */     \
        constexpr int vs_off_ = (2 + (PAR_)) * kW4TileBytes;                  /* V slot read in phase 2 */     \
        W4_STAMP(ts0)                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        W4_STAMP(ts1)                                                                                          \
        W4_TILE_BARRIER                                                                                        \
        W4_FENCE                                                                                               \
        W4_STAMP(ts2)                                                                                          \
        const bool dma_k_ = kDoDma && t_ + 2 < nt;                                                             \
        W4_STAMP(ts3)                                                                                          \
        /* ================= phase 1 ================= */                                                       \
        float psum1_ = 0.f;                                                                                    \
        u32x4_t ka_[2][2], va_;                                                                                \
        u32x2_t vlo_[4], vhi_[4];                                                                              \
        if (has_next_) {                                                                                       \
            W4_LD128(ka_[0][0], ka_addr[0], ks_off_)                                                           \
            W4_LD128(ka_[0][1], ka_addr[0], ks_off_ + 32 * D * 2)                                              \
        }                                                                                                      \
        /* ones / mneg are vector-written registers (re-assembled by the compiler at will): a VALU write straight  \
           in front of an MFMA that reads it is a hazard the recogniser cannot see through the asm */          \
        asm volatile("s_nop 3" : "+v"(mneg[0]), "+v"(mneg[1]) : "v"(ones));                                    \
        W4_FENCE                                                                                               \
        W4_P1x4(SC_, SN_, PAR_, 0) W4_P1x4(SC_, SN_, PAR_, 4) W4_P1x4(SC_, SN_, PAR_, 8)                       \
        W4_P1x4(SC_, SN_, PAR_, 12) W4_P1x4(SC_, SN_, PAR_, 16) W4_P1x4(SC_, SN_, PAR_, 20)                    \
        W4_P1x4(SC_, SN_, PAR_, 24) W4_P1x4(SC_, SN_, PAR_, 28) W4_P1x4(SC_, SN_, PAR_, 32)                    \
        {                                                                  /* odd elements of the last slice */ \
            constexpr int ll_ = w4_lo(w4_s1(D) - 1, w4_s1(D));                                                 \
            if constexpr (ll_ < 32 && (ll_ & 1)) { W4_PACK2(SC_, 1, ll_ - 1) }                                 \
            if constexpr (ll_ + 1 < 32 && ((ll_ + 1) & 1)) { W4_PACK2(SC_, 1, ll_) }                           \
            if constexpr (ll_ + 2 < 32 && ((ll_ + 2) & 1)) { W4_PACK2(SC_, 1, ll_ + 1) }                       \
        }                                                                                                      \
        if constexpr (!kMS) {                                                                                  \
            l_run[0] += psum0;                                                                                 \
            l_run[1] += psum1_ + W4_EL(SC_, 1, 31);                                                            \
        }                                                                                                      \
        W4_FENCE                                                                                               \
        W4_STAMP(ts4)                                                                                          \
        /* ================= phase 2 ================= */                                                       \
        if (has_next_) { W4_MASK(SN_, t_ + 1) }                                                                \
        float mxp_[2] = {-INFINITY, -INFINITY}, dm_[2] = {0.f, 0.f};                                           \
        bool resc_ = false;                                                                                    \
        float psn_ = 0.f;                                                                                      \
        W4_P2STEP(SN_, 0) W4_P2STEP(SN_, 1) W4_P2STEP(SN_, 2) W4_P2STEP(SN_, 3)                                \
        /* the O / l side of a rescale, between tiles (after the last P(t).V(t) product) */                    \
        if (has_next_) {                                                                                       \
            if (__builtin_expect(resc_, 0)) {                                                                  \
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       /* MFMA writes of O -> vector reads */ \
                _Pragma("unroll") for (int qs_ = 0; qs_ < 2; ++qs_) {                                          \
                    const float alpha_ = __builtin_amdgcn_exp2f(-dm_[qs_]);                                    \
                    l_run[qs_] *= alpha_;          /* (kMS: l lives in tuples 2 / 6 and is scaled with O) */   \
                    w4_o_scale(qs_, alpha_);                                                                   \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        if (has_next_ && !kMS) psum0 = psn_ + W4_EL(SN_, 0, 31);        /* the last element's share of the row sum */        \
        W4_FENCE                                                                                               \
        W4_STAMP(ts5)                                                                                          \
        W4_STAMP_ACC                                                                                           \
    }

    // ---- the same tile without the fold (FOLD = false): 32 + 32 MFMA slices, c.s - m by v_fma one slice ahead ----
#define W4_TILE_N(SC_, SN_, TT_, PAR_, HN_)                                                                      \
    {                                                                                                          \
        const int t_ = (TT_);                                                                                  \
        constexpr bool has_next_ = (HN_);                                                                      \
        constexpr int ks_off_ = (1 - (PAR_)) * kW4TileBytes;                  /* K slot read in phase 1 */     \
        constexpr int vs_off_ = (2 + (PAR_)) * kW4TileBytes;                  /* V slot read in phase 2 */     \
        W4_STAMP(ts0)                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
        W4_STAMP(ts1)                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                          \
        W4_FENCE                                                                                               \
        W4_STAMP(ts2)                                                                                          \
        const bool dma_k_ = t_ + 2 < nt;     /* the 8 LDS-DMA pieces go out one per k-step of phase 1 */       \
        W4_STAMP(ts3)                                                                                          \
        /* ================= phase 1: S(t+1) = K(t+1).Q^T  ||  exp2 of sub-block 1 of S(t), packing ========= */ \
        float psum1_ = 0.f, pend1_ = 0.f, y1_[2];                                                              \
        u32x4_t ka_[2][2];                                                                                     \
        u32x2_t vlo_[4], vhi_[4];                                                                              \
        if (has_next_) {                                                                                       \
            W4_LD128(ka_[0][0], ka_addr[0], ks_off_)                                                           \
            W4_LD128(ka_[0][1], ka_addr[0], ks_off_ + 32 * D * 2)                                              \
        }                                                                                                      \
        W4_FMA(SC_, 1, 0, m_run[1], y1_[0])                                                                    \
        W4_FENCE                                                                                               \
        _Pragma("unroll") for (int ks_ = 0; ks_ < kKS; ++ks_) {                                                \
            if (has_next_ && ks_ + 1 < kKS) {                                                                  \
                W4_LD128(ka_[(ks_ + 1) & 1][0], ka_addr[ks_ + 1], ks_off_)                                     \
                W4_LD128(ka_[(ks_ + 1) & 1][1], ka_addr[ks_ + 1], ks_off_ + 32 * D * 2)                        \
            }                                                                                                  \
            if (ks_ == kKS - 2) { W4_LOADV(0, vs_off_) }                                                       \
            if (ks_ == kKS - 1) { W4_LOADV(1, vs_off_) }                                                       \
            /* K(ks) landed: behind it are K(ks+1) (2 reads), at ks = 6 also V pair 0, at ks = 7 V pairs 0, 1 */ \
            if (has_next_) {                                                                                   \
                if (ks_ < kKS - 2) { W4_WAIT_LGKM(2) } else { W4_WAIT_LGKM(4) }                                \
            }                                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
                const int qs_ = i_ & 1, kh_ = i_ >> 1;                                                         \
                const int s_ = 4 * ks_ + i_;                                                                   \
                /* slice s: the fma of element s + 1 of sub-block 1 (FIRST: its exp2 opens the next slice), the  \
                   MFMA, exp2 of element s, one packed pair (sub-block 0 first: exp2'ed a phase ago; then        \
                   sub-block 1, a slice behind its exp2) */                                                    \
                if (s_ + 1 < 32) { W4_FMA(SC_, 1, s_ + 1, m_run[1], y1_[(s_ + 1) & 1]) }                       \
                if (has_next_) {                                                                               \
                    if (ks_ == 0) w4_mfma_qk0<T>(SN_[qs_][kh_], ka_[ks_ & 1][kh_], qf[qs_][ks_]);              \
                    else w4_mfma_qk<T>(SN_[qs_][kh_], ka_[ks_ & 1][kh_], qf[qs_][ks_]);                        \
                }                                                                                              \
                if (i_ == 1) {                                                                                 \
                    if (ks_ < 4) { if (dma_k_) { W4_DMA_K1(t_ + 2, PAR_, ks_) } }                              \
                    else if (has_next_) { W4_DMA_V1(t_ + 1, 1 - (PAR_), ks_ - 4) }                             \
                }                                                                                              \
                W4_EXPY(SC_, 1, s_, y1_[s_ & 1], psum1_, pend1_)                                               \
                if (2 * s_ < 32) { W4_PACK2(SC_, 0, 2 * s_) }                                                  \
                else if (2 * s_ - 32 < s_ - 1) { W4_PACK2(SC_, 1, 2 * s_ - 32) }                               \
                W4_FENCE                                                                                       \
            }                                                                                                  \
        }                                                                                                      \
        l_run[0] += psum0;                                                                                     \
        l_run[1] += psum1_ + pend1_;                                                                           \
        W4_PACK2(SC_, 1, 30)                          /* the pair whose exp2 came in the last slice */         \
        W4_FENCE                                                                                               \
        W4_STAMP(ts4)                                                                                          \
        /* ================= phase 2: O^T += V(t)^T.P(t)^T  ||  row max of S(t+1), exp2 of its sub-block 0 == */ \
        if (has_next_) { W4_MASK(SN_, t_ + 1) }                                                                \
        float mxp_[2] = {-INFINITY, -INFINITY}, m_new_[2] = {m_run[0], m_run[1]};                              \
        float psn_ = 0.f, pendn_ = 0.f, yn_[2] = {0.f, 0.f};                                                   \
        _Pragma("unroll") for (int n_ = 0; n_ < 4 * kDT; ++n_) {                                               \
            if (n_ + 2 < 4 * kDT) { W4_LOADV(n_ + 2, vs_off_) }                                                \
            /* pair n landed: behind it are pairs n+1, n+2 (2 reads each) */                                   \
            if (n_ + 2 < 4 * kDT) { W4_WAIT_LGKM(4) } else if (n_ + 1 < 4 * kDT) { W4_WAIT_LGKM(2) } else { W4_WAIT_LGKM(0) } \
            const u32x4_t va_ = {vlo_[n_ & 3][0], vlo_[n_ & 3][1], vhi_[n_ & 3][0], vhi_[n_ & 3][1]};          \
            _Pragma("unroll") for (int qs_ = 0; qs_ < 2; ++qs_) {                                              \
                const int s_ = 2 * n_ + qs_;                             /* slice 0..31 */                     \
                /* exp2 schedule of sub-block 0 of S(t+1): elements 2(s-5), 2(s-5)+1 in slices 5..9, element s  \
                   from slice 10 on; every fma one slice (or half a slice) ahead of its exp2 */                \
                if (has_next_ && s_ >= 10 && s_ + 1 < 32) { W4_FMA(SN_, 0, s_ + 1, m_new_[0], yn_[(s_ + 1) & 1]) } \
                w4_o_mfma<T>(4 * qs_ + (n_ & 3), va_, pb[qs_][n_ >> 2]);                                       \
                if (has_next_) {                                                                               \
                    if (s_ < 4) {               /* a quarter of both sub-blocks' row maxima: two independent chains */ \
                        const int kh_ = s_ >> 1, o_ = 8 * (s_ & 1);                                            \
                        mxp_[0] = fmx(mxp_[0], W4_MAX8(SN_[0][kh_], o_));                                      \
                        mxp_[1] = fmx(mxp_[1], W4_MAX8(SN_[1][kh_], o_));                                      \
                        asm volatile("" : "+v"(mxp_[0]), "+v"(mxp_[1]));                                       \
                    } else if (s_ == 4) {                                                                      \
                        float mx0_, mx1_;                                                                      \
                        W4_SWAPMAX(mxp_[0], mx0_)                                                              \
                        W4_SWAPMAX(mxp_[1], mx1_)                                                              \
                        m_new_[0] = fmx(m_run[0], mx0_ * c2);                                                  \
                        m_new_[1] = fmx(m_run[1], mx1_ * c2);                                                  \
                        if (!__any(fmx(m_new_[0] - m_run[0], m_new_[1] - m_run[1]) > rescale_thr<T>())) {           \
                            m_new_[0] = m_run[0];                                                              \
                            m_new_[1] = m_run[1];                                                              \
                        }                                                                                      \
                        asm volatile("" : "+v"(m_new_[0]), "+v"(m_new_[1]));                                   \
                        W4_FMA(SN_, 0, 0, m_new_[0], yn_[0])                                                   \
                        W4_FMA(SN_, 0, 1, m_new_[0], yn_[1])                                                   \
                    } else if (s_ < 10) {                                                                      \
                        W4_EXPY(SN_, 0, 2 * (s_ - 5), yn_[0], psn_, pendn_)                                    \
                        W4_EXPY(SN_, 0, 2 * (s_ - 5) + 1, yn_[1], psn_, pendn_)                                \
                        W4_FMA(SN_, 0, 2 * (s_ - 5) + 2, m_new_[0], yn_[0])        /* slice 9: element 10 */        \
                        if (s_ < 9) { W4_FMA(SN_, 0, 2 * (s_ - 5) + 3, m_new_[0], yn_[1]) }                    \
                    } else {                                                                                   \
                        W4_EXPY(SN_, 0, s_, yn_[s_ & 1], psn_, pendn_)                                         \
                    }                                                                                          \
                }                                                                                              \
                W4_FENCE                                                                                       \
            }                                                                                                  \
        }                                                                                                      \
        /* deferred rescale, between tiles: O, l and m of a sub-block move together */                         \
        if (has_next_) {                                                                                       \
            if (__builtin_expect(__any(m_new_[0] != m_run[0] || m_new_[1] != m_run[1]), 0)) {                  \
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       /* MFMA writes of O -> vector reads */ \
                _Pragma("unroll") for (int qs_ = 0; qs_ < 2; ++qs_) {                                          \
                    const float alpha_ = __builtin_amdgcn_exp2f(m_run[qs_] - m_new_[qs_]);                     \
                    m_run[qs_] = m_new_[qs_];                                                                  \
                    l_run[qs_] *= alpha_;                                                                      \
                    w4_o_scale(qs_, alpha_);                                                                   \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        psum0 = psn_ + pendn_;                                                                                 \
        W4_FENCE                                                                                               \
        W4_STAMP(ts5)                                                                                          \
        W4_STAMP_ACC                                                                                           \
    }

#ifdef FINO_ATTN_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0, sa4 = 0;
#define W4_STAMP_ACC { sa0 += ts1 - ts0; sa1 += ts2 - ts1; sa2 += ts3 - ts2; sa3 += ts4 - ts3; sa4 += ts5 - ts4; }
#else
#define W4_STAMP_ACC
#endif
    // tiles 0 .. nt-2 have a successor (two per trip: the S buffers and the ring slots swap roles); the last one does not
#define W4_TILE(A_, B_, T_, P_, H_) if constexpr (FOLD) { W4_TILE_F(A_, B_, T_, P_, H_) } else if constexpr (D == 128) { W4_TILE_N(A_, B_, T_, P_, H_) }
    int t = 0;
    for (; t + 2 < nt; t += 2) {
        W4_TILE(sa, sb, t, 0, true)
        W4_TILE(sb, sa, t + 1, 1, true)
    }
    if (t + 1 < nt) {
        W4_TILE(sa, sb, t, 0, true)
        W4_TILE(sb, sa, t + 1, 1, false)
    } else {
        W4_TILE(sa, sb, t, 0, false)
    }

#ifdef FINO_ATTN_STAMP
    if (blockIdx.x == 40 && lane == 0 && VAR == 0) {
        fino_attn_w4_dbg[wave * 8 + 0] = sa0; fino_attn_w4_dbg[wave * 8 + 1] = sa1; fino_attn_w4_dbg[wave * 8 + 2] = sa2;
        fino_attn_w4_dbg[wave * 8 + 3] = sa3; fino_attn_w4_dbg[wave * 8 + 4] = sa4; fino_attn_w4_dbg[wave * 8 + 5] = (unsigned long long)nt;
    }
#endif
    // ---------------- epilogue: normalise, store O[q][d] (or leave the (O, m, l) partial) ----------------
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");          // last MFMA writes of O -> vector reads
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float l;
        if constexpr (kMS) {                       // every row of the ones tile holds the full row sum of its query
            float fl[16];
            w4_o_read(4 * qs + 2, fl);
            l = fl[0];
        } else {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[qs]), __float_as_uint(l_run[qs]), false, false);
            l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        if (part >= 0) {
            // same layout as the 8-wave kernel's partials: its wave 2w + qs owns these 32 query rows
            float* w = p.ws + (int64_t)part * partial_floats<D>();
            int tid8 = (2 * wave + qs) * 64 + lane;
            asm volatile("" : "+v"(tid8));   // opaque: or the 64 store offsets are computed before the piece loop and spilled
#pragma unroll
            for (int dt = 0; dt < kDT; ++dt) {
                float f[16];
                w4_o_read(4 * qs + dt, f);
#pragma unroll
                for (int j = 0; j < 16; ++j) w[(dt * 16 + j) * (kWaves * 64) + tid8] = f[j];
            }
            w[kDT * 16 * (kWaves * 64) + tid8] = m_run[qs];
            w[kDT * 16 * (kWaves * 64) + kWaves * 64 + tid8] = l;
            continue;
        }
        const float inv = 1.0f / l;
        // whole-row stores through the (now idle) rings, 32 x 2 D bytes per wave: attn_rows_through_lds
        if (qs == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile have landed ...
            __builtin_amdgcn_s_barrier();                      // ... for every wave of the workgroup
        }
        f32x16_t ov[kDT];
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) {
            float f[16];
            w4_o_read(4 * qs + dt, f);
#pragma unroll
            for (int j = 0; j < 16; ++j) ov[dt][j] = f[j];
        }
        int le = lane;
        asm volatile("" : "+v"(le));       // opaque: or the epilogue's per-lane offsets are computed before the loop and spilled
        u32x4_t rows[D / 16];
        attn_rows_through_lds<T, D>(ov, inv, (uint32_t)(wave * (64 * D)), le & 31, le >> 5, le, rows);
        attn_store_rows<D>(rows, op, p.o_rs, qrow[qs] - (lane & 31), p.lq, le);
    }
  }   // piece
}

template <typename T, int D, bool FOLD>
int launch_w4(const AttnParams& p, hipStream_t st) {
    constexpr int smem = 4 * kKV * D * 2;
    static FinoPerDeviceOnce once_a, once_b;
    if (int rc = fino_max_smem_once(once_a, reinterpret_cast<const void*>(&attn_w4_kernel<T, D, 0, FOLD>), smem, "fino_attn_fwd")) return rc;
    if (int rc = fino_max_smem_once(once_b, reinterpret_cast<const void*>(&attn_w4_kernel<T, D, 1, FOLD>), smem, "fino_attn_fwd")) return rc;
    const dim3 grid((unsigned)(8 * (p.full_x + p.nwg)));
    if (p.lk > 1024)
        attn_w4_kernel<T, D, 0, FOLD><<<grid, kW4Threads, smem, st>>>(p);
    else
        attn_w4_kernel<T, D, 1, FOLD><<<grid, kW4Threads, smem, st>>>(p);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

}  // namespace

// the main launch only: p carries the tail-split plan (full_x, rem_x, nwg, per) / all_partial; the caller runs the combine.
// head_dim 64 exists for the folded-scale path only (p.scale_log2 == 1): fino_attn_w4_supports says so.
bool fino_attn_w4_supports(int head_dim, float scale_log2) { return head_dim == 128 || (head_dim == 64 && scale_log2 == 1.0f); }
int fino_attn_launch_w4(const AttnParams& p, int dtype, int head_dim, hipStream_t st) {
    // scale * log2(e) == 1 exactly <=> the caller said FINO_ATTN_SCALE_FOLDED: q carries the softmax scale already
    const bool fold = p.scale_log2 == 1.0f;
    if (head_dim == 64) return dtype == FINO_BF16 ? launch_w4<BF16, 64, true>(p, st) : launch_w4<F16, 64, true>(p, st);
    if (fold) return dtype == FINO_BF16 ? launch_w4<BF16, 128, true>(p, st) : launch_w4<F16, 128, true>(p, st);
    return dtype == FINO_BF16 ? launch_w4<BF16, 128, false>(p, st) : launch_w4<F16, 128, false>(p, st);
}
