// Non-causal flash attention forward with fp8 (OCP e4m3) MATRIX OPERANDS for head_dim 64: BASELINE config 5's "fp8 MFMA
// path" applied to the operator that is 55 % of the CogVideoX-5B step (F.scaled_dot_product_attention at
// architecture/attention_processor.py:2863 of the reference; no reference counterpart for the precision -- SURVEY F11 --
// so parity is stated against fp32 SDPA and against this library's own bf16 kernel).
//
// Both products run on v_mfma_scale_f32_32x32x64_f8f6f4 (block-scaled: one e8m0 scale per 32 K-elements per row; 2x the
// bf16 rate per clock and ONE instruction where the bf16 kernels issue four):
//   S^T = K . Q^T   A = K tile rows (keys), B = Q rows; the K dimension is the whole head (64 = two scale blocks)
//   O^T = V^T . P^T A = V^T rows (head channels), B = P rows; the K dimension is the tile's 64 keys
// Operand / scale / result maps were probed on hardware (tools/fp8/mfma_scale_probe_32x32x64.hip): see kChunk below.
//
//   * K and V are quantised ONCE per call by fino_attn_quantize_kv_fp8 into tile-major images the main kernel stages
//     with plain 16-byte copies: K8 [tile][key][64 B] + one scale per (key, 32-channel block); V8T [tile][channel][64
//     key slots] -- V already TRANSPOSED and its keys in the order the S^T accumulators hold them (register j of lane
//     group g = key (j & 3) + 8 (j >> 2) + 4 g of each 32-key half), so P never moves between lanes and the V^T fragment
//     is two ds_read_b128 (the bf16 kernels need eight ds_read_b64_tr_b16) -- + one scale per (channel, 32-key block).
//   * Q (bf16 / fp16, already multiplied by softmax_scale * log2 e in fp32) is quantised in registers at block start.
//   * P = exp2(s - m + 6) is rounded to e4m3 with a fixed block scale 2^-6: between rescales p <= 2^kThr, so P8 <= 2^8 <
//     448, and what underflows (p < 2^-15 of the running maximum) carries no weight.  l sums the ROUNDED P on the matrix
//     pipe (a ones row appended to V^T), so numerator and denominator see the same numbers.
//   * 8 waves x 32 query rows per workgroup, K/V tiles double-buffered in LDS (9 KB per stage), one barrier per tile; the
//     same (O, m, l) partial layout, tail split and combine kernels as fino_attention.hip.
#include <stdlib.h>

#include "fino_attention_common.h"
using namespace fino_attn_ns;

namespace {

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int kD8 = 64;                      // head_dim
constexpr int kPShift = 6;                   // P8 = e4m3(p * 2^6), block scale 2^-6
constexpr float kThr8 = 2.0f;                // deferred-rescale threshold (log2): p <= 4 between rescales, P8 <= 256
constexpr int kTileK8 = kKV * kD8;           // 4096 B: K8 tile / V8T tile
// Which 32 of the 64 K-elements of its row lane group g (= lane >> 5) holds, as two 16-byte chunks of the row:
// chunk c0 = g, c1 = 2 + g  (k = 16 g .. 16 g + 15 and 32 + 16 g .. 32 + 16 g + 15), and the scale operand of lane group g
// is the scale of K-block g (k in [32 g, 32 g + 32)) -- the 16x16x128 form's rule, re-probed for 32x32x64.
__device__ __forceinline__ int chunk0(int g) { return g; }
__device__ __forceinline__ int chunk1(int g) { return 2 + g; }

// key of k-position kappa (0..63) of a V8T row / of P's B operand: kappa = 32 * half + 16 * g + j
__host__ __device__ __forceinline__ int slot_key(int kappa) {
    const int half = kappa >> 5, g = (kappa >> 4) & 1, j = kappa & 15;
    return 32 * half + (j & 3) + 8 * (j >> 2) + 4 * g;
}

__device__ __forceinline__ int e8m0_of_amax(float amax) {
    // exponent e with amax / 2^e in (224, 448]; byte = e + 127; amax == 0 -> the smallest scale
    if (!(amax > 0.f)) return 0;
    int ex;
    (void)frexpf(amax * (1.0f / 448.0f), &ex);
    ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
    return ex + 127;
}
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}

struct QuantParams {
    const uint16_t* k;
    const uint16_t* v;
    uint8_t* k8;       // [B*H][nt][64][64]
    uint8_t* ks;       // [B*H][nt][64][2]
    uint8_t* v8t;      // [B*H][nt][64][64]
    uint8_t* vs;       // [B*H][nt][64][2]
    int batch, heads, lk, nt;
    int64_t k_bs, k_rs, k_hs, v_bs, v_rs, v_hs;
};

// one workgroup per (batch * head, key tile): 256 threads; threads 0..127 own (key, channel block) of K, 128..255
// (channel, key block) of V
template <typename T>
__global__ __launch_bounds__(256) void attn_quant_kv_fp8_kernel(const QuantParams p) {
    __shared__ float kt[kKV][kD8 + 1];
    __shared__ float vt[kKV][kD8 + 1];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, hb = blockIdx.y;
    const int bi = hb / p.heads, head = hb - bi * p.heads;
    const uint16_t* kp = p.k + bi * p.k_bs + head * p.k_hs;
    const uint16_t* vp = p.v + bi * p.v_bs + head * p.v_hs;
    // 64 rows x 8 chunks of 8 elements per operand = 512 chunk loads each: 2 + 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cid = tid + 256 * i, row = cid >> 3, ch = cid & 7;
        const int key = tile * kKV + row;
        uint4 ku = make_uint4(0, 0, 0, 0), vu = make_uint4(0, 0, 0, 0);
        if (key < p.lk) {
            ku = *reinterpret_cast<const uint4*>(kp + (int64_t)key * p.k_rs + ch * 8);
            vu = *reinterpret_cast<const uint4*>(vp + (int64_t)key * p.v_rs + ch * 8);
        }
        float kf[8], vf[8];
        unpack8<T>(ku, kf);
        unpack8<T>(vu, vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            kt[row][ch * 8 + e] = kf[e];
            vt[row][ch * 8 + e] = vf[e];
        }
    }
    __syncthreads();
    const int64_t tbase = ((int64_t)hb * p.nt + tile);
    const int item = tid & 127, a = item >> 1, blk = item & 1;       // a = key (K) / channel (V)
    float x[32];
    if (tid < 128) {
#pragma unroll
        for (int e = 0; e < 32; ++e) x[e] = kt[a][32 * blk + e];
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) x[e] = vt[slot_key(32 * blk + e)][a];
    }
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf(x[e]));
    const int sb = e8m0_of_amax(amax);
    const float inv = __builtin_amdgcn_ldexpf(1.0f, 127 - sb);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = pack4_fp8(x[4 * i] * inv, x[4 * i + 1] * inv, x[4 * i + 2] * inv, x[4 * i + 3] * inv);
    uint8_t* dst = (tid < 128 ? p.k8 : p.v8t) + tbase * kTileK8 + a * 64 + 32 * blk;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    (tid < 128 ? p.ks : p.vs)[tbase * 128 + a * 2 + blk] = (uint8_t)sb;
}

struct Fp8AttnParams {
    AttnParams a;          // q / o and their strides, lq / lk, batch / heads, nqb, tail split, scale_log2
    const uint8_t* k8;
    const uint8_t* ks;
    const uint8_t* v8t;
    const uint8_t* vs;
    int nt;
};

// LDS: K8 ring (2 x 4 KiB) | V8T ring (2 x 4 KiB) | K scale ring (2 x 128 B) | V scale ring (2 x 128 B)
constexpr int kLdsK = 0, kLdsV = 2 * kTileK8, kLdsKS = 4 * kTileK8, kLdsVS = 4 * kTileK8 + 256;
constexpr int kSmem8 = 4 * kTileK8 + 512;

// PING-PONG (the structure of attn_pp_kernel, fino_attention.hip): the two waves of a SIMD (w, w + 4) run one phase apart --
// one in its SOFTMAX phase (exp2, e4m3 packing of P(t), the rescale decision, its share of the K / V staging), the other in
// its MATRIX phase (S(t+1) = -m + K(t+1).Q^T, O^T += V(t)^T.P(t)^T, l^T += 1.P(t)^T, row maximum of S(t+1) in the MFMAs'
// shadow); two s_barrier per tile.  The running maximum rides into S as one more product on the matrix pipe ("ones" x (-m),
// a bf16 MFMA into the same accumulator: m is kept bf16-representable so the product is exact), so the softmax is a bare
// exp2 + pack.  Staging: K(w + 1) and V(w) may be written during phases 2w - 1 and 2w (their ring slots are free from
// 2w - 1, they are first read in phase 2w + 1): group g writes its half of both in its softmax phase of tile t = w - g.
template <typename T, int VAR>
__global__ __launch_bounds__(kWaves * 64, 2) void attn_fp8_kernel(const Fp8AttnParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnParams& p = fp.a;
    constexpr int kDT = kD8 / 32;            // 2 d-tiles of O^T
    typedef typename T::vec8 vec8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int g = lane >> 5;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int tg = tid & 255;                // thread in its group

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    const int ntall = fp.nt;
    int npieces = 1, first_b = 0;
    int64_t g0 = 0, g1 = 0;
    if (slot >= p.full_x) {
        g0 = (int64_t)(slot - p.full_x) * p.per;
        g1 = g0 + p.per < (int64_t)p.rem_x * ntall ? g0 + p.per : (int64_t)p.rem_x * ntall;
        first_b = (int)(g0 / ntall);
        npieces = (int)((g1 - 1) / ntall) - first_b + 1;
    }
  for (int piece = 0; piece < npieces; ++piece) {
    if (piece > 0) __syncthreads();
    int bx = slot, part = -1, t_begin = 0, t_end = ntall;
    if (slot >= p.full_x) {
        const int tb = first_b + piece;
        const int64_t b0 = (int64_t)tb * ntall;
        t_begin = g0 > b0 ? (int)(g0 - b0) : 0;
        t_end = g1 - b0 < ntall ? (int)(g1 - b0) : ntall;
        bx = p.full_x + tb;
        if (t_begin != 0 || t_end != ntall) part = ((xcd * p.nwg) + (slot - p.full_x)) * 2 + piece;
    }
    int hb, qb;
    if (!attn_map_block(p, xcd, bx, hb, qb)) continue;
    if (p.all_partial) part = hb * p.nqb + qb;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    const int64_t tile0 = (int64_t)hb * fp.nt + t_begin;
    const int lk = (t_end * kKV < p.lk ? t_end * kKV : p.lk) - t_begin * kKV;
    const int nt = t_end - t_begin;

    // ---- Q: lane (row r, group g) holds channels 16 g .. 16 g + 15 and 32 + 16 g .. + 15, pre-scaled, as e4m3 with one
    //      scale per 32-channel block (block b = channels [32 b, 32 b + 32): half of it sits in the partner lane) ----
    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
    i32x8_t qf;
    int q_scale;                              // scale byte of block g (the operand this lane group supplies)
    {
        float x[2][16];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = c == 0 ? chunk0(g) : chunk1(g);
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                uint4 u = *reinterpret_cast<const uint4*>(qp + (int64_t)qrow_c * p.q_rs + 16 * ch + 8 * hlf);
                if (qrow >= p.lq) u = make_uint4(0, 0, 0, 0);
                float f[8];
                unpack8<T>(u, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][8 * hlf + e] = f[e] * p.scale_log2;
            }
        }
        float am[2] = {0.f, 0.f};             // my share of block 0 (chunk c = 0) and block 1 (c = 1)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) am[c] = fmaxf(am[c], fabsf(x[c][e]));
        int sbyte[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(am[c]), __float_as_uint(am[c]), false, false);
            sbyte[c] = e8m0_of_amax(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float inv = __builtin_amdgcn_ldexpf(1.0f, 127 - sbyte[c]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                qf[4 * c + i] = (int)pack4_fp8(x[c][4 * i] * inv, x[c][4 * i + 1] * inv, x[c][4 * i + 2] * inv,
                                               x[c][4 * i + 3] * inv);
        }
        q_scale = g == 0 ? sbyte[0] : sbyte[1];
    }

    // ---- staging roles: threads 0..127 of a group move chunk (128 grp + i) of a K8 tile, 128..255 of a V8T tile (16 B
    //      each); the first 16 of either set also dword (16 grp + i) of that tile's 128 scale bytes ----
    const bool st_k = tg < 128;
    const int st_c = grp * 128 + (tg & 127);
    const bool st_s = (tg & 127) < 16;
    const uint8_t* st_src = (st_k ? fp.k8 : fp.v8t) + tile0 * kTileK8 + st_c * 16;
    const uint8_t* st_ssrc = (st_k ? fp.ks : fp.vs) + tile0 * 128 + (grp * 16 + (tg & 15)) * 4;
    const int st_dst = (st_k ? kLdsK : kLdsV) + st_c * 16;
    const int st_sdst = (st_k ? kLdsKS : kLdsVS) + (grp * 16 + (tg & 15)) * 4;
    const int st_add = st_k ? 1 : 0;         // K runs one tile ahead of V
    uint4 st_reg = make_uint4(0, 0, 0, 0);
    uint32_t st_sreg = 0;
#define F8_LOAD(W_)                                                                                          \
    {                                                                                                        \
        int tt_ = (W_) + st_add;                                                                             \
        tt_ = tt_ < nt ? tt_ : nt - 1;               /* past the last tile: a harmless re-read */            \
        st_reg = *reinterpret_cast<const uint4*>(st_src + (int64_t)tt_ * kTileK8);                           \
        if (st_s) st_sreg = *reinterpret_cast<const uint32_t*>(st_ssrc + (int64_t)tt_ * 128);                \
    }
#define F8_WRITE(W_)                                                                                         \
    {                                                                                                        \
        const int sl_ = ((W_) + st_add) & 1;                                                                 \
        *reinterpret_cast<uint4*>(smem + st_dst + sl_ * kTileK8) = st_reg;                                   \
        if (st_s) *reinterpret_cast<uint32_t*>(smem + st_sdst + sl_ * 128) = st_sreg;                        \
    }
    // ---- prologue: K(0), V(0), K(1) whole (both groups, each its halves) ----
    {
        const uint8_t* k8 = fp.k8 + tile0 * kTileK8;
        const uint8_t* v8 = fp.v8t + tile0 * kTileK8;
        const int c = tid;                                   // 512 threads: K(0) = chunks 0..255, V(0) = 256..511
        const uint4 a0 = *reinterpret_cast<const uint4*>((c < 256 ? k8 : v8) + (c & 255) * 16);
        *reinterpret_cast<uint4*>(smem + (c < 256 ? kLdsK : kLdsV) + (c & 255) * 16) = a0;
        if (c < 256) {
            const int t1 = nt > 1 ? 1 : 0;
            const uint4 a1 = *reinterpret_cast<const uint4*>(k8 + (int64_t)t1 * kTileK8 + c * 16);
            *reinterpret_cast<uint4*>(smem + kLdsK + kTileK8 + c * 16) = a1;
        }
        if (tid < 32) *reinterpret_cast<uint32_t*>(smem + kLdsKS + tid * 4) =
            *reinterpret_cast<const uint32_t*>(fp.ks + tile0 * 128 + tid * 4);
        else if (tid < 64) *reinterpret_cast<uint32_t*>(smem + kLdsVS + (tid - 32) * 4) =
            *reinterpret_cast<const uint32_t*>(fp.vs + tile0 * 128 + (tid - 32) * 4);
        else if (tid < 96 && nt > 1) *reinterpret_cast<uint32_t*>(smem + kLdsKS + 128 + (tid - 64) * 4) =
            *reinterpret_cast<const uint32_t*>(fp.ks + (tile0 + 1) * 128 + (tid - 64) * 4);
    }
    if (grp == 1) { F8_LOAD(1) }             // group 1 writes {K(2), V(1)} in its first softmax phase
    __syncthreads();

    f32x16_t o[kDT], lacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[0][j] = 0.f; o[1][j] = 0.f; lacc[j] = 0.f; }
    const i32x8_t ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    constexpr int kOne = 127, kPs = 127 - kPShift;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // "ones" x (-m): A[key][k = 0] = 1 (k = 0 lives in element 0 of the g = 0 lanes), B[k = 0][q] = -m[q]
    uint4 ones_u = make_uint4(g == 0 ? (uint32_t)T::from_f32(1.0f) : 0u, 0u, 0u, 0u);
    asm volatile("" : "+v"(ones_u.x));

    // K(tile in slot KS_) . Q^T (+ C_) -> two 32-key halves
#define F8_QK(KS_, C0_, C1_, S0_, S1_)                                                                       \
    {                                                                                                        \
        const char* kb_ = smem + kLdsK + (KS_) * kTileK8;                                                    \
        const i32x4_t a00_ = *reinterpret_cast<const i32x4_t*>(kb_ + r * 64 + 16 * chunk0(g));               \
        const i32x4_t a01_ = *reinterpret_cast<const i32x4_t*>(kb_ + r * 64 + 16 * chunk1(g));               \
        const i32x4_t a10_ = *reinterpret_cast<const i32x4_t*>(kb_ + (32 + r) * 64 + 16 * chunk0(g));        \
        const i32x4_t a11_ = *reinterpret_cast<const i32x4_t*>(kb_ + (32 + r) * 64 + 16 * chunk1(g));        \
        const int ks0_ = *reinterpret_cast<const uint8_t*>(smem + kLdsKS + (KS_) * 128 + r * 2 + g);         \
        const int ks1_ = *reinterpret_cast<const uint8_t*>(smem + kLdsKS + (KS_) * 128 + (32 + r) * 2 + g);  \
        const i32x8_t k0_ = __builtin_shufflevector(a00_, a01_, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        const i32x8_t k1_ = __builtin_shufflevector(a10_, a11_, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        S0_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k0_, qf, C0_, 0, 0, 0, ks0_, 0, q_scale);      \
        S1_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k1_, qf, C1_, 0, 0, 0, ks1_, 0, q_scale);      \
    }
    // keys past lk (zero rows of a ragged last tile): out of the maximum, p = exp2(-inf) = 0
#define F8_MASK(T_, S0_, S1_)                                                                                \
    if (__builtin_expect((T_) == nt - 1 && (lk & (kKV - 1)), 0)) {                                           \
        int rem_ = lk - (T_) * kKV - 4 * g;                                                                  \
        asm volatile("" : "+v"(rem_));      /* opaque: keeps this a BRANCH (if-converted it is 62 selects per tile) */ \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                  \
            const int key_ = (j_ & 3) + 8 * (j_ >> 2);                                                       \
            if (key_ >= rem_) S0_[j_] = -INFINITY;                                                           \
            if (key_ + 32 >= rem_) S1_[j_] = -INFINITY;                                                      \
        }                                                                                                    \
    }
#define F8_MAX8(S_, O_) vmax2(vmax3(vmax3(S_[O_], S_[O_ + 1], S_[O_ + 2]), vmax3(S_[O_ + 3], S_[O_ + 4], S_[O_ + 5]), \
                                    S_[O_ + 6]), S_[O_ + 7])
#define F8_SWAPMAX(MX_, OUT_)                                                                                \
    {                                                                                                        \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                      \
    }

    // ---- S(0) unpipelined; m_run = the value SUBTRACTED from s (running maximum minus kPShift, T-representable) ----
    f32x16_t s0, s1;
    F8_QK(0, zero16, zero16, s0, s1)
    F8_MASK(0, s0, s1)
    float m_run;
    {
        float mx = vmax2(vmax3(F8_MAX8(s0, 0), F8_MAX8(s0, 8), F8_MAX8(s1, 0)), F8_MAX8(s1, 8));
        float mxx;
        F8_SWAPMAX(mx, mxx)
        m_run = T::to_f32(T::from_f32(mxx - (float)kPShift));
#pragma unroll
        for (int j = 0; j < 16; ++j) { s0[j] -= m_run; s1[j] -= m_run; }
    }
    float ex_next = 0.f;                      // max over the tile of (s - m_run): what may exceed 8
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on

    for (int t = 0; t < nt; ++t) {
        // ================= softmax phase =================
        const int w = t + grp;
        if (w >= 1) { F8_WRITE(w) }
        {
            // deferred rescale: s already carries -m_run; move m only when P8 would pass 2^(kPShift + kThr8) = 256
            if (__any(ex_next > (float)kPShift + kThr8)) {
                const float mn = T::to_f32(T::from_f32(m_run + fmaxf(ex_next - (float)kPShift, 0.f)));
                const float dm = mn - m_run;
                m_run = mn;
                const float alpha = __builtin_amdgcn_exp2f(-dm);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s0[j] -= dm; s1[j] -= dm;
                    o[0][j] *= alpha; o[1][j] *= alpha; lacc[j] *= alpha;
                }
            }
        }
        i32x8_t pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[i] = (int)pack4_fp8(__builtin_amdgcn_exp2f(s0[4 * i]), __builtin_amdgcn_exp2f(s0[4 * i + 1]),
                                   __builtin_amdgcn_exp2f(s0[4 * i + 2]), __builtin_amdgcn_exp2f(s0[4 * i + 3]));
            pf[4 + i] = (int)pack4_fp8(__builtin_amdgcn_exp2f(s1[4 * i]), __builtin_amdgcn_exp2f(s1[4 * i + 1]),
                                       __builtin_amdgcn_exp2f(s1[4 * i + 2]), __builtin_amdgcn_exp2f(s1[4 * i + 3]));
        }
        F8_LOAD(w + 1)
        {   // the softmax belongs to THIS phase: pin its results here (pure arithmetic otherwise sinks past the barrier)
            asm volatile("" : "+v"(pf));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= matrix phase =================
        __builtin_amdgcn_s_setprio(1);
        if (t + 1 < nt) {
            uint4 mn_u = make_uint4(g == 0 ? (uint32_t)T::from_f32(-m_run) : 0u, 0u, 0u, 0u);
            const vec8 onesv = __builtin_bit_cast(vec8, ones_u), mnegv = __builtin_bit_cast(vec8, mn_u);
            f32x16_t c0 = T::mfma32(onesv, mnegv, zero16);
            f32x16_t c1 = T::mfma32(onesv, mnegv, zero16);
            F8_QK((t + 1) & 1, c0, c1, s0, s1)
        }
        {
            const char* vb = smem + kLdsV + (t & 1) * kTileK8;
            const char* vsb = smem + kLdsVS + (t & 1) * 128;
#pragma unroll
            for (int dt = 0; dt < kDT; ++dt) {
                const i32x4_t v0 = *reinterpret_cast<const i32x4_t*>(vb + (32 * dt + r) * 64 + 16 * chunk0(g));
                const i32x4_t v1 = *reinterpret_cast<const i32x4_t*>(vb + (32 * dt + r) * 64 + 16 * chunk1(g));
                const int vs_ = *reinterpret_cast<const uint8_t*>(vsb + (32 * dt + r) * 2 + g);
                const i32x8_t vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vv, pf, o[dt], 0, 0, 0, vs_, 0, kPs);
            }
            lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones8, pf, lacc, 0, 0, 0, kOne, 0, kPs);
        }
        if (t + 1 < nt) {
            F8_MASK(t + 1, s0, s1)
            const float mx = vmax2(vmax3(F8_MAX8(s0, 0), F8_MAX8(s0, 8), F8_MAX8(s1, 0)), F8_MAX8(s1, 8));
            F8_SWAPMAX(mx, ex_next)
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef F8_LOAD
#undef F8_WRITE
#undef F8_QK
#undef F8_MASK
#undef F8_MAX8
#undef F8_SWAPMAX

    // ---- epilogue: every register of lacc holds l of this lane's query (all rows of the ones tile are equal) ----
    float l_run = lacc[0];
    if (part >= 0) {
        // partial in the bf16 kernels' layout; m, O and l share the 2^kPShift offset, which cancels in the merge
        float* w = p.ws + (int64_t)part * partial_floats<kD8>();
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt)
#pragma unroll
            for (int j = 0; j < 16; ++j) w[(dt * 16 + j) * (kWaves * 64) + tid] = o[dt][j];
        w[kDT * 16 * (kWaves * 64) + tid] = m_run;
        w[kDT * 16 * (kWaves * 64) + kWaves * 64 + tid] = l_run;
        continue;
    }
    const float inv = 1.0f / l_run;
    if (qrow < p.lq) {
        uint16_t* orow = op + (int64_t)qrow * p.o_rs;
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d0 = dt * 32 + 8 * gq + 4 * g;
                uint32_t w0 = (uint32_t)T::from_f32(o[dt][4 * gq + 0] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * gq + 1] * inv) << 16);
                uint32_t w1 = (uint32_t)T::from_f32(o[dt][4 * gq + 2] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * gq + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(orow + d0) = make_uint2(w0, w1);
            }
        }
    }
  }   // piece
}

}  // namespace

extern "C" int64_t fino_attn_fp8_kv_bytes(int batch, int heads, int64_t lk, int head_dim) {
    if (batch <= 0 || heads <= 0 || lk <= 0 || head_dim != kD8) return 0;
    const int64_t nt = (lk + kKV - 1) / kKV;
    return (int64_t)batch * heads * nt * (2 * kTileK8 + 256);
}

extern "C" int fino_attn_fwd_fp8(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                                 int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                                 int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, float scale, int dtype,
                                 void* kv_workspace, int64_t kv_workspace_bytes, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_fwd_fp8: dtype %d", dtype);
    FINO_CHECK(head_dim == kD8, FINO_ERR_UNSUPPORTED, "fino_attn_fwd_fp8: head_dim %d (built for 64)", head_dim);
    FINO_CHECK(q && k && v && o && kv_workspace, FINO_ERR_ARG, "fino_attn_fwd_fp8: null pointer");
    FINO_CHECK(batch > 0 && heads > 0 && lq >= 0 && lk > 0, FINO_ERR_ARG, "fino_attn_fwd_fp8: bad shape");
    FINO_CHECK(fino_aligned16(q) && fino_aligned16(k) && fino_aligned16(v) && fino_aligned16(o) &&
                   fino_aligned16(kv_workspace) && q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 &&
                   q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 8 == 0,
               FINO_ERR_ARG, "fino_attn_fwd_fp8: pointers and strides must be 16-byte aligned");
    FINO_CHECK(scale > 0.f || scale == FINO_ATTN_SCALE_FOLDED, FINO_ERR_ARG, "fino_attn_fwd_fp8: scale");
    const int64_t need = fino_attn_fp8_kv_bytes(batch, heads, lk, head_dim);
    FINO_CHECK(kv_workspace_bytes >= need, FINO_ERR_ARG, "fino_attn_fwd_fp8: workspace %lld B < %lld B",
               (long long)kv_workspace_bytes, (long long)need);
    if (lq == 0) return FINO_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nt = (int)((lk + kKV - 1) / kKV);
    const int64_t bh = (int64_t)batch * heads;
    uint8_t* w8 = (uint8_t*)kv_workspace;
    QuantParams qp;
    qp.k = (const uint16_t*)k; qp.v = (const uint16_t*)v;
    qp.k8 = w8; qp.v8t = w8 + bh * nt * kTileK8; qp.ks = w8 + 2 * bh * nt * kTileK8; qp.vs = qp.ks + bh * nt * 128;
    qp.batch = batch; qp.heads = heads; qp.lk = (int)lk; qp.nt = nt;
    qp.k_bs = k_bs; qp.k_rs = k_rs; qp.k_hs = head_dim; qp.v_bs = v_bs; qp.v_rs = v_rs; qp.v_hs = head_dim;
    if (dtype == FINO_BF16) attn_quant_kv_fp8_kernel<BF16><<<dim3((unsigned)nt, (unsigned)bh), 256, 0, st>>>(qp);
    else attn_quant_kv_fp8_kernel<F16><<<dim3((unsigned)nt, (unsigned)bh), 256, 0, st>>>(qp);
    FINO_LAUNCH_CHECK();

    Fp8AttnParams fp;
    AttnParams& p = fp.a;
    p.q = (const uint16_t*)q; p.k = nullptr; p.v = nullptr; p.o = (uint16_t*)o;
    p.batch = batch; p.heads = heads; p.lq = (int)lq; p.lk = (int)lk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = head_dim; p.k_bs = p.k_rs = p.k_hs = p.v_bs = p.v_rs = p.v_hs = 0;
    p.o_bs = o_bs; p.o_rs = o_rs; p.o_hs = head_dim;
    p.scale_log2 = scale == FINO_ATTN_SCALE_FOLDED ? 1.0f : scale * 1.4426950408889634f;
    p.nqb = (int)((lq + kQBlock - 1) / kQBlock);
    p.ws = nullptr; p.all_partial = 0;
    attn_virtual_heads(p.batch, p.heads, p.nqb, p.vsplit, p.nqb_v);
    const int groups = (p.batch * p.heads * p.vsplit + 7) / 8;
    p.full_x = groups * p.nqb_v; p.rem_x = 0; p.nwg = 0; p.per = 1;
    fp.k8 = qp.k8; fp.ks = qp.ks; fp.v8t = qp.v8t; fp.vs = qp.vs; fp.nt = nt;
    const dim3 grid((unsigned)(8 * p.full_x));
    constexpr int smem = kSmem8;
    if (dtype == FINO_BF16) attn_fp8_kernel<BF16, 0><<<grid, kWaves * 64, smem, st>>>(fp);
    else attn_fp8_kernel<F16, 0><<<grid, kWaves * 64, smem, st>>>(fp);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
