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
//   * Two main kernels.  Default: attn_fp8_fr_kernel -- 4 waves x 32 query rows per workgroup, three workgroups per CU that
//     drift apart freely, one barrier per key tile (a wave of this loop is bound by its own issue rate and dependency chain,
//     not by a pipe: tools/ubench/valu_rate.hip).  FINO_TUNE_ATTN_FP8_KERNEL = 1: attn_fp8_kernel -- 8 waves, the two waves of
//     a SIMD one phase apart (softmax / matrix), two barriers per tile; it also carries the (O, m, l) partial layout of
//     fino_attention.hip.  K8 / V8T tiles travel global -> LDS by DMA into rings, XOR-swizzled through the source offsets.
//     Measurements: DESIGN.md section 4.2, profiles/r03_attn_fp8_*.
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
// LDS image of a K8 / V8T tile: 64 rows of 64 bytes.  A ds_read_b128 serves 16 lanes at a time (one 16-byte chunk of 16
// consecutive rows here): rows r and r + 4 share their banks (4 x 64 B = the 256-byte bank line), a 4-way conflict unless
// the chunk position is permuted by the row: position = chunk ^ swz8(row).  (Rows 32 + r: the same permutation as r.)
__device__ __forceinline__ int swz8(int row) { return (row >> 2) & 3; }

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
    // asm: the first conversion DEFINES the word (its other half is written by the second), so no v_mov 0 in front of it --
    // through the builtin (which takes the old word as an input) that is 8 extra vector instructions per key tile
    uint32_t w;
    asm("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
    asm("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w) : "v"(c), "v"(d));
    return w;
}

// ---- P: how a softmax weight becomes an e4m3 operand byte (template parameter PX of the main kernels) ----
// PX = 0 (FINO_FP8_P_EXP2): p = exp2(s - m) on the transcendental unit, rounded to e4m3 by v_cvt_pk_fp8_f32 -- 32 + 16
//   half-rate instructions per wave and key tile, 60 % of the tile's vector time (tools/ubench/fastexp_probe.hip).
// PX = 1 (FINO_FP8_P_RAMP): the e4m3 byte of 2^x IS, to within 0.69 of a mantissa step, the integer 8 x + 56: the exponent field
//   counts whole octaves and the three mantissa bits interpolate linearly between them (Schraudolph's exponential at 8-bit
//   width).  So the logits are carried in units of 1/8 octave (q pre-multiplied by 8: an exact shift of its e8m0 block scales),
//   the running maximum and the constant 55.5 ride into S through the same "ones" MFMA as before, and ONE v_cvt_pk_u8_f32 per
//   value (round to nearest even, saturating at 0: -inf and underflow give the zero byte) writes the operand byte: 32
//   instructions of the cheaper class instead of 48 of the dearer one.  The weight of a key is then g(s - m) with g(x) = 2^floor(x)
//   (1 + frac(x)) read at a 3-bit mantissa instead of 2^(s - m): within +-3 % of it, the same function in numerator (P.V) and
//   denominator (l sums the same bytes), and -- because g(x - n) = 2^-n g(x) for whole n only -- the running maximum moves in
//   WHOLE octaves, so a deferred rescale and the merge of partials stay exact.  P's own share of the output error goes from
//   2.6e-2 to 3.1e-2 rel-RMS on N(0, 1) logits (K / V / Q quantisation is the larger share either way): tests/test_attention_fp8_gpu.py.
constexpr float kPxC = 55.5f;                // byte = rne(8 (s - m) + kPxC): 56 = (exponent bias 7) x 8, -0.5 centres the ramp's error
template <int PX> struct PxUnit { static constexpr float kS = PX ? 8.0f : 1.0f; };     // S units per octave
__device__ __forceinline__ uint32_t pack4_u8(float a, float b, float c, float d) {
    uint32_t w;
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(w) : "v"(a));          // defines the word (other bytes 0), like pack4_fp8
    asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(b));
    asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(c));
    asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(d));
    return w;
}
template <int PX>
__device__ __forceinline__ uint32_t p_bytes4(float a, float b, float c, float d) {
    if constexpr (PX) return pack4_u8(a, b, c, d);
    else return pack4_fp8(__builtin_amdgcn_exp2f(a), __builtin_amdgcn_exp2f(b), __builtin_amdgcn_exp2f(c), __builtin_amdgcn_exp2f(d));
}
// the value SUBTRACTED from the first tile's raw logits (S units), T-representable (it re-enters S through a T MFMA operand);
// PX: a whole number of octaves
template <typename T, int PX>
__device__ __forceinline__ float px_first_m(float mxx) {
    if constexpr (PX) return 8.0f * T::to_f32(T::from_f32(__builtin_rintf(mxx * 0.125f - (float)kPShift)));
    else return T::to_f32(T::from_f32(mxx - (float)kPShift));
}
// what the first tile's S becomes: raw - m (+ the ramp's constant)
template <int PX> __device__ __forceinline__ float px_first_off(float m_run) { return PX ? kPxC - m_run : -m_run; }
// deferred rescale: S carries -m_run (+ kPxC); m moves only when a weight would pass 2^(kPShift + kThr8)
template <int PX> __device__ __forceinline__ float px_thr() { return PX ? 8.0f * ((float)kPShift + kThr8) + kPxC : (float)kPShift + kThr8; }
template <typename T, int PX>
__device__ __forceinline__ float px_next_m(float m_run, float ex_next) {
    if constexpr (PX)
        return 8.0f * T::to_f32(T::from_f32(m_run * 0.125f + fmaxf(__builtin_rintf((ex_next - kPxC) * 0.125f - (float)kPShift), 0.f)));
    else return T::to_f32(T::from_f32(m_run + fmaxf(ex_next - (float)kPShift, 0.f)));
}
// operands of the "ones" x (-m) MFMA (k = 0, and for PX k = 1: the constant), dword 0 of the g = 0 lanes
template <typename T, int PX> __device__ __forceinline__ uint32_t px_ones_word() {
    return (uint32_t)T::from_f32(1.0f) | (PX ? (uint32_t)T::from_f32(1.0f) << 16 : 0u);
}
template <typename T, int PX> __device__ __forceinline__ uint32_t px_mneg_word(float m_run) {
    return (uint32_t)T::from_f32(-m_run) | (PX ? (uint32_t)T::from_f32(kPxC) << 16 : 0u);
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

// LDS: K8 ring (8 x 4 KiB) | V8T ring (8 x 4 KiB) | K scale ring (8 x 128 B) | V scale ring (8 x 128 B)
constexpr int kRing8 = 8;
#ifndef F8_READS_IN_MATRIX
#define F8_READS_IN_MATRIX 1      // 0: K(t+1) / V(t) fragments are read in the softmax phase, 1: in the matrix phase, 2: K / V split
#endif
constexpr int kLdsK = 0, kLdsV = kRing8 * kTileK8, kLdsKS = 2 * kRing8 * kTileK8, kLdsVS = kLdsKS + kRing8 * 128;
constexpr int kSmem8 = kLdsVS + kRing8 * 128;

// Q row -> e4m3 B operand: lane (row r, group g) holds channels 16 g .. 16 g + 15 and 32 + 16 g .. + 15, pre-scaled, with
// one scale per 32-channel block (block b = channels [32 b, 32 b + 32): half of it sits in the partner lane); q_scale =
// the scale byte of block g (the operand this lane group supplies).  Rows past lq: zeros.
template <typename T>
__device__ __forceinline__ void load_q_fp8(const AttnParams& p, const uint16_t* qp, int qrow, int g, i32x8_t& qf,
                                           int& q_scale) {
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
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
}

#define F8STAMP(V_)

// max of the 16 registers of an S accumulator, ordered BEHIND the MFMA that produced `behind` (an input the asm never reads:
// the data dependence is what keeps the compiler from hoisting the statement above that MFMA).  Plain fmaxf on MFMA results
// costs a canonicalising v_max per input; v_max3 through asm is what the bf16 kernels use as well.
__device__ __forceinline__ float max16_behind(const f32x16_t& s, float behind) {
    float a, b;
    asm("v_max3_f32 %0, %2, %3, %4\n\t"
        "v_max3_f32 %1, %5, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %1, %1, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %14, %15\n\t"
        "v_max3_f32 %0, %0, %16, %17\n\t"
        "v_max_f32 %0, %0, %1"
        : "=&v"(a), "=&v"(b)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]),
          "v"(s[10]), "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]), "v"(behind));
    return a;
}

// PING-PONG (the structure of attn_pp_kernel, fino_attention.hip): the two waves of a SIMD (w, w + 4) run one phase apart --
// one in its SOFTMAX phase (exp2, e4m3 packing of P(t), the rescale decision, its share of the K / V staging), the other in
// its MATRIX phase (S(t+1) = -m + K(t+1).Q^T, O^T += V(t)^T.P(t)^T, l^T += 1.P(t)^T, row maximum of S(t+1) in the MFMAs'
// shadow); two s_barrier per tile.  The running maximum rides into S as one more product on the matrix pipe ("ones" x (-m),
// a bf16 MFMA into the same accumulator: m is kept bf16-representable so the product is exact), so the softmax is a bare
// exp2 + pack.  Staging: rings of eight tiles filled by LDS-DMA; in its softmax phase of tile t a group issues its half of
// K(t+4) and V(t+4) and, before the barrier, waits for its half of t+3 (issued a whole tile earlier: an L2 hit takes about
// one phase, anything further more -- with registers and one phase of distance the wait was 450 cycles per tile).  The
// operands of a matrix phase (K(t+1) and V(t) fragments, their scale bytes: 12 LDS reads) are read at its top
// (F8_READS_IN_MATRIX = 1): read in the softmax phase instead they lengthen the longer phase (1034 vs 1216 TFLOP/s-eq.).
template <typename T, int VAR, int PX>
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
    i32x8_t qf;
    int q_scale;
    load_q_fp8<T>(p, qp, qrow, g, qf, q_scale);

    // ---- staging roles (LDS-DMA, no register round trip): of a group's four waves, 0 / 1 move the group's half of a K8 tile
    //      (64 lanes x 16 B each), 2 / 3 of the V8T tile; lanes 0..15 of each also the group's half of that tile's 128 scale
    //      bytes (waves 0 / 1 and 2 / 3 write the same bytes: one vmcnt per wave and tile, whichever wave it is) ----
    const int wl = __builtin_amdgcn_readfirstlane(wave & 3);
    const bool st_k = wl < 2;
    const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((st_k ? fp.k8 : fp.v8t) + tile0 * kTileK8), 0, nt * kTileK8, 0x00020000);
    const __amdgpu_buffer_rsrc_t st_srsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((st_k ? fp.ks : fp.vs) + tile0 * 128), 0, nt * 128, 0x00020000);
    // LDS position P (16-byte slot of the tile image; row = P >> 2) receives chunk (P & 3) ^ swz(row) of that row: the DMA
    // writes lane-linear, so the permutation is applied to the SOURCE offset (swz8 below: 64-byte rows, conflict-free reads)
    const int st_pos = grp * 128 + (wl & 1) * 64 + lane;
    const uint32_t st_voff = (uint32_t)((st_pos >> 2) * 64 + (((st_pos & 3) ^ swz8(st_pos >> 2)) << 4));
    const uint32_t st_svoff = (uint32_t)((grp * 16 + (lane & 15)) * 4);
    const int st_lds = (st_k ? kLdsK : kLdsV) + (grp * 128 + (wl & 1) * 64) * 16;
    const int st_slds = (st_k ? kLdsKS : kLdsVS) + grp * 64;
    // tile U_ -> its ring slot; past the last tile: a harmless re-read.  Two vector-memory operations per wave.
#define F8_DMA(U_)                                                                                           \
    {                                                                                                        \
        const int tt_ = (U_) < nt ? (U_) : nt - 1;                                                           \
        const int sl_ = (U_) & (kRing8 - 1);                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st_rsrc, (FINO_LDS void*)(smem + st_lds + sl_ * kTileK8), 16, st_voff, \
                                                 tt_ * kTileK8, 0, 0);                                       \
        if (lane < 16)                                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_srsrc, (FINO_LDS void*)(smem + st_slds + sl_ * 128), 4, st_svoff, \
                                                     tt_ * 128, 0, 0);                                       \
    }
    // ---- prologue: tiles 0 .. 3 by the same DMA (each group its halves); 0 and 1 must have landed before the loop ----
    { F8_DMA(0) }
    { F8_DMA(1) }
    { F8_DMA(2) }
    { F8_DMA(3) }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();

    f32x16_t o[kDT], lacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[0][j] = 0.f; o[1][j] = 0.f; lacc[j] = 0.f; }
    const i32x8_t ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    constexpr int kOne = 127, kPs = 127 - kPShift;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // "ones" x (-m): A[key][k = 0] = 1 (k = 0 lives in element 0 of the g = 0 lanes), B[k = 0][q] = -m[q]
    uint4 ones_u = make_uint4(g == 0 ? px_ones_word<T, PX>() : 0u, 0u, 0u, 0u);
    asm volatile("" : "+v"(ones_u.x));

    // K(tile in slot KS_) fragments -> registers; K . Q^T (+ C_) -> two 32-key halves
    i32x4_t ka00, ka01, ka10, ka11;
    int ks0r, ks1r;
#define F8_KREAD(KS_)                                                                                        \
    {                                                                                                        \
        const char* kb_ = smem + kLdsK + (KS_) * kTileK8;                                                    \
        ka00 = *reinterpret_cast<const i32x4_t*>(kb_ + r * 64 + 16 * (chunk0(g) ^ swz8(r)));                             \
        ka01 = *reinterpret_cast<const i32x4_t*>(kb_ + r * 64 + 16 * (chunk1(g) ^ swz8(r)));                             \
        ka10 = *reinterpret_cast<const i32x4_t*>(kb_ + (32 + r) * 64 + 16 * (chunk0(g) ^ swz8(r)));                      \
        ka11 = *reinterpret_cast<const i32x4_t*>(kb_ + (32 + r) * 64 + 16 * (chunk1(g) ^ swz8(r)));                      \
        ks0r = *reinterpret_cast<const uint8_t*>(smem + kLdsKS + (KS_) * 128 + r * 2 + g);                   \
        ks1r = *reinterpret_cast<const uint8_t*>(smem + kLdsKS + (KS_) * 128 + (32 + r) * 2 + g);            \
    }
#define F8_QK(C0_, C1_, S0_, S1_)                                                                            \
    {                                                                                                        \
        const i32x8_t k0_ = __builtin_shufflevector(ka00, ka01, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        const i32x8_t k1_ = __builtin_shufflevector(ka10, ka11, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        S0_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k0_, qf, C0_, 0, 0, 0, ks0r, 0, q_scale);      \
        S1_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k1_, qf, C1_, 0, 0, 0, ks1r, 0, q_scale);      \
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
    F8_KREAD(0)
    F8_QK(zero16, zero16, s0, s1)
    F8_MASK(0, s0, s1)
    float m_run;
    {
        // fmaxf here, not the asm v_max3 of F8_MAX8: the compiler places the MFMA -> VALU wait states for what it can see;
        // an asm reading an accumulator right behind its MFMA gets none (and reads the register before the write-back)
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, fmaxf(s0[j], s1[j]));
        float mxx;
        F8_SWAPMAX(mx, mxx)
        m_run = px_first_m<T, PX>(mxx);
        const float off0 = px_first_off<PX>(m_run);
#pragma unroll
        for (int j = 0; j < 16; ++j) { s0[j] += off0; s1[j] += off0; }
    }
    float ex_next = 0.f;                      // max over the tile of (s - m_run): what may exceed 8
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on

#define F8_BARRIER() __builtin_amdgcn_s_barrier()
    for (int t = 0; t < nt; ++t) {
        F8STAMP(ts0)
        // ================= softmax phase =================
        // operands of the coming matrix phase first (they land under the exp2 work), then this group's half of tile t + 2
        i32x4_t vf0[kDT], vf1[kDT];
        int vsr[kDT];
#define F8_PREFETCH_K() { F8_KREAD((t + 1) & (kRing8 - 1)) }
#define F8_PREFETCH_V()                                                                                      \
        {                                                                                                    \
            const char* vb = smem + kLdsV + (t & (kRing8 - 1)) * kTileK8;                                    \
            const char* vsb = smem + kLdsVS + (t & (kRing8 - 1)) * 128;                                      \
            _Pragma("unroll") for (int dt = 0; dt < kDT; ++dt) {                                             \
                vf0[dt] = *reinterpret_cast<const i32x4_t*>(vb + (32 * dt + r) * 64 + 16 * (chunk0(g) ^ swz8(r)));       \
                vf1[dt] = *reinterpret_cast<const i32x4_t*>(vb + (32 * dt + r) * 64 + 16 * (chunk1(g) ^ swz8(r)));       \
                vsr[dt] = *reinterpret_cast<const uint8_t*>(vsb + (32 * dt + r) * 2 + g);                    \
            }                                                                                                \
        }
#if F8_READS_IN_MATRIX == 0
        F8_PREFETCH_K()
        F8_PREFETCH_V()
#elif F8_READS_IN_MATRIX == 2
        F8_PREFETCH_K()
#endif
        __builtin_amdgcn_sched_barrier(0);
        F8STAMP(tsa)
        {
            // deferred rescale: s already carries -m_run; move m only when P8 would pass 2^(kPShift + kThr8) = 256
            if (__any(ex_next > px_thr<PX>())) {
                const float mn = px_next_m<T, PX>(m_run, ex_next);
                const float dm = mn - m_run;
                m_run = mn;
                const float alpha = __builtin_amdgcn_exp2f(-dm * (1.0f / PxUnit<PX>::kS));
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s0[j] -= dm; s1[j] -= dm;
                    o[0][j] *= alpha; o[1][j] *= alpha; lacc[j] *= alpha;
                }
            }
        }
        i32x8_t pf;
#define F8_EXP(X_) __builtin_amdgcn_exp2f(X_)
#define F8_P4(A_, B_, C_, D_) p_bytes4<PX>(A_, B_, C_, D_)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[i] = (int)F8_P4(s0[4 * i], s0[4 * i + 1], s0[4 * i + 2], s0[4 * i + 3]);
            pf[4 + i] = (int)F8_P4(s1[4 * i], s1[4 * i + 1], s1[4 * i + 2], s1[4 * i + 3]);
        }
#undef F8_EXP
#undef F8_P4
        F8_DMA(t + 4)
        {   // the softmax belongs to THIS phase: pin its results here (pure arithmetic otherwise sinks past the barrier)
            asm volatile("" : "+v"(pf));
        }
        __builtin_amdgcn_sched_barrier(0);
        F8STAMP(ts1)
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");     // tile t + 3 (issued a tile ago) has landed
        F8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        F8STAMP(ts2)
        // ================= matrix phase =================
        // S(t+1) is computed for every t (past the last tile the K ring holds an older tile and the result is dropped).
        // The row maximum of S(t+1) runs in the shadow of the P.V MFMAs and only BEHIND them in program order
        // (max16_behind): v_max3 through asm gets no MFMA -> VALU wait states from the compiler, so each half is read
        // after a LATER MFMA has issued (the pipe is in order: by then the half has been written back).
        __builtin_amdgcn_s_setprio(1);
#if F8_READS_IN_MATRIX == 1
        F8_PREFETCH_K()
        F8_PREFETCH_V()
#elif F8_READS_IN_MATRIX == 2
        F8_PREFETCH_V()
#endif
        {
            uint4 mn_u = make_uint4(g == 0 ? px_mneg_word<T, PX>(m_run) : 0u, 0u, 0u, 0u);
            const vec8 onesv = __builtin_bit_cast(vec8, ones_u), mnegv = __builtin_bit_cast(vec8, mn_u);
            const f32x16_t c0 = T::mfma32(onesv, mnegv, zero16);
            F8_QK(c0, c0, s0, s1)
        }
#define F8_PV(DT_)                                                                                           \
        {                                                                                                    \
            const i32x8_t vv_ = __builtin_shufflevector(vf0[DT_], vf1[DT_], 0, 1, 2, 3, 4, 5, 6, 7);         \
            o[DT_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vv_, pf, o[DT_], 0, 0, 0, vsr[DT_], 0, kPs); \
        }
        F8_PV(0)
        F8_MASK(t + 1, s0, s1)
        float mxa = max16_behind(s0, o[0][0]);          // behind P.V (d-tile 0): S0 was written back two MFMAs ago
        F8_PV(1)
        const float mxb = max16_behind(s1, o[1][0]);    // behind P.V (d-tile 1)
        lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones8, pf, lacc, 0, 0, 0, kOne, 0, kPs);
        mxa = vmax2(mxa, mxb);
        F8_SWAPMAX(mxa, ex_next)
#undef F8_PV
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        F8STAMP(ts3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        F8_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef F8_DMA
#undef F8_PREFETCH_K
#undef F8_PREFETCH_V
#undef F8_QK
#undef F8_KREAD
#undef F8_BARRIER
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
        w[kDT * 16 * (kWaves * 64) + tid] = m_run * (1.0f / PxUnit<PX>::kS);     // octaves, whatever the S unit
        w[kDT * 16 * (kWaves * 64) + kWaves * 64 + tid] = l_run;
        continue;
    }
    const float inv = 1.0f / l_run;
    {
        // whole-row stores through the (now idle) rings, 4 KiB per wave: attn_rows_through_lds
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile have landed ...
        __builtin_amdgcn_s_barrier();                      // ... for every wave of the workgroup
        int le = lane;
        asm volatile("" : "+v"(le));       // opaque: or the epilogue's per-lane offsets are computed before the loop and spilled
        u32x4_t rows[kD8 / 16];
        attn_rows_through_lds<T, kD8>(o, inv, (uint32_t)((tid >> 6) * 4096), le & 31, le >> 5, le, rows);
        attn_store_rows<kD8>(rows, op, p.o_rs, qrow - (lane & 31), p.lq, le);
    }
  }   // piece
}


// ---------------------------------------------------------------------------------------------------------------------------
// FREE-RUNNING variant: 4 waves x 32 query rows per workgroup, THREE workgroups per CU (<= 168 registers), one barrier per
// key tile.  A wave of this loop is bound by its own issue rate (tools/ubench/valu_rate.hip: one vector instruction per
// ~4.9 cycles, v_exp_f32 and v_cvt_pk_fp8_f32 ~8.9, whatever the other waves of the SIMD do) and by the latencies in its
// dependency chain (S -> max -> exp2 -> pack -> P.V), not by a pipe: three waves per SIMD that drift apart freely fill each
// other's gaps, where the ping-pong kernel's two waves wait for each other at two barriers per tile.
// Staging: rings of four tiles by LDS-DMA, three operations per wave and tile (its quarter of K8, of V8T, one of the two
// scale blocks), issued three tiles ahead, awaited one tile later.
constexpr int kFrWaves = 4;
constexpr int kFrQBlock = kFrWaves * kQRowsPerWave;     // 128
constexpr int kFrRing = 4;
constexpr int kFrLdsK = 0, kFrLdsV = kFrRing * kTileK8, kFrLdsKS = 2 * kFrRing * kTileK8, kFrLdsVS = kFrLdsKS + kFrRing * 128;
constexpr int kFrSmem = kFrLdsVS + kFrRing * 128;       // 33 KiB

template <typename T, int PX>
__global__ __launch_bounds__(kFrWaves * 64, 3) void attn_fp8_fr_kernel(const Fp8AttnParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnParams& p = fp.a;
    constexpr int kDT = kD8 / 32;
    typedef typename T::vec8 vec8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int g = lane >> 5;
    const int xcd = blockIdx.x & 7;
    const int bx = blockIdx.x >> 3;
    int hb, qb;
    if (!attn_map_block(p, xcd, bx, hb, qb)) return;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    const int64_t tile0 = (int64_t)hb * fp.nt;
    const int nt = fp.nt;
    const int lk = p.lk;

    const int qrow = qb * kFrQBlock + wave * kQRowsPerWave + r;
    i32x8_t qf;
    int q_scale;
    load_q_fp8<T>(p, qp, qrow, g, qf, q_scale);

    // ---- staging: wave w moves LDS positions 64 w .. 64 w + 63 (16-byte slots) of the K8 and of the V8T image, and (lanes
    //      0..31) the K scales (waves 0, 2) or the V scales (waves 1, 3): three vector-memory operations per wave and tile ----
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(fp.k8 + tile0 * kTileK8), 0, nt * kTileK8, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(fp.v8t + tile0 * kTileK8), 0, nt * kTileK8, 0x00020000);
    const bool sc_k = (wave & 1) == 0;
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((sc_k ? fp.ks : fp.vs) + tile0 * 128), 0, nt * 128, 0x00020000);
    const int st_pos = wave * 64 + lane;
    const uint32_t st_voff = (uint32_t)((st_pos >> 2) * 64 + (((st_pos & 3) ^ swz8(st_pos >> 2)) << 4));
    const int st_slds = sc_k ? kFrLdsKS : kFrLdsVS;
#define FR_DMA(U_)                                                                                           \
    {                                                                                                        \
        const int tt_ = (U_) < nt ? (U_) : nt - 1;                                                           \
        const int sl_ = (U_) & (kFrRing - 1);                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (FINO_LDS void*)(smem + kFrLdsK + sl_ * kTileK8 + wave * 1024), 16, \
                                                 st_voff, tt_ * kTileK8, 0, 0);                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (FINO_LDS void*)(smem + kFrLdsV + sl_ * kTileK8 + wave * 1024), 16, \
                                                 st_voff, tt_ * kTileK8, 0, 0);                              \
        if (lane < 32) {   /* dword (lane) of the scale block; the offset is rebuilt here, not kept in a register */ \
            uint32_t sv_;                                                                                    \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0" : "=v"(sv_));                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (FINO_LDS void*)(smem + st_slds + sl_ * 128), 4, sv_ << 2, \
                                                     tt_ * 128, 0, 0);                                       \
        }                                                                                                    \
    }
    { FR_DMA(0) }
    { FR_DMA(1) }
    { FR_DMA(2) }
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");          // tiles 0 and 1
    __syncthreads();

    f32x16_t o[kDT], lacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[0][j] = 0.f; o[1][j] = 0.f; lacc[j] = 0.f; }
    i32x8_t ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    asm volatile("" : "+v"(ones8));       // opaque: held in 8 registers for the whole loop instead of 7 moves per tile
    constexpr int kOne = 127, kPs = 127 - kPShift;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 ones_u = make_uint4(g == 0 ? px_ones_word<T, PX>() : 0u, 0u, 0u, 0u);
    asm volatile("" : "+v"(ones_u.x));

    // per-lane byte offsets of the two 16-byte chunks of row r inside a tile image (rows 32 + r: + 2048)
    const int la0 = r * 64 + 16 * (chunk0(g) ^ swz8(r));
    const int la1 = r * 64 + 16 * (chunk1(g) ^ swz8(r));
    const int ls0 = r * 2 + g;                               // scale byte of row r (rows 32 + r: + 64)
    i32x4_t ka00, ka01, ka10, ka11;
    int ks0r, ks1r;
#define FR_KREAD(SL_)                                                                                        \
    {                                                                                                        \
        const char* kb_ = smem + kFrLdsK + (SL_) * kTileK8;                                                  \
        ka00 = *reinterpret_cast<const i32x4_t*>(kb_ + la0);                                                 \
        ka01 = *reinterpret_cast<const i32x4_t*>(kb_ + la1);                                                 \
        ka10 = *reinterpret_cast<const i32x4_t*>(kb_ + 2048 + la0);                                          \
        ka11 = *reinterpret_cast<const i32x4_t*>(kb_ + 2048 + la1);                                          \
        ks0r = *reinterpret_cast<const uint8_t*>(smem + kFrLdsKS + (SL_) * 128 + ls0);                       \
        ks1r = *reinterpret_cast<const uint8_t*>(smem + kFrLdsKS + (SL_) * 128 + 64 + ls0);                  \
    }
#define FR_QK(C0_, C1_, S0_, S1_)                                                                            \
    {                                                                                                        \
        const i32x8_t k0_ = __builtin_shufflevector(ka00, ka01, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        const i32x8_t k1_ = __builtin_shufflevector(ka10, ka11, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        S0_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k0_, qf, C0_, 0, 0, 0, ks0r, 0, q_scale);      \
        S1_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k1_, qf, C1_, 0, 0, 0, ks1r, 0, q_scale);      \
    }
#define FR_MASK(T_, S0_, S1_)                                                                                \
    if (__builtin_expect((T_) == nt - 1 && (lk & (kKV - 1)), 0)) {                                           \
        int rem_ = lk - (T_) * kKV - 4 * g;                                                                  \
        asm volatile("" : "+v"(rem_));                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                  \
            const int key_ = (j_ & 3) + 8 * (j_ >> 2);                                                       \
            if (key_ >= rem_) S0_[j_] = -INFINITY;                                                           \
            if (key_ + 32 >= rem_) S1_[j_] = -INFINITY;                                                      \
        }                                                                                                    \
    }
#define FR_SWAPMAX(MX_, OUT_)                                                                                \
    {                                                                                                        \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                      \
    }

    // ---- S(0) unpipelined ----
    f32x16_t s0, s1;
    FR_KREAD(0)
    FR_QK(zero16, zero16, s0, s1)
    FR_MASK(0, s0, s1)
    float m_run;
    {
        float mx = fmaxf(s0[0], s1[0]);                       // fmaxf: see the ping-pong kernel (MFMA -> VALU wait states)
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, fmaxf(s0[j], s1[j]));
        float mxx;
        FR_SWAPMAX(mx, mxx)
        m_run = px_first_m<T, PX>(mxx);
        const float off0 = px_first_off<PX>(m_run);
#pragma unroll
        for (int j = 0; j < 16; ++j) { s0[j] += off0; s1[j] += off0; }
    }
    float ex_next = 0.f;

    // One key tile: P(t) from SI (= S(t) - m), S(t+1) into SO, O += V(t) P(t).  The loop below runs it twice per trip with
    // the two S register sets swapped: with one set the loop-carried S costs 16 v_mov_b64 per tile (the MFMAs cannot
    // write S(t+1) over the registers exp2 is still reading, and the back edge copies it home).
#define FR_PV(DT_)                                                                                           \
        {                                                                                                    \
            const i32x8_t vv_ = __builtin_shufflevector(vf0[DT_], vf1[DT_], 0, 1, 2, 3, 4, 5, 6, 7);         \
            o[DT_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vv_, pf, o[DT_], 0, 0, 0, vsr[DT_], 0, kPs); \
        }
#define FRX_DMA(U_) FR_DMA(U_)
#define FRX_KREAD(SL_) FR_KREAD(SL_)
#define FRX_P4(A_, B_, C_, D_) p_bytes4<PX>(A_, B_, C_, D_)
#define FRX_MAX(S_, B_) max16_behind(S_, B_)
#define FRX_LACC() lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones8, pf, lacc, 0, 0, 0, kOne, 0, kPs);
#define FRX_FOLD(A_, B_) T::mfma32(A_, B_, zero16)
#define FRX_BARRIER() __builtin_amdgcn_s_barrier()
#define FR_TILE(T_, SI0_, SI1_, SO0_, SO1_)                                                                  \
    {                                                                                                        \
        const int t = (T_);                                                                                  \
        /* tile t + 3 into the slot tile t - 1 left (its last reads ended before the previous barrier) */     \
        FRX_DMA(t + 3)                                                                                       \
        /* K fragments of S(t+1): they land under the exp2 work */                                           \
        FRX_KREAD((t + 1) & (kFrRing - 1))                                                                   \
        if (__any(ex_next > px_thr<PX>())) {                  /* deferred rescale (see the ping-pong kernel) */ \
            const float mn = px_next_m<T, PX>(m_run, ex_next);                                               \
            const float dm = mn - m_run;                                                                     \
            m_run = mn;                                                                                      \
            const float alpha = __builtin_amdgcn_exp2f(-dm * (1.0f / PxUnit<PX>::kS));                       \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                 \
                /* s -= dm IN PLACE ("+v"): no fresh registers on this path, no copies at the join */        \
                asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %2" : "+v"(SI0_[j]), "+v"(SI1_[j]) : "v"(dm)); \
                o[0][j] *= alpha; o[1][j] *= alpha; lacc[j] *= alpha;                                        \
            }                                                                                                \
        }                                                                                                    \
        i32x8_t pf;                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
            pf[i] = (int)FRX_P4(SI0_[4 * i], SI0_[4 * i + 1], SI0_[4 * i + 2], SI0_[4 * i + 3]);                 \
            pf[4 + i] = (int)FRX_P4(SI1_[4 * i], SI1_[4 * i + 1], SI1_[4 * i + 2], SI1_[4 * i + 3]);             \
        }                                                                                                    \
        {                                                                                                    \
            uint4 mn_u = make_uint4(g == 0 ? px_mneg_word<T, PX>(m_run) : 0u, 0u, 0u, 0u);                \
            const vec8 onesv = __builtin_bit_cast(vec8, ones_u), mnegv = __builtin_bit_cast(vec8, mn_u);     \
            const f32x16_t c0 = FRX_FOLD(onesv, mnegv);                                                      \
            FR_QK(c0, c0, SO0_, SO1_)                                                                        \
        }                                                                                                    \
        /* V fragments only now: they take the registers the K fragments leave (168 registers per wave is the whole  \
           budget of three waves per SIMD) and land under the two S MFMAs and the other waves' work */       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        i32x4_t vf0[kDT], vf1[kDT];                                                                          \
        int vsr[kDT];                                                                                        \
        {                                                                                                    \
            const char* vb = smem + kFrLdsV + (t & (kFrRing - 1)) * kTileK8;                                 \
            const char* vsb = smem + kFrLdsVS + (t & (kFrRing - 1)) * 128;                                   \
            _Pragma("unroll") for (int dt = 0; dt < kDT; ++dt) {                                             \
                vf0[dt] = *reinterpret_cast<const i32x4_t*>(vb + dt * 2048 + la0);                           \
                vf1[dt] = *reinterpret_cast<const i32x4_t*>(vb + dt * 2048 + la1);                           \
                vsr[dt] = *reinterpret_cast<const uint8_t*>(vsb + dt * 64 + ls0);                            \
            }                                                                                                \
        }                                                                                                    \
        FR_PV(0)                                                                                             \
        FR_MASK(t + 1, SO0_, SO1_)                                                                           \
        float mxa = FRX_MAX(SO0_, o[0][0]);                                                                  \
        FR_PV(1)                                                                                             \
        const float mxb = FRX_MAX(SO1_, o[1][0]);                                                            \
        FRX_LACC()                                                                                           \
        mxa = vmax2(mxa, mxb);                                                                               \
        FR_SWAPMAX(mxa, ex_next)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");     /* tile t + 2 has landed (t + 3 may be in flight) */ \
        FRX_BARRIER();                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    {
        f32x16_t sb0, sb1;
        int tt = 0;
        for (; tt + 1 < nt; tt += 2) {
            FR_TILE(tt, s0, s1, sb0, sb1)
            FR_TILE(tt + 1, sb0, sb1, s0, s1)
        }
        if (tt < nt) FR_TILE(tt, s0, s1, sb0, sb1)
    }
#undef FR_TILE
#undef FRX_DMA
#undef FRX_KREAD
#undef FRX_P4
#undef FRX_MAX
#undef FRX_LACC
#undef FRX_FOLD
#undef FRX_BARRIER
#undef FR_PV
#undef FR_DMA
#undef FR_KREAD
#undef FR_QK
#undef FR_MASK
#undef FR_SWAPMAX

    // (direct 8-byte stores: the staged whole-row form of the other kernels -- attn_rows_through_lds -- pushed this kernel, which
    // lives at the 168-register limit of three workgroups per CU, into scratch spills inside its loop)
    const float inv = 1.0f / lacc[0];
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
}


// ---------------------------------------------------------------------------------------------------------------------------
// head_dim 128 (Wan2.2): the ping-pong kernel over TWO 64-channel sub-heads per head.  The K/V quantisation pre-pass runs
// unchanged with 2 H "heads" of 64 channels (head stride 64), so sub-head s of head h is image 2 h + s: K8 lo | K8 hi and
// V8T lo | V8T hi.  S^T = -m + K_lo . Q_lo^T + K_hi . Q_hi^T (two block-scaled MFMAs per 32-key half chained through the
// accumulator), O^T has four 32-channel d-tiles (two per sub-head), everything else as attn_fp8_kernel: 8 waves, softmax /
// matrix phases one apart, rings of kRing128 tiles filled by LDS-DMA a tile ahead, operand reads at the top of the matrix
// phase.  Whole blocks only (no tail split, no partials).
constexpr int kRing128 = 8;
constexpr int kTile128 = 2 * kTileK8;                        // K8 lo | K8 hi (and V8T lo | V8T hi) of one key tile: 8 KiB
constexpr int kL128K = 0, kL128V = kRing128 * kTile128, kL128KS = 2 * kRing128 * kTile128, kL128VS = kL128KS + kRing128 * 256;
constexpr int kSmem128 = kL128VS + kRing128 * 256;           // 132 KiB

template <typename T, int PX>
__global__ __launch_bounds__(kWaves * 64, 2) void attn_fp8_d128_kernel(const Fp8AttnParams fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnParams& p = fp.a;
    typedef typename T::vec8 vec8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int g = lane >> 5;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    // tail split (as attn_pp_kernel): the first full_x blocks of an XCD run whole; the key tiles of its last rem_x blocks form
    // one stream cut into nwg ranges of `per` tiles, a range touching at most two blocks (two pieces), partials to p.ws
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
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    const int nt = t_end - t_begin;
    const int lk = (t_end * kKV < p.lk ? t_end * kKV : p.lk) - t_begin * kKV;     // keys of this range, re-based to 0
    const int64_t img0 = (int64_t)hb * 2 * ntall + t_begin;  // first tile of sub-head 0; sub-head 1: + ntall

    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    i32x8_t qf[2];
    int q_scale[2];
    load_q_fp8<T>(p, qp, qrow, g, qf[0], q_scale[0]);
    load_q_fp8<T>(p, qp + 64, qrow, g, qf[1], q_scale[1]);

    // ---- staging (LDS-DMA): waves 0 / 1 of a group move the group's half of BOTH K8 images of a tile, 2 / 3 of both V8T
    //      images; lanes 0..15 also the group's half of their scale blocks: four vector-memory operations per wave and tile ----
    const int wl = __builtin_amdgcn_readfirstlane(wave & 3);
    const bool st_k = wl < 2;
    const uint8_t* timg = st_k ? fp.k8 : fp.v8t;
    const uint8_t* simg = st_k ? fp.ks : fp.vs;
    const __amdgpu_buffer_rsrc_t st_r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(timg + img0 * kTileK8), 0, nt * kTileK8, 0x00020000);
    const __amdgpu_buffer_rsrc_t st_r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(timg + (img0 + ntall) * kTileK8), 0, nt * kTileK8, 0x00020000);
    const __amdgpu_buffer_rsrc_t st_s0 = __builtin_amdgcn_make_buffer_rsrc((void*)(simg + img0 * 128), 0, nt * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t st_s1 = __builtin_amdgcn_make_buffer_rsrc((void*)(simg + (img0 + ntall) * 128), 0, nt * 128, 0x00020000);
    const int st_pos = grp * 128 + (wl & 1) * 64 + lane;
    const uint32_t st_voff = (uint32_t)((st_pos >> 2) * 64 + (((st_pos & 3) ^ swz8(st_pos >> 2)) << 4));
    const uint32_t st_svoff = (uint32_t)((grp * 16 + (lane & 15)) * 4);
    const int st_lds = (st_k ? kL128K : kL128V) + (grp * 128 + (wl & 1) * 64) * 16;
    const int st_slds = (st_k ? kL128KS : kL128VS) + grp * 64;
#define D8_DMA(U_)                                                                                           \
    {                                                                                                        \
        const int tt_ = (U_) < nt ? (U_) : nt - 1;                                                           \
        const int sl_ = (U_) & (kRing128 - 1);                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st_r0, (FINO_LDS void*)(smem + st_lds + sl_ * kTile128), 16, st_voff, \
                                                 tt_ * kTileK8, 0, 0);                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(st_r1, (FINO_LDS void*)(smem + st_lds + sl_ * kTile128 + kTileK8), 16, \
                                                 st_voff, tt_ * kTileK8, 0, 0);                              \
        if (lane < 16) {                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_s0, (FINO_LDS void*)(smem + st_slds + sl_ * 256), 4, st_svoff, \
                                                     tt_ * 128, 0, 0);                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_s1, (FINO_LDS void*)(smem + st_slds + sl_ * 256 + 128), 4, \
                                                     st_svoff, tt_ * 128, 0, 0);                             \
        }                                                                                                    \
    }
    { D8_DMA(0) }
    { D8_DMA(1) }
    { D8_DMA(2) }
    { D8_DMA(3) }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tiles 0 and 1
    __syncthreads();

    f32x16_t o[4], lacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { o[0][j] = 0.f; o[1][j] = 0.f; o[2][j] = 0.f; o[3][j] = 0.f; lacc[j] = 0.f; }
    i32x8_t ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    asm volatile("" : "+v"(ones8));       // opaque: held in registers instead of re-materialised every tile
    constexpr int kOne = 127, kPs = 127 - kPShift;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 ones_u = make_uint4(g == 0 ? px_ones_word<T, PX>() : 0u, 0u, 0u, 0u);
    asm volatile("" : "+v"(ones_u.x));
    const int la0 = r * 64 + 16 * (chunk0(g) ^ swz8(r));
    const int la1 = r * 64 + 16 * (chunk1(g) ^ swz8(r));
    const int ls0 = r * 2 + g;

    // K fragments of both sub-heads for both 32-key halves: 8 x ds_read_b128 + 4 scale bytes
    i32x4_t kA[2][2][2];                                     // [sub][half][chunk]
    int kS[2][2];                                            // [sub][half]
#define D8_KREAD(SL_)                                                                                        \
    _Pragma("unroll") for (int sb_ = 0; sb_ < 2; ++sb_) {                                                    \
        const char* kb_ = smem + kL128K + (SL_) * kTile128 + sb_ * kTileK8;                                  \
        const char* sb__ = smem + kL128KS + (SL_) * 256 + sb_ * 128;                                         \
        kA[sb_][0][0] = *reinterpret_cast<const i32x4_t*>(kb_ + la0);                                        \
        kA[sb_][0][1] = *reinterpret_cast<const i32x4_t*>(kb_ + la1);                                        \
        kA[sb_][1][0] = *reinterpret_cast<const i32x4_t*>(kb_ + 2048 + la0);                                 \
        kA[sb_][1][1] = *reinterpret_cast<const i32x4_t*>(kb_ + 2048 + la1);                                 \
        kS[sb_][0] = *reinterpret_cast<const uint8_t*>(sb__ + ls0);                                          \
        kS[sb_][1] = *reinterpret_cast<const uint8_t*>(sb__ + 64 + ls0);                                     \
    }
#define D8_QK(C_, S0_, S1_)                                                                                  \
    {                                                                                                        \
        S0_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                               \
            __builtin_shufflevector(kA[0][0][0], kA[0][0][1], 0, 1, 2, 3, 4, 5, 6, 7), qf[0], C_, 0, 0, 0, kS[0][0], 0, q_scale[0]); \
        S1_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                               \
            __builtin_shufflevector(kA[0][1][0], kA[0][1][1], 0, 1, 2, 3, 4, 5, 6, 7), qf[0], C_, 0, 0, 0, kS[0][1], 0, q_scale[0]); \
        S0_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                               \
            __builtin_shufflevector(kA[1][0][0], kA[1][0][1], 0, 1, 2, 3, 4, 5, 6, 7), qf[1], S0_, 0, 0, 0, kS[1][0], 0, q_scale[1]); \
        S1_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                               \
            __builtin_shufflevector(kA[1][1][0], kA[1][1][1], 0, 1, 2, 3, 4, 5, 6, 7), qf[1], S1_, 0, 0, 0, kS[1][1], 0, q_scale[1]); \
    }
#define D8_MASK(T_, S0_, S1_)                                                                                \
    if (__builtin_expect((T_) == nt - 1 && (lk & (kKV - 1)), 0)) {                                           \
        int rem_ = lk - (T_) * kKV - 4 * g;                                                                  \
        asm volatile("" : "+v"(rem_));                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                  \
            const int key_ = (j_ & 3) + 8 * (j_ >> 2);                                                       \
            if (key_ >= rem_) S0_[j_] = -INFINITY;                                                           \
            if (key_ + 32 >= rem_) S1_[j_] = -INFINITY;                                                      \
        }                                                                                                    \
    }
#define D8_SWAPMAX(MX_, OUT_)                                                                                \
    {                                                                                                        \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                      \
    }

    f32x16_t s0, s1;
    D8_KREAD(0)
    D8_QK(zero16, s0, s1)
    D8_MASK(0, s0, s1)
    float m_run;
    {
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, fmaxf(s0[j], s1[j]));
        float mxx;
        D8_SWAPMAX(mx, mxx)
        m_run = px_first_m<T, PX>(mxx);
        const float off0 = px_first_off<PX>(m_run);
#pragma unroll
        for (int j = 0; j < 16; ++j) { s0[j] += off0; s1[j] += off0; }
    }
    float ex_next = 0.f;
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on

    for (int t = 0; t < nt; ++t) {
        // ================= softmax phase =================
        if (__any(ex_next > px_thr<PX>())) {
            const float mn = px_next_m<T, PX>(m_run, ex_next);
            const float dm = mn - m_run;
            m_run = mn;
            const float alpha = __builtin_amdgcn_exp2f(-dm * (1.0f / PxUnit<PX>::kS));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                s0[j] -= dm; s1[j] -= dm;
                o[0][j] *= alpha; o[1][j] *= alpha; o[2][j] *= alpha; o[3][j] *= alpha; lacc[j] *= alpha;
            }
        }
        i32x8_t pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pf[i] = (int)p_bytes4<PX>(s0[4 * i], s0[4 * i + 1], s0[4 * i + 2], s0[4 * i + 3]);
            pf[4 + i] = (int)p_bytes4<PX>(s1[4 * i], s1[4 * i + 1], s1[4 * i + 2], s1[4 * i + 3]);
        }
        D8_DMA(t + 4)
        asm volatile("" : "+v"(pf));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");     // tile t + 3 (issued a tile ago) has landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= matrix phase =================
        __builtin_amdgcn_s_setprio(1);
        D8_KREAD((t + 1) & (kRing128 - 1))
        i32x4_t vA[4][2];                                      // [d-tile = 2 sub + dt][chunk]
        int vS[4];
        {
            const char* vb = smem + kL128V + (t & (kRing128 - 1)) * kTile128;
            const char* vsb = smem + kL128VS + (t & (kRing128 - 1)) * 256;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                vA[d4][0] = *reinterpret_cast<const i32x4_t*>(vb + (d4 >> 1) * kTileK8 + (d4 & 1) * 2048 + la0);
                vA[d4][1] = *reinterpret_cast<const i32x4_t*>(vb + (d4 >> 1) * kTileK8 + (d4 & 1) * 2048 + la1);
                vS[d4] = *reinterpret_cast<const uint8_t*>(vsb + (d4 >> 1) * 128 + (d4 & 1) * 64 + ls0);
            }
        }
        {
            uint4 mn_u = make_uint4(g == 0 ? px_mneg_word<T, PX>(m_run) : 0u, 0u, 0u, 0u);
            const vec8 onesv = __builtin_bit_cast(vec8, ones_u), mnegv = __builtin_bit_cast(vec8, mn_u);
            const f32x16_t c0 = T::mfma32(onesv, mnegv, zero16);
            D8_QK(c0, s0, s1)
        }
#define D8_PV(D4_)                                                                                           \
        o[D4_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                                            \
            __builtin_shufflevector(vA[D4_][0], vA[D4_][1], 0, 1, 2, 3, 4, 5, 6, 7), pf, o[D4_], 0, 0, 0, vS[D4_], 0, kPs);
        D8_PV(0)
        D8_PV(1)
        D8_MASK(t + 1, s0, s1)
        float mxa = max16_behind(s0, o[1][0]);          // behind two P.V MFMAs: S0 was written back long ago
        D8_PV(2)
        D8_PV(3)
        const float mxb = max16_behind(s1, o[3][0]);
        lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones8, pf, lacc, 0, 0, 0, kOne, 0, kPs);
        mxa = vmax2(mxa, mxb);
        D8_SWAPMAX(mxa, ex_next)
#undef D8_PV
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef D8_DMA
#undef D8_KREAD
#undef D8_QK
#undef D8_MASK
#undef D8_SWAPMAX

    if (part >= 0) {
        // partial in the bf16 kernels' layout (attn_combine_kernel finishes): m, O and l share the 2^kPShift offset, which
        // cancels in the merge
        float* w = p.ws + (int64_t)part * partial_floats<128>();
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4)
#pragma unroll
            for (int j = 0; j < 16; ++j) w[(d4 * 16 + j) * (kWaves * 64) + tid] = o[d4][j];
        w[4 * 16 * (kWaves * 64) + tid] = m_run * (1.0f / PxUnit<PX>::kS);       // octaves, whatever the S unit
        w[4 * 16 * (kWaves * 64) + kWaves * 64 + tid] = lacc[0];
        continue;
    }
    const float inv = 1.0f / lacc[0];
    {
        // whole-row stores through the first 64 KiB of the (now idle) rings, 8 KiB per wave: attn_rows_through_lds
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile have landed ...
        __builtin_amdgcn_s_barrier();                      // ... for every wave
        int le = lane;
        asm volatile("" : "+v"(le));       // opaque: or the epilogue's per-lane offsets are computed before the loop and spilled
        u32x4_t rows[8];
        attn_rows_through_lds<T, 128>(o, inv, (uint32_t)((tid >> 6) * 8192), le & 31, le >> 5, le, rows);
        attn_store_rows<128>(rows, op, p.o_rs, qrow - r, p.lq, le);
    }
  }   // piece
}

}  // namespace

extern "C" int64_t fino_attn_fp8_kv_bytes(int batch, int heads, int64_t lk, int head_dim) {
    if (batch <= 0 || heads <= 0 || lk <= 0 || (head_dim != 64 && head_dim != 128)) return 0;
    const int64_t nt = (lk + kKV - 1) / kKV;
    int64_t b = (int64_t)batch * heads * (head_dim / 64) * nt * (2 * kTileK8 + 256);  // per 64-channel sub-head
    // head_dim 128: + the fp32 (O, m, l) partials of the tail split (at most 32 ranges per XCD, two pieces each), 16-aligned
    if (head_dim == 128) b = ((b + 15) & ~(int64_t)15) + (int64_t)8 * 32 * 2 * partial_floats<128>() * 4;
    return b;
}

extern "C" int fino_attn_fwd_fp8(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                                 int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs,
                                 int64_t v_bs, int64_t v_rs, int64_t o_bs, int64_t o_rs, float scale, int dtype, int p_mode,
                                 void* kv_workspace, int64_t kv_workspace_bytes, void* stream) {
    FINO_CHECK(p_mode == FINO_FP8_P_EXP2 || p_mode == FINO_FP8_P_RAMP, FINO_ERR_ARG, "fino_attn_fwd_fp8: p_mode %d", p_mode);
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_fwd_fp8: dtype %d", dtype);
    FINO_CHECK(head_dim == 64 || head_dim == 128, FINO_ERR_UNSUPPORTED, "fino_attn_fwd_fp8: head_dim %d not in {64, 128}", head_dim);
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
    const int sub = head_dim / 64;                            // 64-channel sub-heads per head: images of the pre-pass
    const int64_t bh = (int64_t)batch * heads * sub;
    uint8_t* w8 = (uint8_t*)kv_workspace;
    QuantParams qp;
    qp.k = (const uint16_t*)k; qp.v = (const uint16_t*)v;
    qp.k8 = w8; qp.v8t = w8 + bh * nt * kTileK8; qp.ks = w8 + 2 * bh * nt * kTileK8; qp.vs = qp.ks + bh * nt * 128;
    qp.batch = batch; qp.heads = heads * sub; qp.lk = (int)lk; qp.nt = nt;
    qp.k_bs = k_bs; qp.k_rs = k_rs; qp.k_hs = 64; qp.v_bs = v_bs; qp.v_rs = v_rs; qp.v_hs = 64;
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
    const bool ramp = p_mode == FINO_FP8_P_RAMP;
    if (ramp) p.scale_log2 *= 8.0f;          // logits in eighths of an octave: an exact shift of q's e8m0 block scales
    const bool free_running = head_dim == 64 && fino_tune_get(FINO_TUNE_ATTN_FP8_KERNEL) != 1;   // default at head_dim 64
    const int qblock = free_running ? kFrQBlock : kQBlock;
    p.nqb = (int)((lq + qblock - 1) / qblock);
    p.ws = nullptr; p.all_partial = 0; p.tail_n = 0;
    attn_virtual_heads(p.batch, p.heads, p.nqb, p.vsplit, p.nqb_v);
    const int groups = (p.batch * p.heads * p.vsplit + 7) / 8;
    p.full_x = groups * p.nqb_v; p.rem_x = 0; p.nwg = 0; p.per = 1;
    fp.k8 = qp.k8; fp.ks = qp.ks; fp.v8t = qp.v8t; fp.vs = qp.vs; fp.nt = nt;
    const dim3 grid((unsigned)(8 * p.full_x));
    if (free_running) {
#define F8_LAUNCH(KERNEL_, GRID_, THREADS_, SMEM_)                                                                        \
    {                                                                                                                     \
        if (dtype == FINO_BF16) { if (ramp) KERNEL_(BF16, 1)<<<GRID_, THREADS_, SMEM_, st>>>(fp); else KERNEL_(BF16, 0)<<<GRID_, THREADS_, SMEM_, st>>>(fp); } \
        else { if (ramp) KERNEL_(F16, 1)<<<GRID_, THREADS_, SMEM_, st>>>(fp); else KERNEL_(F16, 0)<<<GRID_, THREADS_, SMEM_, st>>>(fp); } \
    }
#define F8_SMEM_ONCE(KERNEL_, SMEM_)                                                                                      \
    {                                                                                                                     \
        static FinoPerDeviceOnce once_[4];                                                                                \
        const void* fn_ = dtype == FINO_BF16 ? (ramp ? (const void*)KERNEL_(BF16, 1) : (const void*)KERNEL_(BF16, 0))     \
                                             : (ramp ? (const void*)KERNEL_(F16, 1) : (const void*)KERNEL_(F16, 0));      \
        const int rc_ = fino_max_smem_once(once_[(dtype == FINO_BF16 ? 0 : 2) + (ramp ? 1 : 0)], fn_, SMEM_, "fino_attn_fwd_fp8"); \
        if (rc_ != FINO_OK) return rc_;                                                                                   \
    }
#define K_FR(T_, PX_) attn_fp8_fr_kernel<T_, PX_>
#define K_D128(T_, PX_) attn_fp8_d128_kernel<T_, PX_>
#define K_PP(T_, PX_) attn_fp8_kernel<T_, 0, PX_>
        F8_LAUNCH(K_FR, grid, kFrWaves * 64, kFrSmem)
        FINO_LAUNCH_CHECK();
        return FINO_OK;
    }
    if (head_dim == 128) {
        // tail split: the last, partial round of (head, q-block) workgroups is cut over the keys (fino_attention.hip's plan)
        int full_x, rem_x, nwg, per;
        fino_attn_plan_split(p.batch, p.heads, p.nqb, nt, full_x, rem_x, nwg, per);
        if (rem_x > 0 && nwg <= 32) {
            const int64_t kvb = ((int64_t)bh * nt * (2 * kTileK8 + 256) + 15) & ~(int64_t)15;
            p.ws = (float*)(w8 + kvb);
            p.full_x = full_x; p.rem_x = rem_x; p.nwg = nwg; p.per = per;
        }
        const dim3 grid128((unsigned)(8 * (p.full_x + p.nwg)));
        F8_SMEM_ONCE(K_D128, kSmem128)
        F8_LAUNCH(K_D128, grid128, kWaves * 64, kSmem128)
        FINO_LAUNCH_CHECK();
        return fino_attn_launch_combine(p, dtype, 128, st);
    }
    constexpr int smem = kSmem8;
    F8_SMEM_ONCE(K_PP, smem)        // 66 KiB of dynamic LDS: above the 64 KiB a kernel gets without asking
    F8_LAUNCH(K_PP, grid, kWaves * 64, smem)
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
