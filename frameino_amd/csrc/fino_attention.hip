// Non-causal flash attention forward for the DiT (3D spatio-temporal self-attention, text cross-attention).
// Replaces F.scaled_dot_product_attention at architecture/transformer_wan.py:108 and
// architecture/attention_processor.py:2863/:2934 of the reference.
//
// CDNA4 design (8-wave structure of the gfx950 playbook, written for this path):
//   * workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 query rows.
//   * S^T = K.Q^T ("swapped" product, v_mfma_f32_32x32x16): the query index sits on the lane, so the online
//     softmax is lane-local (row max/sum = 32 registers + one permlane32_swap with the partner half-wave).
//   * O^T = V^T.P^T: the S^T accumulator registers, packed to bf16, ARE the B operand of the second product
//     (k-order permuted: element j of half h is key 16s + 8(j>>2) + 4h + (j&3)); V^T comes from LDS through
//     ds_read_b64_tr_b16 in the same permuted order -- no cross-lane movement of P, no LDS round trip for P.
//   * K/V tiles of 64 keys are register-staged (buffer/global_load_dwordx4, then ds_write_b128) into a double-buffered,
//     XOR-swizzled LDS image that is conflict-free for both the row reads (K) and the transposed reads (V).
//   * Main loops: attn_pp_kernel -- PING-PONG: the two waves of a SIMD alternate a softmax phase and a 32-MFMA matrix
//     phase, one phase apart, two barriers per tile (the earlier one-barrier loop in which every wave interleaved S(t+1)
//     MFMAs with the exp of S(t) left the library in round 5: git history).  Round 4, head_dim 128:
//     attn_ppd_kernel (the default for long key sequences: the ping-pong loop with K/V by LDS-DMA into four-slot rings,
//     LDS reads issued between the MFMAs) and attn_ppw_kernel (short key sequences: one workgroup per CU walks a run of
//     q-blocks without draining).  attn_fr_kernel (round 3): 4 waves, two workgroups per CU -- short key sequences with
//     fewer than two q-blocks per CU (token shards).  All give the same bits.  INTEGRATION.md has the dispatch table.
//   * Tail split: the key tiles of an XCD's last, partial round of blocks are dealt to all its CUs (fino_attn_fwd_ws).
//   * q/k/v are read in place from the fused-QKV GEMM output ([L, 3*H*Dh], strides passed in), o is written
//     token-major [L, H*Dh]: no transposes anywhere.
//   * blockIdx -> (head, q-block) is XCD-aware: all q-blocks of a head share blockIdx%8, i.e. one XCD's L2 holds
//     that head's K/V while its 32 CUs sweep it.
#include <stdlib.h>

#include "fino_common.h"

#ifdef FINO_ATTN_STAMP
__device__ unsigned long long fino_attn_dbg[128];   // [64 ..): whole-workgroup stamps of attn_pp_kernel (entry, Q loaded, tiles staged, loop start, loop end, stored)
extern "C" int fino_attn_debug_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fino_attn_dbg), sizeof(unsigned long long) * 128);
}
#define ASTAMP(V_) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V_) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define ASTAMP(V_)
#endif

#include "fino_attention_common.h"
using namespace fino_attn_ns;

namespace {

// VAR only names the launch site (0: long-KV self-attention, 1: short-KV text cross-attention) so that profilers
// report the two call classes as separate kernel symbols; the code is identical.
template <typename T, int D, int VAR>
__global__ __launch_bounds__(kWaves * 64, 2) void attn_pp_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kTileBytes = kKV * D * 2;
    constexpr int kChunksPerRow = D / 8;
    constexpr int kLoadsPerThread = (kKV * kChunksPerRow) / (kWaves * 64);  // 2 (D=128) or 1 (D=64)
    constexpr int kKS = D / 16;                                             // k-steps of QK^T
    constexpr int kDT = D / 32;                                             // d-tiles of O^T
    typedef typename T::vec8 vec8;

#ifdef FINO_ATTN_STAMP
    unsigned long long wg_ts[6] = {0, 0, 0, 0, 0, 0};
    ASTAMP(wg_ts[0])
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;

    // ---- XCD-aware block -> (head-batch, q-block) ----
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int slot = id >> 3;
    // whole blocks first; then the XCD's last `rem_x` blocks as one stream of rem_x*nt key tiles cut into `nwg` equal
    // ranges (tail split): a range covers pieces of at most two blocks (per < nt).
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
    int hb, qb;
    if (!attn_map_block(p, xcd, bx, hb, qb)) continue;
    if (p.all_partial) part = hb * p.nqb + qb;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;

    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    const uint16_t* kp = p.k + bi * p.k_bs + head * p.k_hs + (int64_t)t_begin * kKV * p.k_rs;
    const uint16_t* vp = p.v + bi * p.v_bs + head * p.v_hs + (int64_t)t_begin * kKV * p.v_rs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    // keys of this workgroup's range, re-based to 0 (a multiple of kKV precedes it, so tail masks are unchanged)
    const int lk = (t_end * kKV < p.lk ? t_end * kKV : p.lk) - t_begin * kKV;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + r][16*ks + 8h .. +7] ----
    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
    vec8 qf[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
        uint4 u = *reinterpret_cast<const uint4*>(qp + (int64_t)qrow_c * p.q_rs + 16 * ks + 8 * h);
        if (qrow >= p.lq) u = make_uint4(0, 0, 0, 0);      // rows past Lq are never stored: zero operands draw the least power
        qf[ks] = __builtin_bit_cast(vec8, u);
    }

#ifdef FINO_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(wg_ts[1])
#endif
    // ---- staging roles: each group (waves 0-3 / 4-7) moves ITS half (32 rows) of a tile ----
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    int st_row[kLoadsPerThread], st_ch[kLoadsPerThread], st_off[kLoadsPerThread], st_off_k[kLoadsPerThread];
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; ++i) {
        const int cid = (tid & 255) + i * 256;
        st_row[i] = grp * 32 + cid / kChunksPerRow;
        st_ch[i] = cid % kChunksPerRow;
        st_off[i] = lds_off<D>(st_row[i], st_ch[i]);
        st_off_k[i] = k_lds_off<D>(st_row[i], st_ch[i]);       // K tiles have their own swizzle at head_dim 64 (k_lds_off)
    }
    u32x4_t kreg[kLoadsPerThread], vreg[kLoadsPerThread];
    // my half of K(W_+1) and V(W_): global -> registers with buffer loads: the per-thread part of the address is a fixed
    // 32-bit offset (row in tile, chunk), the tile advance is the scalar offset, and rows past the last key fail the
    // range check and read as zeros (masked in the ragged last tile) -- no per-tile vector address arithmetic.
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (int)((((int64_t)lk - 1) * p.k_rs + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)vp, 0, (int)((((int64_t)lk - 1) * p.v_rs + D) * 2), 0x00020000);
    uint32_t kvo[kLoadsPerThread], vvo[kLoadsPerThread];
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; ++i) {
        kvo[i] = (uint32_t)((st_row[i] * p.k_rs + st_ch[i] * 8) * 2);
        vvo[i] = (uint32_t)((st_row[i] * p.v_rs + st_ch[i] * 8) * 2);
    }
    const int k_tile_bytes = (int)(kKV * p.k_rs * 2), v_tile_bytes = (int)(kKV * p.v_rs * 2);
#define PP_LOAD(W_)                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < kLoadsPerThread; ++i_) {                                       \
        kreg[i_] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(                       \
                                                   k_rsrc, kvo[i_], ((W_) + 1) * k_tile_bytes, 0));        \
        vreg[i_] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(                       \
                                                   v_rsrc, vvo[i_], (W_) * v_tile_bytes, 0));              \
    }
    // registers -> LDS: K ring slot (W_+1)&1, V ring slot 2 + (W_&1)
#define PP_WRITE(W_)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < kLoadsPerThread; ++i_) {                                       \
        *reinterpret_cast<u32x4_t*>(smem + (((W_) + 1) & 1) * kTileBytes + st_off_k[i_]) = kreg[i_];       \
        *reinterpret_cast<u32x4_t*>(smem + (2 + ((W_) & 1)) * kTileBytes + st_off[i_]) = vreg[i_];         \
    }

    // ---- per-lane LDS read addressing ----
    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    const int g1 = (lane >> 4) & 1;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));

    f32x16_t o[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;
    const float c2 = p.scale_log2;
    const int nt = (lk + kKV - 1) / kKV;

    // ---- prologue: K(0), V(0), K(1) into LDS (both groups, each its half) ----
#pragma unroll
    for (int i = 0; i < kLoadsPerThread; ++i) {
        const u32x4_t k0v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kvo[i], 0, 0));
        const u32x4_t v0v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, vvo[i], 0, 0));
        const u32x4_t k1v =
            __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, kvo[i], k_tile_bytes, 0));
        *reinterpret_cast<u32x4_t*>(smem + 0 * kTileBytes + st_off_k[i]) = k0v;
        *reinterpret_cast<u32x4_t*>(smem + 2 * kTileBytes + st_off[i]) = v0v;
        *reinterpret_cast<u32x4_t*>(smem + 1 * kTileBytes + st_off_k[i]) = k1v;
    }
    if (grp == 1) { PP_LOAD(1) }          // group 1 writes tile 1 in its first softmax phase
    __syncthreads();
#ifdef FINO_ATTN_STAMP
    ASTAMP(wg_ts[2])
#endif

    f32x16_t sc0, sc1;   // S of the tile in flight
#define QK_TILE(KB_)                                                                                        \
    {                                                                                                       \
        const char* kb_ = smem + (KB_) * kTileBytes;                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) { sc0[j_] = 0.f; sc1[j_] = 0.f; }                 \
        _Pragma("unroll") for (int ks_ = 0; ks_ < kKS; ++ks_) {                                             \
            const uint4 a0_ = *reinterpret_cast<const uint4*>(kb_ + k_lds_off<D>(r, 2 * ks_ + h));            \
            const uint4 a1_ = *reinterpret_cast<const uint4*>(kb_ + k_lds_off<D>(32 + r, 2 * ks_ + h));       \
            sc0 = T::mfma32(__builtin_bit_cast(vec8, a0_), qf[ks_], sc0);                                   \
            sc1 = T::mfma32(__builtin_bit_cast(vec8, a1_), qf[ks_], sc1);                                   \
        }                                                                                                   \
    }
    QK_TILE(0)
    // keys past lk (zero K rows in a ragged last tile) stay out of the row max and get p = exp2(-inf) = 0
#define MASK_RAGGED(T_)                                                                                     \
    if ((T_) == nt - 1 && (lk & (kKV - 1))) {            /* key = (j&3) + 8*(j>>2) + 4*h (+32) */           \
        const int kbase_ = (T_) * kKV + 4 * h;                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                 \
            const int key_ = kbase_ + (j_ & 3) + 8 * (j_ >> 2);                                             \
            if (key_ >= lk) sc0[j_] = -INFINITY;                                                            \
            if (key_ + 32 >= lk) sc1[j_] = -INFINITY;                                                       \
        }                                                                                                   \
    }
    // row max of 8 accumulator registers (one of four independent chains).  v_max3_f32 through asm: fmaxf on MFMA
    // results makes the compiler canonicalise every input first (32 extra v_max_f32 per tile).
#define MAX8(S_, O_) vmax2(vmax3(vmax3(S_[O_], S_[O_ + 1], S_[O_ + 2]), vmax3(S_[O_ + 3], S_[O_ + 4], S_[O_ + 5]), \
                                 S_[O_ + 6]), S_[O_ + 7])
#define MAX_FINISH1(MX_, OUT_)                                                                              \
    {                                                                                                       \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                     \
    }
    float mx_next;
    {
        MASK_RAGGED(0)
        const float m0 = vmax2(vmax3(MAX8(sc0, 0), MAX8(sc0, 8), MAX8(sc1, 0)), MAX8(sc1, 8));
        MAX_FINISH1(m0, mx_next)
    }
#ifdef FINO_ATTN_STAMP
    ASTAMP(wg_ts[3])
#endif
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on

    // LDS fragment addresses of the matrix phase: three registers.  The swizzle is an XOR on the chunk bits, so k-step
    // ks / d-tile dt only flip address bits (ks << 5, dt << 6: one v_xor next to the read); the +32 / +16-row operands
    // are instruction offsets; and the ring slot (K: slot (t+1)&1, V: slot 2 + (t&1)) is the kTileBytes bit, toggled
    // once per tile.  Raw LDS addresses: the dynamic segment is the kernel's only LDS, so it starts at 0.
    if ((uint32_t)(uintptr_t)(FINO_LDS char*)smem != 0u) __builtin_trap();
    uint32_t ka0 = 1 * kTileBytes + k_lds_off<D>(r, h);
    uint32_t vl0 = 2 * kTileBytes + lds_off<D>(4 * h + tq, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
    uint32_t vh0 = 2 * kTileBytes + lds_off<D>(4 * h + tq + 8, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
#define LDS_PTR(TYPE_, ADDR_) ((FINO_LDS TYPE_*)(uintptr_t)(uint32_t)(ADDR_))

    // Ping-pong: per key tile every wave alternates a SOFTMAX phase (VALU: max, exp2, sums, bf16 packing; plus its
    // share of the K/V staging) with a MATRIX phase (32 MFMAs: S(t+1) = K(t+1).Q^T and O^T += V(t)^T.P(t)^T, LDS
    // fragment reads in the MFMA gaps).  The two waves of a SIMD are one phase apart, so its matrix pipe and its
    // VALU each serve one wave at a time instead of being arbitrated by age.
#ifdef FINO_ATTN_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0;
    unsigned long long tsa = 0, tsb = 0, tsc = 0, sa5 = 0, sa6 = 0, sa7 = 0;
#endif
    for (int t = 0; t < nt; ++t) {
        ASTAMP(ts0)
        // ================= softmax phase =================
        const int w = t + grp;                      // tile whose V (and K of the next) this group stages now
        if (w >= 1) { PP_WRITE(w) }
        ASTAMP(tsa)
        {
            // deferred rescale (threshold rescale_thr<T>()): O, l and m move together, between tiles.  mx_next = row max of
            // this tile's S, computed in the shadow of the previous matrix phase's P.V MFMAs.
            const float m_cand = fmaxf(m_run, mx_next * c2);
            if (__any((m_cand - m_run) > rescale_thr<T>())) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
                m_run = m_cand;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < kDT; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[i][j] *= alpha;
            }
        }
        ASTAMP(tsb)
        // p = exp2(c.s - m).  Scalar fp32 on purpose: v_pk_fma_f32 / v_pk_add_f32 measured ~2x SLOWER per element here.
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sc0[j] = __builtin_amdgcn_exp2f(sc0[j] * c2 - m_run);
            sc1[j] = __builtin_amdgcn_exp2f(sc1[j] * c2 - m_run);
        }
        ASTAMP(tsc)
        PP_LOAD(w + 1)                               // spaced from the ds_writes above by the exp block
        {
            float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                psum0 += sc0[j];
                psum1 += sc1[j];
            }
            l_run += psum0 + psum1;
        }
        vec8 pb[4];                                  // P(t) packed: pb[2*kt + s2]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pb[0][j] = (typename T::scalar)sc0[j];
            pb[1][j] = (typename T::scalar)sc0[8 + j];
            pb[2][j] = (typename T::scalar)sc1[j];
            pb[3][j] = (typename T::scalar)sc1[8 + j];
        }
        // the whole softmax belongs to THIS phase: pin its results in registers here (pure arithmetic would otherwise
        // be sunk past the barrier into the matrix phase, next to its first use)
        {
            u32x4_t p0 = __builtin_bit_cast(u32x4_t, pb[0]), p1 = __builtin_bit_cast(u32x4_t, pb[1]);
            u32x4_t p2 = __builtin_bit_cast(u32x4_t, pb[2]), p3 = __builtin_bit_cast(u32x4_t, pb[3]);
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(l_run));
            pb[0] = __builtin_bit_cast(vec8, p0); pb[1] = __builtin_bit_cast(vec8, p1);
            pb[2] = __builtin_bit_cast(vec8, p2); pb[3] = __builtin_bit_cast(vec8, p3);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ASTAMP(ts1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ASTAMP(ts2)
        // ================= matrix phase =================
        // 12 stages: 8 k-steps of S(t+1) (2 K row reads + 2 MFMAs each), then 4 key groups of P.V (8 transposed V reads
        // + 4 MFMAs each).  A stage's LDS reads are issued two (K) / one (V) stages ahead of its MFMAs, pinned by
        // sched_barrier, so every MFMA finds its operands in registers.
#ifndef FINO_ATTN_NOPRIO
        __builtin_amdgcn_s_setprio(1);               // the matrix-phase wave wins issue arbitration on its SIMD
#endif
        u32x4_t ka[3][2];
        s16x4_t vlo[2][kDT], vhi[2][kDT];
#define LOADK(KS_, B_)                                                                                      \
    {                                                                                                       \
        const uint32_t a_ = ka0 ^ ((KS_) << 5);                                                             \
        ka[B_][0] = *LDS_PTR(const u32x4_t, a_);                                                            \
        ka[B_][1] = *LDS_PTR(const u32x4_t, a_ + 32 * D * 2);                                               \
    }
#define LOADV(STEP_, B_)                                                                                    \
    _Pragma("unroll") for (int dt_ = 0; dt_ < kDT; ++dt_) {                                                 \
        vlo[B_][dt_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                             \
            LDS_PTR(s16x4_t, (vl0 ^ (dt_ << 6)) + (STEP_) * 16 * D * 2));                                   \
        vhi[B_][dt_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                             \
            LDS_PTR(s16x4_t, (vh0 ^ (dt_ << 6)) + (STEP_) * 16 * D * 2));                                   \
    }
        if (t + 1 < nt) {
            const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            LOADK(0, 0)
            LOADK(1, 1)
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + 2 < kKS) LOADK(ks + 2, (ks + 2) % 3)
                if (ks == kKS - 2) LOADV(0, 0)
                sc0 = T::mfma32(__builtin_bit_cast(vec8, ka[ks % 3][0]), qf[ks], ks == 0 ? zero16 : sc0);
                sc1 = T::mfma32(__builtin_bit_cast(vec8, ka[ks % 3][1]), qf[ks], ks == 0 ? zero16 : sc1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            LOADV(0, 0)
        }
        MASK_RAGGED(t + 1)
        float mxa = -INFINITY;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            if (step + 1 < 4) LOADV(step + 1, (step + 1) & 1)
#pragma unroll
            for (int dt = 0; dt < kDT; ++dt) {
                const s16x8_t va = __builtin_shufflevector(vlo[step & 1][dt], vhi[step & 1][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = T::mfma32(__builtin_bit_cast(vec8, va), pb[step], o[dt]);
            }
            // row max of S(t+1), a quarter per key group, in the shadow of these MFMAs (S(t+1) is complete: its
            // MFMAs precede these in the pipe)
            mxa = vmax2(mxa, step == 0 ? MAX8(sc0, 0) : step == 1 ? MAX8(sc0, 8) : step == 2 ? MAX8(sc1, 0) : MAX8(sc1, 8));
            __builtin_amdgcn_sched_barrier(0);
        }
        MAX_FINISH1(mxa, mx_next)
#undef LOADK
#undef LOADV
        ka0 ^= kTileBytes;
        vl0 ^= kTileBytes;
        vh0 ^= kTileBytes;
#ifndef FINO_ATTN_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ASTAMP(ts3)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef FINO_ATTN_STAMP
        ASTAMP(ts4)
        sa0 += ts1 - ts0; sa1 += ts2 - ts1; sa2 += ts3 - ts2; sa3 += ts4 - ts3;
        sa5 += tsa - ts0; sa6 += tsb - tsa; sa7 += tsc - tsb;
#endif
    }
#ifdef FINO_ATTN_STAMP
    if (blockIdx.x == 40 && lane == 0 && VAR == 0) {
        fino_attn_dbg[wave * 8 + 0] = sa0; fino_attn_dbg[wave * 8 + 1] = sa1; fino_attn_dbg[wave * 8 + 2] = sa2;
        fino_attn_dbg[wave * 8 + 3] = sa3; fino_attn_dbg[wave * 8 + 4] = (unsigned long long)nt;
        fino_attn_dbg[wave * 8 + 5] = sa5; fino_attn_dbg[wave * 8 + 6] = sa6; fino_attn_dbg[wave * 8 + 7] = sa7;
    }
#endif
    if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef FINO_ATTN_STAMP
    ASTAMP(wg_ts[4])
#endif
#undef QK_TILE
#undef PP_LOAD
#undef PP_WRITE
#undef LDS_PTR
#undef MASK_RAGGED
#undef MAX8
#undef MAX_FINISH1

    // ---------------- epilogue: normalise, store O[q][d] ----------------
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (part >= 0) {
        // partial: raw accumulators in thread order (coalesced), m and l per thread; attn_combine_kernel finishes
        float* w = p.ws + (int64_t)part * partial_floats<D>();
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
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * h;
                uint32_t w0 = (uint32_t)T::from_f32(o[dt][4 * g + 0] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * g + 1] * inv) << 16);
                uint32_t w1 = (uint32_t)T::from_f32(o[dt][4 * g + 2] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * g + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(orow + d0) = make_uint2(w0, w1);
            }
        }
    }
#ifdef FINO_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(wg_ts[5])
    if (blockIdx.x == 40 && lane == 0)
        for (int i = 0; i < 6; ++i) fino_attn_dbg[64 + wave * 8 + i] = wg_ts[i];
#endif
  }   // piece
}

// ---------------------------------------------------------------------------------------------------------------------------
// PING-PONG kernel with LDS-DMA staging (round 4; head_dim 128): the softmax / matrix phase structure of attn_pp_kernel, but
// K / V tiles travel global -> LDS by `buffer_load ... lds` into rings of FOUR slots each (128 KiB of the CU's 160), issued
// three (K) / two (V) tiles ahead of their first use by group 0 and four / three by group 1.  What that removes from the
// softmax phase, which the stamps put at the per-wave issue limit (tools/attn_stamp.py: staging 350 of its 1480 cycles): the
// four ds_write_b128, the vmcnt wait in front of them (a register-staged tile has one tile of compute to arrive, a DMA'd one
// two), and 16 staging registers.  A slot is re-filled only after BOTH groups have passed the matrix phase that read it:
//   K(j) is last read by group 1's matrix(j-1) (global phase 2j), V(j) by its matrix(j) (phase 2j+2); group g issues, in its
//   softmax(t) (phase 2t+g), K(t+g+3) into the slot of K(t+g-1) and V(t+g+2) into the slot of V(t+g-2): both dead by then;
//   every wave ends a softmax phase with vmcnt(4) -- everything it issued before this phase (4 DMAs per phase) has landed --
//   so a tile is published a whole phase before its first reader, whose first K fragments are fetched across the barrier
//   (PD_PREK; vmcnt(8) without it).  In the matrix phase the LDS reads are issued BETWEEN the MFMAs that shadow them: an
//   in-order wave issues nothing while its MFMA waits for the pipe, so reads queued behind a run of MFMAs would start late.
//   The row maxima of S(t) open softmax(t) (out of the P.V shadow, where they delayed MFMAs), waves 4-7 run at static
//   priority 1, and the output rows leave through LDS as whole 256-byte rows (attn_rows_through_lds).
// Every LDS read of the loop is inline asm with hand-counted lgkmcnt waits (the compiler would put vmcnt(0) in front of any
// LDS read it can see while a DMA is in flight, see attn_fr_kernel).  Block map, tail split, partial layout and arithmetic
// are attn_pp_kernel's: results are bit-identical (tests/test_kernels_gpu.py).
constexpr int kPdSlots = 4;

// A/B knobs of attn_ppd_kernel (make variant VFLAGS=-DPD_...=x; results are the same bits for every setting):
#ifndef PD_PREK
#define PD_PREK 1          /* 1: the first two k-steps' K fragments are fetched BEFORE the barrier that opens the matrix phase */
#endif
#ifndef PD_PRIO
#define PD_PRIO 1          /* 0: s_setprio 1 around every matrix phase; 1: static priority for waves 4-7, no flips; 2: none */
#endif
#ifndef PD_MAX_SOFTMAX
#define PD_MAX_SOFTMAX 1   /* 1: the row maxima of S(t) open softmax(t) instead of riding in the P.V shadow of matrix(t-1) */
#endif
template <typename T, int D, int VAR>
__global__ __launch_bounds__(kWaves * 64, 2) void attn_ppd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kTileBytes = kKV * D * 2;                  // 16 KiB (head_dim 128) / 8 KiB (64)
    constexpr int kC = D / 8;                                // 16-byte chunks per row
    constexpr int kRPP = 1024 / (2 * D);                     // tile rows per 1-KiB DMA piece
    constexpr int kPW = kTileBytes / 1024 / kWaves;          // DMA pieces per wave and tile, K and V each: 2 / 1
    constexpr int kKS = D / 16;
    constexpr int kDT = D / 32;
    constexpr int kVBase = kPdSlots * kTileBytes;            // V ring behind the K ring
    constexpr uint32_t kRingMask = kPdSlots * kTileBytes - 1;
    typedef typename T::vec8 vec8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;

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
    if (piece > 0) __syncthreads();                 // every wave is done with the previous piece's LDS tiles (its DMAs have landed: loop end)
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
    const uint16_t* kp = p.k + bi * p.k_bs + head * p.k_hs + (int64_t)t_begin * kKV * p.k_rs;
    const uint16_t* vp = p.v + bi * p.v_bs + head * p.v_hs + (int64_t)t_begin * kKV * p.v_rs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    const int lk = (t_end * kKV < p.lk ? t_end * kKV : p.lk) - t_begin * kKV;
    const int nt = (lk + kKV - 1) / kKV;

    // ---- K / V staging by LDS-DMA: a tile is 16 pieces of 1 KiB (4 rows); wave w moves pieces w and w + 8 of K and of V.
    //      The lane's row inside the piece, its chunk position and the swizzle (row bits a step of 8 pieces leaves alone) are
    //      fixed: ONE per-lane offset per operand, the piece and tile advance ride in the scalar offset; rows past the last
    //      key fail the resource's range check and the DMA writes zeros (masked in the ragged last tile) ----
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wv >> 2;
    const int rip = lane / kC, pos = lane % kC;
    const int srow = wv * kRPP + rip;
    const int kswz = (k_lds_off<D>(srow, 0) >> 4) & (kC - 1), vswz = (lds_off<D>(srow, 0) >> 4) & (kC - 1);
    const uint32_t k_voff = (uint32_t)((srow * p.k_rs + ((pos ^ kswz) << 3)) * 2);
    const uint32_t v_voff = (uint32_t)((srow * p.v_rs + ((pos ^ vswz) << 3)) * 2);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (int)((((int64_t)lk - 1) * p.k_rs + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)vp, 0, (int)((((int64_t)lk - 1) * p.v_rs + D) * 2), 0x00020000);
    const int k_piece_bytes = (int)(8 * kRPP * p.k_rs * 2), v_piece_bytes = (int)(8 * kRPP * p.v_rs * 2);
    const int k_tile_bytes = (int)(kKV * p.k_rs * 2), v_tile_bytes = (int)(kKV * p.v_rs * 2);
    // tile index clamped to nt: tile nt lies wholly past the last key (zeros), and the scalar offset stays inside 32 bits
#define PD_DMA_K(TILE_)                                                                                      \
    {                                                                                                        \
        const int tl_ = (TILE_) < nt ? (TILE_) : nt;                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
                k_rsrc, (FINO_LDS void*)(smem + ((TILE_) & (kPdSlots - 1)) * kTileBytes + (8 * i_ + wv) * 1024), 16, \
                k_voff, tl_ * k_tile_bytes + i_ * k_piece_bytes, 0, 0);                                      \
    }
#define PD_DMA_V(TILE_)                                                                                      \
    {                                                                                                        \
        const int tl_ = (TILE_) < nt ? (TILE_) : nt;                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
                v_rsrc, (FINO_LDS void*)(smem + kVBase + ((TILE_) & (kPdSlots - 1)) * kTileBytes + (8 * i_ + wv) * 1024), \
                16, v_voff, tl_ * v_tile_bytes + i_ * v_piece_bytes, 0, 0);                                  \
    }
    // prologue: what the loop's schedule assumes was issued before it starts -- K(0..2), V(0..1) by everyone, and group 1
    // (whose softmax(0) issues K(4) / V(3)) also K(3) / V(2)
    PD_DMA_K(0) PD_DMA_V(0) PD_DMA_K(1) PD_DMA_V(1) PD_DMA_K(2)
    if (grp == 1) { PD_DMA_K(3) PD_DMA_V(2) }

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + r][16*ks + 8h .. +7] ----
    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
    vec8 qf[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
        uint4 u = *reinterpret_cast<const uint4*>(qp + (int64_t)qrow_c * p.q_rs + 16 * ks + 8 * h);
        if (qrow >= p.lq) u = make_uint4(0, 0, 0, 0);      // rows past Lq are never stored: zero operands draw the least power
        qf[ks] = __builtin_bit_cast(vec8, u);
    }

    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    const int g1l = (lane >> 4) & 1;
    f32x16_t o[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;
    const float c2 = p.scale_log2;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // S(0) from K ring slot 0 (plain LDS reads: nothing is in flight here)
    f32x16_t sc0, sc1;
    {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc0[j] = 0.f; sc1[j] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(smem + k_lds_off<D>(r, 2 * ks + h));
            const uint4 a1 = *reinterpret_cast<const uint4*>(smem + k_lds_off<D>(32 + r, 2 * ks + h));
            sc0 = T::mfma32(__builtin_bit_cast(vec8, a0), qf[ks], sc0);
            sc1 = T::mfma32(__builtin_bit_cast(vec8, a1), qf[ks], sc1);
        }
    }
#define MASK_RAGGED(T_)                                                                                     \
    if ((T_) == nt - 1 && (lk & (kKV - 1))) {            /* key = (j&3) + 8*(j>>2) + 4*h (+32) */           \
        const int kbase_ = (T_) * kKV + 4 * h;                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                 \
            const int key_ = kbase_ + (j_ & 3) + 8 * (j_ >> 2);                                             \
            if (key_ >= lk) sc0[j_] = -INFINITY;                                                            \
            if (key_ + 32 >= lk) sc1[j_] = -INFINITY;                                                       \
        }                                                                                                   \
    }
#define MAX8(S_, O_) vmax2(vmax3(vmax3(S_[O_], S_[O_ + 1], S_[O_ + 2]), vmax3(S_[O_ + 3], S_[O_ + 4], S_[O_ + 5]), \
                                 S_[O_ + 6]), S_[O_ + 7])
#define MAX_FINISH1(MX_, OUT_)                                                                              \
    {                                                                                                       \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(MX_), __float_as_uint(MX_), false, false); \
        OUT_ = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                     \
    }
    float mx_next = 0.f;
#if !PD_MAX_SOFTMAX
    {
        MASK_RAGGED(0)
        const float m0 = vmax2(vmax3(MAX8(sc0, 0), MAX8(sc0, 8), MAX8(sc1, 0)), MAX8(sc1, 8));
        MAX_FINISH1(m0, mx_next)
    }
#endif
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on
#if PD_PRIO == 1
    if (grp == 1) __builtin_amdgcn_s_setprio(1);      // the second-dispatched half loses every age arbitration otherwise
#endif

    // LDS fragment addresses of the matrix phase: k-step / d-tile flip address bits (one v_xor next to the read), the
    // +32 / +16-row operands are instruction offsets, the ring slot (K: (t+1) & 3, V: t & 3) advances once per tile.
    if ((uint32_t)(uintptr_t)(FINO_LDS char*)smem != 0u) __builtin_trap();
    uint32_t ka0 = 1 * kTileBytes + k_lds_off<D>(r, h);
    uint32_t vl0 = kVBase + lds_off<D>(4 * h + tq, 2 * g1l + (tp >> 1)) + 8 * (tp & 1);
    uint32_t vh0 = kVBase + lds_off<D>(4 * h + tq + 8, 2 * g1l + (tp >> 1)) + 8 * (tp & 1);

    u32x4_t ka[3][2];
    s16x4_t vlo[2][kDT], vhi[2][kDT];
#define PD_KISSUE(KS_, B_)                                                                                   \
    {                                                                                                        \
        const uint32_t a_ = ka0 ^ ((KS_) << 5);                                                              \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"                                  \
                     : "=&v"(ka[B_][0]), "=&v"(ka[B_][1]) : "v"(a_), "n"(32 * D * 2));                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define PD_KWAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(ka[B_][0]), "+v"(ka[B_][1]));
    // one MFMA, then the reads its 32 cycles shadow -- an in-order wave issues nothing while its MFMA waits for the
    // pipe, so reads queued behind a run of MFMAs start late and the next run waits for them
#define PD_A_S(KS_, B_, H_) __builtin_bit_cast(vec8, ka[B_][H_])
#define PD_A_V(STEP_, B_, DT_) __builtin_bit_cast(vec8, __builtin_shufflevector(vlo[B_][DT_], vhi[B_][DT_], 0, 1, 2, 3, 4, 5, 6, 7))
#define PD_MF0(KS_, B_, FIRST_)                                                                              \
    sc0 = T::mfma32(PD_A_S(KS_, B_, 0), qf[KS_], (FIRST_) ? zero16 : sc0);                                   \
    __builtin_amdgcn_sched_barrier(0);
#define PD_MF1(KS_, B_, FIRST_)                                                                              \
    sc1 = T::mfma32(PD_A_S(KS_, B_, 1), qf[KS_], (FIRST_) ? zero16 : sc1);                                   \
    __builtin_amdgcn_sched_barrier(0);
#define PD_VPAIR(STEP_, B_, DT_)                                                                             \
    {                                                                                                        \
        const uint32_t l_ = vl0 ^ ((DT_) << 6), h_ = vh0 ^ ((DT_) << 6);                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"            \
                     : "=&v"(vlo[B_][DT_]), "=&v"(vhi[B_][DT_]) : "v"(l_), "v"(h_), "n"((STEP_) * 16 * D * 2)); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define PD_VW(N_, B_, DT_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(vlo[B_][DT_]), "+v"(vhi[B_][DT_]));
#define PD_MFV(STEP_, B_, DT_)                                                                               \
    {                                                                                                        \
        o[DT_] = T::mfma32(PD_A_V(STEP_, B_, DT_), pb[STEP_], o[DT_]);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    // one key step of P.V: d-tile DT_'s MFMA, then the next step's fragments for the same d-tile
#define PD_PVSTEP(STEP_, B_, NB_, NSTEP_)                                                                    \
    PD_VW(6, B_, 0) PD_MFV(STEP_, B_, 0) PD_VPAIR(NSTEP_, NB_, 0)                                            \
    PD_VW(6, B_, 1) PD_MFV(STEP_, B_, 1) PD_VPAIR(NSTEP_, NB_, 1)                                            \
    PD_VW(6, B_, 2) PD_MFV(STEP_, B_, 2) PD_VPAIR(NSTEP_, NB_, 2)                                            \
    PD_VW(6, B_, 3) PD_MFV(STEP_, B_, 3) PD_VPAIR(NSTEP_, NB_, 3)
    // head_dim 64: two d-tiles per key step (4 reads in flight per step instead of 8)
#define PD_PVSTEP64(STEP_, B_, NB_, NSTEP_)                                                                  \
    PD_VW(2, B_, 0) PD_MFV(STEP_, B_, 0) PD_VPAIR(NSTEP_, NB_, 0)                                            \
    PD_VW(2, B_, 1) PD_MFV(STEP_, B_, 1) PD_VPAIR(NSTEP_, NB_, 1)
#if PD_MAX_SOFTMAX
#define PD_SHADOW_MAX(MAXEXPR_)
#else
#define PD_SHADOW_MAX(MAXEXPR_) mxa = vmax2(mxa, MAXEXPR_);
#endif

#ifdef FINO_ATTN_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsm = 0, tse = 0, sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0, sa5 = 0, sa6 = 0;
#endif
    for (int t = 0; t < nt; ++t) {
        ASTAMP(ts0)
        // ================= softmax phase =================
        const int w = t + grp;
        PD_DMA_K(w + 3)
        PD_DMA_V(w + 2)
#if PD_MAX_SOFTMAX
        {
            MASK_RAGGED(t)
            const float m0 = vmax2(vmax3(MAX8(sc0, 0), MAX8(sc0, 8), MAX8(sc1, 0)), MAX8(sc1, 8));
            MAX_FINISH1(m0, mx_next)
        }
#endif
        {
            const float m_cand = fmaxf(m_run, mx_next * c2);
            if (__any((m_cand - m_run) > rescale_thr<T>())) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
                m_run = m_cand;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < kDT; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[i][j] *= alpha;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sc0[j] = __builtin_amdgcn_exp2f(sc0[j] * c2 - m_run);
            sc1[j] = __builtin_amdgcn_exp2f(sc1[j] * c2 - m_run);
        }
        {
            float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                psum0 += sc0[j];
                psum1 += sc1[j];
            }
            l_run += psum0 + psum1;
        }
        vec8 pb[4];                                  // P(t) packed: pb[2*kt + s2]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pb[0][j] = (typename T::scalar)sc0[j];
            pb[1][j] = (typename T::scalar)sc0[8 + j];
            pb[2][j] = (typename T::scalar)sc1[j];
            pb[3][j] = (typename T::scalar)sc1[8 + j];
        }
        {
            u32x4_t p0 = __builtin_bit_cast(u32x4_t, pb[0]), p1 = __builtin_bit_cast(u32x4_t, pb[1]);
            u32x4_t p2 = __builtin_bit_cast(u32x4_t, pb[2]), p3 = __builtin_bit_cast(u32x4_t, pb[3]);
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(l_run));
            pb[0] = __builtin_bit_cast(vec8, p0); pb[1] = __builtin_bit_cast(vec8, p1);
            pb[2] = __builtin_bit_cast(vec8, p2); pb[3] = __builtin_bit_cast(vec8, p3);
        }
        __builtin_amdgcn_sched_barrier(0);
        ASTAMP(tse)
#if PD_PREK
        // K(t+1) was published a phase ago (vmcnt(4) below): the first two k-steps' fragments travel across the barrier
        if (t + 1 < nt) {
            PD_KISSUE(0, 0)
            PD_KISSUE(1, 1)
        }
        // all DMAs this wave issued before this phase (4 per phase) have landed
        if constexpr (kPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
        // all but this wave's last two phases' DMAs (4 per phase) have landed: its pieces of K(t+1) and V(t), at the latest
        if constexpr (kPW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
        ASTAMP(ts1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ASTAMP(ts2)
        // ================= matrix phase =================
#if PD_PRIO == 0
        __builtin_amdgcn_s_setprio(1);
#endif
        const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (D == 128) {
        if (t + 1 < nt) {
#if !PD_PREK
            PD_KISSUE(0, 0)
            PD_KISSUE(1, 1)
#endif
            PD_KWAIT(2, 0) PD_MF0(0, 0, true) PD_KISSUE(2, 2) PD_MF1(0, 0, true)
            PD_KWAIT(2, 1) PD_MF0(1, 1, false) PD_KISSUE(3, 0) PD_MF1(1, 1, false)
            PD_KWAIT(2, 2) PD_MF0(2, 2, false) PD_KISSUE(4, 1) PD_MF1(2, 2, false)
            PD_KWAIT(2, 0) PD_MF0(3, 0, false) PD_KISSUE(5, 2) PD_MF1(3, 0, false)
            PD_KWAIT(2, 1) PD_MF0(4, 1, false) PD_KISSUE(6, 0) PD_MF1(4, 1, false)
            PD_KWAIT(2, 2) PD_MF0(5, 2, false) PD_KISSUE(7, 1) PD_MF1(5, 2, false)
            PD_KWAIT(2, 0) PD_MF0(6, 0, false) PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1) PD_MF1(6, 0, false)
            PD_KWAIT(4, 1) PD_MF0(7, 1, false) PD_VPAIR(0, 0, 2) PD_VPAIR(0, 0, 3) PD_MF1(7, 1, false)
        } else {
            PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1) PD_VPAIR(0, 0, 2) PD_VPAIR(0, 0, 3)
        }
        ASTAMP(tsm)
#if !PD_MAX_SOFTMAX
        MASK_RAGGED(t + 1)
        float mxa = -INFINITY;
#endif
        PD_PVSTEP(0, 0, 1, 1)
        PD_SHADOW_MAX(MAX8(sc0, 0))
        PD_PVSTEP(1, 1, 0, 2)
        PD_SHADOW_MAX(MAX8(sc0, 8))
        PD_PVSTEP(2, 0, 1, 3)
        PD_SHADOW_MAX(MAX8(sc1, 0))
        PD_VW(6, 1, 0) PD_MFV(3, 1, 0)
        PD_VW(4, 1, 1) PD_MFV(3, 1, 1)
        PD_VW(2, 1, 2) PD_MFV(3, 1, 2)
        PD_VW(0, 1, 3) PD_MFV(3, 1, 3)
        PD_SHADOW_MAX(MAX8(sc1, 8))
      } else {                                     // head_dim 64: 4 k-steps of S, 4 key steps x 2 d-tiles of P.V
        if (t + 1 < nt) {
#if !PD_PREK
            PD_KISSUE(0, 0)
            PD_KISSUE(1, 1)
#endif
            PD_KWAIT(2, 0) PD_MF0(0, 0, true) PD_KISSUE(2, 2) PD_MF1(0, 0, true)
            PD_KWAIT(2, 1) PD_MF0(1, 1, false) PD_KISSUE(3, 0) PD_MF1(1, 1, false)
            PD_KWAIT(2, 2) PD_MF0(2, 2, false) PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1) PD_MF1(2, 2, false)
            PD_KWAIT(4, 0) PD_MF0(3, 0, false) PD_MF1(3, 0, false)
        } else {
            PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1)
        }
        ASTAMP(tsm)
#if !PD_MAX_SOFTMAX
        MASK_RAGGED(t + 1)
        float mxa = -INFINITY;
#endif
        PD_PVSTEP64(0, 0, 1, 1)
        PD_SHADOW_MAX(MAX8(sc0, 0))
        PD_PVSTEP64(1, 1, 0, 2)
        PD_SHADOW_MAX(MAX8(sc0, 8))
        PD_PVSTEP64(2, 0, 1, 3)
        PD_SHADOW_MAX(MAX8(sc1, 0))
        PD_VW(2, 1, 0) PD_MFV(3, 1, 0)
        PD_VW(0, 1, 1) PD_MFV(3, 1, 1)
        PD_SHADOW_MAX(MAX8(sc1, 8))
#if !PD_MAX_SOFTMAX
        MAX_FINISH1(mxa, mx_next)
#endif
      }
#if !PD_MAX_SOFTMAX
        MAX_FINISH1(mxa, mx_next)
#endif
        ka0 = (ka0 + kTileBytes) & kRingMask;
        vl0 = ((vl0 + kTileBytes) & kRingMask) | kVBase;
        vh0 = ((vh0 + kTileBytes) & kRingMask) | kVBase;
#if PD_PRIO == 0
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        ASTAMP(ts3)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef FINO_ATTN_STAMP
        ASTAMP(ts4)
        sa0 += ts1 - ts0; sa1 += ts2 - ts1; sa2 += ts3 - ts2; sa3 += ts4 - ts3; sa5 += tsm - ts2; sa6 += tse - ts0;
#endif
    }
#ifdef FINO_ATTN_STAMP
    if (blockIdx.x == 40 && lane == 0 && VAR == 0) {
        fino_attn_dbg[wave * 8 + 0] = sa0; fino_attn_dbg[wave * 8 + 1] = sa1; fino_attn_dbg[wave * 8 + 2] = sa2;
        fino_attn_dbg[wave * 8 + 3] = sa3; fino_attn_dbg[wave * 8 + 4] = (unsigned long long)nt;
        fino_attn_dbg[wave * 8 + 5] = sa5; fino_attn_dbg[wave * 8 + 6] = sa6; fino_attn_dbg[wave * 8 + 7] = 0;
    }
#endif
#if PD_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile (zeros) have landed too ...
    __builtin_amdgcn_s_barrier();                      // ... for EVERY wave: the epilogue stages its output rows in the rings
#undef PD_DMA_K
#undef PD_DMA_V
#undef PD_KISSUE
#undef PD_KWAIT
#undef PD_SHADOW_MAX
#undef PD_MF0
#undef PD_A_S
#undef PD_A_V
#undef PD_MF1
#undef PD_VPAIR
#undef PD_VW
#undef PD_MFV
#undef PD_PVSTEP
#undef PD_PVSTEP64
#undef MASK_RAGGED
#undef MAX8
#undef MAX_FINISH1

    // ---------------- epilogue: normalise, store O[q][d] (as attn_pp_kernel) ----------------
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (part >= 0) {
        float* wsp = p.ws + (int64_t)part * partial_floats<D>();
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt)
#pragma unroll
            for (int j = 0; j < 16; ++j) wsp[(dt * 16 + j) * (kWaves * 64) + tid] = o[dt][j];
        wsp[kDT * 16 * (kWaves * 64) + tid] = m_run;
        wsp[kDT * 16 * (kWaves * 64) + kWaves * 64 + tid] = l_run;
        continue;
    }
    const float inv = 1.0f / l_run;
    {
        // every wave is past its last matrix phase (the barrier above): the K ring's first 64 KiB take the output rows, 8 KiB per wave
        int le = lane;
        asm volatile("" : "+v"(le));       // opaque: or the epilogue's per-lane offsets are computed before the loop and kept live
        u32x4_t rows[D / 16];
        attn_rows_through_lds<T, D>(o, inv, (uint32_t)(wv * (64 * D)), le & 31, le >> 5, le, rows);
        attn_store_rows<D>(rows, op, p.o_rs, qb * kQBlock + wave * kQRowsPerWave, p.lq, le);
    }
  }   // piece
}

// ---------------------------------------------------------------------------------------------------------------------------
// WALKING ping-pong kernel (round 4; head_dim 128, short key sequences: the text cross-attention, Lk = 512 = 8 key tiles per
// q-block).  attn_ppd_kernel's loop, but ONE workgroup per CU walks a run of consecutive (head-major) q-blocks without ever
// draining: the K / V rings keep streaming across block boundaries (the DMA streams carry their own block / tile position,
// three and two tiles ahead), the next block's Q rows are prefetched into a second register set at the start of a block, and
// a block's normalise + store happens at the end of its last matrix phase.  What a workgroup of the one-block kernels pays per
// 8 tiles of work -- Q load 4 - 6 us, first tiles 3 us, output store 3 - 5 us with nothing else resident on the CU
// (profiles/r03_attn_cross_stamp.txt) -- is paid once per run instead of once per block.
// vmcnt is in issue order over DMAs, Q loads and stores alike; the invariant of attn_ppd_kernel ("at the end of a softmax phase
// everything issued before that phase has landed") is kept by counting them: a block's first softmax phase issues 4 DMAs + 8 Q
// loads behind the 16 stores of the previous block's epilogue and the 4 DMAs of the phase before.
// Whole blocks only (no tail split, no partials), 2 <= key tiles; arithmetic and order as attn_pp_kernel: same bits.
// TAIL (fino_attn_fwd_tail): per-batch key counts and a logit offset on a batch element's LAST key, which then stands for a run
// of identical keys (the zero-padded tail of a prompt: every padding token has the same K and V row) -- softmax(q.[K; k x M])
// [V; v x M] = softmax(q.[K; k] + [0; ln M]) [V; v].  Only the masking branch of the softmax phase differs; TAIL = false is
// the kernel as it was, bit for bit.
template <typename T, bool TAIL = false>
__global__ __launch_bounds__(kWaves * 64, 2) void attn_ppw_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 128;
    constexpr int kTileBytes = kKV * D * 2;                  // 16 KiB
    constexpr int kKS = D / 16;
    constexpr int kDT = D / 32;
    constexpr int kVBase = kPdSlots * kTileBytes;
    constexpr uint32_t kRingMask = kPdSlots * kTileBytes - 1;
    typedef typename T::vec8 vec8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wv >> 2;

    // ---- my run of q-blocks, head-major: block B = (batch * heads + head) * nqb + q-block ----
    const int total = p.batch * p.heads * p.nqb;
    const int b_begin = (int)((int64_t)blockIdx.x * total / gridDim.x);
    const int nblk = (int)((int64_t)(blockIdx.x + 1) * total / gridDim.x) - b_begin;
    const int lk = p.lk;
    const int nt = (lk + kKV - 1) / kKV;
    const int TT = nblk * nt;                                // key tiles of the run
    // TAIL: the per-batch key counts / offsets as scalars BEFORE the loop (a kernel-argument load inside it would share lgkmcnt
    // with the loop's counted LDS reads)
    int tlk0 = lk, tlk1 = lk, tlk2 = lk, tlk3 = lk;
    float tb0 = 0.f, tb1 = 0.f, tb2 = 0.f, tb3 = 0.f;
    if constexpr (TAIL) {
        tlk0 = p.tail_lk[0]; tlk1 = p.tail_lk[1]; tlk2 = p.tail_lk[2]; tlk3 = p.tail_lk[3];
        tb0 = p.tail_bias[0]; tb1 = p.tail_bias[1]; tb2 = p.tail_bias[2]; tb3 = p.tail_bias[3];
        asm volatile("" : "+s"(tlk0), "+s"(tlk1), "+s"(tlk2), "+s"(tlk3), "+s"(tb0), "+s"(tb1), "+s"(tb2), "+s"(tb3));
    }

    // position of a stream in the run: relative block, its (batch, head, q-block), tile inside the block
    struct Pos { int blk, bi, head, qb, lt; };
    Pos cp, kp_, vp_;                                        // compute, K DMA, V DMA
    {
        const int hb = b_begin / p.nqb;
        cp.blk = 0; cp.qb = b_begin - hb * p.nqb; cp.bi = hb / p.heads; cp.head = hb - cp.bi * p.heads; cp.lt = 0;
        kp_ = cp; vp_ = cp;
    }
#define PW_NEXT_BLOCK(P_)                                                                                    \
    {                                                                                                        \
        ++(P_).blk;                                                                                          \
        if (++(P_).qb == p.nqb) { (P_).qb = 0; if (++(P_).head == p.heads) { (P_).head = 0; ++(P_).bi; } }   \
    }

    // ---- K / V staging by LDS-DMA (as attn_ppd_kernel): wave w moves pieces w and w + 8 of a tile ----
    const int rip = lane >> 4, pos = lane & 15;
    const int srow = wv * 4 + rip;
    const int swz = (lds_off<D>(srow, 0) >> 4) & 15;
    const uint32_t k_voff = (uint32_t)((srow * p.k_rs + ((pos ^ swz) << 3)) * 2);
    const uint32_t v_voff = (uint32_t)((srow * p.v_rs + ((pos ^ swz) << 3)) * 2);
    // ONE resource per operand over the whole tensor (the launch checks each spans < 2 GiB): a per-head resource would be loop
    // state in vector registers and every DMA a waterfall loop.  The block's (batch, head) base rides in the scalar offset.
    // Key rows past lk then read the next batch's rows (or zeros past the tensor) instead of zeros: they only ever meet
    // P = exp2(-inf) = 0 in the ragged last tile, and finite x 0 = 0.
    const int k_rec = (int)((((int64_t)p.batch - 1) * p.k_bs + ((int64_t)p.heads - 1) * p.k_hs + ((int64_t)lk - 1) * p.k_rs + D) * 2);
    const int v_rec = (int)((((int64_t)p.batch - 1) * p.v_bs + ((int64_t)p.heads - 1) * p.v_hs + ((int64_t)lk - 1) * p.v_rs + D) * 2);
    const int o_rec = (int)((((int64_t)p.batch - 1) * p.o_bs + ((int64_t)p.heads - 1) * p.o_hs + ((int64_t)p.lq - 1) * p.o_rs + D) * 2);
    const int k_piece_bytes = (int)(32 * p.k_rs * 2), v_piece_bytes = (int)(32 * p.v_rs * 2);
    const int k_tile_bytes = (int)(kKV * p.k_rs * 2), v_tile_bytes = (int)(kKV * p.v_rs * 2);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, k_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.v, 0, v_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.o, 0, o_rec, 0x00020000);
    // (32-bit scalar arithmetic: the spans fit 31 bits; a 64-bit product would be computed in vector registers)
    const int k_bs2 = (int)(p.k_bs * 2), k_hs2 = (int)(p.k_hs * 2), v_bs2 = (int)(p.v_bs * 2), v_hs2 = (int)(p.v_hs * 2);
    const int o_bs2 = (int)(p.o_bs * 2), o_hs2 = (int)(p.o_hs * 2);
    int k_boff = kp_.bi * k_bs2 + kp_.head * k_hs2, v_boff = vp_.bi * v_bs2 + vp_.head * v_hs2;
    int k_slot = 0, v_slot = 0;                              // ring slots of the next tiles to issue
    // the stream's next tile -> its ring slot; past the run's last tile: tile nt of the last block (wholly past the last key: zeros)
#define PW_DMA_K()                                                                                           \
    {                                                                                                        \
        const int tl_ = kp_.blk < nblk ? kp_.lt : nt;                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
                k_rsrc, (FINO_LDS void*)(smem + k_slot * kTileBytes + (8 * i_ + wv) * 1024), 16, k_voff,     \
                __builtin_amdgcn_readfirstlane(k_boff + tl_ * k_tile_bytes + i_ * k_piece_bytes), 0, 0);     \
        k_slot = (k_slot + 1) & (kPdSlots - 1);                                                              \
        if (kp_.blk < nblk && ++kp_.lt == nt) {                                                              \
            kp_.lt = 0;                                                                                      \
            PW_NEXT_BLOCK(kp_)                                                                               \
            if (kp_.blk < nblk) k_boff = kp_.bi * k_bs2 + kp_.head * k_hs2;                                  \
        }                                                                                                    \
    }
#define PW_DMA_V()                                                                                           \
    {                                                                                                        \
        const int tl_ = vp_.blk < nblk ? vp_.lt : nt;                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                        \
                v_rsrc, (FINO_LDS void*)(smem + kVBase + v_slot * kTileBytes + (8 * i_ + wv) * 1024), 16,    \
                v_voff, __builtin_amdgcn_readfirstlane(v_boff + tl_ * v_tile_bytes + i_ * v_piece_bytes), 0, 0); \
        v_slot = (v_slot + 1) & (kPdSlots - 1);                                                              \
        if (vp_.blk < nblk && ++vp_.lt == nt) {                                                              \
            vp_.lt = 0;                                                                                      \
            PW_NEXT_BLOCK(vp_)                                                                               \
            if (vp_.blk < nblk) v_boff = vp_.bi * v_bs2 + vp_.head * v_hs2;                                  \
        }                                                                                                    \
    }
    // prologue: K(0..2), V(0..1) by everyone; group 1 (whose first softmax phase issues K(4) / V(3)) also K(3) / V(2)
    PW_DMA_K() PW_DMA_V() PW_DMA_K() PW_DMA_V() PW_DMA_K()
    if (grp == 1) { PW_DMA_K() PW_DMA_V() }

    // ---- Q fragments of a block: lane holds Q[q0 + r][16*ks + 8h .. +7], loaded RAW (a select on the loaded value would
    //      make the compiler wait for it on the spot, DMAs included); rows past Lq are zeroed when the set becomes current ----
    //      The loads are inline asm: loads the compiler can see are loop-carried pending events to it (it cannot read the
    //      counted waits), and it would drain the queue -- the previous block's 16 stores included -- before re-using qn.
#define PW_LOAD_Q(DST_, P_)                                                                                  \
    {                                                                                                        \
        const int qrow_ = (P_).qb * kQBlock + wave * kQRowsPerWave + r;                                      \
        const int qrc_ = qrow_ < p.lq ? qrow_ : p.lq - 1;                                                    \
        const uint16_t* qp_ = p.q + (P_).bi * p.q_bs + (P_).head * p.q_hs + (int64_t)qrc_ * p.q_rs + 8 * h;  \
        asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:32\n\t"      \
                     "global_load_dwordx4 %2, %8, off offset:64\n\tglobal_load_dwordx4 %3, %8, off offset:96\n\t" \
                     "global_load_dwordx4 %4, %8, off offset:128\n\tglobal_load_dwordx4 %5, %8, off offset:160\n\t" \
                     "global_load_dwordx4 %6, %8, off offset:192\n\tglobal_load_dwordx4 %7, %8, off offset:224"  \
                     : "=&v"(DST_[0]), "=&v"(DST_[1]), "=&v"(DST_[2]), "=&v"(DST_[3]), "=&v"(DST_[4]),       \
                       "=&v"(DST_[5]), "=&v"(DST_[6]), "=&v"(DST_[7])                                        \
                     : "v"(qp_) : "memory");                                                                 \
    }
#define PW_TAKE_Q(P_)                                                                                        \
    {                                                                                                        \
        const bool live_ = (P_).qb * kQBlock + wave * kQRowsPerWave + r < p.lq;                              \
        _Pragma("unroll") for (int ks_ = 0; ks_ < kKS; ++ks_)                                                \
            qf[ks_] = __builtin_bit_cast(vec8, live_ ? qn[ks_] : u32x4_t{0u, 0u, 0u, 0u});                   \
    }
    vec8 qf[kKS];
    u32x4_t qn[kKS];
    PW_LOAD_Q(qn, cp)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]), "+v"(qn[4]), "+v"(qn[5]), "+v"(qn[6]),
                 "+v"(qn[7]) :: "memory");
    PW_TAKE_Q(cp)

    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    const int g1l = (lane >> 4) & 1;
    f32x16_t o[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;
    const float c2 = p.scale_log2;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // S(0) from K ring slot 0 (plain LDS reads: nothing is in flight here)
    f32x16_t sc0, sc1;
    {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sc0[j] = 0.f; sc1[j] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(smem + lds_off<D>(r, 2 * ks + h));
            const uint4 a1 = *reinterpret_cast<const uint4*>(smem + lds_off<D>(32 + r, 2 * ks + h));
            sc0 = T::mfma32(__builtin_bit_cast(vec8, a0), qf[ks], sc0);
            sc1 = T::mfma32(__builtin_bit_cast(vec8, a1), qf[ks], sc1);
        }
    }
#define MAX8(S_, O_) vmax2(vmax3(vmax3(S_[O_], S_[O_ + 1], S_[O_ + 2]), vmax3(S_[O_ + 3], S_[O_ + 4], S_[O_ + 5]), \
                                 S_[O_ + 6]), S_[O_ + 7])
    float mx_next = 0.f;
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one phase behind group 0 from here on
    if (grp == 1) __builtin_amdgcn_s_setprio(1);      // the second-dispatched half loses every age arbitration otherwise

    if ((uint32_t)(uintptr_t)(FINO_LDS char*)smem != 0u) __builtin_trap();
    uint32_t ka0 = 1 * kTileBytes + lds_off<D>(r, h);
    uint32_t vl0 = kVBase + lds_off<D>(4 * h + tq, 2 * g1l + (tp >> 1)) + 8 * (tp & 1);
    uint32_t vh0 = kVBase + lds_off<D>(4 * h + tq + 8, 2 * g1l + (tp >> 1)) + 8 * (tp & 1);

    u32x4_t ka[3][2];
    s16x4_t vlo[2][kDT], vhi[2][kDT];
#define PD_KISSUE(KS_, B_)                                                                                   \
    {                                                                                                        \
        const uint32_t a_ = ka0 ^ ((KS_) << 5);                                                              \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"                                  \
                     : "=&v"(ka[B_][0]), "=&v"(ka[B_][1]) : "v"(a_), "n"(32 * D * 2));                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define PD_KWAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(ka[B_][0]), "+v"(ka[B_][1]));
#define PD_MF0(KS_, B_, FIRST_)                                                                              \
    sc0 = T::mfma32(__builtin_bit_cast(vec8, ka[B_][0]), qf[KS_], (FIRST_) ? zero16 : sc0);                  \
    __builtin_amdgcn_sched_barrier(0);
#define PD_MF1(KS_, B_, FIRST_)                                                                              \
    sc1 = T::mfma32(__builtin_bit_cast(vec8, ka[B_][1]), qf[KS_], (FIRST_) ? zero16 : sc1);                  \
    __builtin_amdgcn_sched_barrier(0);
#define PD_VPAIR(STEP_, B_, DT_)                                                                             \
    {                                                                                                        \
        const uint32_t l_ = vl0 ^ ((DT_) << 6), h_ = vh0 ^ ((DT_) << 6);                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"            \
                     : "=&v"(vlo[B_][DT_]), "=&v"(vhi[B_][DT_]) : "v"(l_), "v"(h_), "n"((STEP_) * 16 * D * 2)); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define PD_VW(N_, B_, DT_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(vlo[B_][DT_]), "+v"(vhi[B_][DT_]));
#define PD_MFV(STEP_, B_, DT_)                                                                               \
    {                                                                                                        \
        o[DT_] = T::mfma32(__builtin_bit_cast(vec8, __builtin_shufflevector(vlo[B_][DT_], vhi[B_][DT_], 0, 1, 2, 3, 4, 5, 6, 7)), \
                           pb[STEP_], o[DT_]);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define PD_PVSTEP(STEP_, B_, NB_, NSTEP_)                                                                    \
    PD_VW(6, B_, 0) PD_MFV(STEP_, B_, 0) PD_VPAIR(NSTEP_, NB_, 0)                                            \
    PD_VW(6, B_, 1) PD_MFV(STEP_, B_, 1) PD_VPAIR(NSTEP_, NB_, 1)                                            \
    PD_VW(6, B_, 2) PD_MFV(STEP_, B_, 2) PD_VPAIR(NSTEP_, NB_, 2)                                            \
    PD_VW(6, B_, 3) PD_MFV(STEP_, B_, 3) PD_VPAIR(NSTEP_, NB_, 3)

    Pos np = cp;                                       // the block whose Q rows are in qn
    for (int t = 0; t < TT; ++t) {
        // ================= softmax phase of tile cp.lt of block cp.blk =================
        PW_DMA_K()
        PW_DMA_V()
        if (cp.lt == 0) {                              // the next block's Q rows (the last block re-reads its own: a fixed count)
            np = cp;
            if (cp.blk + 1 < nblk) PW_NEXT_BLOCK(np)
            PW_LOAD_Q(qn, np)
        }
        {
            if constexpr (TAIL) {
                const int lk_c = cp.bi == 0 ? tlk0 : (cp.bi == 1 ? tlk1 : (cp.bi == 2 ? tlk2 : tlk3));
                if ((cp.lt + 1) * kKV >= lk_c) {         // the tile holds this batch element's last key, or lies past it
                    const float tb_ = cp.bi == 0 ? tb0 : (cp.bi == 1 ? tb1 : (cp.bi == 2 ? tb2 : tb3));
                    const int kbase_ = cp.lt * kKV + 4 * h;
#pragma unroll
                    for (int j_ = 0; j_ < 16; ++j_) {
                        const int key_ = kbase_ + (j_ & 3) + 8 * (j_ >> 2);
                        sc0[j_] = key_ >= lk_c ? -INFINITY : (key_ == lk_c - 1 ? sc0[j_] + tb_ : sc0[j_]);
                        sc1[j_] = key_ + 32 >= lk_c ? -INFINITY : (key_ + 32 == lk_c - 1 ? sc1[j_] + tb_ : sc1[j_]);
                    }
                }
            } else if (cp.lt == nt - 1 && (lk & (kKV - 1))) {   // keys past lk (zero K rows) stay out of the row max: p = exp2(-inf) = 0
                const int kbase_ = cp.lt * kKV + 4 * h;
#pragma unroll
                for (int j_ = 0; j_ < 16; ++j_) {
                    const int key_ = kbase_ + (j_ & 3) + 8 * (j_ >> 2);
                    if (key_ >= lk) sc0[j_] = -INFINITY;
                    if (key_ + 32 >= lk) sc1[j_] = -INFINITY;
                }
            }
            const float m0 = vmax2(vmax3(MAX8(sc0, 0), MAX8(sc0, 8), MAX8(sc1, 0)), MAX8(sc1, 8));
            const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
            mx_next = vmax2(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
        }
        {
            const float m_cand = fmaxf(m_run, mx_next * c2);
            if (__any((m_cand - m_run) > rescale_thr<T>())) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
                m_run = m_cand;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < kDT; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[i][j] *= alpha;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sc0[j] = __builtin_amdgcn_exp2f(sc0[j] * c2 - m_run);
            sc1[j] = __builtin_amdgcn_exp2f(sc1[j] * c2 - m_run);
        }
        {
            float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                psum0 += sc0[j];
                psum1 += sc1[j];
            }
            l_run += psum0 + psum1;
        }
        vec8 pb[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pb[0][j] = (typename T::scalar)sc0[j];
            pb[1][j] = (typename T::scalar)sc0[8 + j];
            pb[2][j] = (typename T::scalar)sc1[j];
            pb[3][j] = (typename T::scalar)sc1[8 + j];
        }
        {
            u32x4_t p0 = __builtin_bit_cast(u32x4_t, pb[0]), p1 = __builtin_bit_cast(u32x4_t, pb[1]);
            u32x4_t p2 = __builtin_bit_cast(u32x4_t, pb[2]), p3 = __builtin_bit_cast(u32x4_t, pb[3]);
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(l_run));
            pb[0] = __builtin_bit_cast(vec8, p0); pb[1] = __builtin_bit_cast(vec8, p1);
            pb[2] = __builtin_bit_cast(vec8, p2); pb[3] = __builtin_bit_cast(vec8, p3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // Everything this wave issued before the PREVIOUS softmax phase has landed (a block's output stores get two tiles to be
        // acknowledged: all CUs reach their block boundaries together, 16 MB of stores at once); before a block's last matrix
        // phase, everything before THIS phase (the next block's Q rows, issued in the block's first phase, whatever nt is).
        // Per phase, in issue order: 4 DMAs [+ 8 Q loads in a block's first phase] [+ 8 stores behind its last matrix phase].
        if (cp.lt == nt - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (cp.lt == 0 && cp.blk == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (cp.lt == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (cp.lt == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= matrix phase =================
        const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool last = cp.lt == nt - 1;
        if (t + 1 < TT) {
            PD_KISSUE(0, 0)
            PD_KISSUE(1, 1)
            if (last) PW_TAKE_Q(np)                    // S of the NEXT block's first tile: its Q rows have arrived (vmcnt(4) above)
            PD_KWAIT(2, 0) PD_MF0(0, 0, true) PD_KISSUE(2, 2) PD_MF1(0, 0, true)
            PD_KWAIT(2, 1) PD_MF0(1, 1, false) PD_KISSUE(3, 0) PD_MF1(1, 1, false)
            PD_KWAIT(2, 2) PD_MF0(2, 2, false) PD_KISSUE(4, 1) PD_MF1(2, 2, false)
            PD_KWAIT(2, 0) PD_MF0(3, 0, false) PD_KISSUE(5, 2) PD_MF1(3, 0, false)
            PD_KWAIT(2, 1) PD_MF0(4, 1, false) PD_KISSUE(6, 0) PD_MF1(4, 1, false)
            PD_KWAIT(2, 2) PD_MF0(5, 2, false) PD_KISSUE(7, 1) PD_MF1(5, 2, false)
            PD_KWAIT(2, 0) PD_MF0(6, 0, false) PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1) PD_MF1(6, 0, false)
            PD_KWAIT(4, 1) PD_MF0(7, 1, false) PD_VPAIR(0, 0, 2) PD_VPAIR(0, 0, 3) PD_MF1(7, 1, false)
        } else {
            PD_VPAIR(0, 0, 0) PD_VPAIR(0, 0, 1) PD_VPAIR(0, 0, 2) PD_VPAIR(0, 0, 3)
        }
        PD_PVSTEP(0, 0, 1, 1)
        PD_PVSTEP(1, 1, 0, 2)
        PD_PVSTEP(2, 0, 1, 3)
        PD_VW(6, 1, 0) PD_MFV(3, 1, 0)
        PD_VW(4, 1, 1) PD_MFV(3, 1, 1)
        PD_VW(2, 1, 2) PD_MFV(3, 1, 2)
        PD_VW(0, 1, 3) PD_MFV(3, 1, 3)
        ka0 = (ka0 + kTileBytes) & kRingMask;
        vl0 = ((vl0 + kTileBytes) & kRingMask) | kVBase;
        vh0 = ((vh0 + kTileBytes) & kRingMask) | kVBase;
        if (last) {
            // ---- the block is complete: normalise, store O[q][d], start the next one ----
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
            const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
            // rows through this group's 32 KiB of LDS above the rings (the other group's epilogue is a phase -- a barrier -- away),
            // then buffer stores, ALWAYS 8 per wave (the vmcnt bookkeeping counts them): rows past Lq get an offset beyond the
            // resource's num_records and the hardware drops them
            u32x4_t rows[8];
            attn_rows_through_lds<T, D>(o, inv, 2 * kPdSlots * kTileBytes + (wv & 3) * 8192, r, h, lane, rows);
            const int o_boff = __builtin_amdgcn_readfirstlane(cp.bi * o_bs2 + cp.head * o_hs2);
            const int qrow0 = cp.qb * kQBlock + wave * kQRowsPerWave + (lane >> 4);
            const uint32_t ocol = (uint32_t)((lane & 15) * 16);
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const int qrow = qrow0 + 4 * k8;
                const uint32_t off = qrow < p.lq ? (uint32_t)((int64_t)qrow * p.o_rs * 2) + ocol : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(rows[k8], o_rsrc, off, o_boff, 0);
            }
#pragma unroll
            for (int i = 0; i < kDT; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
            m_run = -INFINITY;
            l_run = 0.f;
            cp.lt = 0;
            PW_NEXT_BLOCK(cp)
        } else {
            ++cp.lt;
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PW_NEXT_BLOCK
#undef PW_DMA_K
#undef PW_DMA_V
#undef PW_LOAD_Q
#undef PW_TAKE_Q
#undef PD_KISSUE
#undef PD_KWAIT
#undef PD_MF0
#undef PD_MF1
#undef PD_VPAIR
#undef PD_VW
#undef PD_MFV
#undef PD_PVSTEP
#undef MAX8
}

// Merge the key-range partials of each tail block: m = max m_s, O = sum O_s 2^(m_s-m), l likewise; store bf16.
// grid = (8 * rem_x, D / 32): one workgroup per (tail block, 32-column d-tile) -- a block's 64 KB of fp32 partials per
// piece are four independent column slices, so the few tail blocks (48 at the bench shape) become 4x the workgroups
// (the single-slice form took 43 us per launch on 48 of 256 CUs).
template <typename T, int D>
__global__ __launch_bounds__(kWaves * 64) void attn_combine_kernel(const AttnParams p) {
    constexpr int kDT = D / 32;
    constexpr int NT = kWaves * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, tb = blockIdx.x >> 3;
    const int dt = blockIdx.y;
    const int bx = p.full_x + tb;
    int hb, qb;
    if (!attn_map_block(p, xcd, bx, hb, qb)) return;
    const int ntall = (p.lk + kKV - 1) / kKV;
    // workgroups (ranges) that hold a piece of this block; one range = the block ran whole and is already stored
    const int c_first = (int)(((int64_t)tb * ntall) / p.per);
    const int c_last = (int)((((int64_t)tb + 1) * ntall - 1) / p.per);
    if (c_first == c_last) return;
    const int bi = hb / p.heads, head = hb - bi * p.heads;
    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    if (qrow >= p.lq) return;
#define PART_PTR(C_) (p.ws + (int64_t)(((xcd * p.nwg) + (C_)) * 2 + (tb - (int)(((int64_t)(C_) * p.per) / ntall))) * partial_floats<D>())
    float m = -INFINITY;
    for (int c = c_first; c <= c_last; ++c) m = fmaxf(m, PART_PTR(c)[kDT * 16 * NT + tid]);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float l = 0.f;
    for (int c = c_first; c <= c_last; ++c) {
        const float* w = PART_PTR(c);
        const float a = __builtin_amdgcn_exp2f(w[kDT * 16 * NT + tid] - m);
        l += a * w[kDT * 16 * NT + NT + tid];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += a * w[(dt * 16 + i) * NT + tid];
    }
#undef PART_PTR
    const float inv = 1.0f / l;
    uint16_t* orow = p.o + bi * p.o_bs + head * p.o_hs + (int64_t)qrow * p.o_rs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        const float* a4 = acc + 4 * g;
        uint32_t x0 = (uint32_t)T::from_f32(a4[0] * inv) | ((uint32_t)T::from_f32(a4[1] * inv) << 16);
        uint32_t x1 = (uint32_t)T::from_f32(a4[2] * inv) | ((uint32_t)T::from_f32(a4[3] * inv) << 16);
        *reinterpret_cast<uint2*>(orow + d0) = make_uint2(x0, x1);
    }
}

// Merge the (O, m, l) partials that fino_attn_partial left for the SAME queries over disjoint key ranges (up to 3: the
// token-sharded DiT attends to its own K/V chunk while the other ranks' chunks are still on the wire, then to what
// arrived before / after it): m = max m_s, O = sum_s O_s 2^(m_s - m), l likewise, store O / l.
struct MergeParams {
    const float* part[3];
    int n_parts;
    uint16_t* o;
    int batch, heads, lq, nqb;
    int64_t o_bs, o_rs, o_hs;
};
template <typename T, int D>
__global__ __launch_bounds__(kWaves * 64) void attn_merge_kernel(const MergeParams p) {
    constexpr int kDT = D / 32;
    constexpr int NT = kWaves * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int blk = blockIdx.x, dt = blockIdx.y;
    const int hb = blk / p.nqb, qb = blk - hb * p.nqb;
    const int bi = hb / p.heads, head = hb - bi * p.heads;
    const int qrow = qb * kQBlock + wave * kQRowsPerWave + r;
    if (qrow >= p.lq) return;
    float m = -INFINITY;
    for (int s = 0; s < p.n_parts; ++s)
        m = fmaxf(m, p.part[s][(int64_t)blk * partial_floats<D>() + kDT * 16 * NT + tid]);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float l = 0.f;
    for (int s = 0; s < p.n_parts; ++s) {
        const float* w = p.part[s] + (int64_t)blk * partial_floats<D>();
        const float a = __builtin_amdgcn_exp2f(w[kDT * 16 * NT + tid] - m);
        l += a * w[kDT * 16 * NT + NT + tid];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += a * w[(dt * 16 + i) * NT + tid];
    }
    const float inv = 1.0f / l;
    uint16_t* orow = p.o + bi * p.o_bs + head * p.o_hs + (int64_t)qrow * p.o_rs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * h;
        const float* a4 = acc + 4 * g;
        uint32_t x0 = (uint32_t)T::from_f32(a4[0] * inv) | ((uint32_t)T::from_f32(a4[1] * inv) << 16);
        uint32_t x1 = (uint32_t)T::from_f32(a4[2] * inv) | ((uint32_t)T::from_f32(a4[3] * inv) << 16);
        *reinterpret_cast<uint2*>(orow + d0) = make_uint2(x0, x1);
    }
}

// CU count of the current device (per-device cache; the tail-split plan and its workspace size depend on it)
int device_cus() {
    static std::atomic<int> cus[kFinoMaxDevices];
    const int dev = fino_current_device();
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        int n = 0;
        c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

// Tail split plan.  One workgroup occupies a CU (2 waves/SIMD x 256 registers) and an XCD's CUs take its blocks in
// rounds.  When the last round holds rem_x < CUs-per-XCD blocks, their rem_x*nt key tiles are dealt to nwg = CUs-per-XCD
// workgroups in equal contiguous ranges of `per` tiles (stream-K on the key axis), so that round lasts rem_x/CUs of a
// block instead of a whole one.  Ranges keep >= kMinTiles tiles, and per < nt so a range touches at most two blocks.
struct SplitPlan { int full_x, rem_x, nwg, per; };
inline SplitPlan plan_split(int batch, int heads, int nqb, int nt) {
    constexpr int kMinTiles = 8;
    const int cus_x = device_cus() / 8 > 0 ? device_cus() / 8 : 1;
    int vsplit, nqb_v;
    attn_virtual_heads(batch, heads, nqb, vsplit, nqb_v);
    const int nblk_x = ((batch * heads * vsplit + 7) / 8) * nqb_v;
    SplitPlan sp{nblk_x, 0, 0, 1};
    const int rem = nblk_x % cus_x;
    if (rem == 0) return sp;
    const int64_t total = (int64_t)rem * nt;
    int nwg = cus_x;
    if (nwg > total / kMinTiles) nwg = (int)(total / kMinTiles);
    if (nwg <= rem) return sp;                       // nothing to gain (or too few keys to cut)
    const int per = (int)((total + nwg - 1) / nwg);
    if (per >= nt) return sp;
    sp.full_x = nblk_x - rem;
    sp.rem_x = rem;
    sp.per = per;
    sp.nwg = (int)((total + per - 1) / per);
    return sp;
}


// ---------------------------------------------------------------------------------------------------------------------------
// FREE-RUNNING kernel (round 3): 4 waves x 32 query rows per workgroup, TWO workgroups per CU (<= 256 registers), ONE barrier
// per key tile, K / V tiles by LDS-DMA into the same 2 + 2 XOR-swizzled ring slots as the ping-pong kernel (the swizzle is
// applied to the DMA's source offsets).  Built for SHORT key sequences -- the text cross-attention, 8 key tiles per
// workgroup -- where a workgroup of the ping-pong kernel spends 12 of its 27 us in load / store latency with nothing else
// resident on its CU (profiles/r03_attn_cross_stamp.txt): here the second workgroup's loop runs under the first one's
// prologue and epilogue.  Whole blocks only (no tail split, no partials).
constexpr int kFrWaves = 4;
constexpr int kFrQBlock = kFrWaves * kQRowsPerWave;      // 128

template <typename T, int D>
__global__ __launch_bounds__(kFrWaves * 64, D == 64 ? 3 : 2) void attn_fr_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kTileBytes = kKV * D * 2;
    constexpr int kKS = D / 16;
    constexpr int kDT = D / 32;
    constexpr int kChunks = D / 8;                          // 16-byte chunks per row
    constexpr int kRPP = 1024 / (D * 2);                    // tile rows per 1-KiB DMA piece (one wave instruction)
    constexpr int kPW = kTileBytes / 1024 / kFrWaves;       // DMA pieces per wave and tile (K and V each)
    typedef typename T::vec8 vec8;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;
    const int xcd = blockIdx.x & 7;
    int hb, qb;
    if (!attn_map_block(p, xcd, (int)(blockIdx.x >> 3), hb, qb)) return;
    const int bi = hb / p.heads;
    const int head = hb - bi * p.heads;
    const uint16_t* qp = p.q + bi * p.q_bs + head * p.q_hs;
    const uint16_t* kp = p.k + bi * p.k_bs + head * p.k_hs;
    const uint16_t* vp = p.v + bi * p.v_bs + head * p.v_hs;
    uint16_t* op = p.o + bi * p.o_bs + head * p.o_hs;
    const int lk = p.lk;
    const int nt = (lk + kKV - 1) / kKV;

    // ---- K / V staging: piece (4 i + wave) of a tile = rows [(4 i + wave) kRPP, + kRPP); the lane's row inside the piece and
    //      its chunk position are fixed, and so is the swizzle (it depends on row bits a piece step of 4 kRPP rows leaves
    //      alone): ONE per-lane offset per operand, the piece and tile advance ride in the scalar offset ----
    const int rip = lane / kChunks, pos = lane % kChunks;
    const int srow = wave * kRPP + rip;
    const int kswz = (k_lds_off<D>(srow, 0) >> 4) & (kChunks - 1), vswz = (lds_off<D>(srow, 0) >> 4) & (kChunks - 1);
    const uint32_t k_voff = (uint32_t)((srow * p.k_rs + ((pos ^ kswz) << 3)) * 2);
    const uint32_t v_voff = (uint32_t)((srow * p.v_rs + ((pos ^ vswz) << 3)) * 2);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)kp, 0, (int)((((int64_t)lk - 1) * p.k_rs + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)vp, 0, (int)((((int64_t)lk - 1) * p.v_rs + D) * 2), 0x00020000);
    const int k_piece_bytes = (int)(4 * kRPP * p.k_rs * 2), v_piece_bytes = (int)(4 * kRPP * p.v_rs * 2);
    const int k_tile_bytes = (int)(kKV * p.k_rs * 2), v_tile_bytes = (int)(kKV * p.v_rs * 2);
    // rows past the last key fail the resource's range check: the DMA writes zeros (masked in the ragged last tile)
#define FR_DMA_K(TILE_, SLOT_)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (FINO_LDS void*)(smem + (SLOT_) * kTileBytes + (4 * i_ + wave) * 1024), \
                                                 16, k_voff, (TILE_) * k_tile_bytes + i_ * k_piece_bytes, 0, 0);
#define FR_DMA_V(TILE_, SLOT_)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < kPW; ++i_)                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (FINO_LDS void*)(smem + (2 + (SLOT_)) * kTileBytes + (4 * i_ + wave) * 1024), \
                                                 16, v_voff, (TILE_) * v_tile_bytes + i_ * v_piece_bytes, 0, 0);
    FR_DMA_K(0, 0)
    FR_DMA_V(0, 0)
    FR_DMA_K(1, 1)

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + r][16*ks + 8h .. +7] ----
    const int qrow = qb * kFrQBlock + wave * kQRowsPerWave + r;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
    vec8 qf[kKS];
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
        uint4 u = *reinterpret_cast<const uint4*>(qp + (int64_t)qrow_c * p.q_rs + 16 * ks + 8 * h);
        if (qrow >= p.lq) u = make_uint4(0, 0, 0, 0);
        qf[ks] = __builtin_bit_cast(vec8, u);
    }

    const int tq = (lane & 15) >> 2;
    const int tp = lane & 3;
    const int g1 = (lane >> 4) & 1;
    f32x16_t o[kDT];
#pragma unroll
    for (int i = 0; i < kDT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;
    const float c2 = p.scale_log2;

    if ((uint32_t)(uintptr_t)(FINO_LDS char*)smem != 0u) __builtin_trap();
#define LDS_PTR(TYPE_, ADDR_) ((FINO_LDS TYPE_*)(uintptr_t)(uint32_t)(ADDR_))
    uint32_t ka0 = k_lds_off<D>(r, h);                                       // K ring slot 0 first (S(0)), then toggled
    uint32_t vl0 = 2 * kTileBytes + lds_off<D>(4 * h + tq, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);
    uint32_t vh0 = 2 * kTileBytes + lds_off<D>(4 * h + tq + 8, 2 * g1 + (tp >> 1)) + 8 * (tp & 1);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16_t sc0, sc1;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Every LDS read of the loop is inline asm: the compiler puts s_waitcnt vmcnt(0) in front of any LDS read IT can see
    // while LDS-DMA writes are in flight (it cannot tell the ring slots apart), which would make the DMA of the next tiles
    // land before this tile's first MFMA instead of under the whole tile.  The waits are explicit and tied to the loaded
    // registers ("+v"), so no consumer can be scheduled above them.
    // K fragments of k-steps 2 G_, 2 G_ + 1 (rows r and 32 + r): 4 x ds_read_b128
#define FR_KISSUE(G_, B_)                                                                                    \
    {                                                                                                        \
        const uint32_t a_ = ka0 ^ ((2 * (G_)) << 5), b_ = ka0 ^ ((2 * (G_) + 1) << 5);                       \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%6\n\t"                              \
                     "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:%6"                                  \
                     : "=&v"(kf[B_][0]), "=&v"(kf[B_][1]), "=&v"(kf[B_][2]), "=&v"(kf[B_][3])                \
                     : "v"(a_), "v"(b_), "n"(32 * D * 2));                                                   \
    }
#define FR_KWAIT(N_, B_)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(kf[B_][0]), "+v"(kf[B_][1]), "+v"(kf[B_][2]), "+v"(kf[B_][3]));
    // V^T fragments of key step STEP_ (16 keys) for d-tile pair (2 P_, 2 P_ + 1): 4 x ds_read_b64_tr_b16
#define FR_VISSUE(STEP_, P_, B_)                                                                             \
    {                                                                                                        \
        const uint32_t l0_ = vl0 ^ ((2 * (P_)) << 6), h0_ = vh0 ^ ((2 * (P_)) << 6);                         \
        const uint32_t l1_ = vl0 ^ ((2 * (P_) + 1) << 6), h1_ = vh0 ^ ((2 * (P_) + 1) << 6);                 \
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%8\n\tds_read_b64_tr_b16 %1, %5 offset:%8\n\t"        \
                     "ds_read_b64_tr_b16 %2, %6 offset:%8\n\tds_read_b64_tr_b16 %3, %7 offset:%8"            \
                     : "=&v"(vf[B_][0]), "=&v"(vf[B_][1]), "=&v"(vf[B_][2]), "=&v"(vf[B_][3])                \
                     : "v"(l0_), "v"(h0_), "v"(l1_), "v"(h1_), "n"((STEP_) * 16 * D * 2));                   \
    }
#define FR_VWAIT(N_, B_)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(vf[B_][0]), "+v"(vf[B_][1]), "+v"(vf[B_][2]), "+v"(vf[B_][3]));
    u32x4_t kf[2][4];
    s16x4_t vf[2][4];
    // S^T = K . Q^T of the tile in the K ring slot ka0 addresses: k-steps in pairs, the next pair's reads under this pair's MFMAs
#define FR_QK()                                                                                              \
    {                                                                                                        \
        FR_KISSUE(0, 0)                                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < kKS / 2; ++g_) {                                             \
            if (g_ + 1 < kKS / 2) {                                                                          \
                if ((g_ & 1) == 0) { FR_KISSUE(g_ + 1, 1) FR_KWAIT(4, 0) } else { FR_KISSUE(g_ + 1, 0) FR_KWAIT(4, 1) } \
            } else {                                                                                         \
                if ((g_ & 1) == 0) { FR_KWAIT(0, 0) } else { FR_KWAIT(0, 1) }                                \
            }                                                                                                \
            const int bb_ = g_ & 1;                                                                          \
            sc0 = T::mfma32(__builtin_bit_cast(vec8, kf[bb_][0]), qf[2 * g_], g_ == 0 ? zero16 : sc0);       \
            sc1 = T::mfma32(__builtin_bit_cast(vec8, kf[bb_][1]), qf[2 * g_], g_ == 0 ? zero16 : sc1);       \
            sc0 = T::mfma32(__builtin_bit_cast(vec8, kf[bb_][2]), qf[2 * g_ + 1], sc0);                      \
            sc1 = T::mfma32(__builtin_bit_cast(vec8, kf[bb_][3]), qf[2 * g_ + 1], sc1);                      \
        }                                                                                                    \
    }
#define FR_MASK(T_)                                                                                          \
    if ((T_) == nt - 1 && (lk & (kKV - 1))) {                                                                \
        const int kbase_ = (T_) * kKV + 4 * h;                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                  \
            const int key_ = kbase_ + (j_ & 3) + 8 * (j_ >> 2);                                              \
            if (key_ >= lk) sc0[j_] = -INFINITY;                                                             \
            if (key_ + 32 >= lk) sc1[j_] = -INFINITY;                                                        \
        }                                                                                                    \
    }
#define FR_ROWMAX(OUT_)                                                                                      \
    {                                                                                                        \
        float mx_ = fmaxf(sc0[0], sc1[0]);                                                                   \
        _Pragma("unroll") for (int j_ = 1; j_ < 16; ++j_) mx_ = fmaxf(mx_, fmaxf(sc0[j_], sc1[j_]));         \
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx_), __float_as_uint(mx_), false, false); \
        OUT_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                      \
    }
    FR_QK()
    FR_MASK(0)
    float mx_next;
    FR_ROWMAX(mx_next)
    ka0 ^= kTileBytes;                                       // -> K(1)
    __syncthreads();                                         // every wave is done with K(0): its slot takes K(2)

    for (int t = 0; t < nt; ++t) {
        // K(t+2) into the slot K(t) left, V(t+1) into the slot V(t-1) left: a whole tile of compute to land under
        FR_DMA_K(t + 2, t & 1)
        FR_DMA_V(t + 1, (t + 1) & 1)
        {
            const float m_cand = fmaxf(m_run, mx_next * c2);
            if (__any((m_cand - m_run) > rescale_thr<T>())) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
                m_run = m_cand;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < kDT; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[i][j] *= alpha;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sc0[j] = __builtin_amdgcn_exp2f(sc0[j] * c2 - m_run);
            sc1[j] = __builtin_amdgcn_exp2f(sc1[j] * c2 - m_run);
        }
        {
            float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                psum0 += sc0[j];
                psum1 += sc1[j];
            }
            l_run += psum0 + psum1;
        }
        vec8 pb[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pb[0][j] = (typename T::scalar)sc0[j];
            pb[1][j] = (typename T::scalar)sc0[8 + j];
            pb[2][j] = (typename T::scalar)sc1[j];
            pb[3][j] = (typename T::scalar)sc1[8 + j];
        }
        // S(t+1) for every t: past the last tile the K slot holds the zeros of an out-of-range DMA and the result is dropped
        FR_QK()
        FR_MASK(t + 1)
        // O^T += V(t)^T . P(t)^T: 4 key steps x (kDT / 2) d-tile pairs, the next fragment group's reads under this one's MFMAs
        {
            constexpr int kNG = 4 * (kDT / 2);                     // fragment groups: g = step * (kDT / 2) + pair
            FR_VISSUE(0, 0, 0)
#pragma unroll
            for (int gi = 0; gi < kNG; ++gi) {
                const int step = gi / (kDT / 2), pr = gi % (kDT / 2);
                if (gi + 1 < kNG) {
                    const int ns = (gi + 1) / (kDT / 2), np = (gi + 1) % (kDT / 2);
                    // (macro arguments must be literals for the immediate offset: dispatch on the step)
                    if ((gi & 1) == 0) {
                        if (ns == 0) { FR_VISSUE(0, np, 1) } else if (ns == 1) { FR_VISSUE(1, np, 1) }
                        else if (ns == 2) { FR_VISSUE(2, np, 1) } else { FR_VISSUE(3, np, 1) }
                        FR_VWAIT(4, 0)
                    } else {
                        if (ns == 0) { FR_VISSUE(0, np, 0) } else if (ns == 1) { FR_VISSUE(1, np, 0) }
                        else if (ns == 2) { FR_VISSUE(2, np, 0) } else { FR_VISSUE(3, np, 0) }
                        FR_VWAIT(4, 1)
                    }
                } else {
                    if ((gi & 1) == 0) { FR_VWAIT(0, 0) } else { FR_VWAIT(0, 1) }
                }
                const int bb = gi & 1;
                const s16x8_t va0 = __builtin_shufflevector(vf[bb][0], vf[bb][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8_t va1 = __builtin_shufflevector(vf[bb][2], vf[bb][3], 0, 1, 2, 3, 4, 5, 6, 7);
                o[2 * pr] = T::mfma32(__builtin_bit_cast(vec8, va0), pb[step], o[2 * pr]);
                o[2 * pr + 1] = T::mfma32(__builtin_bit_cast(vec8, va1), pb[step], o[2 * pr + 1]);
            }
        }
        FR_ROWMAX(mx_next)
        ka0 ^= kTileBytes;
        vl0 ^= kTileBytes;
        vh0 ^= kTileBytes;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef FR_KISSUE
#undef FR_KWAIT
#undef FR_VISSUE
#undef FR_VWAIT
#undef FR_DMA_K
#undef FR_DMA_V
#undef FR_QK
#undef FR_MASK
#undef FR_ROWMAX
#undef LDS_PTR

    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_run;
    if (qrow < p.lq) {
        uint16_t* orow = op + (int64_t)qrow * p.o_rs;
#pragma unroll
        for (int dt = 0; dt < kDT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * h;
                uint32_t w0 = (uint32_t)T::from_f32(o[dt][4 * g + 0] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * g + 1] * inv) << 16);
                uint32_t w1 = (uint32_t)T::from_f32(o[dt][4 * g + 2] * inv) |
                              ((uint32_t)T::from_f32(o[dt][4 * g + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(orow + d0) = make_uint2(w0, w1);
            }
        }
    }
}

template <typename T, int D>
int launch_attn_fr(AttnParams p, hipStream_t st) {
    constexpr int smem = 4 * kKV * D * 2;
    static FinoPerDeviceOnce once;
    if (int rc = fino_max_smem_once(once, reinterpret_cast<const void*>(&attn_fr_kernel<T, D>), smem, "fino_attn_fwd")) return rc;
    p.nqb = (p.lq + kFrQBlock - 1) / kFrQBlock;
    p.ws = nullptr; p.all_partial = 0;
    attn_virtual_heads(p.batch, p.heads, p.nqb, p.vsplit, p.nqb_v);
    const int groups = (p.batch * p.heads * p.vsplit + 7) / 8;
    p.full_x = groups * p.nqb_v; p.rem_x = 0; p.nwg = 0; p.per = 1;
    attn_fr_kernel<T, D><<<dim3((unsigned)(8 * p.full_x)), kFrWaves * 64, smem, st>>>(p);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

template <typename T, bool TAIL = false>
int launch_attn_ppw(AttnParams p, hipStream_t st) {
    constexpr int smem = 2 * kPdSlots * kKV * 128 * 2 + 32768;      // rings + 32 KiB for a group's output rows: all of the CU's 160 KiB
    static FinoPerDeviceOnce once;
    if (int rc = fino_max_smem_once(once, reinterpret_cast<const void*>(&attn_ppw_kernel<T, TAIL>), smem, "fino_attn_fwd")) return rc;
    p.ws = nullptr; p.all_partial = 0;
    const int64_t blocks = (int64_t)p.batch * p.heads * p.nqb;
    const int cap = fino_tune_get(FINO_TUNE_ATTN_WALK_GRID) > 0 ? fino_tune_get(FINO_TUNE_ATTN_WALK_GRID) : device_cus();
    const int grid = blocks < cap ? (int)blocks : cap;
    attn_ppw_kernel<T, TAIL><<<dim3((unsigned)grid), kWaves * 64, smem, st>>>(p);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

template <typename T, int D, int VAR>
int launch_attn_v(AttnParams p, int64_t ws_bytes, hipStream_t st) {
    constexpr int smem = 4 * kKV * D * 2;
    // the free-running kernel (two workgroups of 4 waves per CU): FINO_TUNE_ATTN_KERNEL = 3, and by default for the short key
    // sequences of the text cross-attention at head_dim 128 (whole blocks only: never for fino_attn_partial)
    {
        const int tk = fino_tune_get(FINO_TUNE_ATTN_KERNEL);
        // the walking kernel (one workgroup per CU over a run of q-blocks): short key sequences at head_dim 128 with at least
        // two blocks per CU (FINO_TUNE_ATTN_KERNEL = 6: wherever it can run)
        if constexpr (D == 128) {
            const int nt = (p.lk + kKV - 1) / kKV;
            const int64_t blocks = (int64_t)p.batch * p.heads * p.nqb;
            auto span = [&](int64_t bs, int64_t hs, int64_t rs, int64_t rows) {
                return (((int64_t)p.batch - 1) * bs + ((int64_t)p.heads - 1) * hs + (rows - 1 + 2 * kKV) * rs + 128) * 2;
            };
            const bool can = !p.all_partial && nt >= 2 && blocks < (1 << 24) && span(p.k_bs, p.k_hs, p.k_rs, p.lk) < (1ll << 31) &&
                             span(p.v_bs, p.v_hs, p.v_rs, p.lk) < (1ll << 31) && span(p.o_bs, p.o_hs, p.o_rs, p.lq) < (1ll << 31);
            if (can && (tk == 6 || (tk == 0 && VAR == 1 && blocks >= 2 * device_cus()))) return launch_attn_ppw<T>(p, st);
        }
        // the free-running kernel exists for head_dim 128 only (at head_dim 64 it lost to the 4-wave kernel and left in round 5)
        if constexpr (D == 128)
            if (!p.all_partial && (tk == 3 || ((tk == 0 || tk == 5 || tk == 7) && VAR == 1))) return launch_attn_fr<T, D>(p, st);
    }
    static FinoPerDeviceOnce once_b;
    if (int rc = fino_max_smem_once(once_b, reinterpret_cast<const void*>(&attn_pp_kernel<T, D, VAR>), smem, "fino_attn_fwd"))
        return rc;
    attn_virtual_heads(p.batch, p.heads, p.nqb, p.vsplit, p.nqb_v);
    const int groups = (p.batch * p.heads * p.vsplit + 7) / 8;
    SplitPlan sp{groups * p.nqb_v, 0, 0, 1};
    if (p.ws && !p.all_partial) {
        sp = plan_split(p.batch, p.heads, p.nqb, (p.lk + kKV - 1) / kKV);
        const int64_t need = (int64_t)8 * sp.nwg * 2 * partial_floats<D>() * 4;
        if (sp.rem_x > 0 && need > ws_bytes) {
            fino_set_error("fino_attn_fwd_ws: workspace %lld B < %lld B", (long long)ws_bytes, (long long)need);
            return FINO_ERR_ARG;
        }
    }
    if (p.all_partial) sp = SplitPlan{groups * p.nqb_v, 0, 0, 1};     // no tail split: every block is a partial anyway
    p.full_x = sp.full_x; p.rem_x = sp.rem_x; p.nwg = sp.nwg; p.per = sp.per;
    const dim3 grid((unsigned)(8 * (sp.full_x + sp.nwg)));
    // head_dim 128, long key sequences: the LDS-DMA-staged ping-pong kernel (round 4; same bits as the register-staged one,
    // -2.3 ... -2.8 % per launch inside the denoise step, profiles/r04_step_ab_ppd.txt).  The 4-wave kernel: on gaussian operands the 4-wave one is the faster standalone (B = 2, 24 heads, 12320^2:
    // 1186 vs 1166 TFLOP/s, 1253 with the MFMA fold; tools/attn_w4_ab.py).  Inside the denoise step, on the bench's own
    // activations, rocprof says 3237 / 3070 (fold) vs 3114 us per launch for the 8-wave kernel and tools/step_ab.py a tie
    // per step: the default stays the 8-wave kernel, the 4-wave one is FINO_TUNE_ATTN_KERNEL = 2.
    const int tune_k = fino_tune_get(FINO_TUNE_ATTN_KERNEL);
    // head_dim 64 (CogVideoX) is bound by the softmax's vector work, not by power: there the 4-wave kernel with the folded
    // scale wins inside the step too (B = 2, 48 heads, L = 19126: 952 vs 858 TFLOP/s; the CogVideoX-5B step 782 -> 741 ms
    // in bf16, 659 -> 619 with MXFP8 linears) and is the default whenever the caller folds the scale into q.
    const bool w4 = fino_attn_w4_supports(D, p.scale_log2) &&
                    (tune_k == 2 || (tune_k == 0 && D == 64 && p.scale_log2 == 1.0f && p.lk >= 2048));
    // 5: the round-3 policy (register-staged).  head_dim 64: the LDS-DMA-staged kernel by tune 4 (the policy there: below)
    const bool ppd = tune_k == 4 || (D == 128 && (tune_k == 0 || tune_k == 7) && VAR == 0);
    if (w4) {
        if (int rc = fino_attn_launch_w4(p, T::kId, D, st)) return rc;
    } else if (ppd) {
        constexpr int smem_d = 2 * kPdSlots * kKV * D * 2;
        static FinoPerDeviceOnce once_d;
        if (int rc = fino_max_smem_once(once_d, reinterpret_cast<const void*>(&attn_ppd_kernel<T, D, VAR>), smem_d, "fino_attn_fwd"))
            return rc;
        attn_ppd_kernel<T, D, VAR><<<grid, kWaves * 64, smem_d, st>>>(p);
    } else
        attn_pp_kernel<T, D, VAR><<<grid, kWaves * 64, smem, st>>>(p);
    FINO_LAUNCH_CHECK();
    if (sp.rem_x > 0) {
        attn_combine_kernel<T, D><<<dim3((unsigned)(8 * sp.rem_x), (unsigned)(D / 32)), kWaves * 64, 0, st>>>(p);
        FINO_LAUNCH_CHECK();
    }
    return FINO_OK;
}

template <typename T, int D>
int launch_attn(const AttnParams& p, int64_t ws_bytes, hipStream_t st) {
    return p.lk > 1024 ? launch_attn_v<T, D, 0>(p, ws_bytes, st) : launch_attn_v<T, D, 1>(p, ws_bytes, st);
}

}  // namespace

void fino_attn_plan_split(int batch, int heads, int nqb, int nt, int& full_x, int& rem_x, int& nwg, int& per) {
    const SplitPlan sp = plan_split(batch, heads, nqb, nt);
    full_x = sp.full_x; rem_x = sp.rem_x; nwg = sp.nwg; per = sp.per;
}

int fino_attn_launch_combine(const AttnParams& p, int dtype, int head_dim, hipStream_t st) {
    if (p.rem_x <= 0) return FINO_OK;
    const dim3 grid((unsigned)(8 * p.rem_x), (unsigned)(head_dim / 32));
    if (head_dim == 128) {
        if (dtype == FINO_BF16) attn_combine_kernel<BF16, 128><<<grid, kWaves * 64, 0, st>>>(p);
        else attn_combine_kernel<F16, 128><<<grid, kWaves * 64, 0, st>>>(p);
    } else {
        if (dtype == FINO_BF16) attn_combine_kernel<BF16, 64><<<grid, kWaves * 64, 0, st>>>(p);
        else attn_combine_kernel<F16, 64><<<grid, kWaves * 64, 0, st>>>(p);
    }
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int64_t fino_attn_workspace_bytes(int batch, int heads, int64_t lq, int64_t lk, int head_dim) {
    if (batch <= 0 || heads <= 0 || lq <= 0 || lk <= 0 || (head_dim != 64 && head_dim != 128)) return 0;
    const SplitPlan sp = plan_split(batch, heads, (int)((lq + kQBlock - 1) / kQBlock), (int)((lk + kKV - 1) / kKV));
    const int64_t pf = head_dim == 128 ? partial_floats<128>() : partial_floats<64>();
    return (int64_t)8 * sp.nwg * 2 * pf * 4;
}

static int64_t attn_partial_bytes(int batch, int heads, int64_t lq, int head_dim) {
    const int64_t pf = head_dim == 128 ? partial_floats<128>() : partial_floats<64>();
    return (int64_t)batch * heads * ((lq + kQBlock - 1) / kQBlock) * pf * 4;
}

static int attn_common(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                       int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs,
                       int64_t k_rs, int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs,
                       int64_t o_rs, int64_t o_hs, float scale, int dtype, void* workspace,
                       int64_t workspace_bytes, int all_partial, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_fwd: dtype %d", dtype);
    if (all_partial) {
        static const uint16_t dummy_o[8] __attribute__((aligned(16))) = {0};
        o = (void*)dummy_o;                                    // never written in this mode
        o_bs = o_rs = o_hs = 0;
        FINO_CHECK(head_dim == 128 || head_dim == 64, FINO_ERR_UNSUPPORTED, "fino_attn_partial: head_dim %d", head_dim);
        FINO_CHECK(workspace && workspace_bytes >= attn_partial_bytes(batch, heads, lq, head_dim), FINO_ERR_ARG,
                   "fino_attn_partial: workspace of %lld bytes needed",
                   (long long)attn_partial_bytes(batch, heads, lq, head_dim));
    }
    FINO_CHECK(q && k && v && o, FINO_ERR_ARG, "fino_attn_fwd: null pointer");
    FINO_CHECK(batch > 0 && heads > 0 && lq >= 0 && lk > 0, FINO_ERR_ARG,
               "fino_attn_fwd: bad shape B=%d H=%d Lq=%lld Lk=%lld", batch, heads, (long long)lq, (long long)lk);
    FINO_CHECK(head_dim == 128 || head_dim == 64, FINO_ERR_UNSUPPORTED,
               "fino_attn_fwd: head_dim %d not in {64,128}", head_dim);
    FINO_CHECK(lq < (1ll << 31) - 256 && lk < (1ll << 31) - 64, FINO_ERR_ARG, "fino_attn_fwd: sequence too long");
    FINO_CHECK(fino_aligned16(q) && fino_aligned16(k) && fino_aligned16(v) && fino_aligned16(o) && q_rs % 8 == 0 &&
                   k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 &&
                   v_hs % 8 == 0 && o_hs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 8 == 0,
               FINO_ERR_ARG, "fino_attn_fwd: pointers and strides must be 16-byte aligned");
    FINO_CHECK(scale > 0.f || scale == FINO_ATTN_SCALE_FOLDED, FINO_ERR_ARG,
               "fino_attn_fwd: scale must be > 0 (or FINO_ATTN_SCALE_FOLDED)");
    // the kernels address the K/V rows of one (batch, head) through a buffer resource: 32-bit byte count and offsets
    {
        const int64_t kmax = k_rs > v_rs ? k_rs : v_rs;
        FINO_CHECK(((lk - 1) * kmax + head_dim) * 2 < (1ll << 31), FINO_ERR_UNSUPPORTED,
                   "fino_attn_fwd: Lk=%lld keys x row stride %lld exceed the 2 GiB one (batch, head) K/V slice may span",
                   (long long)lk, (long long)kmax);
    }
    if (lq == 0) return FINO_OK;
    AttnParams p;
    p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = (uint16_t*)o;
    p.batch = batch; p.heads = heads; p.lq = (int)lq; p.lk = (int)lk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs;
    p.v_bs = v_bs; p.v_rs = v_rs; p.v_hs = v_hs; p.o_bs = o_bs; p.o_rs = o_rs; p.o_hs = o_hs;
    p.scale_log2 = scale == FINO_ATTN_SCALE_FOLDED ? 1.0f : scale * 1.4426950408889634f;
    p.nqb = (int)((lq + kQBlock - 1) / kQBlock);
    p.ws = (workspace && workspace_bytes > 0) ? (float*)workspace : nullptr;
    p.all_partial = all_partial;
    p.tail_n = 0;
    FINO_CHECK(((uintptr_t)workspace & 15) == 0, FINO_ERR_ARG, "fino_attn_fwd_ws: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t wb = workspace_bytes;
    if (dtype == FINO_BF16)
        return head_dim == 128 ? launch_attn<BF16, 128>(p, wb, st) : launch_attn<BF16, 64>(p, wb, st);
    return head_dim == 128 ? launch_attn<F16, 128>(p, wb, st) : launch_attn<F16, 64>(p, wb, st);
}

extern "C" int fino_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                                int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs,
                                int64_t k_rs, int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs,
                                int64_t o_rs, int64_t o_hs, float scale, int dtype, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    return attn_common(q, k, v, o, batch, heads, lq, lk, head_dim, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs,
                       o_bs, o_rs, o_hs, scale, dtype, workspace, workspace_bytes, 0, stream);
}

extern "C" int64_t fino_attn_partial_bytes(int batch, int heads, int64_t lq, int head_dim) {
    if (batch <= 0 || heads <= 0 || lq <= 0 || (head_dim != 64 && head_dim != 128)) return 0;
    return attn_partial_bytes(batch, heads, lq, head_dim);
}

extern "C" int fino_attn_partial(const void* q, const void* k, const void* v, int batch, int heads, int64_t lq, int64_t lk,
                                 int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs, int64_t k_rs,
                                 int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, float scale, int dtype,
                                 void* partial, int64_t partial_bytes, void* stream) {
    return attn_common(q, k, v, nullptr, batch, heads, lq, lk, head_dim, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs,
                       v_hs, 0, 0, 0, scale, dtype, partial, partial_bytes, 1, stream);
}

extern "C" int fino_attn_merge(void* o, int batch, int heads, int64_t lq, int head_dim, int64_t o_bs, int64_t o_rs,
                               int64_t o_hs, const void* part0, const void* part1, const void* part2, int dtype,
                               void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_merge: dtype %d", dtype);
    FINO_CHECK(o && part0 && batch > 0 && heads > 0 && lq > 0 && (head_dim == 64 || head_dim == 128), FINO_ERR_ARG,
               "fino_attn_merge: bad arguments");
    FINO_CHECK(fino_aligned16(o) && o_rs % 8 == 0 && o_hs % 8 == 0 && o_bs % 8 == 0, FINO_ERR_ARG,
               "fino_attn_merge: output must be 16-byte aligned");
    MergeParams mp;
    mp.part[0] = (const float*)part0; mp.part[1] = (const float*)part1; mp.part[2] = (const float*)part2;
    mp.n_parts = part1 ? (part2 ? 3 : 2) : 1;
    FINO_CHECK(part1 || !part2, FINO_ERR_ARG, "fino_attn_merge: part2 without part1");
    mp.o = (uint16_t*)o; mp.batch = batch; mp.heads = heads; mp.lq = (int)lq;
    mp.nqb = (int)((lq + kQBlock - 1) / kQBlock);
    mp.o_bs = o_bs; mp.o_rs = o_rs; mp.o_hs = o_hs;
    const dim3 grid((unsigned)(batch * heads * mp.nqb), (unsigned)(head_dim / 32));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FINO_BF16) {
        if (head_dim == 128) attn_merge_kernel<BF16, 128><<<grid, kWaves * 64, 0, st>>>(mp);
        else attn_merge_kernel<BF16, 64><<<grid, kWaves * 64, 0, st>>>(mp);
    } else {
        if (head_dim == 128) attn_merge_kernel<F16, 128><<<grid, kWaves * 64, 0, st>>>(mp);
        else attn_merge_kernel<F16, 64><<<grid, kWaves * 64, 0, st>>>(mp);
    }
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

extern "C" int fino_attn_fwd(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                             int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs,
                             int64_t k_rs, int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs,
                             int64_t o_rs, int64_t o_hs, float scale, int dtype, void* stream) {
    return fino_attn_fwd_ws(q, k, v, o, batch, heads, lq, lk, head_dim, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs,
                            v_hs, o_bs, o_rs, o_hs, scale, dtype, nullptr, 0, stream);
}

// Cross-attention over key sequences whose TAIL is one row repeated (a zero-padded prompt: every padding token yields the same
// K and V row, transformer_wan.py:108 via attn2 with the 512-token text of pipeline_wan_i2v_motion_FrameINO.py:235-238): batch
// element b attends to its first lk_b[b] rows of k / v, the last of which stands for tail_mult[b] identical keys; rows from
// lk_b[b] on (up to `lk`, the allocated rows every batch element has) are ignored.  Exactly
//     softmax(q.[K; k x M]^T) [V; v x M] = softmax(q.[K; k]^T + [0; ln M]) [V; v],
// so the result equals fino_attn_fwd on the expanded sequences up to the rounding of one bf16 weight.  Runs on the walking
// kernel (attn_ppw_kernel<T, true>): head_dim 128, batch <= 4, 64 < lk; FINO_ERR_UNSUPPORTED otherwise (fino_attn_tail_supported).
extern "C" int fino_attn_tail_supported(int batch, int heads, int64_t lq, int64_t lk, int head_dim) {
    return head_dim == 128 && batch >= 1 && batch <= 4 && heads > 0 && lq > 0 && lk > kKV && lk < (1 << 20) &&
           (int64_t)batch * heads * ((lq + kQBlock - 1) / kQBlock) < (1 << 24);
}

extern "C" int fino_attn_fwd_tail(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t lq,
                                  int64_t lk, int head_dim, int64_t q_bs, int64_t q_rs, int64_t q_hs, int64_t k_bs,
                                  int64_t k_rs, int64_t k_hs, int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs,
                                  int64_t o_rs, int64_t o_hs, float scale, int dtype, const int* lk_b,
                                  const float* tail_mult, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_attn_fwd_tail: dtype %d", dtype);
    FINO_CHECK(q && k && v && o && lk_b && tail_mult, FINO_ERR_ARG, "fino_attn_fwd_tail: null pointer");
    FINO_CHECK(fino_attn_tail_supported(batch, heads, lq, lk, head_dim), FINO_ERR_UNSUPPORTED,
               "fino_attn_fwd_tail: needs head_dim 128, batch <= 4, lk > 64 (got head_dim %d, batch %d, lk %lld)", head_dim,
               batch, (long long)lk);
    FINO_CHECK(fino_aligned16(q) && fino_aligned16(k) && fino_aligned16(v) && fino_aligned16(o) && q_rs % 8 == 0 &&
                   k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && q_hs % 8 == 0 && k_hs % 8 == 0 &&
                   v_hs % 8 == 0 && o_hs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 8 == 0,
               FINO_ERR_ARG, "fino_attn_fwd_tail: pointers and strides must be 16-byte aligned");
    FINO_CHECK(scale > 0.f || scale == FINO_ATTN_SCALE_FOLDED, FINO_ERR_ARG, "fino_attn_fwd_tail: scale");
    AttnParams p;
    p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = (uint16_t*)o;
    p.batch = batch; p.heads = heads; p.lq = (int)lq; p.lk = (int)lk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs;
    p.v_bs = v_bs; p.v_rs = v_rs; p.v_hs = v_hs; p.o_bs = o_bs; p.o_rs = o_rs; p.o_hs = o_hs;
    p.scale_log2 = scale == FINO_ATTN_SCALE_FOLDED ? 1.0f : scale * 1.4426950408889634f;
    p.nqb = (int)((lq + kQBlock - 1) / kQBlock);
    p.vsplit = 1; p.nqb_v = p.nqb; p.full_x = 0; p.rem_x = 0; p.nwg = 0; p.per = 1;
    p.ws = nullptr; p.all_partial = 0;
    // the walking kernel addresses each operand through ONE buffer resource over the whole tensor
    auto span = [&](int64_t bs, int64_t hs, int64_t rs, int64_t rows) {
        return (((int64_t)batch - 1) * bs + ((int64_t)heads - 1) * hs + (rows - 1 + 2 * kKV) * rs + 128) * 2;
    };
    FINO_CHECK(span(k_bs, k_hs, k_rs, lk) < (1ll << 31) && span(v_bs, v_hs, v_rs, lk) < (1ll << 31) &&
                   span(o_bs, o_hs, o_rs, lq) < (1ll << 31),
               FINO_ERR_UNSUPPORTED, "fino_attn_fwd_tail: an operand spans more than 2 GiB");
    p.tail_n = batch;
    for (int b = 0; b < 4; ++b) {
        const int bb = b < batch ? b : batch - 1;
        FINO_CHECK(lk_b[bb] >= 1 && lk_b[bb] <= lk && tail_mult[bb] >= 1.0f, FINO_ERR_ARG,
                   "fino_attn_fwd_tail: lk_b[%d] = %d (of %lld), tail_mult %g", bb, lk_b[bb], (long long)lk, (double)tail_mult[bb]);
        p.tail_lk[b] = lk_b[bb];
        p.tail_bias[b] = log2f(tail_mult[bb]) / p.scale_log2;
    }
    if (lq == 0) return FINO_OK;
    hipStream_t st = (hipStream_t)stream;
    return dtype == FINO_BF16 ? launch_attn_ppw<BF16, true>(p, st) : launch_attn_ppw<F16, true>(p, st);
}
