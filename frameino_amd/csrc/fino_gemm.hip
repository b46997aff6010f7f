// C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias): the nn.Linear call sites of the DiT (QKV / out / FFN / patch-embed /
// proj_out), bf16|fp16 operands, fp32 accumulate, fused bias + GELU-tanh + (gated) residual epilogues.
//
// CDNA4 design: 256x256 output tile per 8-wave workgroup (2(M) x 4(N) waves, 128x64 per wave), BK = 64,
// v_mfma_f32_16x16x32 in the swapped orientation D = W_frag . A_frag^T so that every lane ends up with 4
// consecutive output columns of one row (8-byte packed values).  Operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a double-buffered LDS image whose 16-byte chunks are
// XOR-swizzled on the SOURCE address (LDS-DMA writes are lane-linear) and on the ds_read_b128 address, which makes
// the fragment reads bank-conflict free.  The next K-tile's DMA is issued before the current tile's MFMAs.
// The epilogue stages the bf16 tile through LDS so that global stores (and the fused residual reads) are whole
// 512-byte rows.  Workgroup ids are remapped per XCD (8 L2s) in 4-tile-high groups for operand reuse in L2.
// A generic variant (register-staged, predicated, zero-filled) covers ragged K and is used for tiny shapes.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "fino_gemm_common.h"

using namespace fino_gemm_ns;

#ifdef FINO_GEMM_STAMP
extern "C" int fino_gemm_debug_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fino_gemm_dbg), sizeof(unsigned long long) * 64);
}
#endif

namespace {

template <typename T, int EPI, bool GENERIC, bool CONV>
__global__ __launch_bounds__(kThreads, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename T::vec8 vec8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2;  // 0..1
    const int wn = wave & 3;   // 0..3

    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // ---- staging: 4 rounds per operand; wave-instruction = 8 rows x 128 B ----
    // thread -> (row_in_round = wave*8 + lane>>3, phys chunk = lane&7); LDS byte = round*8192 + tid*16
    const int srow = wave * 8 + (lane >> 3);
    const int sch = lane & 7;
    const uint16_t* a_src[4];
    const uint16_t* w_src[4];
    bool a_ok[4], w_ok[4];
    int sk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 64 + srow;
        const int lch = sch ^ swz(row);  // logical k-chunk held at this physical slot
        sk[j] = lch * 8;
        int64_t gm = m0 + row, gn = n0 + row;
        a_ok[j] = gm < p.m;
        w_ok[j] = gn < p.n;
        if (gm >= p.m) gm = p.m - 1;
        if (gn >= p.n) gn = p.n - 1;
        a_src[j] = p.a + gm * p.lda + sk[j];
        w_src[j] = p.w + gn * p.ldw + sk[j];
    }
    // CONV: output position of each staged row
    int pos_t[4], pos_h[4], pos_w[4];
    if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t gm = m0 + j * 64 + srow;
            if (gm >= p.m) gm = p.m - 1;
            const int hw = p.ho * p.wo;
            pos_t[j] = (int)(gm / hw);
            const int rem = (int)(gm - (int64_t)pos_t[j] * hw);
            pos_h[j] = rem / p.wo;
            pos_w[j] = rem - pos_h[j] * p.wo;
        }
    }
    auto stage = [&](int buf, int kt) {
        char* ab = smem + buf * kStageBytes;
        char* wb = ab + kTileBytes;
        const int64_t k0 = (int64_t)kt * BK;
        if constexpr (CONV) {
            const int tap = kt / p.cin_chunks;
            const int c0 = (kt - tap * p.cin_chunks) * BK;
            const int dw = tap % p.kw;
            const int dh = (tap / p.kw) % p.kh;
            const int dt = tap / (p.kw * p.kh);
            const int hlim = p.up ? 2 * p.hi : p.hi, wlim = p.up ? 2 * p.wi : p.wi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ti = pos_t[j] * p.st + dt - p.pt;
                int hi = pos_h[j] * p.sh + dh - p.ph;
                int wi = pos_w[j] * p.sw + dw - p.pw;
                const bool ok = a_ok[j] && ti >= 0 && ti < p.ti && hi >= 0 && hi < hlim && wi >= 0 && wi < wlim;
                if (p.up) { hi >>= 1; wi >>= 1; }
                const uint16_t* src = ok ? p.a + (((int64_t)ti * p.hi + hi) * p.wi + wi) * p.lda + c0 + sk[j]
                                         : p.zero_page + sk[j];
                __builtin_amdgcn_global_load_lds((const FINO_GLB void*)src,
                                                 (FINO_LDS void*)(ab + j * 8192 + wave * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const FINO_GLB void*)(w_src[j] + k0),
                                                 (FINO_LDS void*)(wb + j * 8192 + wave * 1024), 16, 0, 0);
            }
        } else if constexpr (!GENERIC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_global_load_lds((const FINO_GLB void*)(a_src[j] + k0),
                                                 (FINO_LDS void*)(ab + j * 8192 + wave * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const FINO_GLB void*)(w_src[j] + k0),
                                                 (FINO_LDS void*)(wb + j * 8192 + wave * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool kin = k0 + sk[j] < p.k;
                uint4 va = make_uint4(0, 0, 0, 0), vw = make_uint4(0, 0, 0, 0);
                if (kin && a_ok[j]) va = *reinterpret_cast<const uint4*>(a_src[j] + k0);
                if (kin && w_ok[j]) vw = *reinterpret_cast<const uint4*>(w_src[j] + k0);
                *reinterpret_cast<uint4*>(ab + j * 8192 + tid * 16) = va;
                *reinterpret_cast<uint4*>(wb + j * 8192 + tid * 16) = vw;
            }
        }
    };

    // one LDS-DMA piece (one wave-instruction = 8 tile rows) of K-tile `kt`: pieces 0..3 = A rounds, 4..7 = W rounds
    auto stage_piece = [&](int buf, int kt, int piece) {
        char* ab = smem + buf * kStageBytes;
        char* wb = ab + kTileBytes;
        const int64_t k0 = (int64_t)kt * BK;
        const int j = piece & 3;
        if (piece >= 4) {
            __builtin_amdgcn_global_load_lds((const FINO_GLB void*)(w_src[j] + k0),
                                             (FINO_LDS void*)(wb + j * 8192 + wave * 1024), 16, 0, 0);
        } else if constexpr (CONV) {
            const int tap = kt / p.cin_chunks;
            const int c0 = (kt - tap * p.cin_chunks) * BK;
            const int dw = tap % p.kw;
            const int dh = (tap / p.kw) % p.kh;
            const int dt = tap / (p.kw * p.kh);
            const int hlim = p.up ? 2 * p.hi : p.hi, wlim = p.up ? 2 * p.wi : p.wi;
            const int ti = pos_t[j] * p.st + dt - p.pt;
            int hi = pos_h[j] * p.sh + dh - p.ph;
            int wi = pos_w[j] * p.sw + dw - p.pw;
            const bool ok = a_ok[j] && ti >= 0 && ti < p.ti && hi >= 0 && hi < hlim && wi >= 0 && wi < wlim;
            if (p.up) { hi >>= 1; wi >>= 1; }
            const uint16_t* src = ok ? p.a + (((int64_t)ti * p.hi + hi) * p.wi + wi) * p.lda + c0 + sk[j]
                                     : p.zero_page + sk[j];
            __builtin_amdgcn_global_load_lds((const FINO_GLB void*)src, (FINO_LDS void*)(ab + j * 8192 + wave * 1024),
                                             16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((const FINO_GLB void*)(a_src[j] + k0),
                                             (FINO_LDS void*)(ab + j * 8192 + wave * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addressing: row = base + 16*tile + (lane&15), chunk = (4*kk + (lane>>4)) ^ swz(row) ----
    const int frow = lane & 15;
    const int pch0 = (lane >> 4) ^ (frow >> 1);  // physical chunk for kk = 0; kk = 1 flips bit 2
    const int a_base = (wm * 128 + frow) * 128;
    const int w_base = kTileBytes + (wn * 64 + frow) * 128;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((p.k + BK - 1) / BK);
#define LOAD_FRAGS(SB_, KK_, WF_, AF_)                                                                           \
    {                                                                                                            \
        const int pch_ = (pch0 ^ ((KK_) << 2)) << 4;                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                         \
            WF_[j_] = *reinterpret_cast<const u32x4_t*>((SB_) + w_base + j_ * 2048 + pch_);                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                         \
            AF_[i_] = *reinterpret_cast<const u32x4_t*>((SB_) + a_base + i_ * 2048 + pch_);                      \
    }
#define MFMA_BLOCK(WF_, AF_)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                         \
            acc[i_][j_] = T::mfma16(__builtin_bit_cast(vec8, WF_[j_]), __builtin_bit_cast(vec8, AF_[i_]), acc[i_][j_]);

    if constexpr (GENERIC) {
        // simple loop (register-staged, predicated): ragged K / tiny shapes
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const char* sb = smem + cur * kStageBytes;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4_t wf[4], af[8];
                LOAD_FRAGS(sb, kk, wf, af)
                MFMA_BLOCK(wf, af)
            }
            __syncthreads();
        }
    } else {
        // Software-pipelined loop.  A K-tile is two k-steps (kk = 0, 1) of 32 MFMAs per wave; fragments of the NEXT
        // k-step are fetched from LDS one row-group at a time into the registers the current k-step has just finished
        // with (A fragment i is dead after its 4 MFMAs), so ds_reads always fly under MFMAs and the register peak is
        // ~1.3 fragment sets instead of 2.  ONE barrier per K-tile, between the two MFMA blocks: it (1) publishes
        // K-tile t+1 (LDS-DMA issued one iteration earlier, so the vmcnt(0) in front of it is free) and (2) retires
        // every wave's reads of K-tile t, so the DMA of K-tile t+2 may overwrite that buffer.
        u32x4_t wfa[4], wfb[4], af[8], an;   // an: the A fragment in flight for the next k-step
#define LD_W(SB_, KK_, WF_)                                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                              \
        WF_[j_] = *reinterpret_cast<const u32x4_t*>((SB_) + w_base + j_ * 2048 + ((pch0 ^ ((KK_) << 2)) << 4));
#define LD_A(SB_, KK_, I_) (*reinterpret_cast<const u32x4_t*>((SB_) + a_base + (I_) * 2048 + ((pch0 ^ ((KK_) << 2)) << 4)))
        // one k-step: 8 groups of {4 MFMAs on af[i]; af[i] <- fragment i of the next k-step (loaded one group early)}
#define K_STEP(WF_, SBN_, KKN_, HAS_NEXT_, DMA_)                                                                  \
    {                                                                                                             \
        if (HAS_NEXT_) an = LD_A(SBN_, KKN_, 0);                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                        \
            if (DMA_) stage_piece(cur, kt + 2, i_); /* LDS-DMA issue spread over the MFMA groups */                \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                      \
                acc[i_][j_] = T::mfma16(__builtin_bit_cast(vec8, WF_[j_]), __builtin_bit_cast(vec8, af[i_]),      \
                                        acc[i_][j_]);                                                             \
            if (HAS_NEXT_) {                                                                                      \
                af[i_] = an;                                                                                      \
                if (i_ < 7) an = LD_A(SBN_, KKN_, i_ + 1);                                                        \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
    }
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        LD_W(smem, 0, wfa)
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = LD_A(smem, 0, i);
#ifdef FINO_GEMM_STAMP
        unsigned long long t0, t1, t2, t3, t4, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#endif
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const char* sb = smem + cur * kStageBytes;
            const char* sbn = smem + (cur ^ 1) * kStageBytes;
#ifdef FINO_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0); STAMP(t0) __builtin_amdgcn_sched_barrier(0);
#endif
            LD_W(sb, 1, wfb)
            K_STEP(wfa, sb, 1, true, false)
#ifdef FINO_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0); STAMP(t1) __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0); STAMP(t2) __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0); STAMP(t3) __builtin_amdgcn_sched_barrier(0);
#else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
            // (on the last K-tile these fragment loads read the stale other buffer: valid LDS, values never used)
            LD_W(sbn, 0, wfa)
            const bool dma = kt + 2 < nk;          // wave-uniform: a scalar branch around each DMA piece
            K_STEP(wfb, sbn, 0, true, dma)
#ifdef FINO_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0); STAMP(t4) __builtin_amdgcn_sched_barrier(0);
            acc0 += t1 - t0; acc1 += t2 - t1; acc2 += t3 - t2; acc3 += t4 - t3;
#endif
        }
#ifdef FINO_GEMM_STAMP
        if (blockIdx.x == 17 && lane == 0) {
            fino_gemm_dbg[wave * 8 + 0] = acc0; fino_gemm_dbg[wave * 8 + 1] = acc1;
            fino_gemm_dbg[wave * 8 + 2] = acc2; fino_gemm_dbg[wave * 8 + 3] = acc3;
            fino_gemm_dbg[wave * 8 + 4] = (unsigned long long)nk;
        }
#endif
        __syncthreads();
    }
#undef LD_W
#undef LD_A
#undef K_STEP
#undef LOAD_FRAGS
#undef MFMA_BLOCK

    gemm_epilogue<T, EPI>(acc, p, smem, m0, n0, tid, lane, wm, wn);
}

// ---- ping-pong main loop (aligned K, plain GEMM) ------------------------------------------------------------------
// The two waves that share a SIMD (wave w of rows 0-127 = group 0, wave w+4 of rows 128-255 = group 1) run HALF A
// K-TILE OUT OF PHASE: while one issues its 64 MFMAs of K-tile t back to back from registers (no LDS access in that
// phase), the other fetches its 24 ds_read_b128 fragments of its next K-tile and issues the LDS-DMA of a later one.
// Phases are separated by s_barrier, so the matrix pipe of each SIMD always has exactly one wave feeding it and the
// age-based arbitration between co-resident waves (the older wave starving the younger, then idling at the barrier:
// 2840 cycles per K-tile measured on the one-barrier loop against 2048 of MFMA) has nothing to arbitrate.
//   phase 2t   : G0 LOAD(t)  + DMA {A rows 0-127, W} of tile t+1     | G1 COMPUTE(t-1)
//   phase 2t+1 : G0 COMPUTE(t)                                        | G1 LOAD(t) + DMA {A rows 128-255} of tile t+1
// LDS: two 64-KiB stages.  Stage (t+1)%2 held tile t-1: its A half of a group is dead after that group's LOAD(t-1),
// its W after G1's LOAD(t-1) (phase 2t-1), so every DMA above starts in a free region and has >= one whole phase to land.
// (Moving 2 or 4 of G0's eight W pieces per wave into G1's COMPUTE phase was measured: no difference.)
//
// CONV = implicit-GEMM convolution on the same loop (Wan VAE, fino_conv3d): row m of A is output position (t, h, w) of a
// channels-last activation, K runs tap-major / channel-minor.  Per lane the gather address of a piece changes only when
// the K-tile crosses into the next tap (every cin_chunks tiles): the 4 offsets are recomputed there (a dozen VALU ops
// each, amortised over >= 4 K-tiles in every Wan conv), the channel advance inside a tap rides in the scalar offset, and
// taps that fall into the padding get an offset beyond the buffer resource's num_records, for which the hardware
// writes zeros into LDS (no zero page, no branch).  The resource is re-based per workgroup to the first input frame
// its rows can touch, so the offsets fit 32 bits on tensors of any size (fino_conv3d checks the per-tile span).
// The main loop of one output tile over K-tiles [kb, kb + nk): everything between the tile coordinates and the
// accumulators (the whole-tile kernel passes kb = 0, nk = K / 64; the stream-K kernel a key... a K range).
// MI = 16-row fragments per wave: the tile is BM_ = 32 * MI rows high, group g (waves 4g .. 4g+3) owns rows [16 MI g,
// 16 MI (g + 1)); 8 = the 256 x 256 tile.  Lower tiles exist for ROW COUNTS, not for speed per FLOP: 3080 rows (a 4-way
// token shard) are 13 x 12 = 156 tiles of 256 x 256 on 256 CUs -- 100 CUs idle for the whole GEMM -- but 20 x 12 = 240
// tiles of 160 x 256 in one round of 0.625 of the time (fino_gemm's planner, plan_tiles below).  A tiles are staged in
// MI pieces of 32 rows: group 0 issues the first ceil(MI / 2) of them (they cover every row IT reads: its loads one phase
// later must find them landed, and only its own vmcnt wait can vouch for that), group 1 the rest.
// byte offset of K-tile kt inside a row of a K-blocked A (GemmParams::a_tpb): block j = kt / a_tpb starts at j * a_blk_elems
__device__ __forceinline__ int ablk_koff(const GemmParams& p, int kt) {
    const int b = (kt * p.a_inv) >> 16;              // K block
    const int j = (b * p.a_ginv) >> 16;              // peer; g = b - j * a_groups: head group
    return (b - j * p.a_groups) * (int)(p.a_grp_elems * 2) + j * (int)(p.a_blk_elems * 2) + (kt - b * p.a_tpb) * (BK * 2);
}

template <typename T, bool CONV, int MI, bool ABLK = false>
__device__ __forceinline__ void pp_mainloop(const GemmParams& p, char* smem, const int64_t m0, const int64_t n0,
                                            const int kb, const int nk, f32x4_t (&acc)[MI][4], const int tid,
                                            const int lane, const int wave, const int wm, const int wn) {
    typedef typename T::vec8 vec8;
    static_assert(MI >= 2 && MI <= 8, "tile height 64 .. 256");
    static_assert(!CONV || MI == 8, "the implicit-GEMM gather is built for 256-row tiles");
    enum : int { NA0 = (MI + 1) / 2, NA1 = MI - (MI + 1) / 2 };      // A pieces (32 rows each) issued by group 0 / group 1

    // ---- LDS-DMA pieces: 8 tile rows x 128 B per wave-instruction; swizzle on the source chunk ----
    // buffer_load_dwordx4 ... lds: the per-lane part of the address is ONE 32-bit byte offset per piece (row base +
    // swizzled chunk, rows past the edge clamped to the last one), the K advance rides in the scalar offset, so the
    // loop carries 12 address registers and no per-piece vector arithmetic (launch checks the operands span < 4 GiB).
    const int r8 = lane >> 3;
    const int sk = ((lane & 7) ^ ((4 * (wn & 1) + (lane >> 4)) & 7)) * 8;   // = (phys chunk ^ swz(row)) * 8 elements
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, 0, (int)(((p.n - 1) * p.ldw + p.k) * 2), 0x00020000);
    // my A piece qi (0 .. NA0-1 or NA1-1) is tile piece aq0 + qi: tile rows (aq0 + qi)*32 + wn*8 + r8;  W piece q (0..7):
    // rows q*32 + wn*8 + r8
    const int aq0 = wm == 0 ? 0 : NA0;
    uint32_t a_off[4], w_off[8];                 // (NA0 <= 4 used; a fixed bound: a dependent one indexed by the unrolled
                                                 // prologue loop below makes the host pass reject the template)
    // CONV state: packed output position of my 4 A rows; the tap (dt, dh, dw) and channel chunk of the NEXT tile to stage
    uint32_t pos[CONV ? 4 : 1];
    int ck = 0, tap_t = 0, tap_h = 0, tap_w = 0, t_first = 0, t_in_first = 0;
    // split-bf16 convolution (GemmParams::a_cc > 0): the product (segment) and the chunk inside it of the NEXT tile to stage, and
    // that tile's byte offset inside an A row -- plane {0,0,1}[seg] or {0,0,0,1,1,2}[seg] of the side-by-side planes
    int cseg = 0, seg = 0, a_koff = 0;
    const uint16_t* a_base_ptr = p.a;
    int64_t a_bytes = ((p.m - 1) * p.lda + p.k) * 2;
    if constexpr (ABLK)
        a_bytes = ((p.a_groups - 1) * p.a_grp_elems + (p.k / (BK * p.a_tpb) / p.a_groups - 1) * p.a_blk_elems +
                   (p.m - 1) * p.lda + BK * p.a_tpb) * 2;
    if constexpr (CONV) {
        const int hw = p.ho * p.wo;
        t_first = (int)(m0 / hw);
        t_in_first = t_first * p.st - p.pt;
        t_in_first = t_in_first < 0 ? 0 : t_in_first;
        const int64_t frame = (int64_t)p.hi * p.wi * p.lda;
        a_base_ptr = p.a + (int64_t)t_in_first * frame;
        a_bytes = (int64_t)(p.ti - t_in_first) * frame * 2;
        a_bytes = a_bytes > 0x7fffffffll ? 0x7fffffffll : a_bytes;
#pragma unroll
        for (int q = 0; q < NA0; ++q) {
            int64_t gm = m0 + (aq0 + q) * 32 + wn * 8 + r8;
            gm = gm < p.m ? gm : p.m - 1;
            const int t = (int)(gm / hw);
            const int rem = (int)(gm - (int64_t)t * hw);
            const int h = rem / p.wo;
            pos[q] = ((uint32_t)(t - t_first) << 24) | ((uint32_t)h << 12) | (uint32_t)(rem - h * p.wo);
        }
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_base_ptr, 0, (int)a_bytes, 0x00020000);
    // offsets of my A pieces for the tap (tap_t, tap_h, tap_w); padding taps -> beyond num_records -> zeros
    auto conv_offsets = [&]() {
        const int hlim = p.up ? 2 * p.hi : p.hi, wlim = p.up ? 2 * p.wi : p.wi;
#pragma unroll
        for (int q = 0; q < NA0; ++q) {
            const int ti = (t_first + (int)(pos[CONV ? q : 0] >> 24)) * p.st + tap_t - p.pt;
            int hi = (int)((pos[CONV ? q : 0] >> 12) & 0xfffu) * p.sh + tap_h - p.ph;
            int wi = (int)(pos[CONV ? q : 0] & 0xfffu) * p.sw + tap_w - p.pw;
            const bool ok = (unsigned)ti < (unsigned)p.ti && (unsigned)hi < (unsigned)hlim && (unsigned)wi < (unsigned)wlim;
            if (p.up) { hi >>= 1; wi >>= 1; }
            const uint32_t off = (uint32_t)(((((ti - t_in_first) * p.hi + hi) * p.wi + wi) * (int)p.lda + sk) * 2);
            a_off[q] = ok ? off : 0x80000000u;
        }
    };
    // state -> the K-tile after the one just staged
    auto conv_next = [&]() {
        if (p.a_cc > 0 && ++cseg == p.a_cc) { cseg = 0; ++seg; }
        if (++ck == p.cin_chunks) {
            ck = 0; cseg = 0; seg = 0;
            if (++tap_w == p.kw) {
                tap_w = 0;
                if (++tap_h == p.kh) { tap_h = 0; ++tap_t; }
            }
            conv_offsets();
        }
        if (p.a_cc > 0) {
            const int plane = p.a_nplanes == 3 ? (seg >= 3) + (seg >= 5) : (seg >= 2);
            a_koff = (plane * p.a_cc + cseg) * (BK * 2);
        } else {
            a_koff = ck * (BK * 2);
        }
    };
    if constexpr (CONV) {
        conv_offsets();
    } else {
#pragma unroll
        for (int q = 0; q < NA0; ++q) {
            int64_t gm = m0 + (aq0 + q) * 32 + wn * 8 + r8;
            gm = gm < p.m ? gm : p.m - 1;
            a_off[q] = (uint32_t)((gm * p.lda + sk) * 2);
        }
    }
    // W pieces: CONV keeps ONE per-lane offset (piece 0) and moves the 32-row piece stride into the scalar offset
    // (7 registers the gather state needs); pieces wholly past N (fino_conv3d requires N % 32 == 0) are skipped --
    // their LDS rows only feed output columns >= N, which the epilogue never stores.
    const int w_piece_bytes = (int)(32 * p.ldw * 2);
    const int w_pieces = CONV ? (int)((p.n - n0 + 31) / 32) : 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // (CONV uses w_off[0] only; the rest is dead code there)
        int64_t gn = n0 + q * 32 + wn * 8 + r8;
        gn = gn < p.n ? gn : p.n - 1;
        w_off[q] = (uint32_t)((gn * p.ldw + sk) * 2);
    }
    // my A piece QI_ (a compile-time index; group 1 has NA1 <= NA0 of them)
#ifndef GP_DMA_AUX
#define GP_DMA_AUX 0       /* A/B knob: cache-policy bits of the operand DMAs (1 = sc0, 2 = nt, 16 = sc1) */
#endif
#define PP_DMA_A(STAGE_, KT_, QI_)                                                                                \
    if ((QI_) < NA1 || wm == 0)                                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                 \
            a_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageBytes + ((aq0 + (QI_)) * 32 + wn * 8) * 128), 16,    \
            a_off[QI_], CONV ? a_koff : (ABLK ? ablk_koff(p, kb + (KT_)) : (kb + (KT_)) * (BK * 2)), 0, GP_DMA_AUX);
#define PP_DMA_W(STAGE_, KT_, Q_)                                                                                 \
    if (!CONV || (Q_) < w_pieces)                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                 \
            w_rsrc, (FINO_LDS void*)(smem + (STAGE_) * kStageBytes + kTileBytes + ((Q_) * 32 + wn * 8) * 128), 16, \
            w_off[CONV ? 0 : (Q_)], (kb + (KT_)) * (BK * 2) + (CONV ? (Q_) * w_piece_bytes : 0), 0, GP_DMA_AUX);

    const int frow = lane & 15;
    const int pch0 = (lane >> 4) ^ (frow >> 1);
    const int a_base = (wm * (16 * MI) + frow) * 128;
    const int w_base = kTileBytes + (wn * 64 + frow) * 128;

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0 and 1 whole (each group its A pieces and half of W)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < NA0) { PP_DMA_A(0, 0, q) }
        if (wm == 0) { PP_DMA_W(0, 0, q) } else { PP_DMA_W(0, 0, 4 + q) }
    }
    if constexpr (CONV) conv_next();
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < NA0) { PP_DMA_A(1, 1, q) }
            if (wm == 0) { PP_DMA_W(1, 1, q) } else { PP_DMA_W(1, 1, 4 + q) }
        }
        if constexpr (CONV) conv_next();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();       // phase 0: group 1 has nothing to compute yet

    u32x4_t af[2][MI], wf[2][4];
#ifdef FINO_GEMM_STAMP
    unsigned long long t0, t1, t2, t3, t4, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#define PSTAMP(V_) { __builtin_amdgcn_sched_barrier(0); STAMP(V_) __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(V_)
#endif
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const char* sb = smem + cur * kStageBytes;
        PSTAMP(t0)
        // ---------------- LOAD(t): all fragments of my (16 MI)x64 (A) and 64x64 (W) operand blocks ----------------
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pch = (pch0 ^ (kk << 2)) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[kk][j] = *reinterpret_cast<const u32x4_t*>(sb + w_base + j * 2048 + pch);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[kk][i] = *reinterpret_cast<const u32x4_t*>(sb + a_base + i * 2048 + pch);
        }
        if (t >= 1 && t + 1 < nk) {                  // tile t+1 into the stage tile t-1 has left
#pragma unroll
            for (int q = 0; q < NA0; ++q) PP_DMA_A(cur ^ 1, t + 1, q)
            if (wm == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { PP_DMA_W(cur ^ 1, t + 1, q) }
            }
            if constexpr (CONV) conv_next();
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PSTAMP(t1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(t2)
        // ---------------- COMPUTE(t): 8 MI MFMAs from registers ----------------
#ifndef GP_ORDER
#define GP_ORDER 3         /* A/B knob (same bits): 0 = the W fragment (MFMA A operand) changes with every MFMA, the activation fragment
                              (B operand) every fourth; 1 = the activation fragment changes with every MFMA, the W fragment every MI-th;
                              2 / 3 = serpentine walks in which exactly ONE operand changes between consecutive MFMAs.  Six block GEMMs
                              at M = 24640 (profiles/r04_gemm_mfma_order.txt): 5500 / 5480 / 5460 / 5454 us -- fewer operand-register
                              switches per MFMA draw less power, and the step is power-capped */
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#if GP_ORDER == 0
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = T::mfma16(__builtin_bit_cast(vec8, wf[kk][j]), __builtin_bit_cast(vec8, af[kk][i]),
                                          acc[i][j]);
            }
#elif GP_ORDER == 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i][j] = T::mfma16(__builtin_bit_cast(vec8, wf[kk][j]), __builtin_bit_cast(vec8, af[kk][i]),
                                          acc[i][j]);
            }
#elif GP_ORDER == 2        /* serpentine over i inside j: exactly ONE operand changes between consecutive MFMAs */
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int ii = 0; ii < MI; ++ii) {
                    const int i = ((j + 4 * kk) & 1) ? MI - 1 - ii : ii;
                    acc[i][j] = T::mfma16(__builtin_bit_cast(vec8, wf[kk][j]), __builtin_bit_cast(vec8, af[kk][i]),
                                          acc[i][j]);
                }
            }
#else                      /* 3: serpentine over j inside i */
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = ((i + MI * kk) & 1) ? 3 - jj : jj;
                    acc[i][j] = T::mfma16(__builtin_bit_cast(vec8, wf[kk][j]), __builtin_bit_cast(vec8, af[kk][i]),
                                          acc[i][j]);
                }
            }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my pieces of tile t+1, issued one phase ago
        PSTAMP(t3)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef FINO_GEMM_STAMP
        STAMP(t4)
        acc0 += t1 - t0; acc1 += t2 - t1; acc2 += t3 - t2; acc3 += t4 - t3;
#endif
    }
#ifdef FINO_GEMM_STAMP
    if (blockIdx.x == 17 && lane == 0) {
        fino_gemm_dbg[wave * 8 + 0] = acc0; fino_gemm_dbg[wave * 8 + 1] = acc1;
        fino_gemm_dbg[wave * 8 + 2] = acc2; fino_gemm_dbg[wave * 8 + 3] = acc3;
        fino_gemm_dbg[wave * 8 + 4] = (unsigned long long)nk;
    }
#endif
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PP_DMA_A
#undef PP_DMA_W
}

// Round 4 measured two ways of taking the per-tile fixed cost (prologue ~10k cycles + epilogue 11k .. 44k against ~113k of
// loop at K = 3072; profiles/r04_gemm_tile_stamps_*.txt) out of the critical path, and neither is here (git history has
// both, DESIGN.md section 4.3 the numbers): PERSISTENT workgroups that chain their tiles -- the next tile's first two
// K-tiles issued in front of the epilogue, which then runs in four slices through the 32 KiB of LDS above the operand
// stages, the chained prologue waiting with a counted vmcnt so that the previous tile's stores stay in flight -- took 0.8 %
// off the layer's GEMMs and nothing off the denoise step; an L2 warm-up touch of the next tile's A rows took nothing off
// either.  What did pay is inside the epilogue (fino_gemm_common.h: order of issue).
template <typename T, int EPI, bool CONV = false, int MI = 8, bool ABLK = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_pp_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;  // group
    const int wn = wave & 3;
    int tm, tn;
    tile_coords(p, tm, tn);
    const int64_t m0 = (int64_t)tm * (32 * MI), n0 = (int64_t)tn * BN;
    int nk = (int)(p.k / BK);
    // A RAGGED LAST TILE ROW runs as a lower tile (round 6).  M = 24640 ends in a tile row of 64 rows, a 4-way token shard's 3080
    // rows in one of 8: as a 256-row tile it issues every MFMA of 256 rows -- on a power-capped chip joules spent on rows that do not
    // exist (1/13 of the GEMM's matrix work on that shard).  The loop and the epilogue are templates on the tile height, so the
    // workgroup simply runs the 64- or 128-row instantiation on its rows: same K walk, same MFMA order per element, bit-identical
    // results (tests/test_gemm_tiles_gpu.py), a quarter / half of the matrix instructions and of the A staging.
#ifndef GP_RAGGED
#define GP_RAGGED 1        /* A/B knob (same bits): 0 = the ragged last tile row as a whole 256-row tile (rounds 1 - 5) */
#endif
    if constexpr (GP_RAGGED && MI == 8 && !CONV && EPI != FINO_EPI_F32 && EPI != FINO_EPI_F32_RESIDUAL) {
        const int64_t rows = p.m - m0;                      // (block-uniform)
        if (rows <= 64) {
            f32x4_t acc2[2][4];
            pp_mainloop<T, false, 2, ABLK>(p, smem, m0, n0, 0, nk, acc2, tid, lane, wave, wm, wn);
            gemm_epilogue<T, EPI, false, 2>(acc2, p, smem, m0, n0, tid, lane, wm, wn);
            return;
        }
        if (rows <= 128) {
            f32x4_t acc4[4][4];
            pp_mainloop<T, false, 4, ABLK>(p, smem, m0, n0, 0, nk, acc4, tid, lane, wave, wm, wn);
            gemm_epilogue<T, EPI, false, 4>(acc4, p, smem, m0, n0, tid, lane, wm, wn);
            return;
        }
    }
    f32x4_t acc[MI][4];
#ifdef FINO_GEMM_STAMP
    unsigned long long ts0, ts1;
    STAMP(ts0)
#endif
    pp_mainloop<T, CONV, MI, ABLK>(p, smem, m0, n0, 0, nk, acc, tid, lane, wave, wm, wn);
#ifdef FINO_GEMM_STAMP
    STAMP(ts1)
    if (blockIdx.x == 17 && lane == 0) fino_gemm_dbg[wave * 8 + 5] = ts1 - ts0;     // prologue + loop + re-sync
#endif
    gemm_epilogue<T, EPI, false, MI>(acc, p, smem, m0, n0, tid, lane, wm, wn);
}

template <typename T, int EPI, bool GENERIC, bool CONV = false>
int launch_gemm_t(const GemmParams& p, hipStream_t st) {
    static FinoPerDeviceOnce once;
    if (int rc = fino_max_smem_once(once, reinterpret_cast<const void*>(&gemm_kernel<T, EPI, GENERIC, CONV>), kSmemBytes, "fino_gemm")) return rc;
    gemm_kernel<T, EPI, GENERIC, CONV><<<dim3((unsigned)(p.tiles_m * p.tiles_n)), kThreads, kSmemBytes, st>>>(p);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

template <typename T, int EPI, bool CONV = false, int MI = 8, bool ABLK = false>
int launch_gemm_pp(const GemmParams& p, hipStream_t st) {
    static FinoPerDeviceOnce once;
    if (int rc = fino_max_smem_once(once, reinterpret_cast<const void*>(&gemm_pp_kernel<T, EPI, CONV, MI, ABLK>), kSmemBytes, "fino_gemm")) return rc;
    gemm_pp_kernel<T, EPI, CONV, MI, ABLK><<<dim3((unsigned)(p.tiles_m * p.tiles_n)), kThreads, kSmemBytes, st>>>(p);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}

inline int gemm_device_cus() {
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

// ---- tile-height plan: which rows run as 256-row tiles, which as lower ones ------------------------------------------
// Every tile of a launch lasts the same time, so a launch costs its ROUNDS of the CUs whatever the last round holds: the 156
// tiles of a 3080-row token shard (N = 3072) keep 100 of 256 CUs idle for the whole GEMM, 1164 tiles (M = 24640) are 4.55
// rounds.  Measured in round 3 (profiles/r03_gemm_tiles.md):
//   * dealing the K-tiles of the last round to all CUs (stream-K, one launch, write-through hand-off) LOSES even with the
//     hand-off compiled out: blocks that start at different K offsets stop walking K in lock step, and the A / W panels
//     that concurrent tiles share in L2 at every step stop being shared -- the 256 x 256 x 64 step sits right at what
//     the L2 -> LDS path delivers per CU (~28 B / clock: 64 KB in ~2300 cycles against 2048 cycles of MFMA).
//   * what keeps the lock step is the tile HEIGHT.  A full round of (32 mi)-row tiles lasts ~(32 mi + 256) / 512 of a
//     256-row round (operand delivery: every tile streams its whole W panel whatever its height), a tile alone on its CU
//     32 mi / 256 of one (MFMA), and a round of n <= CUs tiles max(MFMA, delivery * n / CUs).
// The plan: leading rows as 256-row tiles, the rest as ONE more launch of lower tiles (two launches of the same kernel
// family, results bit-identical to any other tiling: one fp32 dot product per element in the same K order), whichever
// (rows1, height) minimises that model.  The model is for a GEMM ALONE on the chip: a rank that runs a second stream of
// kernels beside it (the interleaved multi-GPU plan) keeps 256-row tiles (FINO_TUNE_GEMM_TILE_M = 8) -- the other stream
// fills the CUs a partial round leaves idle, and lower tiles only add operand traffic (tools/plan_sim.py: 77.0 vs 78.6 ms
// per rank at 4 GPUs, 47.0 vs 48.2 at 8).  A SECOND launch for a handful of leftover rows never pays: a launch lasts at
// least one tile's walk over K (~16 us at K = 3072, ~75 us at K = 14336, however few rows it has): measured +25 %.
struct TilePlan { int64_t rows1; int mi2; };          // rows1 rows as 256-row tiles (may be 0 or M), the rest with mi2
inline double plan_launch_cost(int64_t rows, int mi, int tiles_n, int cus) {
    if (rows <= 0) return 0.0;
    const double bm = 32.0 * mi;
    const int64_t tiles = ((rows + 32 * mi - 1) / (32 * mi)) * tiles_n;
    // in 256-row full rounds.  The 0.80 floor: a tile's walk over K cannot go faster than its DMA -> LDS -> fragment
    // latency chain (~1 us per K-step), however few rows it has -- one-round launches of 64 .. 128-row tiles all take 0.84
    // of the 256-row one (profiles/r03_gemm_tile_heights.txt); without it the planner bought second launches of a few tile
    // rows at a third of their real price (the N = 8 shard's K|V projection: 77 + 49 us where one launch takes 107)
    const double mfma_raw = 0.85 * bm / 256.0, mfma = mfma_raw > 0.80 ? mfma_raw : 0.80, deliver = (bm + 256.0) / 512.0;
    const int64_t full = tiles / cus, rem = tiles % cus;
    double c = (double)full * (mfma > deliver ? mfma : deliver);
    if (rem) {
        const double d = deliver * (double)rem / cus;
        c += mfma > d ? mfma : d;
    }
    return c + 0.07;                                                             // a launch: ~5 us of a 74-us round
}
// `tile_m`: the caller's per-call choice (fino_gemm_split_n / fino_gemm_blocked_a: 2..7 -> one launch of 32 x that many rows
// per tile, 8 -> 256-row tiles only); 0 = planned, or what the A/B knob FINO_TUNE_GEMM_TILE_M says (tools/ only).
// The search is (M / 256) x 7 cost evaluations and depends on (m, tiles_n, cus) only: memoised in a small direct-mapped
// table (a step makes ~240 GEMM calls of a handful of shapes).  Key AND plan live in ONE 64-bit atomic word per slot --
// [m : 24 | tiles_n : 9 | cus : 9 | rows1 / 256 (0xffff = all rows) : 16 | mi2 - 1 : 3 | unused : 2 | valid : 1] -- so two host threads that
// race on a slot (ctypes releases the GIL: a VAE thread beside a DiT thread) can only replace a whole entry, never pair one
// shape's key with another shape's plan (round 4 kept key and value in two words); shapes beyond the fields skip the memo.
inline TilePlan plan_tiles_search(int64_t m, int tiles_n, int cus);
inline TilePlan plan_tiles(int64_t m, int tiles_n, int cus, int tile_m = 0) {
    const int forced = tile_m ? tile_m : fino_tune_get(FINO_TUNE_GEMM_TILE_M);
    if (forced >= 2 && forced <= 7) return TilePlan{0, forced};
    if (forced == 8 || m <= 0) return TilePlan{m, 8};
    static std::atomic<uint64_t> memo[64];
    const bool memoable = m < (1ll << 24) && tiles_n > 0 && tiles_n < 512 && cus > 0 && cus < 512;
    const uint64_t key = ((uint64_t)m << 18) | ((uint64_t)tiles_n << 9) | (uint64_t)cus;            // 42 bits
    std::atomic<uint64_t>& slot = memo[(key * 0x9E3779B97F4A7C15ull) >> 58];
    if (memoable) {
        const uint64_t e = slot.load(std::memory_order_relaxed);
        if ((e & 1) && (e >> 22) == key) {
            const uint64_t r = (e >> 6) & 0xffff;                    // 0xffff: every row in 256-row tiles (rows1 = m)
            return TilePlan{r == 0xffff ? m : (int64_t)r * 256, (int)((e >> 3) & 7) + 1};
        }
    }
    const TilePlan tp = plan_tiles_search(m, tiles_n, cus);
    if (memoable && (tp.rows1 == m || (tp.rows1 % 256 == 0 && tp.rows1 / 256 < 0xffff)) && tp.mi2 >= 2 && tp.mi2 <= 8)
        slot.store((key << 22) | ((tp.rows1 == m ? 0xffffull : (uint64_t)(tp.rows1 / 256)) << 6) |
                       ((uint64_t)(tp.mi2 - 1) << 3) | 1ull, std::memory_order_relaxed);
    return tp;
}
inline TilePlan plan_tiles_search(int64_t m, int tiles_n, int cus) {
    TilePlan best{m, 8};
    double best_c = plan_launch_cost(m, 8, tiles_n, cus);
    const double base_c = best_c;
    const int64_t rows256 = (m + 255) / 256;
    for (int64_t r1 = 0; r1 < rows256; ++r1) {
        const int64_t rows1 = r1 * 256;
        const double c1 = plan_launch_cost(rows1, 8, tiles_n, cus);
        for (int mi = 2; mi <= 8; ++mi) {
            if (rows1 > 0 && mi == 8) continue;
            const double c = c1 + plan_launch_cost(m - rows1, mi, tiles_n, cus);
            if (c < best_c - 1e-9) { best_c = c; best = TilePlan{rows1, mi}; }
        }
    }
    if (best_c > (best.rows1 > 0 ? 0.97 : 0.985) * base_c) return TilePlan{m, 8};      // not worth a second launch / another code path
    return best;
}

inline bool use_pingpong() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FINO_GEMM_PP");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

template <typename T, int EPI>
int launch_gemm_pp_mi(const GemmParams& p, int mi, hipStream_t st) {
    if (p.a_tpb > 0) {     // K-blocked A: built for the out-projection's epilogue only (fino_gemm_blocked_a checks)
        if constexpr (EPI == FINO_EPI_GATED_RESIDUAL) {
            switch (mi) {
                case 2: return launch_gemm_pp<T, EPI, false, 2, true>(p, st);
                case 3: return launch_gemm_pp<T, EPI, false, 3, true>(p, st);
                case 4: return launch_gemm_pp<T, EPI, false, 4, true>(p, st);
                case 5: return launch_gemm_pp<T, EPI, false, 5, true>(p, st);
                case 6: return launch_gemm_pp<T, EPI, false, 6, true>(p, st);
                case 7: return launch_gemm_pp<T, EPI, false, 7, true>(p, st);
                default: return launch_gemm_pp<T, EPI, false, 8, true>(p, st);
            }
        } else {
            fino_set_error("fino_gemm_blocked_a: epilogue must be FINO_EPI_GATED_RESIDUAL");
            return FINO_ERR_UNSUPPORTED;
        }
    }
    switch (mi) {
        case 2: return launch_gemm_pp<T, EPI, false, 2>(p, st);
        case 3: return launch_gemm_pp<T, EPI, false, 3>(p, st);
        case 4: return launch_gemm_pp<T, EPI, false, 4>(p, st);
        case 5: return launch_gemm_pp<T, EPI, false, 5>(p, st);
        case 6: return launch_gemm_pp<T, EPI, false, 6>(p, st);
        case 7: return launch_gemm_pp<T, EPI, false, 7>(p, st);
        default: return launch_gemm_pp<T, EPI, false, 8>(p, st);
    }
}

// rows [r0, r0 + rows) of the GEMM as one launch of 32 * mi-row tiles
template <typename T>
int launch_gemm_rows(GemmParams p, int64_t r0, int64_t rows, int mi, int epi, hipStream_t st) {
    p.a += r0 * p.lda;
    p.c += r0 * p.ldc;
    if (p.c2) p.c2 += r0 * p.ldc2;
    if (p.r) p.r += r0 * p.ldr;
    if (p.sel) p.sel += r0;
    p.m = rows;
    p.tiles_m = (int)((rows + 32 * mi - 1) / (32 * mi));
    switch (epi) {
        case FINO_EPI_NONE: return launch_gemm_pp_mi<T, FINO_EPI_NONE>(p, mi, st);
        case FINO_EPI_GELU_TANH: return launch_gemm_pp_mi<T, FINO_EPI_GELU_TANH>(p, mi, st);
        case FINO_EPI_RESIDUAL: return launch_gemm_pp_mi<T, FINO_EPI_RESIDUAL>(p, mi, st);
        case FINO_EPI_GATED_RESIDUAL_STAGED: return launch_gemm_pp_mi<T, FINO_EPI_GATED_RESIDUAL_STAGED>(p, mi, st);
        default: return launch_gemm_pp_mi<T, FINO_EPI_GATED_RESIDUAL>(p, mi, st);
    }
}

template <typename T, bool GENERIC>
int launch_gemm_e(const GemmParams& p, int epi, int tile_m, hipStream_t st) {
    const bool fits32 = ((p.m - 1) * p.lda + p.k) * 2 < (1ll << 31) && ((p.n - 1) * p.ldw + p.k) * 2 < (1ll << 31);
    if (epi == FINO_EPI_F32 || epi == FINO_EPI_F32_RESIDUAL) {
        // fp32 output (the split-bf16 products of the Wan VAE's mid-block attention): one launch of 256-row tiles on the
        // ping-pong loop -- the row-offset arithmetic of the planned two-launch form counts T elements
        if (GENERIC || !fits32) {
            fino_set_error("fino_gemm: the fp32-output epilogues need K %% 64 == 0, 16-byte aligned operands and operands < 2 GiB");
            return FINO_ERR_UNSUPPORTED;
        }
        return epi == FINO_EPI_F32 ? launch_gemm_pp<T, FINO_EPI_F32, false, 8>(p, st)
                                   : launch_gemm_pp<T, FINO_EPI_F32_RESIDUAL, false, 8>(p, st);
    }
    if (!GENERIC && use_pingpong() && fits32) {
        const TilePlan tp = plan_tiles(p.m, p.tiles_n, gemm_device_cus(), tile_m);
        if (tp.rows1 > 0)
            if (int rc = launch_gemm_rows<T>(p, 0, tp.rows1, 8, epi, st)) return rc;
        if (tp.rows1 < p.m) return launch_gemm_rows<T>(p, tp.rows1, p.m - tp.rows1, tp.mi2, epi, st);
        return FINO_OK;
    }
    switch (epi) {
        case FINO_EPI_NONE: return launch_gemm_t<T, FINO_EPI_NONE, GENERIC>(p, st);
        case FINO_EPI_GELU_TANH: return launch_gemm_t<T, FINO_EPI_GELU_TANH, GENERIC>(p, st);
        case FINO_EPI_RESIDUAL: return launch_gemm_t<T, FINO_EPI_RESIDUAL, GENERIC>(p, st);
        case FINO_EPI_GATED_RESIDUAL_STAGED: return launch_gemm_t<T, FINO_EPI_GATED_RESIDUAL_STAGED, GENERIC>(p, st);
        default: return launch_gemm_t<T, FINO_EPI_GATED_RESIDUAL, GENERIC>(p, st);
    }
}

// ---- skinny fp32 linear: one wave per output column, M <= 16 rows kept in registers ---------------------------
template <int WT>  // -1 fp32, FINO_BF16, FINO_F16
__global__ __launch_bounds__(256) void skinny_linear_kernel(const float* __restrict__ x, const void* __restrict__ w,
                                                            const void* __restrict__ b, float* __restrict__ y, int m,
                                                            int64_t n, int64_t k, int silu_input) {
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= n) return;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int64_t kk = lane; kk < k; kk += 64) {
        float wv;
        if (WT == -1) wv = ((const float*)w)[col * k + kk];
        else if (WT == FINO_BF16) wv = BF16::to_f32(((const uint16_t*)w)[col * k + kk]);
        else wv = F16::to_f32(((const uint16_t*)w)[col * k + kk]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < m) {
                float xv = x[(int64_t)i * k + kk];
                if (silu_input) xv = xv / (1.0f + __expf(-xv));
                acc[i] += xv * wv;
            }
        }
    }
    float bias = 0.f;
    if (b) {
        if (WT == -1) bias = ((const float*)b)[col];
        else if (WT == FINO_BF16) bias = BF16::to_f32(((const uint16_t*)b)[col]);
        else bias = F16::to_f32(((const uint16_t*)b)[col]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < m) {
            const float s = wave_sum(acc[i]);
            if (lane == 0) y[(int64_t)i * n + col] = s + bias;
        }
    }
}

}  // namespace

extern "C" int fino_gemm_plan(int64_t m, int64_t n, int64_t* rows_256, int* tile_rows_rest) {
    if (m <= 0 || n <= 0) return FINO_ERR_ARG;
    const TilePlan tp = plan_tiles(m, (int)((n + BN - 1) / BN), gemm_device_cus());
    if (rows_256) *rows_256 = tp.rows1;
    if (tile_rows_rest) *tile_rows_rest = tp.rows1 < m ? 32 * tp.mi2 : 0;
    return FINO_OK;
}

extern "C" int fino_gemm(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k,
                         int64_t lda, int64_t ldw, int64_t ldc, int epilogue, const void* r, int64_t ldr,
                         const float* gate, int64_t mod_stride, const int32_t* sel, int dtype, void* stream) {
    return fino_gemm_split_n(a, w, bias, c, m, n, k, lda, ldw, ldc, epilogue, r, ldr, gate, mod_stride, sel, dtype, nullptr,
                             0, 0, 0, stream);
}

extern "C" int fino_gemm_split_n(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k,
                                 int64_t lda, int64_t ldw, int64_t ldc, int epilogue, const void* r, int64_t ldr,
                                 const float* gate, int64_t mod_stride, const int32_t* sel, int dtype, void* c2,
                                 int64_t ldc2, int64_t n_split, int tile_m, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_gemm: dtype %d", dtype);
    FINO_CHECK(tile_m == 0 || (tile_m >= 2 && tile_m <= 8), FINO_ERR_ARG, "fino_gemm: tile_m=%d (0 = planned, 2 .. 8)", tile_m);
    if (c2 || n_split) {
        FINO_CHECK(c2 && n_split > 0 && n_split < n && n_split % BN == 0 && ldc2 % 8 == 0 && ldc2 >= n - n_split &&
                       fino_aligned16(c2) && k % BK == 0 && epilogue <= FINO_EPI_GELU_TANH,
                   FINO_ERR_ARG,
                   "fino_gemm_split_n: needs c2, 0 < n_split < N a multiple of %d, ldc2 >= N - n_split, K %% %d == 0, no "
                   "residual epilogue", BN, BK);
        FINO_CHECK(ldc >= n_split, FINO_ERR_ARG, "fino_gemm_split_n: ldc must cover the first n_split columns");
    }
    FINO_CHECK(a && w && c, FINO_ERR_ARG, "fino_gemm: null pointer");
    FINO_CHECK(m >= 0 && n > 0 && k > 0, FINO_ERR_ARG, "fino_gemm: bad shape M=%lld N=%lld K=%lld", (long long)m,
               (long long)n, (long long)k);
    FINO_CHECK(n % 8 == 0 && k % 8 == 0, FINO_ERR_ARG, "fino_gemm: N=%lld and K=%lld must be multiples of 8",
               (long long)n, (long long)k);
    FINO_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= k && ldw >= k && ldc >= (n_split > 0 ? n_split : n),
               FINO_ERR_ARG, "fino_gemm: leading dimensions must be multiples of 8 and cover the row");
    FINO_CHECK(fino_aligned16(a) && fino_aligned16(w) && fino_aligned16(c), FINO_ERR_ARG,
               "fino_gemm: A/W/C must be 16-byte aligned");
    FINO_CHECK(epilogue >= FINO_EPI_NONE && epilogue <= FINO_EPI_F32_RESIDUAL, FINO_ERR_ARG,
               "fino_gemm: epilogue %d", epilogue);
    if (epilogue == FINO_EPI_F32 || epilogue == FINO_EPI_F32_RESIDUAL)
        FINO_CHECK(!c2 && tile_m == 0 && fino_aligned16(bias), FINO_ERR_ARG,
                   "fino_gemm: the fp32-output epilogues take one output, the planned tiling and a 16-byte aligned fp32 bias");
    if (epilogue >= FINO_EPI_RESIDUAL && epilogue != FINO_EPI_F32)
        FINO_CHECK(r && ldr % 8 == 0 && ldr >= n && fino_aligned16(r), FINO_ERR_ARG, "fino_gemm: residual operand");
    if (epilogue == FINO_EPI_GATED_RESIDUAL || epilogue == FINO_EPI_GATED_RESIDUAL_STAGED)
        FINO_CHECK(gate && fino_aligned16(gate) && mod_stride % 4 == 0, FINO_ERR_ARG, "fino_gemm: gate operand");
    if (m == 0) return FINO_OK;
    GemmParams p = {};
    p.c2 = (uint16_t*)c2; p.ldc2 = ldc2; p.n_split = n_split;
    p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.c = (uint16_t*)c;
    p.r = (const uint16_t*)r; p.gate = gate; p.sel = sel;
    p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.mod_stride = mod_stride;
    p.tiles_m = (int)((m + BM - 1) / BM);
    p.tiles_n = (int)((n + BN - 1) / BN);
    p.group_m = fino_tune_get(FINO_TUNE_GEMM_GROUP_M);
    hipStream_t st = (hipStream_t)stream;
    const bool generic = (k % BK) != 0;
    if (p.group_m <= 0) p.group_m = gemm_default_group_m(p.tiles_n, k);
    // Long-K GEMMs (FFN-down: K = 14336, a 706 MB A operand written by the launch before it) walk their tile rows LAST TO FIRST: the
    // rows the producer wrote last are the ones still in the 256 MiB Infinity Cache, and a first-to-last walk evicts them before it gets
    // there.  Same bits; FFN-up + FFN-down 3373.8 -> 3340.4 us per pair at M = 24640 (tools/ffn_pair_ab.py, profiles/r06_ffn_pair_ab.txt).
    // FINO_TUNE_GEMM_RASTER: 0 = this default, 1 = last to first for every K >= 8192 (the same), 2 = first to last (rounds 1 - 5).
    {
        const int rk = fino_tune_get(FINO_TUNE_GEMM_RASTER);
        if (rk == 1 || (rk == 3 && (k >= 8192 || n >= 8192)) || (rk != 2 && rk != 3 && k >= 8192)) p.group_m |= 0x100;
    }
    if (dtype == FINO_BF16)
        return generic ? launch_gemm_e<BF16, true>(p, epilogue, tile_m, st) : launch_gemm_e<BF16, false>(p, epilogue, tile_m, st);
    return generic ? launch_gemm_e<F16, true>(p, epilogue, tile_m, st) : launch_gemm_e<F16, false>(p, epilogue, tile_m, st);
}

extern "C" int fino_gemm_blocked_a(const void* a, const void* w, const void* bias, void* c, int64_t m, int64_t n, int64_t k,
                                   int64_t a_block_k, int64_t a_block_stride, int a_groups, int64_t a_group_stride,
                                   int64_t lda, int64_t ldw, int64_t ldc,
                                   const void* r, int64_t ldr, const float* gate, int64_t mod_stride, const int32_t* sel,
                                   int dtype, int tile_m, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_gemm_blocked_a: dtype %d", dtype);
    FINO_CHECK(tile_m == 0 || (tile_m >= 2 && tile_m <= 8), FINO_ERR_ARG, "fino_gemm_blocked_a: tile_m=%d (0 = planned, 2 .. 8)",
               tile_m);
    FINO_CHECK(a && w && c && r && gate, FINO_ERR_ARG, "fino_gemm_blocked_a: null pointer");
    FINO_CHECK(m >= 0 && n > 0 && k > 0 && n % 8 == 0, FINO_ERR_ARG, "fino_gemm_blocked_a: bad shape");
    FINO_CHECK(a_block_k > 0 && a_block_k % BK == 0 && k % a_block_k == 0 && a_block_k / BK <= 256 && k / BK <= 4096,
               FINO_ERR_ARG, "fino_gemm_blocked_a: a_block_k=%lld must be a multiple of %d dividing K=%lld",
               (long long)a_block_k, BK, (long long)k);
    FINO_CHECK(lda % 8 == 0 && lda >= a_block_k && a_block_stride % 8 == 0 && ldw % 8 == 0 && ldw >= k && ldc % 8 == 0 &&
                   ldc >= n && ldr % 8 == 0 && ldr >= n && mod_stride % 4 == 0,
               FINO_ERR_ARG, "fino_gemm_blocked_a: leading dimensions");
    FINO_CHECK(fino_aligned16(a) && fino_aligned16(w) && fino_aligned16(c) && fino_aligned16(r) && fino_aligned16(gate),
               FINO_ERR_ARG, "fino_gemm_blocked_a: 16-byte alignment required");
    const int64_t nblk = k / a_block_k;
    FINO_CHECK(a_groups >= 1 && a_groups <= 16 && nblk % a_groups == 0 && a_group_stride % 8 == 0, FINO_ERR_ARG,
               "fino_gemm_blocked_a: a_groups=%d must divide the %lld K blocks", a_groups, (long long)nblk);
    const int64_t span = ((a_groups - 1) * a_group_stride + (nblk / a_groups - 1) * a_block_stride +
                          (m > 0 ? m - 1 : 0) * lda + a_block_k) * 2;
    FINO_CHECK(span < (1ll << 31) && ((n - 1) * ldw + k) * 2 < (1ll << 31), FINO_ERR_ARG,
               "fino_gemm_blocked_a: operands must span < 2 GiB");
    if (m == 0) return FINO_OK;
    GemmParams p = {};
    p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.c = (uint16_t*)c;
    p.r = (const uint16_t*)r; p.gate = gate; p.sel = sel;
    p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.mod_stride = mod_stride;
    p.a_tpb = (int)(a_block_k / BK); p.a_inv = (65536 + p.a_tpb - 1) / p.a_tpb; p.a_blk_elems = a_block_stride;
    p.a_groups = a_groups; p.a_ginv = (65536 + a_groups - 1) / a_groups; p.a_grp_elems = a_group_stride;
    // (x * ceil(65536 / d)) >> 16 == x / d holds while x * (d * ceil(65536 / d) - 65536) < 65536: check it for the largest
    // K-tile index and the largest K-block index instead of trusting the shape limits above
    FINO_CHECK((k / BK - 1) * ((int64_t)p.a_inv * p.a_tpb - 65536) < 65536 &&
                   (nblk - 1) * ((int64_t)p.a_ginv * a_groups - 65536) < 65536,
               FINO_ERR_UNSUPPORTED,
               "fino_gemm_blocked_a: a_block_k=%lld / a_groups=%d with K=%lld is beyond the exact range of the kernel's "
               "reciprocal index arithmetic", (long long)a_block_k, a_groups, (long long)k);
    p.tiles_m = (int)((m + BM - 1) / BM);
    p.tiles_n = (int)((n + BN - 1) / BN);
    p.group_m = fino_tune_get(FINO_TUNE_GEMM_GROUP_M);
    if (p.group_m <= 0) p.group_m = gemm_default_group_m(p.tiles_n, k);
    hipStream_t st = (hipStream_t)stream;
    const TilePlan tp = plan_tiles(p.m, p.tiles_n, gemm_device_cus(), tile_m);
    auto rows = [&](int64_t r0, int64_t nr, int mi) {
        return dtype == FINO_BF16 ? launch_gemm_rows<BF16>(p, r0, nr, mi, FINO_EPI_GATED_RESIDUAL, st)
                                  : launch_gemm_rows<F16>(p, r0, nr, mi, FINO_EPI_GATED_RESIDUAL, st);
    };
    if (tp.rows1 > 0)
        if (int rc = rows(0, tp.rows1, 8)) return rc;
    if (tp.rows1 < p.m) return rows(tp.rows1, p.m - tp.rows1, tp.mi2);
    return FINO_OK;
}

// fino_conv3d and fino_conv3d_split (planes = 0 / 2 / 3): one body.  With planes > 0, c_in_pad is the width of ONE plane,
// the A rows hold `planes` of them side by side and W one plane per product (3 for 2 planes, 6 for 3); bias, y and r are fp32.
static int conv3d_impl(const void* x, const void* w, const void* bias, void* y, int t_in, int h_in, int w_in,
                       int c_in_pad, int t_out, int h_out, int w_out, int c_out_pad, int kt, int kh, int kw, int st,
                       int sh, int sw, int pt, int ph, int pw, int upsample2x, int epilogue, const void* r,
                       const void* zero_page, int dtype, int planes, void* stream);

extern "C" int fino_conv3d(const void* x, const void* w, const void* bias, void* y, int t_in, int h_in, int w_in,
                           int c_in_pad, int t_out, int h_out, int w_out, int c_out_pad, int kt, int kh, int kw, int st,
                           int sh, int sw, int pt, int ph, int pw, int upsample2x, int epilogue, const void* r,
                           const void* zero_page, int dtype, void* stream) {
    return conv3d_impl(x, w, bias, y, t_in, h_in, w_in, c_in_pad, t_out, h_out, w_out, c_out_pad, kt, kh, kw, st, sh, sw, pt, ph,
                       pw, upsample2x, epilogue, r, zero_page, dtype, 0, stream);
}

extern "C" int fino_conv3d_split(const void* x_planes, const void* w_products, const float* bias, float* y, int t_in, int h_in,
                                 int w_in, int c_in_pad, int planes, int t_out, int h_out, int w_out, int c_out_pad, int kt,
                                 int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int upsample2x, int epilogue,
                                 const float* r, const void* zero_page, void* stream) {
    FINO_CHECK(planes == 2 || planes == 3, FINO_ERR_ARG, "fino_conv3d_split: planes = %d (2: hi | lo, 3: hi | mid | lo)", planes);
    return conv3d_impl(x_planes, w_products, bias, y, t_in, h_in, w_in, c_in_pad, t_out, h_out, w_out, c_out_pad, kt, kh, kw, st,
                       sh, sw, pt, ph, pw, upsample2x, epilogue, r, zero_page, FINO_BF16, planes, stream);
}

static int conv3d_impl(const void* x, const void* w, const void* bias, void* y, int t_in, int h_in, int w_in,
                       int c_in_pad, int t_out, int h_out, int w_out, int c_out_pad, int kt, int kh, int kw, int st,
                       int sh, int sw, int pt, int ph, int pw, int upsample2x, int epilogue, const void* r,
                       const void* zero_page, int dtype, int planes, void* stream) {
    FINO_CHECK(dtype == FINO_BF16 || dtype == FINO_F16, FINO_ERR_ARG, "fino_conv3d: dtype %d", dtype);
    FINO_CHECK(x && w && y && zero_page, FINO_ERR_ARG, "fino_conv3d: null pointer");
    FINO_CHECK(c_in_pad > 0 && c_in_pad % BK == 0, FINO_ERR_ARG,
               "fino_conv3d: c_in_pad=%d must be a multiple of %d (pad channels with zeros)", c_in_pad, BK);
    FINO_CHECK(c_out_pad > 0 && c_out_pad % 8 == 0, FINO_ERR_ARG, "fino_conv3d: c_out_pad=%d %% 8 != 0", c_out_pad);
    FINO_CHECK(t_in > 0 && h_in > 0 && w_in > 0 && t_out > 0 && h_out > 0 && w_out > 0 && kt > 0 && kh > 0 && kw > 0 &&
                   st > 0 && sh > 0 && sw > 0 && pt >= 0 && ph >= 0 && pw >= 0,
               FINO_ERR_ARG, "fino_conv3d: bad geometry");
    FINO_CHECK(epilogue == FINO_EPI_NONE || (epilogue == FINO_EPI_RESIDUAL && r), FINO_ERR_ARG,
               "fino_conv3d: epilogue %d (NONE or RESIDUAL with r)", epilogue);
    FINO_CHECK(fino_aligned16(x) && fino_aligned16(w) && fino_aligned16(y) && fino_aligned16(r) &&
                   fino_aligned16(zero_page),
               FINO_ERR_ARG, "fino_conv3d: 16-byte alignment required");
    // (the split products' fp32 epilogue loads the bias as float4: gemm_epilogue_f32)
    FINO_CHECK(planes == 0 || fino_aligned16(bias), FINO_ERR_ARG, "fino_conv3d_split: bias must be 16-byte aligned");
    const int products = planes == 0 ? 1 : (planes == 2 ? 3 : 6);
    const int64_t a_width = (int64_t)(planes == 0 ? 1 : planes) * c_in_pad;         // elements of one A row
    GemmParams p = {};
    p.a = (const uint16_t*)x; p.w = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.c = (uint16_t*)y;
    p.r = (const uint16_t*)r; p.gate = nullptr; p.sel = nullptr;
    p.m = (int64_t)t_out * h_out * w_out; p.n = c_out_pad; p.k = (int64_t)kt * kh * kw * c_in_pad * products;
    p.lda = a_width; p.ldw = p.k; p.ldc = c_out_pad; p.ldr = c_out_pad; p.mod_stride = 0;
    p.a_cc = planes == 0 ? 0 : c_in_pad / BK; p.a_nplanes = planes;
    p.tiles_m = (int)((p.m + BM - 1) / BM); p.tiles_n = (int)((p.n + BN - 1) / BN);
    p.group_m = fino_tune_get(FINO_TUNE_GEMM_GROUP_M);
    p.to = t_out; p.ho = h_out; p.wo = w_out; p.ti = t_in; p.hi = h_in; p.wi = w_in;
    p.kt = kt; p.kh = kh; p.kw = kw; p.st = st; p.sh = sh; p.sw = sw; p.pt = pt; p.ph = ph; p.pw = pw;
    p.up = upsample2x; p.cin_chunks = c_in_pad / BK * products; p.zero_page = (const uint16_t*)zero_page;
    hipStream_t s = (hipStream_t)stream;
    // ping-pong loop with per-tap gather offsets: 32-bit offsets relative to the first input frame a tile can touch,
    // output positions packed 8/12/12 bits
    const int64_t hw_out = (int64_t)h_out * w_out;
    const int64_t t_span = (BM - 1 + hw_out - 1) / hw_out + 1;                      // output frames a 256-row tile can straddle
    const int64_t in_span_bytes = (t_span * st + kt) * (int64_t)h_in * w_in * a_width * 2;
    const bool pp = use_pingpong() && fino_tune_get(FINO_TUNE_CONV_LOOP) != 1 && in_span_bytes < (1ll << 31) &&
                    h_out < 4096 && w_out < 4096 && t_span < 256 && c_out_pad % 32 == 0 && ((p.n - 1) * p.ldw + p.k) * 2 < (1ll << 31);
    if (planes > 0) {
        // the split products run on the ping-pong loop only (its K walk knows the plane map); what does not fit its 32-bit
        // gather offsets -- one tile's input span of `planes` planes beyond 2 GiB -- is refused, not silently slow
        if (!pp) {
            fino_set_error("fino_conv3d_split: a 256-row tile's input span (%lld bytes) exceeds the 2 GiB the gather offsets cover, "
                           "or c_out_pad %% 32 != 0 -- run the layer in chunks of fewer frames", (long long)in_span_bytes);
            return FINO_ERR_UNSUPPORTED;
        }
        return epilogue == FINO_EPI_NONE ? launch_gemm_pp<BF16, FINO_EPI_F32, true>(p, s)
                                         : launch_gemm_pp<BF16, FINO_EPI_F32_RESIDUAL, true>(p, s);
    }
    if (pp) {
        if (dtype == FINO_BF16)
            return epilogue == FINO_EPI_NONE ? launch_gemm_pp<BF16, FINO_EPI_NONE, true>(p, s)
                                             : launch_gemm_pp<BF16, FINO_EPI_RESIDUAL, true>(p, s);
        return epilogue == FINO_EPI_NONE ? launch_gemm_pp<F16, FINO_EPI_NONE, true>(p, s)
                                         : launch_gemm_pp<F16, FINO_EPI_RESIDUAL, true>(p, s);
    }
    if (dtype == FINO_BF16)
        return epilogue == FINO_EPI_NONE ? launch_gemm_t<BF16, FINO_EPI_NONE, false, true>(p, s)
                                         : launch_gemm_t<BF16, FINO_EPI_RESIDUAL, false, true>(p, s);
    return epilogue == FINO_EPI_NONE ? launch_gemm_t<F16, FINO_EPI_NONE, false, true>(p, s)
                                     : launch_gemm_t<F16, FINO_EPI_RESIDUAL, false, true>(p, s);
}

extern "C" int fino_skinny_linear(const float* x, const void* w, const void* b, float* y, int m, int64_t n,
                                  int64_t k, int w_dtype, int silu_input, void* stream) {
    FINO_CHECK(x && w && y, FINO_ERR_ARG, "fino_skinny_linear: null pointer");
    FINO_CHECK(m >= 1 && m <= 16 && n > 0 && k > 0, FINO_ERR_ARG, "fino_skinny_linear: need 1 <= M <= 16 (M=%d)", m);
    FINO_CHECK(w_dtype == -1 || w_dtype == FINO_BF16 || w_dtype == FINO_F16, FINO_ERR_ARG,
               "fino_skinny_linear: w_dtype %d", w_dtype);
    const dim3 grid((unsigned)((n + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (w_dtype == -1) skinny_linear_kernel<-1><<<grid, 256, 0, st>>>(x, w, b, y, m, n, k, silu_input);
    else if (w_dtype == FINO_BF16) skinny_linear_kernel<FINO_BF16><<<grid, 256, 0, st>>>(x, w, b, y, m, n, k, silu_input);
    else skinny_linear_kernel<FINO_F16><<<grid, 256, 0, st>>>(x, w, b, y, m, n, k, silu_input);
    FINO_LAUNCH_CHECK();
    return FINO_OK;
}
