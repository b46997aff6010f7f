// Micro-benchmark (diagnostic, standalone):
//     hipcc --offload-arch=gfx950 -O3 l2_lds_stream.hip -o bin/l2_lds_stream && bin/l2_lds_stream
// What does the L2 -> LDS path deliver per CU for the GEMM kernel's own staging pattern (csrc/fino_gemm.hip, pp_mainloop:
// LDS-DMA pieces of 8 rows x 128 B per wave-instruction, row pitch = K x 2 bytes, XOR swizzle on the source chunk, the K
// advance in the scalar offset; 64 pieces = 64 KiB per 256 x 256 x 64 K-step per CU)?  DESIGN.md section 4.2 claimed "a
// 256 x 256 x 64 step already sits at what L2 -> LDS delivers per CU (~28 B/clk)"; this prints the measured ceiling for
//   * where the rows come from:  one 3-MB panel set shared by the whole chip (pure L2 hits);  the GEMM raster's sharing (an
//     XCD's 32 CUs form a G x 32/G window of tiles: G + 32/G panels of 1.5 MB per sweep, 144 MB in all: Infinity-Cache
//     backed, what the real GEMM reads);  private panels per CU (768 MB: HBM)
//   * one workgroup of 8 waves per CU, or two of 4 waves (the same 64 KiB per step per CU)
//   * DMA alone / beside the K-step's 64 MFMAs per wave (16x16x32 bf16 from registers) / + its 24 ds_read_b128 per wave
//   * one K-step in flight behind a counted vmcnt, with and without the per-step barrier
// Output: shader cycles per 64-KiB step (s_memtime, median CU), bytes per clock per CU, chip-wide TB/s from wall time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4_t;
#define LDS_AS __attribute__((address_space(3)))

constexpr int KELEMS = 3072;                 // row pitch of the panels (the block GEMMs' K)
constexpr int ROW_BYTES = KELEMS * 2;
constexpr int STEPS = KELEMS / 64;           // 48 K-steps per sweep
constexpr int PANEL_ROWS = 256;
constexpr int64_t PANEL_BYTES = (int64_t)PANEL_ROWS * ROW_BYTES;      // 1.5 MB

struct Params {
    const char* base;
    int64_t total_bytes;
    int mode;         // 0 shared, 1 window (G = group), 2 private
    int group;
    int sweeps;
    uint64_t* cycles; // per workgroup
};

// WAVES waves; every wave issues 8 pieces per step: 2 panels x 256 rows = 512 rows = 64 pieces (WAVES = 8) or one workgroup
// of a pair takes 32 of them (WAVES = 4: its own A half and W half).
template <int WAVES, bool MFMA, bool LDSREAD, bool BARRIER>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = WAVES * 8 * 1024;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // which CU-slot am I: blocks b and b + 8 share an XCD (round-robin dealing); slot = position inside the XCD
    const int wg = blockIdx.x;
    const int cu = WAVES == 8 ? wg : wg >> 1;            // two 4-wave workgroups share a "tile"
    const int half = WAVES == 8 ? 0 : (wg & 1);
    const int xcd = cu & 7, slot = cu >> 3;              // 32 slots per XCD
    int64_t a_panel, w_panel;
    if (p.mode == 0) { a_panel = 0; w_panel = 1; }
    else if (p.mode == 1) {
        const int g = p.group, cols = 32 / g;
        a_panel = xcd * (g + cols) + (slot % g);
        w_panel = xcd * (g + cols) + g + (slot / g) % cols;
    } else { a_panel = 2 * cu; w_panel = 2 * cu + 1; }
    const int r8 = lane >> 3;
    const int sk = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16;       // swizzled 16-B chunk of the 128-B row piece
    uint32_t off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // pieces 0-3: A panel, 4-7: W panel (8 waves: rows q*64 + wave*8; 4 waves: my half's 128 rows)
        const int64_t panel = q < 4 ? a_panel : w_panel;
        const int row = WAVES == 8 ? ((q & 3) * 64 + wave * 8 + r8) : (half * 128 + (q & 3) * 32 + wave * 8 + r8);
        off[q] = (uint32_t)(panel * PANEL_BYTES + (int64_t)row * ROW_BYTES + sk);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.base, 0, (int)p.total_bytes, 0x00020000);
    f32x4_t acc[16];
    bf16x8_t fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4_t t = {0x3f803f80u + lane * 0x00010001u, 0x3e803f00u ^ (lane * 2654435761u), 0xbf004000u + i, 0x3dcc3f99u * (lane + 1)};
        fa[i] = __builtin_bit_cast(bf16x8_t, t);
        t[1] ^= 0x00550033u;
        fb[i] = __builtin_bit_cast(bf16x8_t, t);
    }
    u32x4_t frag[24];
#define DMA(STAGE_, KT_)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)(smem + (STAGE_) * STAGE + (q * WAVES + wave) * 1024), \
                                                 16, off[q], (KT_) * 128, 0, 0);
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int sw = 0; sw < p.sweeps; ++sw) {
        DMA(0, 0)
        for (int kt = 0; kt < STEPS; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < STEPS) { DMA(cur ^ 1, kt + 1) }
            // step kt's pieces (issued one step ago) must have landed: 8 younger ones may stay in flight
            if (kt + 1 < STEPS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (BARRIER) __builtin_amdgcn_s_barrier();
            if (LDSREAD) {
                const char* sb = smem + cur * STAGE;
                const int base = ((lane & 15) * 128) + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4);
#pragma unroll
                for (int i = 0; i < 24; ++i)
                    frag[i] = *reinterpret_cast<const u32x4_t*>(sb + ((i * 2048 + wave * 4096) % STAGE) + base);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MFMA) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        bf16x8_t a = fa[i & 3], b = fb[(i >> 2) & 3];
                        if (LDSREAD) { a = __builtin_bit_cast(bf16x8_t, frag[(r * 6 + i) % 24]); }
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                    }
            } else if (LDSREAD) {
#pragma unroll
                for (int i = 0; i < 24; ++i) asm volatile("" ::"v"(frag[i]));
            }
            if (BARRIER) __builtin_amdgcn_s_barrier();      // (the ping-pong loop has two per step)
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1234.5f) p.cycles[0] = 1;
    if (threadIdx.x == 0) p.cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES, bool MFMA, bool LDSREAD, bool BARRIER>
void run(const char* what, Params p, const char* src) {
    const int wgs = WAVES == 8 ? 256 : 512;
    const int smem = 2 * WAVES * 8 * 1024;
    auto k = stream_kernel<WAVES, MFMA, LDSREAD, BARRIER>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {                       // first launch warms the caches
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), smem, 0, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(wgs);
    CK(hipMemcpy(h.data(), p.cycles, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double steps = (double)p.sweeps * STEPS;
    const double cyc = h[wgs / 2] / steps, cyc_max = h[wgs - 1] / steps;
    const double bytes_cu_step = 65536.0;
    printf("%-7s %-34s | %7.0f cycles per 64-KiB step (slowest CU %7.0f) = %5.1f B/clk/CU | chip %5.2f TB/s | %6.2f ms\n", src,
           what, cyc, cyc_max, bytes_cu_step / cyc, 256.0 * bytes_cu_step * steps / (ms * 1e-3) / 1e12, ms);
}

int main() {
    const int64_t total = 768ll << 20;
    char* buf;
    uint64_t* cyc;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&cyc, 512 * 8));
    // gaussian-ish bf16 bit patterns (operand bits matter for the clock the chip holds beside MFMAs)
    {
        std::vector<uint32_t> h(1 << 20);
        uint32_t x = 12345u;
        for (auto& v : h) {
            x = x * 1664525u + 1013904223u;
            const uint32_t lo = 0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u);
            x = x * 1664525u + 1013904223u;
            const uint32_t hi = 0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u);
            v = lo | (hi << 16);
        }
        for (int64_t o = 0; o < total; o += (int64_t)h.size() * 4) CK(hipMemcpy(buf + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    Params p{buf, total, 0, 4, 20, cyc};
    printf("# L2 -> LDS delivery for the GEMM's staging pattern; the ping-pong GEMM needs 64 KiB per ~2250-2400 cycles = 27-29 B/clk/CU\n");
    const char* names[3] = {"L2", "raster4", "HBM"};
    for (int mode = 0; mode < 3; ++mode) {
        p.mode = mode;
        p.sweeps = mode == 2 ? 4 : 20;
        run<8, false, false, false>("1 WG x 8 waves: DMA only", p, names[mode]);
        run<8, false, false, true>("1 WG x 8 waves: DMA + 2 barriers", p, names[mode]);
        run<8, true, false, true>("1 WG x 8 waves: DMA + MFMA + bar", p, names[mode]);
        run<8, true, true, true>("1 WG x 8 waves: DMA + reads + MFMA + bar", p, names[mode]);
        run<4, false, false, false>("2 WG x 4 waves: DMA only", p, names[mode]);
        run<4, true, false, true>("2 WG x 4 waves: DMA + MFMA + bar", p, names[mode]);
        run<4, true, true, true>("2 WG x 4 waves: DMA + reads + MFMA + bar", p, names[mode]);
    }
    p.mode = 1;
    p.sweeps = 20;
    for (int g : {1, 2, 8, 16}) {
        p.group = g;
        char nm[64];
        snprintf(nm, sizeof nm, "raster%d", g);
        run<8, false, false, false>("1 WG x 8 waves: DMA only", p, nm);
        run<8, true, true, true>("1 WG x 8 waves: DMA + reads + MFMA + bar", p, nm);
    }
    return 0;
}
