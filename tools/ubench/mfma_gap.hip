// Micro-benchmark (diagnostic, standalone: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result mfma_gap.hip -o mfma_gap): what one wave per
// SIMD pays for vector instructions placed between back-to-back v_mfma_f32_32x32x16_bf16 -- the gap of the 4-wave
// attention kernel's software pipeline.  Every variant runs `iters` x 4 gaps "MFMA ; fillers" on all CUs (256 threads per
// block, one block per CU) and reports shader cycles per gap (s_memtime) and the wall time per gap.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define MFMA(T_) "v_mfma_f32_32x32x16_bf16 %" #T_ ", %4, %5, %" #T_ "\n\t"
// operands: 0-3 accumulators, 4 a, 5 b, 6.. x0..x7 (f32 fillers), 14..17 w0..w3 (packed outputs)
#define EXP(I_) "v_exp_f32 %" #I_ ", %" #I_ "\n\t"
#define ADD(I_) "v_add_f32 %" #I_ ", %" #I_ ", %" #I_ "\n\t"
#define MAX3(I_, J_, K_) "v_max3_f32 %" #I_ ", %" #I_ ", %" #J_ ", %" #K_ "\n\t"
#define CVT(W_, I_, J_) "v_cvt_pk_bf16_f32 %" #W_ ", %" #I_ ", %" #J_ "\n\t"
#define CVTH(W_, I_, J_) "v_cvt_pk_f16_f32 %" #W_ ", %" #I_ ", %" #J_ "\n\t"
#define PERM(W_, I_, J_) "v_perm_b32 %" #W_ ", %" #I_ ", %" #J_ ", %18\n\t"
#define ANDOR(W_, I_, J_) "v_and_or_b32 %" #W_ ", %" #I_ ", %19, %" #J_ "\n\t"
#define NOP "s_nop 0\n\t"
#define LD128(R_, OFF_) "ds_read_b128 %" #R_ ", %20 offset:" #OFF_ "\n\t"
#define LDTR(R_, OFF_) "ds_read_b64_tr_b16 %" #R_ ", %20 offset:" #OFF_ "\n\t"
#define WAITL(N_) "s_waitcnt lgkmcnt(" #N_ ")\n\t"
#define MFMA_AB(T_) "v_mfma_f32_32x32x16_bf16 %" #T_ ", %4, %23, %" #T_ "\n\t"
#define BODYA(F0_, F1_, F2_, F3_)                                                                                    \
    asm volatile(MFMA_AB(0) F0_ MFMA_AB(1) F1_ MFMA_AB(2) F2_ MFMA_AB(3) F3_                                         \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                            \
                 : "v"(a), "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), \
                   "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(sel), "v"(msk), "v"(laddr), "v"(l0), "v"(l1), "a"(ba), "s"(s0), "s"(s1), "v"(h0), "v"(h1) : "memory");

#define BODY(F0_, F1_, F2_, F3_)                                                                                     \
    asm volatile(MFMA(0) F0_ MFMA(1) F1_ MFMA(2) F2_ MFMA(3) F3_                                                     \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                            \
                 : "v"(a), "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), \
                   "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(sel), "v"(msk), "v"(laddr), "v"(l0), "v"(l1), "a"(ba), "s"(s0), "s"(s1), "v"(h0), "v"(h1) : "memory");

template <int V, int U>
__global__ __launch_bounds__(256) void gap_kernel(uint64_t* out, const uint4* opnd, int iters) {
    const u32x4_t a = __builtin_bit_cast(u32x4_t, opnd[threadIdx.x]), b = __builtin_bit_cast(u32x4_t, opnd[256 + threadIdx.x]);
    f32x16_t acc[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = -1.0f - 0.01f * (threadIdx.x + i);
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t sel = 0x07060302u, msk = 0xffff0000u;
    asm volatile("" : "+v"(sel), "+v"(msk), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = opnd[i & 511];
    __syncthreads();
    uint32_t laddr = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
    u32x4_t l0 = a, l1 = b, ba = b;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t h0 = {a[0], a[1]}, h1 = {b[0], b[1]};
    asm volatile("" : "+v"(h0), "+v"(h1));
    uint32_t s0 = 5, s1 = 3;
    asm volatile("" : "+v"(laddr), "+v"(l0), "+v"(l1), "+a"(ba), "+s"(s0), "+s"(s1));
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (V == 0) BODY("", "", "", "")
        if constexpr (V == 1) BODY(EXP(6) EXP(7), EXP(8) EXP(9), EXP(10) EXP(11), EXP(12) EXP(13))
        if constexpr (V == 2) BODY(CVT(14, 6, 7) CVT(15, 8, 9), CVT(16, 10, 11) CVT(17, 12, 13), CVT(14, 6, 7) CVT(15, 8, 9), CVT(16, 10, 11) CVT(17, 12, 13))
        if constexpr (V == 3) BODY(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11))
        if constexpr (V == 4) BODY(ADD(6) ADD(7) ADD(8) ADD(9), ADD(10) ADD(11) ADD(12) ADD(13), ADD(6) ADD(7) ADD(8) ADD(9), ADD(10) ADD(11) ADD(12) ADD(13))
        if constexpr (V == 5) BODY(PERM(14, 6, 7) PERM(15, 8, 9), PERM(16, 10, 11) PERM(17, 12, 13), PERM(14, 6, 7) PERM(15, 8, 9), PERM(16, 10, 11) PERM(17, 12, 13))
        if constexpr (V == 6) BODY(EXP(6) EXP(7) PERM(14, 8, 9) PERM(15, 10, 11), EXP(8) EXP(9) PERM(16, 12, 13) PERM(17, 6, 7), EXP(10) EXP(11) PERM(14, 12, 13) PERM(15, 6, 7), EXP(12) EXP(13) PERM(16, 8, 9) PERM(17, 10, 11))
        if constexpr (V == 7) BODY(EXP(6) EXP(7) EXP(8) EXP(9), EXP(10) EXP(11) EXP(12) EXP(13), EXP(6) EXP(7) EXP(8) EXP(9), EXP(10) EXP(11) EXP(12) EXP(13))
        if constexpr (V == 8) BODY(EXP(6) EXP(7) EXP(8), EXP(9) EXP(10) EXP(11), EXP(12) EXP(13) EXP(6), EXP(7) EXP(8) EXP(9))
        if constexpr (V == 9) BODY(EXP(6) CVT(14, 8, 9), EXP(7) CVT(15, 10, 11), EXP(8) CVT(16, 12, 13), EXP(9) CVT(17, 6, 7))
        if constexpr (V == 10) BODY(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11) NOP CVT(16, 12, 13), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7) NOP CVT(14, 10, 11), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7) NOP CVT(16, 8, 9), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11) NOP CVT(14, 6, 7))
        if constexpr (V == 11) BODY(CVT(14, 6, 7), CVT(15, 8, 9), CVT(16, 10, 11), CVT(17, 12, 13))
        if constexpr (V == 12) BODY(CVT(14, 6, 7) CVT(15, 8, 9) CVT(16, 10, 11) CVT(17, 12, 13), CVT(14, 6, 7) CVT(15, 8, 9) CVT(16, 10, 11) CVT(17, 12, 13), CVT(14, 6, 7) CVT(15, 8, 9) CVT(16, 10, 11) CVT(17, 12, 13), CVT(14, 6, 7) CVT(15, 8, 9) CVT(16, 10, 11) CVT(17, 12, 13))
        if constexpr (V == 13) BODY(CVTH(14, 6, 7) CVTH(15, 8, 9), CVTH(16, 10, 11) CVTH(17, 12, 13), CVTH(14, 6, 7) CVTH(15, 8, 9), CVTH(16, 10, 11) CVTH(17, 12, 13))
        if constexpr (V == 14) BODY(ANDOR(14, 6, 7) ANDOR(15, 8, 9), ANDOR(16, 10, 11) ANDOR(17, 12, 13), ANDOR(14, 6, 7) ANDOR(15, 8, 9), ANDOR(16, 10, 11) ANDOR(17, 12, 13))
        if constexpr (V == 15) BODY(MAX3(6, 7, 8) MAX3(9, 10, 11) MAX3(12, 13, 6) MAX3(7, 8, 9), MAX3(6, 7, 8) MAX3(9, 10, 11) MAX3(12, 13, 6) MAX3(7, 8, 9), MAX3(6, 7, 8) MAX3(9, 10, 11) MAX3(12, 13, 6) MAX3(7, 8, 9), MAX3(6, 7, 8) MAX3(9, 10, 11) MAX3(12, 13, 6) MAX3(7, 8, 9))
        if constexpr (V == 16) BODY(EXP(6) EXP(7) ADD(8) ADD(9), EXP(8) EXP(9) ADD(10) ADD(11), EXP(10) EXP(11) ADD(12) ADD(13), EXP(12) EXP(13) ADD(6) ADD(7))
        if constexpr (V == 20) BODY(LD128(21, 0) LD128(22, 1024), LD128(21, 2048) LD128(22, 3072), LD128(21, 4096) LD128(22, 5120), LD128(21, 6144) LD128(22, 7168) WAITL(0))
        if constexpr (V == 21) BODY(LDTR(26, 0) LDTR(27, 1024), LDTR(26, 2048) LDTR(27, 3072), LDTR(26, 4096) LDTR(27, 5120), LDTR(26, 6144) LDTR(27, 7168) WAITL(0))
        if constexpr (V == 22) BODY(NOP NOP NOP NOP, NOP NOP NOP NOP, NOP NOP NOP NOP, NOP NOP NOP NOP)
        if constexpr (V == 25) BODY(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11) LD128(21, 0), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7) LD128(22, 1024), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7) LDTR(26, 2048), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11) LDTR(27, 3072) WAITL(2))
        if constexpr (V == 32) BODY(LD128(21, 0), LD128(22, 1024), LD128(21, 2048), LD128(22, 3072))
        if constexpr (V == 33) BODY(LD128(21, 0) LD128(22, 1024), LD128(21, 2048) LD128(22, 3072), LD128(21, 4096) LD128(22, 5120), LD128(21, 6144) LD128(22, 7168))
        if constexpr (V == 34) BODY(LDTR(26, 0), LDTR(27, 1024), LDTR(26, 2048), LDTR(27, 3072))
        if constexpr (V == 35) BODY(LDTR(26, 0) LDTR(27, 1024), LDTR(26, 2048) LDTR(27, 3072), LDTR(26, 4096) LDTR(27, 5120), LDTR(26, 6144) LDTR(27, 7168))
        if constexpr (V == 36) BODY(LD128(21, 0) LD128(22, 1024), "", "", "")
        if constexpr (V == 37) BODY(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11) LD128(21, 0) LD128(22, 1024), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11))
        if constexpr (V == 38) BODY(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11) LDTR(26, 0) LDTR(27, 1024), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7) LDTR(26, 2048) LDTR(27, 3072), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11))
        if constexpr (V == 39) BODY(EXP(6) EXP(7) LD128(21, 0) LD128(22, 1024), EXP(8) EXP(9), EXP(10) EXP(11), EXP(12) EXP(13))
        if constexpr (V == 40) BODY(LD128(21, 0) EXP(6) EXP(7) LD128(22, 1024), EXP(8) EXP(9), EXP(10) EXP(11), EXP(12) EXP(13))
        // dependency distance of a pack on its exps (issue sum 8 + 16 + 9 + 4 = 37 in every variant)
        if constexpr (V == 41) BODY(EXP(6) EXP(7) CVT(14, 12, 13) CVT(15, 12, 13) ADD(10), EXP(8) EXP(9) CVT(16, 6, 7) CVT(17, 6, 7) ADD(10), EXP(11) EXP(6) CVT(14, 8, 9) CVT(15, 8, 9) ADD(10), EXP(12) EXP(13) CVT(16, 11, 6) CVT(17, 11, 6) ADD(10))
        if constexpr (V == 42) BODY(EXP(6) EXP(7) CVT(14, 18, 19) CVT(15, 18, 19) ADD(10), EXP(8) EXP(9) CVT(16, 18, 19) CVT(17, 18, 19) ADD(10), EXP(11) EXP(6) CVT(14, 18, 19) CVT(15, 18, 19) ADD(10), EXP(12) EXP(13) CVT(16, 18, 19) CVT(17, 18, 19) ADD(10))
        if constexpr (V == 43) BODY(EXP(6) EXP(7) CVT(14, 6, 7) CVT(15, 6, 7) ADD(10), EXP(8) EXP(9) CVT(16, 8, 9) CVT(17, 8, 9) ADD(10), EXP(11) EXP(12) CVT(14, 11, 12) CVT(15, 11, 12) ADD(10), EXP(13) EXP(6) CVT(16, 13, 6) CVT(17, 13, 6) ADD(10))
        if constexpr (V == 44) BODY(EXP(6) EXP(7) ADD(10) ADD(10) ADD(10), EXP(8) EXP(9) ADD(10) ADD(10) ADD(10), EXP(11) EXP(12) ADD(10) ADD(10) ADD(10), EXP(13) EXP(6) ADD(10) ADD(10) ADD(10))
        if constexpr (V == 45) BODY(EXP(6) EXP(7) MAX3(10, 12, 13) MAX3(10, 12, 13) ADD(10), EXP(8) EXP(9) MAX3(10, 6, 7) MAX3(10, 6, 7) ADD(10), EXP(11) EXP(6) MAX3(10, 8, 9) MAX3(10, 8, 9) ADD(10), EXP(12) EXP(13) MAX3(10, 11, 6) MAX3(10, 11, 6) ADD(10))
        if constexpr (V == 26) BODYA("", "", "", "")
        if constexpr (V == 27) BODYA(EXP(6) EXP(7) CVT(14, 8, 9) CVT(15, 10, 11), EXP(8) EXP(9) CVT(16, 12, 13) CVT(17, 6, 7), EXP(10) EXP(11) CVT(14, 12, 13) CVT(15, 6, 7), EXP(12) EXP(13) CVT(16, 8, 9) CVT(17, 10, 11))
        if constexpr (V == 28) BODY(EXP(6) EXP(7) CVT(14, 12, 13), EXP(8) EXP(9) CVT(15, 6, 7), EXP(10) EXP(11) CVT(16, 8, 9), EXP(12) EXP(13) CVT(17, 10, 11))
        if constexpr (V == 29) BODY(EXP(6) EXP(7) CVT(14, 6, 7), EXP(8) EXP(9) CVT(15, 8, 9), EXP(10) EXP(11) CVT(16, 10, 11), EXP(12) EXP(13) CVT(17, 12, 13))
        if constexpr (V == 30) BODY(EXP(6) EXP(7) NOP CVT(14, 6, 7), EXP(8) EXP(9) NOP CVT(15, 8, 9), EXP(10) EXP(11) NOP CVT(16, 10, 11), EXP(12) EXP(13) NOP CVT(17, 12, 13))
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += (float)w[i] + (float)l0[i] + (float)l1[i] + (float)h0[i & 1] + (float)h1[i & 1];
    s += (float)s0;
    if (s == 12345.678f) out[1] = (uint64_t)s;
    if ((threadIdx.x & 63) == 0) out[8 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;      // every wave's ticks
}

template <int V, int U = 1>
void run(const char* name, uint64_t* out, const uint4* opnd, int cus, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gap_kernel<V, U><<<cus, 256>>>(out, opnd, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gap_kernel<V, U><<<cus, 256>>>(out, opnd, iters / U);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    static uint64_t h[8 + 4 * 1024];
    hipMemcpy(h, out, (8 + 4 * cus) * 8, hipMemcpyDeviceToHost);
    uint64_t mn = ~0ull, mx = 0; double sum = 0;
    for (int i = 0; i < 4 * cus; ++i) { mn = h[8 + i] < mn ? h[8 + i] : mn; mx = h[8 + i] > mx ? h[8 + i] : mx; sum += (double)h[8 + i]; }
    const double g = 4.0 * iters;
    printf("%-46s ticks/gap min %6.1f avg %6.1f max %6.1f   %6.2f ns/gap  (%.0f MHz if a tick of the slowest wave is a cycle)\n", name,
           mn / g, sum / (4 * cus) / g, mx / g, ms * 1e6 / g, (double)mx / (ms * 1e3));
    if (getenv("GAP_DUMP") && mx > mn + mn / 8) {
        int n = 0;
        printf("    slow waves (block.wave ticks/gap):");
        for (int i = 0; i < 4 * cus; ++i) if (h[8 + i] > mn + mn / 8) { if (n++ < 48) printf(" %d.%d %.1f", i / 4, i % 4, h[8 + i] / g); }
        printf("  [%d of %d]\n", n, 4 * cus);
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const bool zeros = argc > 2 && atoi(argv[2]) == 0;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint64_t* out; uint4* opnd;
    hipMalloc(&out, (8 + 4 * 1024) * 8); hipMalloc(&opnd, 512 * 16);
    uint32_t h[512 * 4];
    srand(1);
    for (int i = 0; i < 512 * 4; ++i) {        // bf16 pairs of moderate magnitude (or zeros)
        const uint32_t lo = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15), hi = 0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15);
        h[i] = zeros ? 0u : (lo | (hi << 16));
    }
    hipMemcpy(opnd, h, sizeof(h), hipMemcpyHostToDevice);
    printf("one wave per SIMD, %d CUs, %d x 4 gaps, operands %s\n", cus, iters, zeros ? "zero" : "random");
    run<0>("MFMA only", out, opnd, cus, iters);
    run<4>("+ 4 v_add_f32", out, opnd, cus, iters);
    run<15>("+ 4 v_max3_f32", out, opnd, cus, iters);
    run<1>("+ 2 v_exp_f32", out, opnd, cus, iters);
    run<8>("+ 3 v_exp_f32", out, opnd, cus, iters);
    run<7>("+ 4 v_exp_f32", out, opnd, cus, iters);
    run<16>("+ 2 v_exp_f32 + 2 v_add_f32", out, opnd, cus, iters);
    run<11>("+ 1 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<2>("+ 2 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<12>("+ 4 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<13>("+ 2 v_cvt_pk_f16_f32", out, opnd, cus, iters);
    run<5>("+ 2 v_perm_b32", out, opnd, cus, iters);
    run<14>("+ 2 v_and_or_b32", out, opnd, cus, iters);
    run<9>("+ 1 v_exp_f32 + 1 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<3>("+ 2 v_exp_f32 + 2 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<6>("+ 2 v_exp_f32 + 2 v_perm_b32", out, opnd, cus, iters);
    run<10>("+ 2 v_exp + 3 v_cvt_pk_bf16 + s_nop 0", out, opnd, cus, iters);
    printf("loop body unrolled 8 x (32 gaps per branch)\n");
    run<0, 8>("MFMA only", out, opnd, cus, iters);
    run<4, 8>("+ 4 v_add_f32", out, opnd, cus, iters);
    run<1, 8>("+ 2 v_exp_f32", out, opnd, cus, iters);
    run<8, 8>("+ 3 v_exp_f32", out, opnd, cus, iters);
    run<2, 8>("+ 2 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<12, 8>("+ 4 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<3, 8>("+ 2 v_exp_f32 + 2 v_cvt_pk_bf16_f32", out, opnd, cus, iters);
    run<6, 8>("+ 2 v_exp_f32 + 2 v_perm_b32", out, opnd, cus, iters);
    run<10, 8>("+ 2 v_exp + 3 v_cvt_pk_bf16 + s_nop 0", out, opnd, cus, iters);
    run<20, 8>("+ 2 ds_read_b128", out, opnd, cus, iters);
    run<21, 8>("+ 2 ds_read_b64_tr_b16", out, opnd, cus, iters);
    run<32, 8>("+ 1 ds_read_b128, no waits", out, opnd, cus, iters);
    run<33, 8>("+ 2 ds_read_b128, no waits", out, opnd, cus, iters);
    run<34, 8>("+ 1 ds_read_b64_tr_b16, no waits", out, opnd, cus, iters);
    run<35, 8>("+ 2 ds_read_b64_tr_b16, no waits", out, opnd, cus, iters);
    run<36, 8>("+ 2 ds_read_b128 in one gap of four", out, opnd, cus, iters);
    run<37, 8>("+ 2 exp + 2 cvt, 2 ds_read_b128 in one gap of four", out, opnd, cus, iters);
    run<38, 8>("+ 2 exp + 2 cvt, 2 ds_read_tr in two gaps of four", out, opnd, cus, iters);
    run<39, 8>("+ 2 exp, 2 ds_read_b128 after the exps in one gap of 4", out, opnd, cus, iters);
    run<40, 8>("+ 2 exp, ds_read_b128 / exps / ds_read_b128 in one gap of 4", out, opnd, cus, iters);
    run<44, 8>("+ 2 exp + 3 add (no dependencies)", out, opnd, cus, iters);
    run<42, 8>("+ 2 exp + 2 cvt of constants + add", out, opnd, cus, iters);
    run<41, 8>("+ 2 exp + 2 cvt of the PREVIOUS gap's exps + add", out, opnd, cus, iters);
    run<43, 8>("+ 2 exp + 2 cvt of THIS gap's exps + add", out, opnd, cus, iters);
    run<45, 8>("+ 2 exp + 2 max3 reading the previous gap's exps + add", out, opnd, cus, iters);
    run<22, 8>("+ 4 s_nop 0", out, opnd, cus, iters);
    run<25, 8>("+ 2 exp + 2 cvt + 1 ds_read", out, opnd, cus, iters);
    run<26, 8>("MFMA only, B operand in AGPRs", out, opnd, cus, iters);
    run<27, 8>("B in AGPRs + 2 exp + 2 cvt", out, opnd, cus, iters);
    run<28, 8>("+ 2 exp + cvt of the previous gap's exps", out, opnd, cus, iters);
    run<29, 8>("+ 2 exp + cvt of THIS gap's exps", out, opnd, cus, iters);
    run<30, 8>("+ 2 exp + s_nop 0 + cvt of this gap's exps", out, opnd, cus, iters);
    return 0;
}
