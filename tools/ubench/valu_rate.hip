// Micro-benchmark (diagnostic, standalone: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate): what a SIMD pays per
// vector instruction when W waves share it (W = 1, 2), for the instruction mixes of the attention softmax: v_exp_f32 alone,
// v_exp_f32 alternating with a full-rate instruction, v_cvt_pk_fp8_f32, v_max3_f32, v_mul_f32.  256 blocks x (W x 256)
// threads, every wave runs `iters` x 32 instructions on 8 independent registers; prints shader cycles (s_memtime) per
// instruction PER SIMD (= per wave / W when the waves interleave perfectly).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define EXP(I) "v_exp_f32 %" #I ", %" #I "\n\t"
#define MUL(I) "v_mul_f32 %" #I ", %" #I ", %" #I "\n\t"
#define MAX3(I) "v_max3_f32 %" #I ", %" #I ", %" #I ", %" #I "\n\t"
#define CVT(I) "v_cvt_pk_fp8_f32 %" #I ", %" #I ", %" #I "\n\t"
#define EXPMUL(I) "v_exp_f32 %" #I ", %" #I "\n\tv_mul_f32 %8, %8, %8\n\t"
#define EXPMUL2(I) "v_exp_f32 %" #I ", %" #I "\n\tv_mul_f32 %8, %8, %8\n\tv_mul_f32 %9, %9, %9\n\t"
#define EXPMUL3(I) "v_exp_f32 %" #I ", %" #I "\n\tv_mul_f32 %8, %8, %8\n\tv_mul_f32 %9, %9, %9\n\tv_mul_f32 %8, %8, %8\n\t"

template <int V>
__global__ void k(uint64_t* out, int iters) {
    float x[8], y = 1.0f, z = 1.0f;
    for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i + 1);
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
    asm volatile("" : "+v"(y), "+v"(z));
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define BODY(OP) asm volatile(R8(OP) R8(OP) R8(OP) R8(OP) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(y), "+v"(z));
        if (V == 0) BODY(EXP)
        if (V == 1) BODY(MUL)
        if (V == 2) BODY(MAX3)
        if (V == 3) BODY(CVT)
        if (V == 4) BODY(EXPMUL)
        if (V == 5) BODY(EXPMUL2)
        if (V == 6) BODY(EXPMUL3)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = y + z;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.f) out[1] = 1;
    if (blockIdx.x == 7 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int V>
void run(const char* name, int per_group, uint64_t* d) {
    for (int w = 1; w <= 2; ++w) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * w), 0, 0, d, iters);
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * w), 0, 0, d, iters);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * w), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        uint64_t h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double groups = 32.0 * iters;
        printf("%-44s waves/SIMD %d: %7.2f s_memtime ticks per group of %d per wave, %7.2f per SIMD; wall %7.2f ns per group per SIMD\n",
               name, w, h[0] / groups, per_group, h[0] / groups / w, ms * 1e6 / groups / w);
    }
}

int main() {
    uint64_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run<0>("v_exp_f32", 1, d);
    run<1>("v_mul_f32", 1, d);
    run<2>("v_max3_f32", 1, d);
    run<3>("v_cvt_pk_fp8_f32", 1, d);
    run<4>("v_exp_f32 + 1 v_mul_f32", 2, d);
    run<5>("v_exp_f32 + 2 v_mul_f32", 3, d);
    run<6>("v_exp_f32 + 3 v_mul_f32", 4, d);
    return 0;
}
