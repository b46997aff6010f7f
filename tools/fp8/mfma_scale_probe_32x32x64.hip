// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (gfx950), as mfma_scale_probe.hip did for the
// 16x16x128 form: one wave computes D = A.B^T for a 32x64 A and a 32x64 B (fp8 bytes, small integers) with per-(row, 32-k
// block) e8m0 scales; the host checks which bytes / which scale a lane must hold and the C/D register map.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_scale_probe_32x32x64.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// variant 0: lane l holds row l&31, bytes k = 32*(l>>5) + [0,32)
// variant 1: lane l holds row l&31, bytes k = 16*(l>>5) + [0,16) and 32 + 16*(l>>5) + [0,16)
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sA, const uint8_t* sB, float* D, int variant) {
    const int l = threadIdx.x, row = l & 31, g = l >> 5;
    uint8_t a[32], b[32];
    for (int j = 0; j < 32; ++j) {
        const int k = variant == 0 ? 32 * g + j : (j < 16 ? 16 * g + j : 32 + 16 * g + (j - 16));
        a[j] = A[row * 64 + k];
        b[j] = B[row * 64 + k];
    }
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = a[4 * i] | (a[4 * i + 1] << 8) | (a[4 * i + 2] << 16) | (a[4 * i + 3] << 24);
        bv[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | (b[4 * i + 3] << 24);
    }
    const int sa = sA[row * 2 + g], sb = sB[row * 2 + g];
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    uint8_t hA[32 * 64], hB[32 * 64], hsA[64], hsB[64];
    const uint8_t vals[6] = {0x00, 0x38, 0x40, 0x44, 0xB8, 0x30};   // 0, 1, 2, 3, -1, 0.5
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = vals[rand() % 6]; hB[i] = vals[rand() % 6]; }
    for (int i = 0; i < 64; ++i) { hsA[i] = 126 + rand() % 3; hsB[i] = 127 + rand() % 2; }
    uint8_t *dA, *dB, *dsA, *dsB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsA, hsA, 64, hipMemcpyHostToDevice); hipMemcpy(dsB, hsB, 64, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        float hD[1024];
        probe<<<1, 64>>>(dA, dB, dsA, dsB, dD, variant);
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        for (int blkmode = 0; blkmode < 2; ++blkmode) {       // 0: blk = k/32; 1: blk = (k%32)/16 (variant-1 grouping)
            for (int cd = 0; cd < 2; ++cd) {                  // C/D: lane (c = l&31, h = l>>5), reg j: r = (j&3) + 8(j>>2) + 4h; 0: D[m=r][n=c], 1: D[m=c][n=r]
                double err = 0, mag = 0;
                for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
                    const int c = l & 31, r = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5);
                    const int m = cd == 0 ? r : c, n = cd == 0 ? c : r;
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) {
                        const int blk = blkmode == 0 ? k / 32 : (k % 32) / 16;
                        ref += (double)e4m3(hA[m * 64 + k]) * e4m3(hB[n * 64 + k]) * ldexp(1.0, hsA[m * 2 + blk] - 127) *
                               ldexp(1.0, hsB[n * 2 + blk] - 127);
                    }
                    err += fabs(ref - hD[l * 16 + j]); mag += fabs(ref);
                }
                printf("operand variant %d, scale-block mode %d, C/D map %d: sum|err| = %.3f (sum|ref| = %.1f)%s\n", variant,
                       blkmode, cd, err, mag, err < 1e-3 ? "   <-- MATCH" : "");
            }
        }
    }
    return 0;
}
