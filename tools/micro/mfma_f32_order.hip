// In which order does v_mfma_f32_16x16x4_f32 add its four products to the accumulator on gfx950 -- and is every step a fused
// multiply-add rounded to f32?  (decides whether the response layer, whose summation order the oracle fixes as a chain of fmaf
// calls, can run on the matrix pipe bit for bit)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mfma_f32_order.hip -o tools/micro/mfma_order.bin && tools/micro/mfma_order.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// one wave: A [16][4], B [4][16], C [16][16] -> D; lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], C/D rows 4 (l >> 4) + r, column l & 15
__global__ void k_one(const float *A, const float *B, const float *C, float *D, int n) {
    const int l = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        const float a = A[i * 64 + (l & 15) * 4 + (l >> 4)], b = B[i * 64 + (l >> 4) * 16 + (l & 15)];
        f4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[i * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[i * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
    }
}
template <int WHICH>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
    f4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    float side[8] = {0.f, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WHICH == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            else if (WHICH == 2) {  // both: one matrix instruction and eight scalar FMAs on other registers per step
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) side[r] = fmaf(a, b, side[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] = fmaf(a, b, acc[j][r]);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + side[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static uint32_t rng = 12345u;
static float rnd(int mode) {
    rng = rng * 1664525u + 1013904223u;
    const float u = (float)(rng >> 8) / 16777216.0f * 2.0f - 1.0f;
    rng = rng * 1664525u + 1013904223u;
    if (mode == 0) return u;
    const int e = (int)(rng >> 27) - 16;   // wide exponent range: cancellation and absorbed terms
    return ldexpf(u, e);
}
int main() {
    const int n = 4096;
    float *A = new float[n * 64], *B = new float[n * 64], *C = new float[n * 256], *D = new float[n * 256];
    for (int i = 0; i < n * 64; ++i) { A[i] = rnd(i & 1); B[i] = rnd((i >> 1) & 1); }
    for (int i = 0; i < n * 256; ++i) C[i] = rnd(i & 1);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, n * 64 * 4); hipMalloc(&dB, n * 64 * 4); hipMalloc(&dC, n * 256 * 4); hipMalloc(&dD, n * 256 * 4);
    hipMemcpy(dA, A, n * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B, n * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C, n * 256 * 4, hipMemcpyHostToDevice);
    k_one<<<1, 64>>>(dA, dB, dC, dD, n);
    hipMemcpy(D, dD, n * 256 * 4, hipMemcpyDeviceToHost);
    // candidate orders: every permutation of k = 0..3 as a chain of fmaf from C; plus one rounding of the exact sum (long double)
    int perm[24][4], np = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) for (int d = 0; d < 4; ++d)
        if (a != b && a != c && a != d && b != c && b != d && c != d) { perm[np][0] = a; perm[np][1] = b; perm[np][2] = c; perm[np][3] = d; ++np; }
    long long miss[24] = {0}, miss_once = 0, total = 0;
    for (int i = 0; i < n; ++i)
        for (int m = 0; m < 16; ++m)
            for (int c = 0; c < 16; ++c) {
                const float got = D[i * 256 + m * 16 + c];
                ++total;
                for (int p = 0; p < 24; ++p) {
                    float acc = C[i * 256 + m * 16 + c];
                    for (int s = 0; s < 4; ++s) { const int k = perm[p][s]; acc = fmaf(A[i * 64 + m * 4 + k], B[i * 64 + k * 16 + c], acc); }
                    if (memcmp(&acc, &got, 4)) ++miss[p];
                }
                long double e = C[i * 256 + m * 16 + c];
                for (int k = 0; k < 4; ++k) e += (long double)A[i * 64 + m * 4 + k] * (long double)B[i * 64 + k * 16 + c];
                const float once = (float)e;
                if (memcmp(&once, &got, 4)) ++miss_once;
            }
    printf("%lld results\n", total);
    for (int p = 0; p < 24; ++p)
        if (miss[p] * 20 < total || p == 0 || p == 23) printf("fmaf chain k = %d %d %d %d : %lld differ\n", perm[p][0], perm[p][1], perm[p][2], perm[p][3], miss[p]);
    printf("one rounding of the exact sum: %lld differ\n", miss_once);
    printf("MFMA_F32_IS_ASCENDING_FMAF_CHAIN=%d\n", miss[0] == 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float *dout; hipMalloc(&dout, 2048 * 256 * 4);
    for (int w = 0; w < 6; ++w) {
        // w >= 2: the matrix instruction alone with 1, 2, 4, 8 waves per SIMD (a wave keeps 8 independent accumulators)
        const int iters = 20000, blocks = w < 2 ? 1024 : 256 << (w - 2);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (w != 1) k_rate<0><<<blocks, 256>>>(dout, iters); else k_rate<1><<<blocks, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop = (double)blocks * 4 * iters * 8 * (w != 1 ? 2.0 * 16 * 16 * 4 : 2.0 * 4 * 64);
        printf("%s, %d workgroups of 4 waves: %.1f TFLOP/s\n", w != 1 ? "v_mfma_f32_16x16x4_f32" : "v_fma_f32 (4 per lane)", blocks, flop / (ms * 1e-3) / 1e12);
    }
    // do the matrix instruction and VALU FMAs of the same SIMD overlap?  8 + 64 per step against 8 + 0
    for (int w = 0; w < 2; ++w) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (w == 0) k_rate<0><<<1024, 256>>>(dout, 20000); else k_rate<2><<<1024, 256>>>(dout, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.2f ms\n", w == 0 ? "8 v_mfma per step" : "8 v_mfma + 64 v_fma_f32 per step (256 cycles of VALU beside 256 of MFMA)", ms);
    }
    return 0;
}
