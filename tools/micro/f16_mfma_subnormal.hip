// Does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL inputs on gfx950, and how fast is it next to the bf16 form?
// (decides whether f32 products can run as 2-way f16 splits: the low part of a value below 0.125 is an f16 subnormal)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f16_mfma_subnormal.hip -o /tmp/f16sub && /tmp/f16sub
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_sub(const float *av, const float *bv, float *out, int n) {
    for (int i = 0; i < n; ++i) {
        h8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)av[i]; b[k] = (_Float16)bv[i]; }
        f4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        if (threadIdx.x == 0) out[i] = c[0];
    }
}
template <int F16>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
    f4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 ah, bh; b8 ab, bb;
    for (int k = 0; k < 8; ++k) { ah[k] = (_Float16)(threadIdx.x * 0.001f); bh[k] = (_Float16)1.0f; ab[k] = (__bf16)(threadIdx.x * 0.001f); bb[k] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (F16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int n = 8;
    // f16 min normal 2^-14 = 6.1e-5; subnormals down to 2^-24
    float a[n] = {ldexpf(1.f, -20), ldexpf(1.f, -24), ldexpf(1.5f, -16), 1.0f, ldexpf(1.f, -14), ldexpf(1.f, -20), 3.0f, ldexpf(1.f, -10)};
    float b[n] = {1.0f, 1.0f, 2.0f, ldexpf(1.f, -20), 1.0f, ldexpf(1.f, -4), ldexpf(1.f, -24), ldexpf(1.f, -10)};
    float *da, *db, *dout, out[n];
    hipMalloc(&da, sizeof(a)); hipMalloc(&db, sizeof(b)); hipMalloc(&dout, 256 * 1024 * 4);
    hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice);
    k_sub<<<1, 64>>>(da, db, dout, n);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < n; ++i) {
        const double want = 32.0 * (double)a[i] * (double)b[i];
        printf("a=%.6e b=%.6e  mfma=%.9e  exact=%.9e  %s\n", a[i], b[i], out[i], want, out[i] == (float)want ? "ok" : "DIFFERENT");
        if (out[i] != (float)want) ok = 0;
    }
    printf("F16_SUBNORMALS_HONOURED=%d\n", ok);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f16 = 0; f16 < 2; ++f16) {
        const int iters = 20000, blocks = 1024;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) k_rate<1><<<blocks, 256>>>(dout, iters); else k_rate<0><<<blocks, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
        printf("%s 16x16x32: %.3f ms, %.1f TFLOP/s\n", f16 ? "f16 " : "bf16", ms, flop / ms / 1e9);
    }
    return 0;
}
