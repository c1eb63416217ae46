// Sustained rate of the f16 matrix instructions in a bare loop, by shape, operand data and waves per SIMD (MI355X):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rates.hip -o /tmp/mfma_rates && /tmp/mfma_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE, int ZERO>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = ZERO ? (_Float16)0.0f : (_Float16)(0.37f + threadIdx.x * 0.001f + k * 0.01f); b[k] = ZERO ? (_Float16)0.0f : (_Float16)(1.0f - k * 0.03f); }
    float s = 0.f;
    if (SHAPE == 0) {
        f4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));   // (inline: hipcc's own loop shuffles the accumulators through AGPR copies, ~50 extra instructions per 8 MFMAs)
        for (int j = 0; j < 8; ++j) s += acc[j][0];
    } else {
        f16v acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        for (int j = 0; j < 4; ++j) s += acc[j][0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE, int ZERO>
static void run(const char *name, float *d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k_rate<SHAPE, ZERO><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double per = SHAPE == 0 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;
    printf("%-34s %4d workgroups of 4 waves: %7.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, (double)blocks * 4 * iters * per / ms / 1e9);
}
int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0, 0>("16x16x32 f16, random-ish operands", d, blocks);
        run<1, 0>("32x32x16 f16, random-ish operands", d, blocks);
    }
    run<0, 1>("16x16x32 f16, zero operands", d, 1024);
    run<1, 1>("32x32x16 f16, zero operands", d, 1024);
    return 0;
}
