#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *out, int iters) {
    float a = threadIdx.x, b = 1.0001f;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; a = a * b + 0.25f; a = a * b + 0.125f; a = a * b + 0.0625f; }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)a; }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    for (int rep = 0; rep < 6; ++rep) {
        int iters = rep < 3 ? 20000 : 2000000;
        k<<<1024, 256>>>(d, iters);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("iters %d: clock64 %llu ticks, wall %llu (100MHz) = %.1f us -> clock64 rate %.1f MHz; dependent fma chain %.2f ticks/fma\n", iters, h[0], h[1], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / (4.0 * iters));
    }
    return 0;
}
