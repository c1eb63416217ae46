// Where does the dispatcher put the waves of a grid that is SMALLER than the chip's capacity?  Every wave records its
// (XCC, SE, CU, SIMD) from the hardware id registers and then spins for ~20 us so that all of the grid is resident at once.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_placement.hip -o tools/micro/wave_placement.bin && tools/micro/wave_placement.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k_place(unsigned int *ids, int lds_words, long long spin) {
    extern __shared__ unsigned int lds[];
    if (lds_words) lds[threadIdx.x % lds_words] = threadIdx.x;
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) ids[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (hw & 0xFFFFu) | ((xcc & 0xFu) << 16);
}
static void run(int blocks, int threads, int lds_bytes, unsigned int *d) {
    const int waves = blocks * threads / 64;
    hipMemset(d, 0xFF, waves * 4);
    hipFuncSetAttribute((const void *)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    k_place<<<blocks, threads, lds_bytes>>>(d, lds_bytes / 4, 2000);  // 100 MHz ticks: 20 us
    hipDeviceSynchronize();
    unsigned int *h = new unsigned int[waves];
    hipMemcpy(h, d, waves * 4, hipMemcpyDeviceToHost);
    static int per_simd[16 * 8 * 2 * 16 * 4];  // xcc, se, sh, cu, simd
    memset(per_simd, 0, sizeof(per_simd));
    for (int i = 0; i < waves; ++i) {
        const unsigned v = h[i];
        const int simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 15;
        per_simd[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd]++;
    }
    int hist[40] = {0}, used = 0;
    for (int i = 0; i < 16 * 8 * 2 * 16 * 4; ++i)
        if (per_simd[i]) { ++used; hist[per_simd[i] < 39 ? per_simd[i] : 39]++; }
    printf("%5d workgroups x %4d threads, %6d B LDS: %5d waves on %4d SIMDs; SIMDs holding k waves:", blocks, threads, lds_bytes, waves, used);
    for (int k = 1; k < 40; ++k) if (hist[k]) printf("  %d: %d", k, hist[k]);
    printf("\n");
    delete[] h;
}
int main() {
    unsigned int *d;
    hipMalloc(&d, 1 << 20);
    run(2048, 64, 4864, d);
    run(1024, 64, 4864, d);
    run(4096, 64, 4864, d);
    run(512, 256, 18600, d);
    run(1024, 256, 18600, d);
    run(256, 256, 18600, d);
    run(256, 1024, 65536, d);
    run(512, 512, 70000, d);
    run(2048, 64, 0, d);
    return 0;
}
