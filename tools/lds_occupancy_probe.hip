// How many 256-thread workgroups with D bytes of dynamic LDS does a CU admit?  (API answer + a census kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_probe(unsigned* counters, unsigned long long spin) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = ((xcc & 7) << 8) | ((hw >> 8) & 0xf) | (((hw >> 13) & 7) << 4);  // xcc | se | cu
        const unsigned now = atomicAdd(&counters[cu], 1u) + 1;
        atomicMax(&counters[4096 + cu], now);
        lds[0] = (char)now;
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
        atomicSub(&counters[cu], 1u);
    }
    __syncthreads();
}
int main() {
    unsigned* d; hipMalloc(&d, 8192 * 4);
    for (int lds = 24 * 1024; lds <= 42 * 1024; lds += 512) {
        int api = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_probe, 256, lds);
        hipMemset(d, 0, 8192 * 4);
        hipLaunchKernelGGL(k_probe, dim3(256 * 12), dim3(256), lds, 0, d, 20000ull);  // 200 us spin
        hipDeviceSynchronize();
        unsigned h[8192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        unsigned mx = 0; for (int i = 4096; i < 8192; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("dynamic LDS %6d B: API %d blocks/CU, census max %u resident per CU\n", lds, api, mx);
    }
    return 0;
}
