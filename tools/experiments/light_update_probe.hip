// Probe for tools/ab_light_update.py: a stand-in for a LIGHT update launch — workgroups of 256 threads with a small LDS
// footprint that just stay resident for a given time (a dependent chain, like the real update's latency-bound phases).
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/light_update_probe.hip -o tools/_build/liblight_update_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_linger(double* sink, int ticks_100mhz, int lds_doubles) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x) lds[i] = (double)i;
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc = lds[threadIdx.x % (lds_doubles > 0 ? lds_doubles : 1)];
    while (wall_clock64() - t0 < ticks_100mhz) {
#pragma unroll 1
        for (int k = 0; k < 64; ++k) acc = acc * 1.0000001 + 1e-9;  // dependent f64 chain: low issue pressure, like the update
    }
    if (acc == 123.456) sink[blockIdx.x] = acc;
}

extern "C" int probe_linger(double* sink, int wgs, int micros, int lds_bytes, void* stream) {
    if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linger), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_linger, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, sink, micros * 100, lds_bytes / 8);
    return (int)hipGetLastError();
}
