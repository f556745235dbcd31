// hold_probe.hip — experiment aid (tools/ab_noupdate.py): N workgroups of 512 threads with `lds` bytes of dynamic LDS that do nothing
// but sleep for `ms` milliseconds (100 MHz wall clock): stand-ins for resident update-server workgroups, to price the goal-set
// kernel's loss of CU slots before building anything.   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_build/libhold.so tools/hold_probe.hip
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void k_hold(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[(threadIdx.x + 1) & 511] == -1) sink[0] = 1;
}
extern "C" int hold_launch(int n, int lds, int ms, void* stream) {
    hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static int* sink = nullptr;
    if (!sink) hipMalloc(&sink, 4);
    hipLaunchKernelGGL(k_hold, dim3(n), dim3(512), lds, (hipStream_t)stream, (unsigned long long)ms * 100000ull, sink);
    return (int)hipGetLastError();
}
