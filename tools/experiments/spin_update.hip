// Stand-in for the update launch in tools/ab_light_update.py: workgroups that hold a given footprint (threads, LDS bytes, VGPRs) for a
// given time and do nothing — to price what an update workgroup of another SHAPE would do to the step before it is written.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int VGPRS>
__global__ void k_spin(uint32_t ticks /* 100 MHz */, uint32_t* sink) {
    extern __shared__ unsigned char smem[];
    if (VGPRS > 100) asm volatile("v_mov_b32 v150, 0" ::: "v150");
    else asm volatile("v_mov_b32 v78, 0" ::: "v78");
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0xffffffffu) sink[0] = smem[threadIdx.x];
}

extern "C" int spin_launch(int wgs, int threads, int lds, int vgprs, int usec, void* stream) {
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void*)k_spin<151>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_spin<79>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (vgprs > 100) hipLaunchKernelGGL(k_spin<151>, dim3(wgs), dim3(threads), lds, (hipStream_t)stream, (uint32_t)usec * 100u, nullptr);
    else hipLaunchKernelGGL(k_spin<79>, dim3(wgs), dim3(threads), lds, (hipStream_t)stream, (uint32_t)usec * 100u, nullptr);
    return (int)hipGetLastError();
}
