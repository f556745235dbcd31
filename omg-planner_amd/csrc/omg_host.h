// omg_host.h — host-side helpers shared by the translation units of libomg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/omg_hip.h"

// Records "<what>: <hip error string>" for omgx_last_error() and returns OMGX_ERR_LAUNCH.
int omgx_set_error(const char* what, hipError_t e);

#define OMGX_CHECK_LAUNCH(what)                                \
    do {                                                       \
        hipError_t e_ = hipGetLastError();                     \
        if (e_ != hipSuccess) return omgx_set_error(what, e_); \
    } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS (gfx950: 160 KB per workgroup).
// Once per (kernel, device): the attribute belongs to the device's copy of the function, and entry points may be called from
// several host threads (one bit per device in an atomic word; a lost race only repeats the idempotent call).
template <int TAG, class K>  // TAG: one flag word per kernel instantiation (instantiations of one template share the type K)
static int allow_big_lds(K kernel, const char* what) {
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return omgx_set_error("hipGetDevice", e);
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return OMGX_OK;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return omgx_set_error(what, e);
    if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return OMGX_OK;
}

