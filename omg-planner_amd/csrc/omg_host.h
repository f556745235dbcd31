// omg_host.h — host-side helpers shared by the translation units of libomg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/omg_hip.h"

// Records "<what>: <hip error string>" for omgx_last_error() and returns OMGX_ERR_LAUNCH.
int omgx_set_error(const char* what, hipError_t e);

#define OMGX_CHECK_LAUNCH(what)                                \
    do {                                                       \
        hipError_t e_ = hipGetLastError();                     \
        if (e_ != hipSuccess) return omgx_set_error(what, e_); \
    } while (0)
