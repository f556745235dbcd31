// ref_helpers_wrapper.hip — C entry points around the REFERENCE's own interpolation helpers.
//
// TEST INFRASTRUCTURE ONLY.  The file REF_HELPERS_INC is produced at build time by oracle/Makefile as lines 15-86
// of /root/reference/layers/sdf_matching_loss_kernel.cu (lerp, float3 +/-, getValue, getValueInterpolated,
// getGradientInterpolated: all `__device__ __host__`, no ATen / Eigen / Sophus), compiled here for the HOST.
// Nothing of it is stored in the repository; only the resulting oracle/_ref/libsdf_ref_helpers.so exists (git-ignored).
// The kernel body (.cu:96-181) and launcher need ATen + Eigen + Sophus + nvcc and stay unbuildable.
#include <hip/hip_runtime.h>

#include REF_HELPERS_INC

extern "C" float ref_value_interpolated(float gx, float gy, float gz, int dx, int dy, int dz, const float* grid) {
    return getValueInterpolated<float>(make_float3(gx, gy, gz), make_int3(dx, dy, dz), grid);
}

extern "C" void ref_gradient_interpolated(float gx, float gy, float gz, int dx, int dy, int dz, const float* grid, float delta,
                                          float* out3) {
    const float3 g = getGradientInterpolated<float>(make_float3(gx, gy, gz), make_int3(dx, dy, dz), grid, delta);
    out3[0] = g.x; out3[1] = g.y; out3[2] = g.z;
}
