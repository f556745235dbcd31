// omg_device.h — device-side building blocks of the gfx950 CHOMP engine.
//
//   * SDF lookup: trilinear value + one-voxel central-difference gradient + hinge potential for one
//     (point, object) pair — the body of SDFdistanceForward, layers/sdf_matching_loss_kernel.cu:96-181
//     and its helpers (.cu:15-86).  The float32 arithmetic is written operation for operation like
//     oracle/omg_oracle.c (explicit fmaf, -ffp-contract=off) so results are bit-identical to the oracle.
//   * Panda forward kinematics in double — ycb_render/robotPose/robot_pykdl.py:148-215 re-associated
//     around host-precomputed constants (see RobotView).
//
// gfx950 only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omg_hip.h"

#define OMG_WAVE 64

namespace omg {

// -------------------------------------------------------------------------------------------------
// SDF lookup
// -------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) F2 { float a, b; };        // dword-aligned 8-byte row piece
struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };  // dword-aligned 16-byte row piece

__device__ __forceinline__ float lerpf(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }  // .cu:15-18

// getValueInterpolated's split of one grid coordinate (.cu:39-41): the reference evaluates
//   s = (double)g - 0.5;  i0 = (int)s (truncation toward zero);  f = (float)(s - i0)
// The same result is obtained in float arithmetic for every g whose grid is smaller than 2^22 voxels
// per axis (proof sketch: for g >= 0.25 the float subtraction g - 0.5f is exact; for 0 <= g < 0.25 both
// give i0 = 0 and f = fl32(g - 0.5); for g <= -0.5 both are out of range; the single discrepancy is
// g in (-0.5, -0.5 + 2^-25) where fl32(g - 0.5) rounds to -1.0 although the exact s > -1 truncates to 0
// with f = fl32(s) = -1.0 — patched explicitly).  Checked exhaustively near the critical values in
// tests/test_axis_split.py.  `ok` is false where the (int) cast would be undefined (|s| >= 1e9, NaN):
// the oracle returns 1.0 there.
struct Axis { int i0; float f; bool ok; };
__device__ __forceinline__ Axis axis_of(float g) {
    Axis a;
    a.ok = (g > -1.0e9f) && (g <= 1.0e9f);  // == (g-0.5 in (-1e9,1e9)) for float g; false for NaN
    const float s = g - 0.5f;
    a.i0 = (int)s;
    a.f = s - (float)a.i0;
    const bool edge = (s == -1.0f) && (g > -0.5f);
    a.i0 = edge ? 0 : a.i0;
    a.f = edge ? -1.0f : a.f;
    return a;
}

__device__ __forceinline__ float trilerp(float v000, float v001, float v010, float v011, float v100, float v101,
                                         float v110, float v111, float fx, float fy, float fz) {
    const float dx00 = lerpf(v000, v100, fx);
    const float dx01 = lerpf(v001, v101, fx);
    const float dx10 = lerpf(v010, v110, fx);
    const float dx11 = lerpf(v011, v111, fx);
    const float dxy0 = lerpf(dx00, dx10, fy);
    const float dxy1 = lerpf(dx01, dx11, fy);
    return lerpf(dxy0, dxy1, fz);
}

struct Grid {
    const float* __restrict__ g;
    int dx, dy, dz;
    __device__ __forceinline__ int idx(int x, int y, int z) const { return (x * dy + y) * dz + z; }
};

// getValueInterpolated (.cu:36-64) from pre-split axes: generic path, own range test, 1.0 outside.
__device__ __forceinline__ float sdf_value(const Grid& G, const Axis& ax, const Axis& ay, const Axis& az) {
    const bool in = ax.ok && ay.ok && az.ok && ax.i0 >= 0 && ax.i0 < G.dx - 1 && ay.i0 >= 0 && ay.i0 < G.dy - 1 &&
                    az.i0 >= 0 && az.i0 < G.dz - 1;
    if (!in) return 1.0f;
    const int b = G.idx(ax.i0, ay.i0, az.i0);
    const int sy = G.dz, sx = G.dy * G.dz;
    const F2 r00 = *reinterpret_cast<const F2*>(G.g + b);
    const F2 r01 = *reinterpret_cast<const F2*>(G.g + b + sy);
    const F2 r10 = *reinterpret_cast<const F2*>(G.g + b + sx);
    const F2 r11 = *reinterpret_cast<const F2*>(G.g + b + sx + sy);
    return trilerp(r00.a, r00.b, r01.a, r01.b, r10.a, r10.b, r11.a, r11.b, ax.f, ay.f, az.f);
}

struct ObjParams {  // wave-uniform (SGPR-resident) per-object parameters
    float T[12];    // inverse pose rows
    float lo[3], hi[3];
    int dim[3];
    float delta, eps, pad, clr;
    double rw[3];   // 1 / (double)(float)(hi - lo): t / w == (float)((double)t * rw) exactly (see pair_exact)
    float rc[3], rh[3], rr2;  // influence region (rounded box) in offset-from-lo coordinates, see rbox_inside
    double rdelta;         // 1 / (double)delta
    float i2eps, ieps;     // 1.0f / (2.0f * eps), 1.0f / eps in float32
};

// The influence region of an object (include/omg_hip.h: rb_c, rb_h, rb_r2): a rounded box.  R = 0 makes it a plain box.
// NaN offsets give d = 0 (fmaxf drops the NaN) and pass — the exact path then rejects them through its ordered comparisons,
// like the oracle, which returns 1.0 for them; infinite offsets fail unless the region itself is infinite.
__device__ __forceinline__ bool rbox_inside(float tx, float ty, float tz, const float* c, const float* h, float r2) {
    const float dx = __builtin_fmaxf(__builtin_fabsf(tx - c[0]) - h[0], 0.0f);
    const float dy = __builtin_fmaxf(__builtin_fabsf(ty - c[1]) - h[1], 0.0f);
    const float dz = __builtin_fmaxf(__builtin_fabsf(tz - c[2]) - h[2], 0.0f);
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= r2;
}
// ... grown by a ball of radius `rad` (row-level culling: a link's bounding ball around its frame origin)
__device__ __forceinline__ bool rbox_near(float ux, float uy, float uz, float rad, const float* c, const float* h, float r) {
    const float dx = __builtin_fmaxf(__builtin_fabsf(ux - c[0]) - h[0], 0.0f);
    const float dy = __builtin_fmaxf(__builtin_fabsf(uy - c[1]) - h[1], 0.0f);
    const float dz = __builtin_fmaxf(__builtin_fabsf(uz - c[2]) - h[2], 0.0f);
    const float rr = r + rad;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= rr * rr * 1.000001f;
}

// influence region of an object from its limits (used where the record does not carry it: the raw-tensor API): the grid
// with 1.5 voxels of slack
__device__ __forceinline__ void derive_far_box(ObjParams& o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = o.hi[k] - o.lo[k];
        const bool ok = w > 0.0f && o.dim[k] > 0;
        const float vox = w / (float)(o.dim[k] > 0 ? o.dim[k] : 1);
        o.rc[k] = ok ? 0.5f * w : 0.0f;
        o.rh[k] = ok ? 0.5f * w + 1.5f * vox : __builtin_inff();
    }
    o.rr2 = 0.0f;
    o.rdelta = 1.0 / (double)o.delta;
    o.i2eps = 1.0f / (2.0f * o.eps);
    o.ieps = 1.0f / o.eps;
}

struct Accum { float pot, gx, gy, gz, col; };

// One (point, object) pair: body of SDFdistanceForward (.cu:111-180); adds into acc.
//
// Work is ordered so that the common cases retire early (the kernel is VALU-bound, profiles/r01a_*):
//   1. conservative far-reject in object space (1.5 voxels of slack on a half-voxel requirement): the
//      centre lookup is provably out of range -> value 1.0 -> nothing to add (when eps < 1, clr <= 1);
//   2. exact grid coordinates (IEEE division, double -0.5 split) and the centre trilinear value;
//   3. only when a gradient is requested AND value <= eps: the six +-1-voxel lookups.
// WANT_GRAD=false (goal-set cost: Learner.cost_vector never reads batch_obstacle_cost's gradient,
// online_learner.py:134-148) therefore needs 4 row loads per in-range pair instead of the reference's 56.
struct PairPrep { float tx, ty, tz; bool far; };

// Step 1 of a pair: object-space offset from the grid's min corner + the conservative far test (influence region).
__device__ __forceinline__ PairPrep pair_prepare(const ObjParams& o, float px, float py, float pz) {
    const float* T = o.T;
    // SE3(pose) * point (.cu:125-133)
    const float ux = __builtin_fmaf(T[2], pz, __builtin_fmaf(T[1], py, __builtin_fmaf(T[0], px, T[3])));
    const float uy = __builtin_fmaf(T[6], pz, __builtin_fmaf(T[5], py, __builtin_fmaf(T[4], px, T[7])));
    const float uz = __builtin_fmaf(T[10], pz, __builtin_fmaf(T[9], py, __builtin_fmaf(T[8], px, T[11])));
    PairPrep r;
    r.tx = ux - o.lo[0]; r.ty = uy - o.lo[1]; r.tz = uz - o.lo[2];
    // an out-of-range lookup returns 1.0, which adds nothing when eps < 1 and clr <= 1 (wave-uniform)
    const bool inside = rbox_inside(r.tx, r.ty, r.tz, o.rc, o.rh, o.rr2);
    r.far = (o.eps < 1.0f && o.clr <= 1.0f) ? !inside : false;
    return r;
}

// Steps 2-3 of a pair that was not far-rejected.
template <bool WANT_GRAD>
__device__ __forceinline__ void pair_exact(const ObjParams& o, const float* __restrict__ grid, float tx, float ty, float tz,
                                           Accum& acc) {
    const float* T = o.T;
    // grid coordinates (.cu:137-142): (u - lo) / (hi - lo) * dim.  The IEEE float quotient t / w is
    // obtained as (float)((double)t * rw) with rw = fl64(1 / w): the double product is within 2^-52 of
    // t / w while a quotient of two 24-bit floats is never closer than 2^-49 (relative) to a float
    // rounding boundary, so the final rounding is the correct one — 3 instructions instead of the
    // ~12-instruction v_div_scale/v_rcp/v_fma/v_div_fmas/v_div_fixup sequence.
    const float gx = (float)((double)tx * o.rw[0]) * (float)o.dim[0];
    const float gy = (float)((double)ty * o.rw[1]) * (float)o.dim[1];
    const float gz = (float)((double)tz * o.rw[2]) * (float)o.dim[2];
    Grid G{grid, o.dim[0], o.dim[1], o.dim[2]};
    const Axis ax = axis_of(gx), ay = axis_of(gy), az = axis_of(gz);
    const bool in_c = ax.ok && ay.ok && az.ok && ax.i0 >= 0 && ax.i0 < G.dx - 1 && ay.i0 >= 0 && ay.i0 < G.dy - 1 &&
                      az.i0 >= 0 && az.i0 < G.dz - 1;
    if (!in_c) {  // centre lookup is out of range -> 1.0 (.cu:49-50)
        if (1.0f < o.clr) acc.col += 1.0f;
        if (!(1.0f <= o.eps)) return;
    }
    const int sy = G.dz, sx = G.dy * G.dz;
    const int b = G.idx(ax.i0, ay.i0, az.i0);

    if (!WANT_GRAD) {  // branch-free: clamped (always valid) addresses + selects
        const int bb = in_c ? b : 0;
        const F2 r00 = *reinterpret_cast<const F2*>(G.g + bb);
        const F2 r01 = *reinterpret_cast<const F2*>(G.g + bb + sy);
        const F2 r10 = *reinterpret_cast<const F2*>(G.g + bb + sx);
        const F2 r11 = *reinterpret_cast<const F2*>(G.g + bb + sx + sy);
        const float tv = trilerp(r00.a, r00.b, r01.a, r01.b, r10.a, r10.b, r11.a, r11.b, ax.f, ay.f, az.f);
        const float value = in_c ? tv : 1.0f;
        acc.col += (in_c && value < o.clr) ? 1.0f : 0.0f;                       // .cu:150-151
        const float p_in = (float)(-(double)value + 0.5 * (double)o.eps);        // .cu:158-160
        const float d = value - o.eps;
        const float p_band = o.i2eps * d * d * o.pad;                           // .cu:165-167
        acc.pot += value <= 0.0f ? p_in : (value <= o.eps ? p_band : 0.0f);
        return;
    }

    // Interior fast path: the whole 4x4x4-minus-corners stencil (32 voxels) is in range and the
    // +-1 voxel shifted coordinates split regularly (i0 +- 1).  Then the 7 trilinear lookups share 12
    // row loads (4 x 16 B + 8 x 8 B) instead of 7 x 8 scalar loads.
    const Axis axp = axis_of(gx + 1.0f), axm = axis_of(gx - 1.0f);
    const Axis ayp = axis_of(gy + 1.0f), aym = axis_of(gy - 1.0f);
    const Axis azp = axis_of(gz + 1.0f), azm = axis_of(gz - 1.0f);
    const bool regular = axp.i0 == ax.i0 + 1 && axm.i0 == ax.i0 - 1 && ayp.i0 == ay.i0 + 1 && aym.i0 == ay.i0 - 1 &&
                         azp.i0 == az.i0 + 1 && azm.i0 == az.i0 - 1 && axp.ok && axm.ok && ayp.ok && aym.ok && azp.ok && azm.ok;
    const bool interior = in_c && regular && ax.i0 >= 1 && ax.i0 < G.dx - 2 && ay.i0 >= 1 && ay.i0 < G.dy - 2 &&
                          az.i0 >= 1 && az.i0 < G.dz - 2;

    float value, fpx, fmx, fpy, fmy, fpz, fmz;
    if (interior) {
        // centre rows, z0-1 .. z0+2
        const F4 c00 = *reinterpret_cast<const F4*>(G.g + b - 1);
        const F4 c01 = *reinterpret_cast<const F4*>(G.g + b + sy - 1);
        const F4 c10 = *reinterpret_cast<const F4*>(G.g + b + sx - 1);
        const F4 c11 = *reinterpret_cast<const F4*>(G.g + b + sx + sy - 1);
        value = trilerp(c00.b, c00.c, c01.b, c01.c, c10.b, c10.c, c11.b, c11.c, ax.f, ay.f, az.f);
        if (value < o.clr) acc.col += 1.0f;  // .cu:150-151
        if (!(value <= o.eps)) return;       // .cu:170-171
        // +-z: same rows shifted by one voxel
        fpz = trilerp(c00.c, c00.d, c01.c, c01.d, c10.c, c10.d, c11.c, c11.d, ax.f, ay.f, azp.f);
        fmz = trilerp(c00.a, c00.b, c01.a, c01.b, c10.a, c10.b, c11.a, c11.b, ax.f, ay.f, azm.f);
        // +-x: planes x0+2 and x0-1
        const F2 xp0 = *reinterpret_cast<const F2*>(G.g + b + 2 * sx);
        const F2 xp1 = *reinterpret_cast<const F2*>(G.g + b + 2 * sx + sy);
        const F2 xm0 = *reinterpret_cast<const F2*>(G.g + b - sx);
        const F2 xm1 = *reinterpret_cast<const F2*>(G.g + b - sx + sy);
        fpx = trilerp(c10.b, c10.c, c11.b, c11.c, xp0.a, xp0.b, xp1.a, xp1.b, axp.f, ay.f, az.f);
        fmx = trilerp(xm0.a, xm0.b, xm1.a, xm1.b, c00.b, c00.c, c01.b, c01.c, axm.f, ay.f, az.f);
        // +-y: rows y0+2 and y0-1
        const F2 yp0 = *reinterpret_cast<const F2*>(G.g + b + 2 * sy);
        const F2 yp1 = *reinterpret_cast<const F2*>(G.g + b + sx + 2 * sy);
        const F2 ym0 = *reinterpret_cast<const F2*>(G.g + b - sy);
        const F2 ym1 = *reinterpret_cast<const F2*>(G.g + b + sx - sy);
        fpy = trilerp(c01.b, c01.c, yp0.a, yp0.b, c11.b, c11.c, yp1.a, yp1.b, ax.f, ayp.f, az.f);
        fmy = trilerp(ym0.a, ym0.b, c00.b, c00.c, ym1.a, ym1.b, c10.b, c10.c, ax.f, aym.f, az.f);
    } else {
        if (in_c) {
            value = sdf_value(G, ax, ay, az);
            if (value < o.clr) acc.col += 1.0f;
            if (!(value <= o.eps)) return;
        } else {
            value = 1.0f;  // only reachable when eps >= 1
        }
        fpx = sdf_value(G, axp, ay, az);
        fpy = sdf_value(G, ax, ayp, az);
        fpz = sdf_value(G, ax, ay, azp);
        fmx = sdf_value(G, axm, ay, az);
        fmy = sdf_value(G, ax, aym, az);
        fmz = sdf_value(G, ax, ay, azm);
    }
    // .cu:82-84: 0.5 * (f_p - f_m) / delta in double, narrowed.  x / delta == (float)(x * fl64(1/delta)) after the
    // narrowing for the same reason as in the grid-coordinate quotient (x and delta are 24-bit values).
    const float g0 = (float)(0.5 * (double)(fpx - fmx) * o.rdelta);
    const float g1 = (float)(0.5 * (double)(fpy - fmy) * o.rdelta);
    const float g2 = (float)(0.5 * (double)(fpz - fmz) * o.rdelta);
    float v0, v1, v2;
    if (value <= 0.0f) {  // .cu:158-164
        acc.pot += (float)(-(double)value + 0.5 * (double)o.eps);
        v0 = -g0; v1 = -g1; v2 = -g2;
    } else {  // 0 < value <= eps (.cu:165-171)
        const float d = value - o.eps;
        acc.pot += o.i2eps * d * d * o.pad;
        const float ie = o.ieps;
        v0 = ie * g0 * d * o.pad; v1 = ie * g1 * d * o.pad; v2 = ie * g2 * d * o.pad;
    }
    // rotationMatrix.transpose() * vgrad (.cu:176-179)
    acc.gx += __builtin_fmaf(T[8], v2, __builtin_fmaf(T[4], v1, T[0] * v0));
    acc.gy += __builtin_fmaf(T[9], v2, __builtin_fmaf(T[5], v1, T[1] * v0));
    acc.gz += __builtin_fmaf(T[10], v2, __builtin_fmaf(T[6], v1, T[2] * v0));
}

template <bool WANT_GRAD>
__device__ __forceinline__ void sdf_pair(const ObjParams& o, const float* __restrict__ grid, float px, float py, float pz,
                                         Accum& acc) {
    const PairPrep pp = pair_prepare(o, px, py, pz);
    if (!pp.far) pair_exact<WANT_GRAD>(o, grid, pp.tx, pp.ty, pp.tz, acc);
}

// Wave-uniform tables (object records, scene_begin) are read through the CONSTANT address space: the
// compiler then emits scalar loads (s_load -> SGPRs) even though the kernel also stores to global
// memory it cannot prove disjoint; plain global pointers fell back to per-lane vector loads into
// ~26 VGPRs per object.  The tables are never written by a kernel that reads them.
#define OMG_CONST_AS __attribute__((address_space(4)))
typedef const OMG_CONST_AS omgx_object* ObjTablePtr;
typedef const OMG_CONST_AS int32_t* IntTablePtr;
#define OMG_GLOBAL_AS __attribute__((address_space(1)))
typedef const OMG_GLOBAL_AS char* GlobalBytes;  // a pointer the compiler must treat as global memory (global_load, vmcnt only), whatever it went through
typedef float F2v __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ F2 global_f2(GlobalBytes p) { const F2v v = *(const OMG_GLOBAL_AS F2v*)p; return F2{v.x, v.y}; }
__device__ __forceinline__ ObjTablePtr as_const(const omgx_object* p) { return (ObjTablePtr)(uintptr_t)p; }
__device__ __forceinline__ IntTablePtr as_const(const int32_t* p) { return (IntTablePtr)(uintptr_t)p; }

__device__ __forceinline__ ObjParams load_object(ObjTablePtr ob) {
    ObjParams o;
#pragma unroll
    for (int k = 0; k < 12; ++k) o.T[k] = ob->pose_inv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.lo[k] = ob->lo[k]; o.hi[k] = ob->hi[k]; o.dim[k] = ob->dim[k]; }
    o.delta = ob->delta; o.eps = ob->epsilon; o.pad = ob->padding_scale; o.clr = ob->clearance;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.rw[k] = ob->inv_extent[k]; o.rc[k] = ob->rb_c[k]; o.rh[k] = ob->rb_h[k]; }
    o.rr2 = ob->rb_r2;
    o.rdelta = ob->inv_delta; o.i2eps = ob->inv_2eps; o.ieps = ob->inv_eps;
    return o;
}

// All objects of one scene for one point (objects summed in index order; the reference's atomicAdd
// order is unspecified, .cu:185-195).
template <bool WANT_GRAD>
__device__ __forceinline__ Accum sdf_point(const omgx_object* __restrict__ objs, int o_begin, int o_end,
                                           const float* __restrict__ pool, float px, float py, float pz) {
    Accum acc{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int o = o_begin; o < o_end; ++o) {  // wave-uniform trip count and addresses -> scalar loads
        ObjTablePtr ob = as_const(objs) + o;
        if (ob->disabled > 0) continue;  // .cu:115-116
        const ObjParams op = load_object(ob);
        sdf_pair<WANT_GRAD>(op, pool + ob->grid_offset, px, py, pz, acc);
    }
    return acc;
}

// -------------------------------------------------------------------------------------------------
// Forward kinematics (double)
// -------------------------------------------------------------------------------------------------
// Derived constants appended to the robot blob by the host (omg-planner_amd/robot.py: PandaModel.blob):
//   raw tables [0, 528 + 30 P)  as documented in include/omg_hip.h, then at D = 528 + 30 P:
//   D+0    UVW  [7][3][9]   b_i(q) = pose_0[i] . Rz(q) . Rx(off_i) . N_i has rotation c*U_i + s*V_i + W_i
//   D+189  TP   [7][3]      ... and translation pose_0[i][:3,3]
//   D+210  H    [12]        pose_0[7] rows (hand)            (robot_pykdl.py:186)
//   D+222  LF   [12]        pose_0[8] rows (left finger)     (:181-182)
//   D+234  RF   [12]        pose_0[9] rows (right finger)    (:183-184)
//   D+246  PTS  [10][P][3]  center_offset[l] applied to collision_points[l][p]  (:203-204 folded into cost.py:60-72)
//   D+246+30P AX [10][3]    tip2joint[l][:3,:3] . joint_axis[l]                  (:190-197)
//   D+276+30P OG [10][3]    tip2joint[l][:3,3]
//   D+306+30P RAD [10]      bounding-sphere radius of each link's centred points
// Latency mode: a workgroup alone on a cold CU takes its scalar-cache misses one after the other — the culling stage's object loop
// needs two dependent round trips per object (the `disabled` flag, then the record): 5 objects = 7 us of an 8 us stage, measured.
// gq_warm_scalar_cache requests every 64-byte line of up to 8 object records and of the links' bounding balls (40 doubles) through the scalar cache AT ONCE
// and waits for them (one round trip, taken while the workgroup's first vector loads are in flight anyway).  One asm statement:
// the loads' throw-away targets are clobbers, and nothing of it is in flight when it ends (the compiler neither tracks inline-asm
// loads nor knows when a scalar load lands).
__device__ __forceinline__ void gq_warm_scalar_cache(const omgx_object* objects, int o_begin, int o_end, const double* balls) {
    const omgx_object* ob[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ob[k] = objects + (o_begin + k < o_end ? o_begin + k : o_end - 1);  // fewer than 8: the last one again (a hit)
#define OMG_WARM_OBJ(N) "s_load_dword s96, %" #N ", 0x0\n\ts_load_dword s97, %" #N ", 0x40\n\ts_load_dword s98, %" #N ", 0x80\n\ts_load_dword s99, %" #N ", 0xb4\n\t"
    asm volatile(OMG_WARM_OBJ(0) OMG_WARM_OBJ(1) OMG_WARM_OBJ(2) OMG_WARM_OBJ(3) OMG_WARM_OBJ(4) OMG_WARM_OBJ(5) OMG_WARM_OBJ(6) OMG_WARM_OBJ(7)
                 "s_load_dword s96, %8, 0x0\n\ts_load_dword s97, %8, 0x40\n\ts_load_dword s98, %8, 0x80\n\ts_load_dword s99, %8, 0xc0\n\t"
                 "s_load_dword s96, %8, 0x100\n\ts_load_dword s97, %8, 0x13c\n\ts_waitcnt lgkmcnt(0)"
                 :
                 : "s"(ob[0]), "s"(ob[1]), "s"(ob[2]), "s"(ob[3]), "s"(ob[4]), "s"(ob[5]), "s"(ob[6]), "s"(ob[7]), "s"(balls)
                 : "s96", "s97", "s98", "s99", "memory");
#undef OMG_WARM_OBJ
}

// DPtr = pointer type of the derived constants: plain global memory, a copy elsewhere (LDS), or the CONSTANT address space —
// then wave-uniform reads (the chain constants of joint i, the hand / finger rows, radii) become scalar loads into SGPRs
// instead of per-lane vector loads of the same address (measured in the goal-set kernel's kinematics: the vector loads and
// their spills were 20 of a goal workgroup's 69 us).  pts() is indexed per lane and always comes from global memory.
template <class DPtr>
struct RobotViewT {
    const double* __restrict__ raw;
    DPtr d;
    const double* __restrict__ g;  // derived constants in global memory (for pts)
    int P;
    __device__ __forceinline__ RobotViewT(const double* blob, int P_) : raw(blob), d((DPtr)(uintptr_t)(blob + OMGX_ROBOT_POINTS + 30 * P_)), g(blob + OMGX_ROBOT_POINTS + 30 * P_), P(P_) {}
    // the first 246 derived doubles (UVW, TP, H, LF, RF: all the kinematic chain needs) from a copy elsewhere, e.g. LDS
    __device__ __forceinline__ RobotViewT(const double* blob, int P_, DPtr chain_constants) : raw(blob), d(chain_constants), g(blob + OMGX_ROBOT_POINTS + 30 * P_), P(P_) {}
    __device__ __forceinline__ DPtr uvw(int i) const { return d + 27 * i; }
    __device__ __forceinline__ DPtr tp(int i) const { return d + 189 + 3 * i; }
    __device__ __forceinline__ DPtr hand() const { return d + 210; }
    __device__ __forceinline__ DPtr lf() const { return d + 222; }
    __device__ __forceinline__ DPtr rf() const { return d + 234; }
    __device__ __forceinline__ const double* pts(int l, int p) const { return g + 246 + 3 * (l * P + p); }
    __device__ __forceinline__ DPtr ax(int l) const { return d + 246 + 30 * P + 3 * l; }
    __device__ __forceinline__ DPtr og(int l) const { return d + 276 + 30 * P + 3 * l; }
    __device__ __forceinline__ double radius(int l) const { return d[306 + 30 * P + l]; }
    __device__ __forceinline__ DPtr ball(int l) const { return d + 316 + 30 * P + 4 * l; }  // (c_x, c_y, c_z, r) of the link's own points
    __device__ __forceinline__ const double* lower() const { return raw + OMGX_ROBOT_LOWER; }
    __device__ __forceinline__ const double* upper() const { return raw + OMGX_ROBOT_UPPER; }
};
typedef RobotViewT<const double*> RobotView;                 // global memory or an LDS copy
typedef RobotViewT<const OMG_CONST_AS double*> RobotViewS;   // scalar loads

struct Pose { double R[9]; double t[3]; };  // link frame BEFORE center_offset (robot_pykdl output_pose)

// The double-precision kinematics may fuse multiply-adds (the file is compiled with -ffp-contract=off
// for the float32 SDF arithmetic only; FK parity is checked at 1e-12, not bitwise).
template <class BP>
__device__ __forceinline__ void pose_mul(const Pose& A, BP B /*rows [3][4]*/, Pose& C) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C.R[3 * r + c] = A.R[3 * r] * B[c] + A.R[3 * r + 1] * B[4 + c] + A.R[3 * r + 2] * B[8 + c];
        C.t[r] = A.R[3 * r] * B[3] + A.R[3 * r + 1] * B[7] + A.R[3 * r + 2] * B[11] + A.t[r];
    }
}

// wrap_values' rad->deg and forward_kinematics_parallel's deg->rad (omg/util.py:194-202, robot_pykdl.py:164)
// x / c for a compile-time constant c without the ~15-instruction IEEE division sequence: q = x * RN(1 / c) is within two
// ulps of the quotient, the FMA residual r = x - q c is exact, and q + r RN(1 / c) rounds to the correctly rounded quotient
// (Markstein's correction step).  For c = pi and c = 180 it equals the division on all of 4 x 10^8 sampled arguments
// (tests/test_oracle_golden.py::test_division_by_constant_through_fma_equals_the_quotient).
__device__ __forceinline__ double div_by_const(double x, double c, double rc) {
    const double q = x * rc;
    const double r = __builtin_fma(-q, c, x);
    return __builtin_fma(r, rc, q);
}
__device__ __forceinline__ double deg_round_trip(double q) {
    return div_by_const(div_by_const(q, M_PI, 1.0 / M_PI) * 180.0, 180.0, 1.0 / 180.0) * M_PI;  // (q / pi * 180) / 180 * pi
}

// Visits the 10 link poses of configuration q[9] (radians) in order; f(l, pose).
template <class RV, class F>
__device__ __forceinline__ void fk_chain(const RV& rv, const double* __restrict__ q, F&& f) {
#pragma clang fp contract(fast)
    Pose cur;
#pragma unroll
    for (int k = 0; k < 9; ++k) cur.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    cur.t[0] = cur.t[1] = cur.t[2] = 0.0;
    for (int i = 0; i < 7; ++i) {
        double s, c;
        sincos(deg_round_trip(q[i]), &s, &c);
        const auto uvw = rv.uvw(i);
        const auto tp = rv.tp(i);
        double B[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) B[k] = c * uvw[k] + s * uvw[9 + k] + uvw[18 + k];
        Pose nxt;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                nxt.R[3 * r + cc] = cur.R[3 * r] * B[cc] + cur.R[3 * r + 1] * B[3 + cc] + cur.R[3 * r + 2] * B[6 + cc];
            nxt.t[r] = cur.R[3 * r] * tp[0] + cur.R[3 * r + 1] * tp[1] + cur.R[3 * r + 2] * tp[2] + cur.t[r];
        }
        cur = nxt;
        f(i, cur);
    }
    Pose hand;
    pose_mul(cur, rv.hand(), hand);
    f(7, hand);
    {
        double Fm[12];
        const auto L = rv.lf();
#pragma unroll
        for (int k = 0; k < 12; ++k) Fm[k] = L[k];
        Fm[7] += deg_round_trip(q[7]);  // left_finger_pose[:, 1, 3] += joints[:, -2]
        Pose fp;
        pose_mul(hand, Fm, fp);
        f(8, fp);
        const auto Rr = rv.rf();
#pragma unroll
        for (int k = 0; k < 12; ++k) Fm[k] = Rr[k];
        Fm[7] -= deg_round_trip(q[8]);  // right_finger_pose[:, 1, 3] -= joints[:, -1]
        pose_mul(hand, Fm, fp);
        f(9, fp);
    }
}

// The same kinematics split for a workgroup (k_goalset_queue, k_chomp_optimize), where one lane per configuration
// leaves most of the workgroup idle behind a ~2200-instruction serial chain:
//   stage 1, one lane per (configuration, arm joint):  fk_joint_sincos -> (sin, cos) of the round-tripped angle;
//   stage 2, one lane per (configuration, row r < 3):  row r of every link pose.  Row r of a product A.B depends on
//            row r of A only, so the three rows of the chain are independent: 7 x (18 + 12) FMAs per lane.
// f(l, R_r0, R_r1, R_r2, t_r) receives row r of link l's pose (before center_offset, like fk_chain).
__device__ __forceinline__ void fk_joint_sincos(double q, double& s, double& c) { sincos(deg_round_trip(q), &s, &c); }

// The rounding sequence of the workgroup kinematics, fixed in SOURCE: the three code shapes below — fk_chain_row,
// fk_joint_matrix + fk_chain_row_B — must produce the same bits (the launches hand poses to each other), which under
// `fp contract(fast)` depended on the backend fusing the same multiply-adds in each.  Here every fused operation is an explicit fma and
// contraction is off: a 3-term dot product is fma(a2, b2, fma(a1, b1, a0 b0)).  (The backend had fused differently: the plan digests of
// tools/bits_of_a_plan.py changed once with this rewrite — last-bit differences of the poses, inside every parity tolerance — and
// are now a property of this file, not of the compiler version.)
__device__ __forceinline__ double fk_dot3(double a0, double a1, double a2, double b0, double b1, double b2) {
#pragma clang fp contract(off)
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}
// joint matrix entry: c U + s V + W
__device__ __forceinline__ double fk_bentry(double c, double s, double U, double V, double W) {
#pragma clang fp contract(off)
    return __builtin_fma(s, V, c * U) + W;
}
// hand = link7 . pose_0[7]; fingers = hand . pose_0[8|9] with y -+ q (robot_pykdl.py:181-188): the tail of every chain
template <class RV, class F>
__device__ __forceinline__ void fk_chain_tail(const RV& rv, double a0, double a1, double a2, double at, double q7, double q8, F&& f) {
#pragma clang fp contract(off)
    const auto H = rv.hand();
    const double h0 = fk_dot3(a0, a1, a2, H[0], H[4], H[8]), h1 = fk_dot3(a0, a1, a2, H[1], H[5], H[9]), h2 = fk_dot3(a0, a1, a2, H[2], H[6], H[10]);
    const double ht = fk_dot3(a0, a1, a2, H[3], H[7], H[11]) + at;
    f(7, h0, h1, h2, ht);
    const auto Lf = rv.lf();
    f(8, fk_dot3(h0, h1, h2, Lf[0], Lf[4], Lf[8]), fk_dot3(h0, h1, h2, Lf[1], Lf[5], Lf[9]), fk_dot3(h0, h1, h2, Lf[2], Lf[6], Lf[10]),
      fk_dot3(h0, h1, h2, Lf[3], Lf[7] + deg_round_trip(q7), Lf[11]) + ht);
    const auto Rf = rv.rf();
    f(9, fk_dot3(h0, h1, h2, Rf[0], Rf[4], Rf[8]), fk_dot3(h0, h1, h2, Rf[1], Rf[5], Rf[9]), fk_dot3(h0, h1, h2, Rf[2], Rf[6], Rf[10]),
      fk_dot3(h0, h1, h2, Rf[3], Rf[7] - deg_round_trip(q8), Rf[11]) + ht);
}

template <class RV, class F>
__device__ __forceinline__ void fk_chain_row(const RV& rv, int r, const double* __restrict__ sc /* [7][2] sin, cos */,
                                             double q7, double q8, F&& f) {
#pragma clang fp contract(off)
    double a0 = r == 0 ? 1.0 : 0.0, a1 = r == 1 ? 1.0 : 0.0, a2 = r == 2 ? 1.0 : 0.0, at = 0.0;
#pragma unroll 1  // unrolled, the 7 x 27 wave-uniform constants are hoisted into (spilled) SGPRs all at once
    for (int i = 0; i < 7; ++i) {
        const double s = sc[2 * i], c = sc[2 * i + 1];
        const auto uvw = rv.uvw(i);
        const auto tp = rv.tp(i);
        double B[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) B[k] = fk_bentry(c, s, uvw[k], uvw[9 + k], uvw[18 + k]);
        const double n0 = fk_dot3(a0, a1, a2, B[0], B[3], B[6]);
        const double n1 = fk_dot3(a0, a1, a2, B[1], B[4], B[7]);
        const double n2 = fk_dot3(a0, a1, a2, B[2], B[5], B[8]);
        at = fk_dot3(a0, a1, a2, tp[0], tp[1], tp[2]) + at;
        a0 = n0; a1 = n1; a2 = n2;
        f(i, a0, a1, a2, at);
    }
    fk_chain_tail(rv, a0, a1, a2, at, q7, q8, f);
}

// The same chain with the joints' matrices B_i = c_i U_i + s_i V_i + W_i computed beforehand by one lane per (configuration, joint)
// (fk_joint_matrix: the expression of fk_chain_row, so the same bits) and read from Bt [7][9]: the chain's lanes then do 12 instead
// of 30 multiply-adds and read 12 instead of 30 constants per joint.  Latency mode only: the table costs 504 B of LDS per configuration.
template <class RV>
__device__ __forceinline__ void fk_joint_matrix(const RV& rv, int i, double s, double c, double* __restrict__ B) {
    const auto uvw = rv.uvw(i);
#pragma unroll
    for (int k = 0; k < 9; ++k) B[k] = fk_bentry(c, s, uvw[k], uvw[9 + k], uvw[18 + k]);
}

// bstride: doubles between the matrices of consecutive joints (9: a [7][9] table per configuration; the batch kernel keeps joint i's
// matrices where link i's poses will be written: one link block apart).  Bt is NOT restrict there: f() overwrites what was read.
template <class RV, class F>
__device__ __forceinline__ void fk_chain_row_B(const RV& rv, int r, const double* Bt /* [7][bstride] */, double q7, double q8, F&& f, int bstride = 9) {
#pragma clang fp contract(off)
    double a0 = r == 0 ? 1.0 : 0.0, a1 = r == 1 ? 1.0 : 0.0, a2 = r == 2 ? 1.0 : 0.0, at = 0.0;
#pragma unroll 1
    for (int i = 0; i < 7; ++i) {
        const double* B = Bt + bstride * i;
        const auto tp = rv.tp(i);
        const double n0 = fk_dot3(a0, a1, a2, B[0], B[3], B[6]);
        const double n1 = fk_dot3(a0, a1, a2, B[1], B[4], B[7]);
        const double n2 = fk_dot3(a0, a1, a2, B[2], B[5], B[8]);
        at = fk_dot3(a0, a1, a2, tp[0], tp[1], tp[2]) + at;
        a0 = n0; a1 = n1; a2 = n2;
        f(i, a0, a1, a2, at);
    }
    fk_chain_tail(rv, a0, a1, a2, at, q7, q8, f);
}

__device__ __forceinline__ void pose_apply(const Pose& A, const double* __restrict__ p, double& x, double& y, double& z) {
#pragma clang fp contract(fast)
    x = A.R[0] * p[0] + A.R[1] * p[1] + A.R[2] * p[2] + A.t[0];
    y = A.R[3] * p[0] + A.R[4] * p[1] + A.R[5] * p[2] + A.t[1];
    z = A.R[6] * p[0] + A.R[7] * p[1] + A.R[8] * p[2] + A.t[2];
}

// x = R p + t for a pose stored as 12 doubles (R row-major [9], t [3]); result rounded to float32 like
// `torch.from_numpy(ws_positions).cuda().float()` (cost.py:136,218).
__device__ __forceinline__ void pose12_apply(const double* __restrict__ A, const double* __restrict__ p, float& x, float& y, float& z) {
#pragma clang fp contract(fast)
    x = (float)(A[0] * p[0] + A[1] * p[1] + A[2] * p[2] + A[9]);
    y = (float)(A[3] * p[0] + A[4] * p[1] + A[5] * p[2] + A[10]);
    z = (float)(A[6] * p[0] + A[7] * p[1] + A[8] * p[2] + A[11]);
}

// Same for a pose stored as 9 doubles (rotation rows 0 and 1, translation): the third row of a proper rotation is
// r0 x r1.  Saves a quarter of the LDS a workgroup needs for its poses (5 instead of 4 workgroups per CU); the
// reconstructed row differs from the FK product's by ~1e-16 relative, like any re-association of the float64 FK.
__device__ __forceinline__ void pose9_apply(const double* __restrict__ A, const double* __restrict__ p, float& x, float& y, float& z) {
#pragma clang fp contract(fast)
    const double r20 = A[1] * A[5] - A[2] * A[4], r21 = A[2] * A[3] - A[0] * A[5], r22 = A[0] * A[4] - A[1] * A[3];
    x = (float)(A[0] * p[0] + A[1] * p[1] + A[2] * p[2] + A[6]);
    y = (float)(A[3] * p[0] + A[4] * p[1] + A[5] * p[2] + A[7]);
    z = (float)(r20 * p[0] + r21 * p[1] + r22 * p[2] + A[8]);
}

// Centre of link l's bounding ball (RobotViewT::ball: centre c in the link frame) in the workspace, for the row-level culling:
// R c + t, rounded to float32.  Explicit fma, contraction off: every kernel that writes row masks computes the same centre.
// r0 / r1: rotation rows 0 and 1, t: translation; the third row is r0 x r1 as in pose9_apply.
template <class BP>
__device__ __forceinline__ void link_ball_center(const double* r0, const double* r1, const double* t, BP b, float& cx, float& cy, float& cz) {
#pragma clang fp contract(off)
    const double c0 = b[0], c1 = b[1], c2 = b[2];
    const double r20 = __builtin_fma(r0[1], r1[2], -(r0[2] * r1[1])), r21 = __builtin_fma(r0[2], r1[0], -(r0[0] * r1[2])),
                 r22 = __builtin_fma(r0[0], r1[1], -(r0[1] * r1[0]));
    cx = (float)(fk_dot3(r0[0], r0[1], r0[2], c0, c1, c2) + t[0]);
    cy = (float)(fk_dot3(r1[0], r1[1], r1[2], c0, c1, c2) + t[1]);
    cz = (float)(fk_dot3(r20, r21, r22, c0, c1, c2) + t[2]);
}

// np.argmin / np.argmax order: does (v, i) beat (bv, bi)?  The first occurrence wins, and a NaN counts as the extreme
// for BOTH (numpy propagates NaN: np.argmax([1, nan, 3]) == np.argmin([1, nan, 0]) == 1) — a degenerate cost vector
// (0/0 after normalisation) therefore selects index 0 like the reference instead of leaving the index undefined.
template <bool MIN>
__device__ __forceinline__ bool np_arg_better(double v, int i, double bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn | bn) return vn && (!bn || i < bi);
    return (MIN ? v < bv : v > bv) || (v == bv && i < bi);
}

// -------------------------------------------------------------------------------------------------
// wave / block reductions with a fixed combination order (deterministic results)
// -------------------------------------------------------------------------------------------------
// DPP cross-lane moves (one VALU instruction, no LDS round trip; __shfl_* lowers to ds_bpermute, ~10x slower
// on a dependent chain).  CTRL: 0xB1 = quad_perm[1,0,3,2] (lane^1), 0x4E = quad_perm[2,3,0,1] (lane^2),
// 0x141 = row_half_mirror (i <-> 7-i), 0x140 = row_mirror (i <-> 15-i).  All lanes of the row must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
// inclusive SUFFIX sum over the 64 lanes of a wave: lane i receives v[i] + v[i+1] + ... + v[63].  Four row shifts inside the DPP rows
// of 16 (lanes without a source read 0) and the totals of the higher rows through three lane reads — ~15 instructions where six
// rounds of __shfl_down (ds_bpermute: a trip through the LDS crossbar each) took ~700 cycles.  Call from all 64 lanes.
__device__ __forceinline__ int wave_suffix_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true);  // row_shl:1 — lane i reads lane i + 1 of its row
    v += __builtin_amdgcn_update_dpp(0, v, 0x102, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x108, 0xf, 0xf, true);
    const int t1 = __builtin_amdgcn_readlane(v, 16), t2 = __builtin_amdgcn_readlane(v, 32), t3 = __builtin_amdgcn_readlane(v, 48);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row < 1 ? t1 : 0) + (row < 2 ? t2 : 0) + (row < 3 ? t3 : 0);
}
// sum over each aligned group of 16 lanes (a DPP row), result in every lane of the group
__device__ __forceinline__ double row16_allsum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over the 64 lanes of a wave, identical in every lane, fixed order
__device__ __forceinline__ double wave_allsum(double v) {
    v = row16_allsum(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_allmax(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, OMG_WAVE);
    return v;  // valid in lane 0
}

}  // namespace omg
