/*
 * omg_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm of liruiw/OMG-Planner's CHOMP trajectory-update
 * path.  It exists to CHECK the HIP kernels in omg-planner_amd/csrc; nothing in the product path may
 * import, link or call it (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - FK, points, Jacobians, derivatives, functional gradient, collision/smooth/total loss and the
 *     optimiser step are PINNED against outputs of the reference Python itself
 *     (tests/golden/make_golden.py imports /root/reference/omg/{cost,optimizer,config,util}.py and
 *     ycb_render/robotPose/robot_pykdl.py and dumps the .npz fixtures under tests/golden/).
 *   - The SDF op (orc_sdf_loss_forward): its interpolation helpers (orc_sdf_value = getValueInterpolated, the
 *     central differences = getGradientInterpolated, .cu:15-86) are PINNED against the reference's own source compiled
 *     for the host (oracle/_ref, `make -C oracle ref`; tests/test_oracle_ref_helpers.py: bit-exact against the
 *     FMA-contracted build).  The kernel BODY (.cu:96-181: pose transform, hinge, rotate-back, reduction) is PARITY
 *     UNPINNED by the reference: it needs nvcc + ATen + Eigen + Sophus, none present, and the reference holds no test
 *     vectors for it; it follows the source line by line and is checked by closed-form known-answer tests
 *     (tests/test_oracle_sdf.py).
 *
 * Floating-point conventions, shared bit-for-bit with the HIP kernels (both are compiled with
 * -ffp-contract=off so only the explicit fma()s below fuse):
 *   - lerp(a,b,t) = fmaf(t, b-a, a)        (.cu:15-18; nvcc contracts this form to one FFMA)
 *   - affine maps are fma chains seeded with the translation / first product (see orc_xform*)
 *   - the double sub-expressions the C literals promote (.cu:39-41, 82-84, 160) are kept in double.
 *
 * Citations are file:line into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/omg_hip.h"

#define NL OMGX_NUM_LINKS
#define ND OMGX_NUM_DOF
#define MAXP OMGX_MAX_POINTS
#define MAXN OMGX_MAX_WAYPOINTS

/* =============================================================================================
 * 1. SDF op — layers/sdf_matching_loss_kernel.cu
 * =========================================================================================== */

static inline float orc_lerp(float a, float b, float t) { return fmaf(t, b - a, a); } /* .cu:15-18 */

/* getValueInterpolated, .cu:36-64.  pGrid - 0.5 is evaluated in double (literal promotion), the cast
 * truncates toward zero (so (-1,0) -> 0 with a negative weight), out of range returns 1.0. */
static float orc_sdf_value(float gx, float gy, float gz, int dx, int dy, int dz, const float* g) {
    const double sx = (double)gx - 0.5, sy = (double)gy - 0.5, sz = (double)gz - 0.5;
    /* guard the (int) cast against UB for absurd coordinates; such points are out of range anyway */
    if (!(sx > -1.0e9 && sx < 1.0e9 && sy > -1.0e9 && sy < 1.0e9 && sz > -1.0e9 && sz < 1.0e9)) return 1.0f;
    const int x0 = (int)sx, y0 = (int)sy, z0 = (int)sz;
    const float fx = (float)(sx - (double)x0), fy = (float)(sy - (double)y0), fz = (float)(sz - (double)z0);
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    if (!(x0 >= 0 && x1 < dx && y0 >= 0 && y1 < dy && z0 >= 0 && z1 < dz)) return 1.0f;
#define G(x, y, z) g[(x) * dy * dz + (y) * dz + (z)] /* getValue, .cu:30-34 */
    const float dx00 = orc_lerp(G(x0, y0, z0), G(x1, y0, z0), fx);
    const float dx01 = orc_lerp(G(x0, y0, z1), G(x1, y0, z1), fx);
    const float dx10 = orc_lerp(G(x0, y1, z0), G(x1, y1, z0), fx);
    const float dx11 = orc_lerp(G(x0, y1, z1), G(x1, y1, z1), fx);
#undef G
    const float dxy0 = orc_lerp(dx00, dx10, fy);
    const float dxy1 = orc_lerp(dx01, dx11, fy);
    return orc_lerp(dxy0, dxy1, fz);
}

/* Exported twins of the reference helpers, checked against oracle/_ref (the reference's own .cu:15-86 compiled for the
 * host) in tests/test_oracle_ref_helpers.py. */
float orc_value_interpolated(float gx, float gy, float gz, int dx, int dy, int dz, const float* g) {
    return orc_sdf_value(gx, gy, gz, dx, dy, dz, g);
}
void orc_gradient_interpolated(float gx, float gy, float gz, int dx, int dy, int dz, const float* g, float delta, float* out3) {
    /* getGradientInterpolated, .cu:66-86 */
    const float fpx = orc_sdf_value(gx + 1.0f, gy, gz, dx, dy, dz, g), fmx = orc_sdf_value(gx - 1.0f, gy, gz, dx, dy, dz, g);
    const float fpy = orc_sdf_value(gx, gy + 1.0f, gz, dx, dy, dz, g), fmy = orc_sdf_value(gx, gy - 1.0f, gz, dx, dy, dz, g);
    const float fpz = orc_sdf_value(gx, gy, gz + 1.0f, dx, dy, dz, g), fmz = orc_sdf_value(gx, gy, gz - 1.0f, dx, dy, dz, g);
    out3[0] = (float)(0.5 * (double)(fpx - fmx) / (double)delta);
    out3[1] = (float)(0.5 * (double)(fpy - fmy) / (double)delta);
    out3[2] = (float)(0.5 * (double)(fpz - fmz) / (double)delta);
}

/* SOPHUS MODE (round 6; off by default).  The reference builds `Sophus::SE3<float>(Matrix4f)` (.cu:125-126): the rotation becomes a unit
 * quaternion (Eigen's matrix -> quaternion assignment, Eigen/src/Geometry/Quaternion.h: quaternionbase_assign_impl<.., 3, 3>), the
 * point is rotated with the quaternion (QuaternionBase::_transformVector: uv = 2 q.vec x p; p + w uv + q.vec x uv) and translated
 * (.cu:133), and the gradient is rotated back with the matrix REGENERATED from the quaternion (so3().matrix() = toRotationMatrix,
 * .cu:126,176).  Sophus is an empty submodule of the reference tree and Eigen is not in this image, so this is a restatement of the
 * PUBLISHED algorithms (Eigen 3.3 / Sophus 1.0 as the reference's submodule pins them), all in float32 without contraction; the default
 * mode multiplies by the float32 matrix, which is what the HIP kernels do.  bench.py's parity_sample runs the engine against the
 * oracle in BOTH modes and reports the difference. */
static int g_sophus_mode = 0;
void orc_set_sophus_mode(int on) { g_sophus_mode = on ? 1 : 0; }
int orc_get_sophus_mode(void) { return g_sophus_mode; }

static void orc_quat_from_rows(const float* T, float* q /* x y z w */) {
#define M(r, c) T[4 * (r) + (c)]
    float t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0f) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (M(2, 1) - M(1, 2)) * t;
        q[1] = (M(0, 2) - M(2, 0)) * t;
        q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(M(i, i) - M(j, j) - M(k, k) + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (M(k, j) - M(j, k)) * t;
        q[j] = (M(j, i) + M(i, j)) * t;
        q[k] = (M(k, i) + M(i, k)) * t;
    }
#undef M
}
static void orc_cross(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void orc_quat_matrix(const float* q, float* m /* [9] row-major */) {  /* QuaternionBase::toRotationMatrix */
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0] = 1.0f - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
    m[3] = txy + twz; m[4] = 1.0f - (txx + tzz); m[5] = tyz - twx;
    m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1.0f - (txx + tyy);
}

/* One (point, object) pair: the body of SDFdistanceForward, .cu:111-180.  Adds into pot/grad/col. */
static void orc_sdf_pair(const float* T /*[3][4] rows of the inverse pose*/, const float* lo, const float* hi,
                         const int* dim, float delta, float eps, float pad, float clr, const float* grid,
                         const float* p, float* pot, float* grad, float* col) {
    /* SE3(pose) * point, .cu:125-133 (Sophus; restated as R p + t — or, in Sophus mode, through the quaternion) */
    float ux, uy, uz, Rq[9], qd[4];
    if (g_sophus_mode) {
        float uv[3], c2[3];
        orc_quat_from_rows(T, qd);
        orc_cross(qd, p, uv);
        uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
        orc_cross(qd, uv, c2);
        ux = ((p[0] + qd[3] * uv[0]) + c2[0]) + T[3];
        uy = ((p[1] + qd[3] * uv[1]) + c2[1]) + T[7];
        uz = ((p[2] + qd[3] * uv[2]) + c2[2]) + T[11];
    } else {
        ux = fmaf(T[2], p[2], fmaf(T[1], p[1], fmaf(T[0], p[0], T[3])));
        uy = fmaf(T[6], p[2], fmaf(T[5], p[1], fmaf(T[4], p[0], T[7])));
        uz = fmaf(T[10], p[2], fmaf(T[9], p[1], fmaf(T[8], p[0], T[11])));
    }
    /* grid coordinates, .cu:137-142 */
    const int d0 = dim[0], d1 = dim[1], d2 = dim[2];
    const float gx = (ux - lo[0]) / (hi[0] - lo[0]) * (float)d0;
    const float gy = (uy - lo[1]) / (hi[1] - lo[1]) * (float)d1;
    const float gz = (uz - lo[2]) / (hi[2] - lo[2]) * (float)d2;
    const float value = orc_sdf_value(gx, gy, gz, d0, d1, d2, grid); /* .cu:147 */
    if (value < clr) *col += 1.0f;                                      /* .cu:150-151 */
    if (!(value <= eps)) return; /* .cu:170-171 `else continue` — the gradient below would be unused */
    /* getGradientInterpolated, .cu:66-86: central differences one VOXEL apart, divided by delta */
    const float fpx = orc_sdf_value(gx + 1.0f, gy, gz, d0, d1, d2, grid);
    const float fpy = orc_sdf_value(gx, gy + 1.0f, gz, d0, d1, d2, grid);
    const float fpz = orc_sdf_value(gx, gy, gz + 1.0f, d0, d1, d2, grid);
    const float fmx = orc_sdf_value(gx - 1.0f, gy, gz, d0, d1, d2, grid);
    const float fmy = orc_sdf_value(gx, gy - 1.0f, gz, d0, d1, d2, grid);
    const float fmz = orc_sdf_value(gx, gy, gz - 1.0f, d0, d1, d2, grid);
    const float g0 = (float)(0.5 * (double)(fpx - fmx) / (double)delta);
    const float g1 = (float)(0.5 * (double)(fpy - fmy) / (double)delta);
    const float g2 = (float)(0.5 * (double)(fpz - fmz) / (double)delta);
    float v0, v1, v2;
    if (value <= 0.0f) { /* .cu:158-164 */
        *pot += (float)(-(double)value + 0.5 * (double)eps);
        v0 = -g0; v1 = -g1; v2 = -g2;
    } else { /* 0 < value <= eps, .cu:165-171 */
        const float d = value - eps;
        *pot += 1.0f / (2.0f * eps) * d * d * pad;
        const float ie = 1.0f / eps;
        v0 = ie * g0 * d * pad; v1 = ie * g1 * d * pad; v2 = ie * g2 * d * pad;
    }
    /* rotationMatrix.transpose() * vgrad, .cu:176-179 */
    if (g_sophus_mode) {  /* the matrix regenerated from the quaternion; Eigen's product: a plain sum of three products per row */
        orc_quat_matrix(qd, Rq);
        grad[0] += (Rq[0] * v0 + Rq[3] * v1) + Rq[6] * v2;
        grad[1] += (Rq[1] * v0 + Rq[4] * v1) + Rq[7] * v2;
        grad[2] += (Rq[2] * v0 + Rq[5] * v1) + Rq[8] * v2;
        return;
    }
    grad[0] += fmaf(T[8], v2, fmaf(T[4], v1, T[0] * v0));
    grad[1] += fmaf(T[9], v2, fmaf(T[5], v1, T[1] * v0));
    grad[2] += fmaf(T[10], v2, fmaf(T[6], v1, T[2] * v0));
}

/* sdf_loss_cuda_forward, .cu:204-262, with the reference's padded [O,X,Y,Z] layout.  The reference
 * reduces over objects with atomicAdd in nondeterministic order (.cu:185-195); here objects are
 * summed in index order. */
int orc_sdf_loss_forward(const float* pose_init, const float* sdf_grids, const float* sdf_limits,
                         const float* points, const float* epsilons, const float* padding_scales,
                         const float* clearances, const float* disables, int64_t N, int32_t O,
                         float* potentials, float* potential_grads, float* collides) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float pot = 0.0f, col = 0.0f, gr[3] = {0.0f, 0.0f, 0.0f};
        for (int o = 0; o < O; ++o) {
            if (disables[o] > 0.0f) continue; /* .cu:115-116 */
            const float* L = sdf_limits + 10 * o;
            const int dim[3] = {(int)L[6], (int)L[7], (int)L[8]};
            const float* grid = sdf_grids + (int64_t)o * dim[0] * dim[1] * dim[2];
            orc_sdf_pair(pose_init + 16 * o, L, L + 3, dim, L[9], epsilons[o], padding_scales[o], clearances[o],
                         grid, points + 3 * i, &pot, gr, &col);
        }
        potentials[i] = pot;
        collides[i] = col;
        potential_grads[3 * i + 0] = gr[0];
        potential_grads[3 * i + 1] = gr[1];
        potential_grads[3 * i + 2] = gr[2];
    }
    return 0;
}

/* Same op over the engine's object table (ragged pool). */
static void orc_sdf_point_table(const omgx_object* objs, int o_begin, int o_end, const float* pool, const float* p,
                                float* pot, float* grad, float* col) {
    *pot = 0.0f; *col = 0.0f; grad[0] = grad[1] = grad[2] = 0.0f;
    for (int o = o_begin; o < o_end; ++o) {
        const omgx_object* ob = objs + o;
        if (ob->disabled > 0) continue;
        orc_sdf_pair(ob->pose_inv, ob->lo, ob->hi, ob->dim, ob->delta, ob->epsilon, ob->padding_scale, ob->clearance,
                     pool + ob->grid_offset, p, pot, grad, col);
    }
}

/* =============================================================================================
 * 2. Forward kinematics — ycb_render/robotPose/robot_pykdl.py:148-215
 * =========================================================================================== */

static void m4_mul(const double* A, const double* B, double* C) { /* C = A B, row-major 4x4 */
    double t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * r + k] * B[4 * k + c];
            t[4 * r + c] = s;
        }
    memcpy(C, t, sizeof t);
}

/* q: 9-dof radians.  Applies wrap_values (omg/util.py:194-202: rad->deg, insert the dummy hand joint at
 * index 7) and the deg->rad of robot_pykdl.py:164.  Outputs: link_pose [10][16] (after center_offset,
 * offset=True), and when jorigin/jaxis != NULL the joint info computed BEFORE the offset
 * (robot_pykdl.py:190-201), including the quirk that `_joint_origin` is loaded from the `_joint_axis`
 * key (robot_pykdl.py:104): origin = R*axis_local + t. */
void orc_fk(const double* robot, const double* q, double* link_pose, double* jorigin, double* jaxis) {
    const double* pose0 = robot + OMGX_ROBOT_POSE0;
    const double* tip2joint = robot + OMGX_ROBOT_TIP2JOINT;
    const double* coff = robot + OMGX_ROBOT_CENTER_OFFSET;
    const double* ax = robot + OMGX_ROBOT_JOINT_AXIS;
    static const double offs[7] = {0.0, -M_PI, M_PI, M_PI, -M_PI, M_PI, M_PI}; /* robot_pykdl.py:167 */
    double j[10];
    for (int i = 0; i < 7; ++i) j[i] = (q[i] / M_PI * 180.0) / 180.0 * M_PI;
    j[7] = 0.0;
    j[8] = (q[7] / M_PI * 180.0) / 180.0 * M_PI;
    j[9] = (q[8] / M_PI * 180.0) / 180.0 * M_PI;

    double out[10][16];
    double cur[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 7; ++i) {
        const double c = cos(j[i]), s = sin(j[i]), co = cos(offs[i]), so = sin(offs[i]);
        const double Rz[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const double Rx[16] = {1, 0, 0, 0, 0, co, -so, 0, 0, so, co, 0, 0, 0, 0, 1};
        double m[16], b[16];
        m4_mul(Rz, Rx, m);             /* DH(): np.matmul(M, rotx), robot_pykdl.py:52-56 */
        m4_mul(pose0 + 16 * i, m, b);
        if (i > 0)
            for (int r = 0; r < 4; ++r) { b[4 * r + 1] = -b[4 * r + 1]; b[4 * r + 2] = -b[4 * r + 2]; } /* :176 */
        m4_mul(cur, b, cur);
        memcpy(out[i], cur, sizeof cur);
    }
    double lf[16], rf[16];
    memcpy(lf, pose0 + 16 * 8, sizeof lf); lf[4 * 1 + 3] += j[8]; /* :181-182 */
    memcpy(rf, pose0 + 16 * 9, sizeof rf); rf[4 * 1 + 3] -= j[9]; /* :183-184 */
    m4_mul(out[6], pose0 + 16 * 7, out[7]);
    m4_mul(out[7], lf, out[8]);
    m4_mul(out[7], rf, out[9]);

    for (int l = 0; l < NL; ++l) {
        if (jorigin && jaxis) {
            double jp[16];
            m4_mul(out[l], tip2joint + 16 * l, jp);
            const double* a = ax + 3 * l;
            for (int r = 0; r < 3; ++r) {
                const double w = jp[4 * r + 0] * a[0] + jp[4 * r + 1] * a[1] + jp[4 * r + 2] * a[2];
                jaxis[3 * l + r] = w;
                jorigin[3 * l + r] = w + jp[4 * r + 3]; /* pose2origin == pose2axis (quirk) */
            }
        }
        m4_mul(out[l], coff + 16 * l, link_pose + 16 * l); /* :203-204 */
    }
}

/* Cost.forward_points, omg/cost.py:60-72: x = R pts + t */
static inline void orc_point(const double* pose, const double* pt, double* x) {
    for (int r = 0; r < 3; ++r)
        x[r] = pose[4 * r + 0] * pt[0] + pose[4 * r + 1] * pt[1] + pose[4 * r + 2] * pt[2] + pose[4 * r + 3];
}

void orc_fk_batch(const double* robot, const double* joints, int64_t B, double* link_pose, double* jorigin,
                  double* jaxis) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        orc_fk(robot, joints + ND * b, link_pose + 160 * b, jorigin ? jorigin + 30 * b : NULL,
               jaxis ? jaxis + 30 * b : NULL);
}

/* =============================================================================================
 * 3. FK -> points -> SDF layer for a batch of configurations
 *    Cost.batch_obstacle_cost (arc_length <= 0), omg/cost.py:192-232 + compute_obstacle_cost_layer 288-360
 * =========================================================================================== */
int orc_fk_sdf(const double* robot, int32_t P, const omgx_object* objects, const int32_t* scene_begin,
               const float* sdf_pool, const double* joints, int32_t S, int32_t C, int32_t soften_fingers,
               float* potentials, float* grads, float* collides) {
    if (P < 1 || P > MAXP) return OMGX_ERR_UNSUPPORTED;
    const double* pts = robot + OMGX_ROBOT_POINTS;
    const int64_t B = (int64_t)S * C;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const int s = (int)(b / C);
        double pose[160];
        orc_fk(robot, joints + ND * b, pose, NULL, NULL);
        for (int l = 0; l < NL; ++l)
            for (int p = 0; p < P; ++p) {
                double x[3];
                orc_point(pose + 16 * l, pts + 3 * (l * P + p), x);
                const float xf[3] = {(float)x[0], (float)x[1], (float)x[2]}; /* .cuda().float(), cost.py:218 */
                float pot, col, g[3];
                orc_sdf_point_table(objects, scene_begin[s], scene_begin[s + 1], sdf_pool, xf, &pot, g, &col);
                if (soften_fingers && l >= NL - 2) { /* cost.py:350-353 */
                    pot *= 0.1f; g[0] *= 0.1f; g[1] *= 0.1f; g[2] *= 0.1f; col = 0.0f;
                }
                const int64_t k = (b * NL + l) * P + p;
                if (potentials) potentials[k] = pot;
                if (collides) collides[k] = col;
                if (grads) { grads[3 * k] = g[0]; grads[3 * k + 1] = g[1]; grads[3 * k + 2] = g[2]; }
            }
    }
    return 0;
}

/* =============================================================================================
 * 4. Goal-set cost — Learner.cost_vector's device work, omg/online_learner.py:104-148
 * =========================================================================================== */
int orc_goalset_cost(const double* robot, int32_t P, const omgx_object* objects, const int32_t* scene_begin,
                     const float* sdf_pool, const double* traj_start, const double* goals, int32_t S, int32_t G,
                     int32_t n, double dt, int32_t soften_fingers, float* goal_cost, float* potentials,
                     float* collides) {
    if (P < 1 || P > MAXP || n < 1 || n > MAXN) return OMGX_ERR_UNSUPPORTED;
    const double* pts = robot + OMGX_ROBOT_POINTS;
    const float inv_dt = (float)(1.0 / dt); /* diff_matrices_torch entries, config.py:222-225 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t sg = 0; sg < (int64_t)S * G; ++sg) {
        const int s = (int)(sg / G);
        const double* q0 = traj_start + ND * s;
        const double* qg = goals + ND * sg;
        float prev[NL * MAXP * 3], cur[NL * MAXP * 3];
        double pose[160], x[3];
        /* ws_positions_start: FK of traj_start, cost.py:240-255 */
        orc_fk(robot, q0, pose, NULL, NULL);
        for (int l = 0; l < NL; ++l)
            for (int p = 0; p < P; ++p) {
                orc_point(pose + 16 * l, pts + 3 * (l * P + p), x);
                float* d = prev + 3 * (l * P + p);
                d[0] = (float)x[0]; d[1] = (float)x[1]; d[2] = (float)x[2];
            }
        float total = 0.0f, ncol = 0.0f;
        for (int i = 0; i < n; ++i) {
            /* multi_interpolate_waypoints "linear": t = linspace(0,1,n+2)[1:-1], util.py:261-290 */
            const double t = (double)(i + 1) * (1.0 / (double)(n + 1)); /* numpy linspace: i * step, step = fl(1 / (n + 1)) */
            double q[ND];
            for (int d = 0; d < ND; ++d) q[d] = q0[d] + t * (qg[d] - q0[d]);
            orc_fk(robot, q, pose, NULL, NULL);
            float wsum = 0.0f;
            for (int l = 0; l < NL; ++l)
                for (int p = 0; p < P; ++p) {
                    orc_point(pose + 16 * l, pts + 3 * (l * P + p), x);
                    float* c = cur + 3 * (l * P + p);
                    const float* pv = prev + 3 * (l * P + p);
                    c[0] = (float)x[0]; c[1] = (float)x[1]; c[2] = (float)x[2];
                    float pot, col, g[3];
                    orc_sdf_point_table(objects, scene_begin[s], scene_begin[s + 1], sdf_pool, c, &pot, g, &col);
                    if (soften_fingers && l >= NL - 2) { pot *= 0.1f; col = 0.0f; }
                    /* get_derivative_torch (config.py:162-187): (x_i - x_{i-1}) / dt in float32 */
                    const float vx = (c[0] - pv[0]) * inv_dt, vy = (c[1] - pv[1]) * inv_dt, vz = (c[2] - pv[2]) * inv_dt;
                    const float w = pot * sqrtf(vx * vx + vy * vy + vz * vz); /* cost.py:264-275 */
                    if (potentials) potentials[((sg * n + i) * NL + l) * P + p] = w;
                    wsum += w;
                    ncol += col;
                }
            total += wsum; /* torch.sum(..., (-2,-1)).reshape([-1,n]).sum(-1), online_learner.py:145-148 */
            memcpy(prev, cur, sizeof(float) * NL * P * 3);
        }
        goal_cost[sg] = total;
        if (collides) collides[sg] = ncol;
    }
    return 0;
}

/* =============================================================================================
 * 5. CHOMP cost + gradient + step for one trajectory
 * =========================================================================================== */

/* wrap_joint(j+1), omg/util.py:213-220: indices into the 10-joint arrays.  Returns count. */
static int orc_wrap_joint(int link, int* idx) {
    int k = 0;
    if (link < 7) { for (int i = 0; i <= link; ++i) idx[k++] = i; return k; }
    for (int i = 0; i < 7; ++i) idx[k++] = i;
    if (link == 8) idx[k++] = 8;
    if (link == 9) idx[k++] = 9;
    return k;
}
/* wrap_index(j+1), omg/util.py:205-210: columns of the 9-dof trajectory. */
static int orc_wrap_index(int link, int* idx) {
    int k = 0;
    if (link < 7) { for (int i = 0; i <= link; ++i) idx[k++] = i; return k; }
    for (int i = 0; i < 7; ++i) idx[k++] = i;
    if (link == 8) idx[k++] = 7;
    if (link == 9) idx[k++] = 8;
    return k;
}

/* Cost.functional_grad for ONE point (omg/cost.py:24-43) combined with compute_point_jacobian
 * (cost.py:92-110).  c, dc are the float32 SDF potential / gradient, promoted to double.
 * Returns c*||v|| and writes J.g for the k joints of this link. */
static double orc_point_grad(const double* x, const double* v, const double* a, double c, const double* dc,
                             int link, const double* jorigin /*[10][3]*/, const double* jaxis /*[10][3]*/,
                             double* out /*[<=8]*/, int* kout) {
    const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); /* np.linalg.norm */
    double nv[3], Pm[9], Pa[3], Pg[3], g[3];
    for (int r = 0; r < 3; ++r) nv[r] = v[r] / (vn + 1e-8); /* safe_div */
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) Pm[3 * r + q] = (r == q ? 1.0 : 0.0) - nv[r] * nv[q];
    for (int r = 0; r < 3; ++r) {
        Pa[r] = Pm[3 * r] * a[0] + Pm[3 * r + 1] * a[1] + Pm[3 * r + 2] * a[2];
        Pg[r] = Pm[3 * r] * dc[0] + Pm[3 * r + 1] * dc[1] + Pm[3 * r + 2] * dc[2];
    }
    for (int r = 0; r < 3; ++r) {
        const double kappa = c * (Pa[r] / (vn * vn + 1e-8)); /* safe_div(P a, ||v||**2), cost.py:35-37 */
        g[r] = vn * Pg[r] - kappa;
    }
    int jidx[8];
    const int k = orc_wrap_joint(link, jidx);
    for (int t = 0; t < k; ++t) {
        const double* ax = jaxis + 3 * jidx[t];
        const double* og = jorigin + 3 * jidx[t];
        double J[3];
        if (link >= 8 && t == k - 1) { /* "prsimatic" finger joint, cost.py:106-108 */
            J[0] = ax[0]; J[1] = ax[1]; J[2] = ax[2];
        } else {
            const double d[3] = {x[0] - og[0], x[1] - og[1], x[2] - og[2]};
            J[0] = ax[1] * d[2] - ax[2] * d[1];
            J[1] = ax[2] * d[0] - ax[0] * d[2];
            J[2] = ax[0] * d[1] - ax[1] * d[0];
        }
        out[t] = J[0] * g[0] + J[1] * g[1] + J[2] * g[2];
    }
    *kout = k;
    return c * vn;
}

typedef struct { float v; int idx; } orc_kv;
static int orc_kv_cmp(const void* a, const void* b) {
    const orc_kv* x = (const orc_kv*)a; const orc_kv* y = (const orc_kv*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* stable: ties in ascending flat index */
}

/* Dense inverse by Gauss-Jordan with partial pivoting (np.linalg.inv stand-in). */
static void orc_inv(const double* A, int n, double* X) {
    double* M = (double*)malloc(sizeof(double) * n * 2 * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] = c < n ? A[r * n + c] : (c - n == r ? 1.0 : 0.0);
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c])) piv = r;
        if (piv != c)
            for (int k = 0; k < 2 * n; ++k) { double t = M[c * 2 * n + k]; M[c * 2 * n + k] = M[piv * 2 * n + k]; M[piv * 2 * n + k] = t; }
        const double d = M[c * 2 * n + c];
        for (int k = 0; k < 2 * n; ++k) M[c * 2 * n + k] /= d;
        for (int r = 0; r < n; ++r)
            if (r != c) {
                const double f = M[r * 2 * n + c];
                if (f != 0.0)
                    for (int k = 0; k < 2 * n; ++k) M[r * 2 * n + k] -= f * M[c * 2 * n + k];
            }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) X[r * n + c] = M[r * 2 * n + n + c];
    free(M);
}

/* get_diff_matrix order 1 (omg/util.py:165-178) and A = D^T D, Ainv (omg/config.py:199-220). */
void orc_smooth_matrices(int n, double dt, int goal_set_proj, double* D /*[(n+1)*n]*/, double* A, double* Ainv) {
    memset(D, 0, sizeof(double) * (n + 1) * n);
    for (int i = 0; i <= n; ++i) {
        if (i - 1 >= 0 && i - 1 < n) D[i * n + i - 1] = -1.0;
        if (i < n) D[i * n + i] = 1.0;
    }
    if (goal_set_proj) D[n * n + n - 1] = 0.0; /* with_end == False */
    for (int i = 0; i < (n + 1) * n; ++i) D[i] /= dt;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            double s = 0.0;
            for (int k = 0; k <= n; ++k) s += D[k * n + r] * D[k * n + c];
            A[r * n + c] = s;
        }
    orc_inv(A, n, Ainv);
}

/* One Optimizer.optimize step for one trajectory (omg/optimizer.py:115-135), given the SDF layer
 * outputs for its n waypoint configurations.  See include/omg_hip.h (4) for the argument contract. */
int orc_chomp_optimize_one(const double* robot, const omgx_chomp_params* prm, double* traj, const double* start,
                           const double* end, const double* goal, const double* goal_point, const float* pot,
                           const float* pgrad, const float* col, double* grad_out, double* cost_traj, double* info) {
    const int n = prm->n_waypoints, P = prm->n_points, c = prm->constraint_num;
    if (n < 1 || n > MAXN || P < 1 || P > MAXP || c < 1 || c > OMGX_MAX_CONSTRAINTS || c > n) return OMGX_ERR_UNSUPPORTED;
    const double dt = prm->time_interval;
    const double* pts = robot + OMGX_ROBOT_POINTS;
    const double* lower = robot + OMGX_ROBOT_LOWER;
    const double* upper = robot + OMGX_ROBOT_UPPER;

    /* ---- forward_kinematics_obstacle, cost.py:112-190 ---- */
    double* pose = (double*)malloc(sizeof(double) * (n + 2) * 160);
    double* jorg = (double*)malloc(sizeof(double) * n * 30);
    double* jax = (double*)malloc(sizeof(double) * n * 30);
    double* X = (double*)malloc(sizeof(double) * (n + 2) * NL * P * 3); /* row 0 = start, 1..n, n+1 = end */
    orc_fk(robot, start, pose, NULL, NULL);
    for (int i = 0; i < n; ++i) orc_fk(robot, traj + ND * i, pose + 160 * (i + 1), jorg + 30 * i, jax + 30 * i);
    orc_fk(robot, end, pose + 160 * (n + 1), NULL, NULL);
    for (int i = 0; i < n + 2; ++i)
        for (int l = 0; l < NL; ++l)
            for (int p = 0; p < P; ++p) orc_point(pose + 160 * i + 16 * l, pts + 3 * (l * P + p), X + 3 * ((i * NL + l) * P + p));
#define XP(i, l, p) (X + 3 * ((((i) + 1) * NL + (l)) * P + (p)))
    const double idt = dt, idt2 = dt * dt;

    /* ---- compute_collision_loss, cost.py:362-423 ---- */
    double* obs_grad = (double*)calloc((size_t)n * ND, sizeof(double));
    double* obs_cost = (double*)calloc((size_t)n * NL, sizeof(double));
    const int total = n * NL * P;
    if (prm->top_k == 0) { /* clean branch, cost.py:380-388: every point of every link (fingers included) */
        for (int l = 0; l < NL; ++l) {
            int cols[8]; const int kc = orc_wrap_index(l, cols);
            for (int i = 0; i < n; ++i) {
                double acc[8] = {0}; double csum = 0.0;
                for (int p = 0; p < P; ++p) {
                    const double *x = XP(i, l, p), *xm = XP(i - 1, l, p), *xp = XP(i + 1, l, p);
                    double v[3], a[3], dc[3], o[8]; int k;
                    for (int r = 0; r < 3; ++r) { v[r] = (x[r] - xm[r]) / idt; a[r] = (xm[r] - 2.0 * x[r] + xp[r]) / idt2; }
                    const int64_t f = ((int64_t)i * NL + l) * P + p;
                    for (int r = 0; r < 3; ++r) dc[r] = (double)pgrad[3 * f + r];
                    csum += orc_point_grad(x, v, a, (double)pot[f], dc, l, jorg + 30 * i, jax + 30 * i, o, &k);
                    for (int t = 0; t < k; ++t) acc[t] += o[t];
                }
                obs_cost[i * NL + l] += csum;
                for (int t = 0; t < kc; ++t) obs_grad[i * ND + cols[t]] += acc[t];
            }
        }
    } else { /* top-k branch, cost.py:390-421 */
        orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * total);
        for (int f = 0; f < total; ++f) { kv[f].v = pot[f]; kv[f].idx = f; }
        qsort(kv, total, sizeof(orc_kv), orc_kv_cmp); /* np.argsort(potentials.flatten()) */
        const int K = prm->top_k < total ? prm->top_k : total;
        const orc_kv* top = kv + (total - K); /* [-top_k:] in ascending order */
        const int mlinks = prm->consider_finger ? NL : NL - 2; /* cost.py:401-404 */
        for (int l = 0; l < mlinks; ++l) {
            int cols[8]; const int kc = orc_wrap_index(l, cols);
            double csum = 0.0; int any = 0;
            /* duplicate (waypoint, link) targets of the fancy-index `+=` resolve to the LAST selected
             * point (cost.py:421): remember the base value once, overwrite with base + contribution. */
            double* base = (double*)malloc(sizeof(double) * n * ND);
            memcpy(base, obs_grad, sizeof(double) * n * ND);
            for (int t = 0; t < K; ++t) {
                const int f = top[t].idx;
                const int i = f / (NL * P), lf = (f / P) % NL, p = f % P;
                if (lf != l) continue;
                any = 1;
                const double *x = XP(i, l, p), *xm = XP(i - 1, l, p), *xp = XP(i + 1, l, p);
                double v[3], a[3], dc[3], o[8]; int k;
                for (int r = 0; r < 3; ++r) { v[r] = (x[r] - xm[r]) / idt; a[r] = (xm[r] - 2.0 * x[r] + xp[r]) / idt2; }
                for (int r = 0; r < 3; ++r) dc[r] = (double)pgrad[3 * f + r];
                csum += orc_point_grad(x, v, a, (double)pot[f], dc, l, jorg + 30 * i, jax + 30 * i, o, &k);
                for (int u = 0; u < kc; ++u) obs_grad[i * ND + cols[u]] = base[i * ND + cols[u]] + o[u];
            }
            free(base);
            if (any)
                for (int i = 0; i < n; ++i) obs_cost[i * NL + l] += csum; /* scalar broadcast, cost.py:416 */
        }
        free(kv);
    }
    double collide = 0.0; /* collide.sum(), cost.py:187 (float32 tensor of small integers) */
    for (int f = 0; f < total; ++f) collide += (double)col[f];

    /* ---- compute_smooth_loss, cost.py:425-449 ---- */
    const double* w = prm->link_smooth_weight;
    double* sm_grad = (double*)calloc((size_t)n * ND, sizeof(double));
    double* sm_loss = (double*)calloc((size_t)n + 1, sizeof(double));
    double *D = (double*)malloc(sizeof(double) * (n + 1) * n), *A = (double*)malloc(sizeof(double) * n * n),
           *Ainv = (double*)malloc(sizeof(double) * n * n);
    orc_smooth_matrices(n, dt, prm->goal_set_proj, D, A, Ainv);
    double* ed = (double*)calloc((size_t)(n + 1) * ND, sizeof(double));
    for (int d = 0; d < ND; ++d) {
        ed[d] = -1.0 * start[d] / dt;
        if (!prm->goal_set_proj) ed[n * ND + d] = 1.0 * end[d] / dt;
    }
    for (int i = 0; i <= n; ++i) {
        double s2 = 0.0;
        for (int d = 0; d < ND; ++d) {
            double vel = 0.0;
            for (int k = 0; k < n; ++k) vel += D[i * n + k] * traj[k * ND + d];
            const double e = (vel + ed[i * ND + d]) * w[d];
            s2 += e * e;
        }
        const double nrm = sqrt(s2);
        sm_loss[i] = 0.5 * nrm * nrm;
    }
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < ND; ++d) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * traj[k * ND + d];
            double s2 = 0.0;
            for (int k = 0; k <= n; ++k) s2 += D[k * n + i] * ed[k * ND + d];
            sm_grad[i * ND + d] = (s + s2) * w[d];
        }

    /* ---- compute_total_loss, cost.py:451-532 ---- */
    double smooth_sum = 0.0, obs_sum = 0.0;
    for (int i = 0; i <= n; ++i) smooth_sum += sm_loss[i];
    for (int i = 0; i < n * NL; ++i) obs_sum += obs_cost[i];
    const double w_obs = prm->obstacle_weight * obs_sum, w_sm = prm->smoothness_weight * smooth_sum;
    double n_og = 0.0, n_sg = 0.0, n_g = 0.0;
    for (int i = 0; i < n * ND; ++i) {
        double og = prm->obstacle_weight * obs_grad[i];
        if (og > prm->clip_grad_scale) og = prm->clip_grad_scale;
        if (og < -prm->clip_grad_scale) og = -prm->clip_grad_scale;
        const double sg = prm->smoothness_weight * sm_grad[i];
        grad_out[i] = og + sg;
        n_og += og * og; n_sg += sg * sg; n_g += grad_out[i] * grad_out[i];
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int l = 0; l < NL; ++l) s += obs_cost[i * NL + l];
        cost_traj[i] = prm->obstacle_weight * s + prm->smoothness_weight * sm_loss[i];
    }
    double goal_dist = 0.0;
    if (prm->goal_set_proj) { /* ||traj.data[-1] - goal_set[goal_idx]||, cost.py:483-487 */
        for (int d = 0; d < ND; ++d) { const double e = traj[(n - 1) * ND + d] - goal_point[d]; goal_dist += e * e; }
        goal_dist = sqrt(goal_dist);
    }
    int terminate = (collide <= prm->allow_collision_point) && prm->pre_terminate && (goal_dist < 0.01) &&
                    (smooth_sum < prm->terminate_smooth_loss);
    const int failure = (collide >= prm->allow_collision_point * 10) || (smooth_sum >= prm->terminate_smooth_loss * 2.5);
    const int execute = (collide <= prm->allow_collision_point) && (smooth_sum < prm->terminate_smooth_loss);

    /* ---- check_joint_limit, optimizer.py:166-174 (needs BOTH a low and a high violation) ---- */
    int any_low = 0, any_both = 0;
    for (int i = 0; i < n * ND; ++i) if (traj[i] < lower[i % ND] - 5e-3) any_low = 1;
    for (int i = 0; i < n * ND; ++i) if (any_low && traj[i] > upper[i % ND] + 5e-3) any_both = 1;
    terminate = terminate && !any_both;

    info[OMGX_INFO_COST] = w_obs + w_sm;
    info[OMGX_INFO_OBS] = obs_sum;
    info[OMGX_INFO_SMOOTH] = smooth_sum;
    info[OMGX_INFO_WEIGHTED_OBS] = w_obs;
    info[OMGX_INFO_WEIGHTED_SMOOTH] = w_sm;
    info[OMGX_INFO_WEIGHTED_OBS_GRAD] = sqrt(n_og);
    info[OMGX_INFO_WEIGHTED_SMOOTH_GRAD] = sqrt(n_sg);
    info[OMGX_INFO_GRAD] = sqrt(n_g);
    info[OMGX_INFO_COLLIDE] = collide;
    info[OMGX_INFO_REACH] = goal_dist;
    info[OMGX_INFO_TERMINATE] = terminate;
    info[OMGX_INFO_FAILURE_TERMINATE] = failure;
    info[OMGX_INFO_EXECUTE] = execute;
    info[OMGX_INFO_STANDOFF_IDX] = prm->use_standoff ? n - prm->constraint_num : n - 1;
    info[OMGX_INFO_VIOLATE_LIMIT] = any_both;
    info[OMGX_INFO_LIMIT_STEPS] = 0;

    /* optimizer.py:126-127: `if (info["terminate"] and not force_update) or info_only: return` — do_update 2 = no force_update */
    if (prm->do_update == 1 || (prm->do_update == 2 && !terminate)) {
        /* ---- goal_set_projection (optimizer.py:88-113) or plain step (:132) ---- */
        double* Ag = (double*)malloc(sizeof(double) * n * ND);
        double* upd = (double*)malloc(sizeof(double) * n * ND);
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < ND; ++d) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += Ainv[i * n + k] * grad_out[k * ND + d];
                Ag[i * ND + d] = s;
            }
        if (prm->goal_set_proj) {
            /* M = Ainv C^T (C Ainv C^T)^-1 with C = [0 I_c] */
            double B[OMGX_MAX_CONSTRAINTS * OMGX_MAX_CONSTRAINTS], Bi[OMGX_MAX_CONSTRAINTS * OMGX_MAX_CONSTRAINTS];
            for (int r = 0; r < c; ++r)
                for (int q = 0; q < c; ++q) B[r * c + q] = Ainv[(n - c + r) * n + (n - c + q)];
            orc_inv(B, c, Bi);
            double* M = (double*)malloc(sizeof(double) * n * c);
            for (int i = 0; i < n; ++i)
                for (int q = 0; q < c; ++q) {
                    double s = 0.0;
                    for (int r = 0; r < c; ++r) s += Ainv[i * n + (n - c + r)] * Bi[r * c + q];
                    M[i * c + q] = s;
                }
            for (int i = 0; i < n; ++i)
                for (int d = 0; d < ND; ++d) {
                    double s1 = 0.0, s2 = 0.0;
                    for (int q = 0; q < c; ++q) {
                        s1 += M[i * c + q] * Ag[(n - c + q) * ND + d];                       /* M C Ainv g */
                        s2 += M[i * c + q] * (traj[(n - c + q) * ND + d] - goal[q * ND + d]); /* M b */
                    }
                    upd[i * ND + d] = -prm->step_size * Ag[i * ND + d] + prm->step_size * s1 - s2;
                }
            free(M);
        } else {
            for (int i = 0; i < n * ND; ++i) upd[i] = -prm->step_size * Ag[i];
        }
        /* Trajectory.update, omg/core.py:43-51 */
        for (int i = 0; i < n; ++i) {
            const int nd = prm->consider_finger ? ND : ND - 2;
            for (int d = 0; d < nd; ++d) traj[i * ND + d] += upd[i * ND + d];
            for (int d = ND - 2; d < ND; ++d) traj[i * ND + d] = fmin(fmax(traj[i * ND + d], 0.0), 0.04);
        }
        /* handle_joint_limit, optimizer.py:148-164 */
        double* tv = (double*)malloc(sizeof(double) * n * ND);
        double* tvs = (double*)malloc(sizeof(double) * n * ND);
        int cnt = 0;
        for (;;) {
            double nrm = 0.0;
            for (int i = 0; i < n * ND; ++i) { /* compute_traj_v */
                const int d = i % ND;
                tv[i] = (traj[i] < lower[d] ? lower[d] - traj[i] : 0.0) + (traj[i] > upper[d] ? upper[d] - traj[i] : 0.0);
                nrm += tv[i] * tv[i];
            }
            if (!(sqrt(nrm) > 1e-2) || cnt >= prm->joint_limit_max_steps) break;
            for (int i = 0; i < n; ++i)
                for (int d = 0; d < ND; ++d) {
                    double s = 0.0;
                    for (int k = 0; k < n; ++k) s += Ainv[i * n + k] * tv[k * ND + d];
                    tvs[i * ND + d] = s;
                }
            int am = 0; /* np.abs(traj_v).argmax(): first maximum in flat order, NaN wins */
            for (int i = 1; i < n * ND; ++i) {
                if (tv[am] != tv[am]) break;
                if (tv[i] != tv[i] || fabs(tv[i]) > fabs(tv[am])) am = i;
            }
            const double scale = fabs(tv[am]) / (fabs(tvs[am]) + 1e-8);
            for (int i = 0; i < n * ND; ++i) traj[i] += scale * tvs[i];
            ++cnt;
        }
        info[OMGX_INFO_LIMIT_STEPS] = cnt;
        free(tv); free(tvs); free(Ag); free(upd);
    }
#undef XP
    free(pose); free(jorg); free(jax); free(X); free(obs_grad); free(obs_cost); free(sm_grad); free(sm_loss);
    free(D); free(A); free(Ainv); free(ed);
    return 0;
}

/* Batched form of (4): S independent trajectories (scenes). */
int orc_chomp_optimize(const double* robot, const omgx_chomp_params* prm, double* traj, const double* start,
                       const double* end, const double* goal, const double* goal_point, const float* pot,
                       const float* pgrad, const float* col, const int32_t* active, int32_t S, double* grad, double* cost_traj, double* info) {
    const int n = prm->n_waypoints, P = prm->n_points, c = prm->constraint_num;
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < S; ++s) {
        if (active && !active[s]) continue;
        const int64_t np = (int64_t)s * n * NL * P;
        const int r = orc_chomp_optimize_one(robot, prm, traj + (int64_t)s * n * ND, start + ND * s, end + ND * s,
                                             goal + (int64_t)s * c * ND, goal_point + ND * s, pot + np, pgrad + 3 * np, col + np,
                                             grad + (int64_t)s * n * ND, cost_traj + (int64_t)s * n,
                                             info + (int64_t)s * OMGX_INFO_STRIDE);
        if (r != 0) rc = r;
    }
    return rc;
}

/* =============================================================================================
 * 6. Learner.update_goal — omg/online_learner.py
 * =========================================================================================== */

/* numpy's pairwise summation (add.reduce on a contiguous float64 vector): plain loop below 8 elements,
 * 8 strided partial sums up to 128, recursion above. */
static double np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return n ? r : 0.0;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

static double orc_sign(double y) { return (y > 0) - (y < 0); }

/* bp (online_learner.py:32-58) with find_zero (16-29) inlined: Bregman projection onto the simplex. */
static void orc_bp(const double* x, const double* v, const double* delta, const double* w, int n, double* y) {
    const int max_iter = 100; const double err = 1e-6;
    double *alpha = (double*)calloc(n, sizeof(double)), *z = (double*)malloc(sizeof(double) * n),
           *shiftx = (double*)malloc(sizeof(double) * n), *tmp = (double*)malloc(sizeof(double) * n),
           *ap = (double*)malloc(sizeof(double) * n);
    for (int it = 0; it < max_iter; ++it) {
        for (int i = 0; i < n; ++i) { z[i] = (alpha[i] - v[i]) / w[i]; shiftx[i] = x[i] + delta[i]; }
        const double target = 1.0 + np_sum(delta, n);
        double x1 = w[0] + v[0];
        for (int i = 1; i < n; ++i) if (w[i] + v[i] > x1) x1 = w[i] + v[i];
        /* find_zero(f, 0, max(w + v), err, max_iter) */
        double L = (0.0 + x1) / 2.0, sstep = (x1 - 0.0) / 4.0;
        for (int k = 0; k < max_iter; ++k) {
            for (int i = 0; i < n; ++i) tmp[i] = shiftx[i] * exp(L / w[i] + z[i]);
            const double fy = np_sum(tmp, n) - target;
            if (fabs(fy) < err) break;
            L -= sstep * orc_sign(fy);
            sstep /= 2.0;
        }
        for (int i = 0; i < n; ++i) y[i] = shiftx[i] * exp((L + alpha[i] - v[i]) / w[i]) - delta[i];
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) {
            ap[i] = fmax(0.0, v[i] - L + w[i] * log(delta[i] / shiftx[i]));
            nrm += (alpha[i] - ap[i]) * (alpha[i] - ap[i]);
        }
        if (sqrt(nrm) < err) break;
        memcpy(alpha, ap, sizeof(double) * n);
    }
    for (int i = 0; i < n; ++i) tmp[i] = y[i] = fmax(y[i], 0.0);
    const double sy = np_sum(tmp, n);
    for (int i = 0; i < n; ++i) y[i] /= sy;
    free(alpha); free(z); free(shiftx); free(tmp); free(ap);
}

/* np.argmin / np.argmax: first occurrence of the extreme; a NaN counts as the extreme for both (numpy propagates NaN) */
static int np_argext(const double* a, int n, int want_min) {
    int idx = 0;
    for (int g = 1; g < n; ++g) {
        if (a[idx] != a[idx]) break; /* the first NaN stays */
        if (a[g] != a[g] || (want_min ? a[g] < a[idx] : a[g] > a[idx])) idx = g;
    }
    return idx;
}

int orc_goal_update(const omgx_learner_params* prm, const double* traj, const double* goal_set, const double* reach,
                    const float* goal_cost, double* state, int32_t S, int32_t* goal_idx, double* end, double* goal_rows,
                    double* goal_point, double* cost_vector) {
    const int G = prm->num_goals, n = prm->n_waypoints, c = prm->constraint_num;
    if (G < 1 || G > OMGX_MAX_GOALS) return OMGX_ERR_UNSUPPORTED;
    const int64_t SS = 7 * (int64_t)G + 10;
    for (int s = 0; s < S; ++s) {
        double* st = state + s * SS;
        double *sum_costs = st, *p = st + G, *experts_p = st + 2 * G, *q = st + 7 * G, *ecost = st + 7 * G + 5;
        const double* gs = goal_set + (int64_t)s * G * ND;
        double* cv = (double*)malloc(sizeof(double) * G);
        double* tmp = (double*)malloc(sizeof(double) * G);
        int idx = 0;
        if (prm->alg == OMGX_ALG_PROJ) { /* online_learner.py:196-206: closest goal to the last waypoint */
            const double* last = traj + ((int64_t)s * n + n - 1) * ND;
            for (int g = 0; g < G; ++g) {
                double d2 = 0.0;
                for (int d = 0; d < ND; ++d) { const double e = last[d] - gs[g * ND + d]; d2 += e * e; }
                tmp[g] = sqrt(d2);
                p[g] = 0.0;
            }
            idx = np_argext(tmp, G, 1);
            p[idx] = 1.0;
        } else {
            /* cost_vector tail, online_learner.py:145-160 */
            const double* ts = traj + ((int64_t)s * n + prm->start_idx) * ND;
            for (int g = 0; g < G; ++g) {
                double s2 = 0.0;
                for (int d = 0; d + 1 < ND; ++d) { /* np.diff(traj_start - goal_set, axis=-1) */
                    const double a = (ts[d + 1] - gs[g * ND + d + 1]) - (ts[d] - gs[g * ND + d]);
                    s2 += a * a;
                }
                const double nr = sqrt(s2);
                const float wc = (float)prm->base_obstacle_weight * goal_cost[(int64_t)s * G + g]; /* float32 product */
                cv[g] = (double)wc + prm->smooth_weight * (nr * nr);
            }
            if (prm->normalize_cost) {
                double nn = 0.0;
                for (int g = 0; g < G; ++g) nn += cv[g] * cv[g];
                nn = sqrt(nn);
                for (int g = 0; g < G; ++g) cv[g] /= nn;
            }
            if (cost_vector) memcpy(cost_vector + (int64_t)s * G, cv, sizeof(double) * G);
            if (prm->alg == OMGX_ALG_FTL || prm->alg == OMGX_ALG_FTC) { /* :175-189 */
                const double* key = cv;
                if (prm->alg == OMGX_ALG_FTL) { for (int g = 0; g < G; ++g) sum_costs[g] += cv[g]; key = sum_costs; }
                idx = np_argext(key, G, 1);
                for (int g = 0; g < G; ++g) p[g] = 0.0;
                p[idx] = 1.0;
            } else if (prm->alg == OMGX_ALG_EXP) { /* :208-217 */
                for (int g = 0; g < G; ++g) sum_costs[g] += cv[g];
                const double tot = np_sum(sum_costs, G);
                for (int g = 0; g < G; ++g) {
                    const double pn = exp(-prm->eta * cv[g]) * p[g];
                    p[g] = pn * 0.999 + (sum_costs[g] / (tot + 1e-8)) * 0.001;
                }
                const double ps = np_sum(p, G);
                for (int g = 0; g < G; ++g) p[g] = p[g] / (ps + 1e-8);
            } else { /* MD, :219-235 */
                static const double pw[5] = {0.25, 0.5, 1.0, 4.0, 16.0}; /* eta * 2**[-2,-1,0,2,4] */
                double *v = (double*)malloc(sizeof(double) * G), *delta = (double*)malloc(sizeof(double) * G),
                       *w = (double*)malloc(sizeof(double) * G), *pn = (double*)malloc(sizeof(double) * G);
                for (int g = 0; g < G; ++g) { delta[g] = 1.0 / (4.0 * G + 1.0); w[g] = 1.0; }
                for (int i = 0; i < 5; ++i) {
                    double* ep = experts_p + (int64_t)i * G;
                    for (int g = 0; g < G; ++g) v[g] = prm->eta * pw[i] * cv[g];
                    orc_bp(ep, v, delta, w, G, pn);
                    double dcv = 0.0, dw = 0.0;
                    for (int g = 0; g < G; ++g) { dcv += cv[g] * pn[g]; dw += w[g] * fabs(pn[g] - ep[g]); }
                    ecost[i] = dcv + dw;
                    memcpy(ep, pn, sizeof(double) * G);
                    /* the mixture update sits INSIDE the expert loop (online_learner.py:231-235) */
                    for (int k = 0; k < 5; ++k) q[k] = q[k] * exp(-1.0 * ecost[k]);
                    const double qs = np_sum(q, 5);
                    for (int k = 0; k < 5; ++k) q[k] /= qs;
                    for (int g = 0; g < G; ++g) {
                        double m = 0.0;
                        for (int k = 0; k < 5; ++k) m += experts_p[(int64_t)k * G + g] * q[k];
                        p[g] = m;
                    }
                    const double ps = np_sum(p, G);
                    for (int g = 0; g < G; ++g) p[g] /= ps;
                }
                free(v); free(delta); free(w); free(pn);
            }
            if (prm->alg == OMGX_ALG_EXP || prm->alg == OMGX_ALG_MD) { /* np.argmax(self.p), :243 */
                idx = np_argext(p, G, 0);
            }
        }
        goal_idx[s] = idx;
        for (int d = 0; d < ND; ++d) { end[s * ND + d] = gs[idx * ND + d]; goal_point[s * ND + d] = gs[idx * ND + d]; }
        for (int r = 0; r < c; ++r)
            for (int d = 0; d < ND; ++d)
                goal_rows[((int64_t)s * c + r) * ND + d] =
                    prm->use_standoff ? reach[(((int64_t)s * G + idx) * c + r) * ND + d] : gs[idx * ND + d];
        free(cv); free(tmp);
    }
    return 0;
}

/* =============================================================================================
 * 7. Point-cloud SDF — PointEnv.compute_sdf_from_points, omg/core.py:426-457 (cKDTree.query k=1, p=2, brute force here)
 * =========================================================================================== */
int orc_point_cloud_sdf(const double* points, int32_t N, const double* origin, double res, const int32_t* dims, float* out) {
    const int64_t total = (int64_t)dims[0] * dims[1] * dims[2];
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < total; ++id) {
        const int k = (int)(id % dims[2]), j = (int)((id / dims[2]) % dims[1]), i = (int)(id / ((int64_t)dims[2] * dims[1]));
        /* np.arange fills start, start + step, start + i * ((start + step) - start)  (numpy DOUBLE_fill) */
        double xyz[3];
        const int ijk[3] = {i, j, k};
        for (int a = 0; a < 3; ++a) {
            const double second = origin[a] + res, delta = second - origin[a];
            xyz[a] = ijk[a] == 0 ? origin[a] : (ijk[a] == 1 ? second : origin[a] + (double)ijk[a] * delta);
        }
        const double x = xyz[0], y = xyz[1], z = xyz[2];
        double best = 1.0e300;
        for (int q = 0; q < N; ++q) {
            const double a = points[3 * q] - x, b = points[3 * q + 1] - y, c = points[3 * q + 2] - z;
            const double d2 = (a * a + b * b) + c * c;
            if (d2 < best) best = d2;
        }
        out[id] = (float)sqrt(best);
    }
    return 0;
}

/* Exposed pieces for fine-grained golden checks. */
void orc_points_of_config(const double* robot, int32_t P, const double* q, double* x /*[10][P][3]*/) {
    double pose[160];
    orc_fk(robot, q, pose, NULL, NULL);
    for (int l = 0; l < NL; ++l)
        for (int p = 0; p < P; ++p) orc_point(pose + 16 * l, robot + OMGX_ROBOT_POINTS + 3 * (l * P + p), x + 3 * (l * P + p));
}

int orc_sizeof_object(void) { return (int)sizeof(omgx_object); }
int orc_sizeof_params(void) { return (int)sizeof(omgx_chomp_params); }
int orc_sizeof_learner_params(void) { return (int)sizeof(omgx_learner_params); }

/* =============================================================================================
 * Checker of an arithmetic identity the HIP kernels rely on (omg_device.h: div_by_const): for a constant c,
 *   q = x * RN(1/c);  r = fma(-q, c, x);  q' = fma(r, RN(1/c), q)
 * is the IEEE quotient x / c.  Returns how many of n pseudo-random arguments (joint-angle, degree and wide exponent
 * ranges) it is not.  The oracle itself divides (orc_fk); this function only counts.
 * =========================================================================================== */
int64_t orc_div_by_const_mismatches(double c, int64_t n, uint64_t seed) {
    const double rc = 1.0 / c;
    uint64_t s = seed ? seed : 88172645463325252ull;
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        double x;
        switch (i & 3) {
            case 0: x = (u - 0.5) * 14.0; break;
            case 1: x = (u - 0.5) * 2000.0; break;
            case 2: x = ldexp(1.0 + u, (int)(s % 60) - 30) * (((s >> 3) & 1) ? 1.0 : -1.0); break;
            default: x = (u - 0.5) * 0.1;
        }
        const double q = x * rc;
        const double r = fma(-q, c, x);
        if (fma(r, rc, q) != x / c) ++bad;
    }
    return bad;
}
