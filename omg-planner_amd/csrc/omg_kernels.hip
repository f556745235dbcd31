// omg_kernels.hip — gfx950 kernels + the C ABI of include/omg_hip.h (part 1: SDF / FK / goal-set).
//
// Kernels in this file
//   k_sdf_loss        points x objects SDF potential/gradient/collides, one launch (API 1)
//   k_goalset_queue   (omg_goalset_queue.h) the goal-set batch — kinematics, culling, SDF lookups, arc-length cost per goal — and,
//                     as extra workgroups of the same launch, the SDF layer of the current trajectories: the dominant kernel
//   k_fk_poses        Panda FK, one lane per robot configuration -> float64 link poses
//   k_sdf_chunks      the same SDF evaluation with per-point outputs over the collision points of those poses (or of its own
//                     kinematics), per (scene, chunk) workgroup, optional arc-length weighting + per-chunk reduction
//   k_forward_kinematics, k_point_cloud_sdf
//
// Hot-path data layout in HBM:
//   objects[]  176-byte omgx_object records, wave-uniform reads -> SGPRs via scalar loads
//   sdf pool   float32 grids, x-major, z fastest: a trilinear row (z0-1..z0+2) is one 16-byte load
//   pose ws    [scene][chunk][link][config-in-chunk][12] float64 link poses: the 16 lanes of a row (the P
//              points of one link at one waypoint) share one 96-byte pose; a wave = 4 consecutive
//              waypoints of one link -> spatially coherent gathers
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "omg_device.h"
#include "omg_host.h"

using namespace omg;

// =================================================================================================
// (1) k_sdf_loss — reference boundary op
// =================================================================================================
struct RawObjects {  // the eight tensors of omg_cuda.sdf_loss_forward
    const float* __restrict__ pose_init;
    const float* __restrict__ sdf_grids;
    const float* __restrict__ sdf_limits;
    const float* __restrict__ eps;
    const float* __restrict__ pad;
    const float* __restrict__ clr;
    const float* __restrict__ dis;
};

#define SDF_LOSS_TILE 32  // objects staged in LDS per pass

// The raw tensors carry no derived constants, so each workgroup first turns (a tile of) the objects into
// ObjParams records in LDS — including the double reciprocals of the grid extents — and then streams its
// points (grid-stride) against them.
__global__ __launch_bounds__(256) void k_sdf_loss(RawObjects R, const float* __restrict__ points, int64_t N, int O,
                                                   float* __restrict__ pot, float* __restrict__ grad,
                                                   float* __restrict__ col) {
    __shared__ ObjParams objs[SDF_LOSS_TILE];
    __shared__ int64_t goff[SDF_LOSS_TILE];
    __shared__ int live[SDF_LOSS_TILE];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int o0 = 0; o0 < O; o0 += SDF_LOSS_TILE) {
        const int cnt = min(SDF_LOSS_TILE, O - o0);
        __syncthreads();
        if (threadIdx.x < cnt) {
            const int o = o0 + threadIdx.x;
            ObjParams op;
#pragma unroll
            for (int k = 0; k < 12; ++k) op.T[k] = R.pose_init[16 * o + k];
            const float* L = R.sdf_limits + 10 * o;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                op.lo[k] = L[k]; op.hi[k] = L[3 + k]; op.dim[k] = (int)L[6 + k];
                op.rw[k] = 1.0 / (double)(op.hi[k] - op.lo[k]);
            }
            op.delta = L[9]; op.eps = R.eps[o]; op.pad = R.pad[o]; op.clr = R.clr[o];
            derive_far_box(op);
            objs[threadIdx.x] = op;
            goff[threadIdx.x] = (int64_t)o * op.dim[0] * op.dim[1] * op.dim[2];  // .cu:147
            live[threadIdx.x] = !(R.dis[o] > 0.0f);                               // .cu:115-116
        }
        __syncthreads();
        for (int64_t i = first; i < N; i += stride) {
            const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
            Accum acc{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (o0 > 0) { acc.pot = pot[i]; acc.col = col[i]; acc.gx = grad[3 * i]; acc.gy = grad[3 * i + 1]; acc.gz = grad[3 * i + 2]; }
            for (int k = 0; k < cnt; ++k)
                if (live[k]) sdf_pair<true>(objs[k], R.sdf_grids + goff[k], px, py, pz, acc);
            pot[i] = acc.pot;
            col[i] = acc.col;
            grad[3 * i] = acc.gx; grad[3 * i + 1] = acc.gy; grad[3 * i + 2] = acc.gz;
        }
    }
    if (O == 0)
        for (int64_t i = first; i < N; i += stride) { pot[i] = 0.0f; col[i] = 0.0f; grad[3 * i] = grad[3 * i + 1] = grad[3 * i + 2] = 0.0f; }
}

// =================================================================================================
// (2) k_fk_poses — one lane per configuration -> link poses (double [12]: R row-major, t)
// =================================================================================================
// Config sources:
//   mode 0: joints[S][C][9] given                                    (omgx_fk_sdf)
//   mode 2: the same + one extra configuration per scene = arc_start, written to ws_start (omgx_fk_sdf with arc_length > 0)
// Output layout ws[S][NCH][10][CH][12]: for a fixed link the 64 configurations of a wave are contiguous
// (6 KiB), so each link's poses are staged in LDS and stored cooperatively, fully coalesced.
struct FkArgs {
    const double* robot;
    int P;
    int mode;
    const double* joints;      // [S][C][9]
    const double* traj_start;  // mode 2: row s at traj_start + s * ts_stride
    int64_t ts_stride;
    int S, C;                  // C configs per scene
    int CH;                    // configs per chunk
    double* ws;                // [S][NCH][10][CH][12]
    double* ws_start;          // mode 2: [S][10][12]
};

__global__ __launch_bounds__(64) void k_fk_poses(FkArgs a) {
    __shared__ double stage[64 * 13];  // [64][12], row stride 13 doubles
    __shared__ double* rowptr[64];
    __shared__ int rowstride[64];
    const RobotView rv(a.robot, a.P);
    const int per_scene = a.C + (a.mode != 0 ? 1 : 0);
    const int64_t total = (int64_t)a.S * per_scene;
    const int lane = threadIdx.x;
    const int64_t id0 = (int64_t)blockIdx.x * 64;
    const int64_t id = id0 + lane;
    const int nlive = (int)((total - id0) < 64 ? (total - id0) : 64);
    const int64_t idc = id < total ? id : total - 1;
    const int s = (int)(idc / per_scene), c = (int)(idc % per_scene);
    const int CH = a.CH;
    const int NCH = (a.C + CH - 1) / CH;
    double q[9];
    {   // explicit joints; mode 2 adds one start configuration per scene
        if (a.mode == 2 && c == a.C) {
            const double* q0 = a.traj_start + a.ts_stride * (int64_t)s;
#pragma unroll
            for (int d = 0; d < 9; ++d) q[d] = q0[d];
            rowptr[lane] = a.ws_start + (int64_t)s * 120;
            rowstride[lane] = 12;
        } else {
            const double* src = a.joints + ((int64_t)s * a.C + c) * 9;
#pragma unroll
            for (int d = 0; d < 9; ++d) q[d] = src[d];
            rowptr[lane] = a.ws + ((((int64_t)s * NCH + c / CH) * 10) * CH + c % CH) * 12;
            rowstride[lane] = CH * 12;
        }
    }
    fk_chain(rv, q, [&](int l, const Pose& pose) {
        double* mine = stage + lane * 13;
#pragma unroll
        for (int k = 0; k < 9; ++k) mine[k] = pose.R[k];
        mine[9] = pose.t[0]; mine[10] = pose.t[1]; mine[11] = pose.t[2];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = k * 64 + lane;
            const int row = idx / 12, colm = idx - row * 12;
            if (row < nlive) rowptr[row][(int64_t)l * rowstride[row] + colm] = stage[row * 13 + colm];
        }
        __syncthreads();
    });
}

// =================================================================================================
// (3) k_sdf_chunks — SDF over the collision points of FK-produced link poses, one workgroup per (scene, chunk)
// =================================================================================================
struct ChunkArgs {
    const double* robot;
    const omgx_object* objects;
    const int32_t* scene_begin;
    const float* pool;
    const int32_t* active;     // [S] or null (k_goalset_queue): scenes with 0 are skipped, their outputs stay as they are
    const int32_t* goal_count; // [S] or null (k_goalset_queue): goals >= goal_count[s] of the padded goal array are skipped
    const double* ws;        // [S][NCH][10][CH][12] link poses
    const double* ws_start;  // [S][10][12] or null
    int S, C, CH, NCH, P;
    int LPW;                // links per workgroup (10, or fewer to expose more workgroups for small batches)
    int soften;             // uncheck_finger_collision == -1 (cost.py:350-353)
    int arc;                // weight potentials by ||(x_i - x_{i-1}) / dt|| (cost.py:235-275)
    float inv_dt;
    float* pot;             // [S][C][10][P] or null
    float* grad;            // [S][C][10][P][3] or null
    float* col;             // [S][C][10][P] or null
    float* chunk_cost;      // [S][NCH] or null: sum of (weighted) potentials of the chunk
    float* chunk_col;       // [S][NCH] or null: sum of collides of the chunk
    // fused FK (goal-set batch): the workgroup computes its own link poses into LDS instead of reading ws
    const double* traj_start;  // row s at traj_start + s * ts_stride
    int64_t ts_stride;
    const double* goals;       // [S][NCH][9]
    // SDF layer of the current trajectories inside the goal-set launch (k_goalset_queue only): one extra workgroup per scene
    const double* wp_traj;     // [S][wp_n][9] or null
    int wp_n, wp_soften;
    float* wp_pot;             // [S][wp_n][10][P]
    float* wp_grad;            // [S][wp_n][10][P][3]
    float* wp_col;             // [S][wp_n][10][P]
    int PS, MR;                // LDS pose stride (configurations per link) and mask rows per link: max over both kinds of workgroup
    // dispatch order of the goal workgroups (k_goalset_queue only).  schedule[k] = scene * NCH + goal of the k-th goal
    // workgroup in blockIdx order (< 0: nothing), normally the goals of the previous launch sorted by the time they took
    // (longest first, omgx_goalset_schedule) so that the launch does not end on a few long workgroups.  work[scene * NCH +
    // goal] receives the workgroup's duration in 10 ns ticks (0 for skipped goals).
    const int32_t* schedule;
    int sched_len;             // entries of `schedule` = goal workgroups of the launch (a multiple of 8)
    int tbl_n;                 // exact-path records staged in LDS (k_goalset_queue; chosen by the launcher)
    // work decomposition of k_goalset_queue (set by launch_goalset).  Batch: one workgroup per goal (NP = 1), 5 layer workgroups per
    // scene (2 links x all waypoints each), a scene's workgroups on one XCD.  Latency mode (a few scenes): a goal's tiles dealt over
    // NP workgroups (chunk = goal * NP + part; NCH = NG * NP chunks per scene), layer_lg x layer_nb layer workgroups per scene
    // (10 / layer_lg links x layer_cb waypoints), workgroups in plain order over all XCDs (spread).
    int NG, NP;
    int range_h;  // k_goalset_range: a goal's window cut into two RANGES of waypoints — part 0 the first range_h, part 1 the rest (omg_goalset_queue.h: RANGE)
    int layer_parts, layer_lg, layer_nb, layer_cb, spread;
    double* wp_pose_out;       // [S][wp_n][10][12] or null: the layer workgroups of link group 0 leave the waypoints' poses here
    uint32_t* work;
    // kinematics pre-pass (k_goalset_kin -> k_goalset_queue<..., PRE>; omg_goalset_kin.h): the goals' link poses
    // [S * NG][10][9][CH + 1] and row masks [S * NG][10][CH] in the caller's workspace, or null: every goal workgroup runs its own
    double* pre_poses;
    uint32_t* pre_masks;
};

// Thread layout: 256 threads = 16 rows x 16 lanes.  Lane = collision point p of a link (P <= 16), row =
// configuration ci of the chunk (strided by 16); a wave therefore holds the points of ONE link at 4
// consecutive waypoints — spatially coherent, so its SDF gathers share cache lines.  Each thread walks
// the 10 links in batches of LB: the wave-uniform object record is fetched (scalar loads -> SGPRs) once
// per (batch, object) and the LB far tests are straight-line code before any exact pair runs.
// The point itself is x = R_link p' + t_link (double, rounded to float32) from the FK kernel's pose
// (96 bytes shared by the 16 lanes of a row) and the centred collision point of the robot blob.
// No integer division by run-time sizes anywhere.
// FUSED (goal-set batch only): chunk = goal, its CH = n_remaining configurations are the linear interpolation
// start + (i+1)/(n+1) (goal - start) (util.py:261-290) and the workgroup runs their FK itself (one lane per
// configuration, plus the start configuration) into dynamic LDS [(CH+1)][10][12] — no pose workspace round trip
// through HBM and no separate FK launch.
template <bool WANT_GRAD, int LB, bool FUSED, int TPB = 256>
__global__ __launch_bounds__(TPB) void k_sdf_chunks(ChunkArgs a) {
    // dynamic LDS: [FUSED: (CH+1) x 10 poses of 12 doubles] then 10*CH row masks (candidate objects of each row)
    extern __shared__ __attribute__((aligned(16))) double lds_pose[];
    __shared__ float red[2][TPB / 64];
    uint32_t* rowmask = reinterpret_cast<uint32_t*>(lds_pose + (FUSED ? (size_t)(a.CH + 1) * 120 : 0));
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed; used for L2 affinity only).
    // All chunks of a scene go to the same XCD so the scene's SDF volumes stay in one 4 MiB L2.
    const int xcd = blockIdx.x & 7;
    const int nlg = 10 / a.LPW;                 // link groups per chunk
    const int jj = blockIdx.x >> 3;
    const int j = jj / nlg, lg = jj - j * nlg;  // wave-uniform divisions, once per workgroup
    const int sgrp = j / a.NCH;
    const int s = sgrp * 8 + xcd, chunk = j - sgrp * a.NCH;
    if (s >= a.S) return;
    const int l_begin = lg * a.LPW, l_end = l_begin + a.LPW;
    const int o_begin = as_const(a.scene_begin)[s], o_end = as_const(a.scene_begin)[s + 1];
    const int P = a.P, CH = a.CH;
    const int nvalid = min(CH, a.C - chunk * CH);  // configs in this (possibly last, partial) chunk
    const int p = threadIdx.x & 15, r = threadIdx.x >> 4;
    const RobotView rv(a.robot, P);
    if (FUSED) {
        for (int cfg = threadIdx.x; cfg < CH + 1; cfg += TPB) {
            const double* q0 = a.traj_start + a.ts_stride * (int64_t)s;
            const double* qg = a.goals + ((int64_t)s * a.NCH + chunk) * 9;
            const double t = (double)cfg * (1.0 / (double)(CH + 1));  // cfg 0 = start itself, cfg i+1 = linspace(0,1,n+2)[1:-1][i] = (i + 1) * fl(1 / (n + 1))
            double q[9];
#pragma unroll
            for (int d = 0; d < 9; ++d) q[d] = cfg == 0 ? q0[d] : q0[d] + t * (qg[d] - q0[d]);
            fk_chain(rv, q, [&](int l, const Pose& pose) {
                // layout [link][config][12] (config 0 = start) so that phase B indexes like the workspace
                double* dst = lds_pose + ((size_t)l * (CH + 1) + cfg) * 12;
#pragma unroll
                for (int k = 0; k < 9; ++k) dst[k] = pose.R[k];
                dst[9] = pose.t[0]; dst[10] = pose.t[1]; dst[11] = pose.t[2];
            });
        }
        __syncthreads();
    }
    // pose of (link l, config ci): FUSED -> LDS [l][ci+1]; else workspace [l][ci]
    const double* base = FUSED ? lds_pose + 12 : a.ws + ((int64_t)s * a.NCH + chunk) * (int64_t)(10 * CH) * 12;
    const int pstride = FUSED ? (CH + 1) : CH;  // configs per link in the pose array
    const double* sbase = FUSED ? lds_pose : (a.arc ? a.ws_start + (int64_t)s * 120 : nullptr);
    const int sstride = FUSED ? (CH + 1) * 12 : 12;  // distance between the start poses of consecutive links

    // ---- phase A: row-level culling.  All P points of a row lie in the ball (link origin, RAD[l]); an object
    // whose far box (grown by that radius + 1e-4 m for float rounding) misses the ball centre on any axis
    // cannot be in range for any of them.  One lane per row; objects >= 31 share the last mask bit.
    for (int row = l_begin * CH + threadIdx.x; row < l_end * CH; row += TPB) {
        const int l = row / CH, ci = row - l * CH;
        uint32_t m = 0;
        if (ci < nvalid) {
            const double* A = base + ((int64_t)l * pstride + ci) * 12;
            const auto bl = rv.ball(l);  // the ball around the link's own points (robot blob BALL)
            float cx, cy, cz;
            link_ball_center(A, A + 3, A + 9, bl, cx, cy, cz);
            const float rad = (float)bl[3] + 1.0e-4f;
            for (int o = o_begin; o < o_end; ++o) {
                ObjTablePtr ob = as_const(a.objects) + o;
                if (ob->disabled > 0) continue;
                const int oo = o - o_begin;
                const uint32_t bit = 1u << (oo < 31 ? oo : 31);
                const float ux = __builtin_fmaf(ob->pose_inv[2], cz, __builtin_fmaf(ob->pose_inv[1], cy, __builtin_fmaf(ob->pose_inv[0], cx, ob->pose_inv[3]))) - ob->lo[0];
                const float uy = __builtin_fmaf(ob->pose_inv[6], cz, __builtin_fmaf(ob->pose_inv[5], cy, __builtin_fmaf(ob->pose_inv[4], cx, ob->pose_inv[7]))) - ob->lo[1];
                const float uz = __builtin_fmaf(ob->pose_inv[10], cz, __builtin_fmaf(ob->pose_inv[9], cy, __builtin_fmaf(ob->pose_inv[8], cx, ob->pose_inv[11]))) - ob->lo[2];
                const float rbc[3] = {ob->rb_c[0], ob->rb_c[1], ob->rb_c[2]}, rbh[3] = {ob->rb_h[0], ob->rb_h[1], ob->rb_h[2]};
                const bool near = rbox_near(ux, uy, uz, rad, rbc, rbh, ob->rb_r);
                const bool cullable = ob->epsilon < 1.0f && ob->clearance <= 1.0f;  // else out-of-range values matter
                if (near || !cullable) m |= bit;
            }
        }
        rowmask[row] = m;
    }
    __syncthreads();

    // ---- phase B: points
    float tsum = 0.0f, tcol = 0.0f;
    if (p < P) {
        for (int ci = r; ci < nvalid; ci += TPB / 16) {
            const int64_t out_cfg = ((int64_t)s * a.C + chunk * CH + ci) * 10;
#pragma unroll 1
            for (int l0 = l_begin; l0 < l_end; l0 += LB) {
                float px[LB], py[LB], pz[LB];
                uint32_t msk[LB];
                uint32_t many = 0;
                Accum acc[LB];
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    msk[k] = rowmask[(l0 + k) * CH + ci];
                    many |= msk[k];
                    acc[k] = Accum{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                }
                const bool work = __any(many != 0);
                if (work) {
#pragma unroll
                    for (int k = 0; k < LB; ++k)
                        pose12_apply(base + ((int64_t)(l0 + k) * pstride + ci) * 12, rv.pts(l0 + k, p), px[k], py[k], pz[k]);
                }
                if (work) {
                    for (int o = o_begin; o < o_end; ++o) {  // wave-uniform trip count and addresses
                        const int oo = o - o_begin;
                        const uint32_t bit = 1u << (oo < 31 ? oo : 31);
                        if (!__any((many & bit) != 0)) continue;  // no row of this wave can reach the object
                        ObjTablePtr ob = as_const(a.objects) + o;
                        if (ob->disabled > 0) continue;  // .cu:115-116
                        const ObjParams op = load_object(ob);
                        const float* grid = a.pool + ob->grid_offset;
                        PairPrep pp[LB];
#pragma unroll
                        for (int k = 0; k < LB; ++k) {
                            pp[k] = pair_prepare(op, px[k], py[k], pz[k]);
                            pp[k].far = pp[k].far || !(msk[k] & bit);
                        }
#pragma unroll
                        for (int k = 0; k < LB; ++k)
                            if (!pp[k].far) pair_exact<WANT_GRAD>(op, grid, pp[k].tx, pp[k].ty, pp[k].tz, acc[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    const int l = l0 + k;
                    if (a.soften && l >= 8) {  // cost.py:350-353
                        acc[k].pot *= 0.1f; acc[k].gx *= 0.1f; acc[k].gy *= 0.1f; acc[k].gz *= 0.1f; acc[k].col = 0.0f;
                    }
                    if (a.arc && acc[k].pot != 0.0f) {  // ||(x_i - x_{i-1}) / dt|| in float32 (config.py:162-187, cost.py:260-275)
                        float qx, qy, qz;
                        pose12_apply(ci > 0 ? base + ((int64_t)l * pstride + ci - 1) * 12 : sbase + (int64_t)l * sstride, rv.pts(l, p), qx, qy, qz);
                        const float vx = (px[k] - qx) * a.inv_dt, vy = (py[k] - qy) * a.inv_dt, vz = (pz[k] - qz) * a.inv_dt;
                        acc[k].pot = acc[k].pot * sqrtf(vx * vx + vy * vy + vz * vz);
                    }
                    const int64_t kk = (out_cfg + l) * P + p;
                    if (a.pot) a.pot[kk] = acc[k].pot;
                    if (a.col) a.col[kk] = acc[k].col;
                    if (WANT_GRAD) { a.grad[3 * kk] = acc[k].gx; a.grad[3 * kk + 1] = acc[k].gy; a.grad[3 * kk + 2] = acc[k].gz; }
                    tsum += acc[k].pot;
                    tcol += acc[k].col;
                }
            }
        }
    }
    if (a.chunk_cost || a.chunk_col) {  // fixed-order block reduction: lanes -> waves -> thread 0
        const float ws_ = wave_sum(tsum), wc_ = wave_sum(tcol);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = ws_; red[1][w] = wc_; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t k = (int64_t)s * a.NCH + chunk;
            float c0 = red[0][0], c1 = red[1][0];
#pragma unroll
            for (int w2 = 1; w2 < TPB / 64; ++w2) { c0 += red[0][w2]; c1 += red[1][w2]; }  // fixed order
            if (a.chunk_cost) a.chunk_cost[k] = c0;
            if (a.chunk_col) a.chunk_col[k] = c1;
        }
    }
}
// =================================================================================================
// (3b) waypoint_layer_block — the SDF layer of a scene's current trajectory inside the goal-set launch
// =================================================================================================
// What omgx_fk_sdf computes for Optimizer.optimize (potentials, gradients, collisions of wp_n x 10 x P points) as
// extra workgroups per scene (ChunkArgs::layer_parts) of k_goalset_queue (omg_goalset_queue.h).  The optimiser step that follows on
// the same stream then depends on a single kernel: no side stream, no events.  Arithmetic is that of k_sdf_chunks<true>
// (same sdf_pair calls on the same float32 points); kinematics in two stages like the goal workgroups.
// PERSIST: the trajectory is read with agent-scope (sc1) loads — another workgroup of the same launch stored it (omg_persist.h)
template <bool LAT, bool PERSIST = false>  // LAT: the chain's constants from LDS (fkc, filled here) instead of through the scalar cache: see k_goalset_queue
__device__ __forceinline__ void waypoint_layer_block(const ChunkArgs& a, const int s, const int l_begin, const int l_end,
                                                     const int c_begin, const int c_end, double* lds_pose, uint32_t* rowmask,
                                                     const int o_begin, const int o_end, const RobotViewS& rv, double* fkc,
                                                     const bool warming, float* objc, double* btab) {
    // links [l_begin, l_end) at the waypoint configurations [c_begin, c_end): every output element is computed on its own, so any
    // split of the (link, configuration) grid over workgroups writes the same bits
    const int n = a.wp_n, P = a.P, PS = a.PS, MR = a.MR;
    const int nloc = c_end - c_begin;
    const double* tr = a.wp_traj + ((int64_t)s * n + c_begin) * 9;
    double* sc = reinterpret_cast<double*>(rowmask);  // [nloc][7][2], dead before the masks are written
    auto ldtr = [&](int k) -> double {
        if constexpr (PERSIST) return __hip_atomic_load(tr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return tr[k];
    };
    const double fkv = (LAT && threadIdx.x < 246) ? rv.g[threadIdx.x] : 0.0;
    if (warming) gq_warm_scalar_cache(a.objects, o_begin, o_end, a.robot + OMGX_ROBOT_POINTS + 30 * P + 316 + 30 * P);
    for (int t = threadIdx.x; t < nloc * 7; t += 256) {
        const int cfg = t / 7, i = t - cfg * 7;
        double sn, cs;
        fk_joint_sincos(ldtr(cfg * 9 + i), sn, cs);
        sc[2 * t] = sn; sc[2 * t + 1] = cs;
    }
    if (LAT && threadIdx.x < 246) fkc[threadIdx.x] = fkv;
    __syncthreads();
    // (the step that follows can take the waypoints' poses from here instead of running the same kinematics again: omgx_chomp_params)
    double* const gpose = (a.wp_pose_out && l_begin == 0) ? a.wp_pose_out + ((int64_t)s * n + c_begin) * 120 : nullptr;
    auto run_chain = [&](const auto& view) {
        for (int t = threadIdx.x; t < nloc * 3; t += 256) {
            const int cfg = t / 3, r = t - cfg * 3;
            fk_chain_row(view, r, sc + 14 * cfg, ldtr(cfg * 9 + 7), ldtr(cfg * 9 + 8), [&](int l, double r0, double r1, double r2, double tt) {
                double* dst = lds_pose + ((size_t)l * PS + cfg) * 9;
                if (r < 2) { dst[3 * r] = r0; dst[3 * r + 1] = r1; dst[3 * r + 2] = r2; }
                dst[6 + r] = tt;
                if (gpose) {
                    double* g = gpose + ((size_t)cfg * 10 + l) * 12;
                    g[3 * r] = r0; g[3 * r + 1] = r1; g[3 * r + 2] = r2; g[9 + r] = tt;
                }
            });
        }
    };
    if constexpr (LAT) {  // the joints' matrices tabulated first (fk_chain_row_B; see k_goalset_queue)
        const RobotView rvl(a.robot, P, fkc);
        for (int t = threadIdx.x; t < nloc * 7; t += 256) fk_joint_matrix(rvl, t - (t / 7) * 7, sc[2 * t], sc[2 * t + 1], btab + 9 * t);
        __syncthreads();
        for (int t = threadIdx.x; t < nloc * 3; t += 256) {
            const int cfg = t / 3, r = t - cfg * 3;
            fk_chain_row_B(rvl, r, btab + 63 * cfg, ldtr(cfg * 9 + 7), ldtr(cfg * 9 + 8), [&](int l, double r0, double r1, double r2, double tt) {
                double* dst = lds_pose + ((size_t)l * PS + cfg) * 9;
                if (r < 2) { dst[3 * r] = r0; dst[3 * r + 1] = r1; dst[3 * r + 2] = r2; }
                dst[6 + r] = tt;
                if (gpose) {
                    double* g = gpose + ((size_t)cfg * 10 + l) * 12;
                    g[3 * r] = r0; g[3 * r + 1] = r1; g[3 * r + 2] = r2; g[9 + r] = tt;
                }
            });
        }
    } else run_chain(rv);
    __syncthreads();
    for (int row = l_begin * nloc + threadIdx.x; row < l_end * nloc; row += 256) {  // row-level culling of this workgroup's links
        const int l = row / nloc, ci = row - l * nloc;
        const double* A = lds_pose + ((int64_t)l * PS + ci) * 9;
        const auto bl = rv.ball(l);  // the ball around the link's own points (robot blob BALL)
        float cx, cy, cz;
        link_ball_center(A, A + 3, A + 6, bl, cx, cy, cz);
        const float rad = (float)bl[3] + 1.0e-4f;
        uint32_t m = 0;
        for (int o = o_begin; o < o_end; ++o) {
            ObjTablePtr ob = as_const(a.objects) + o;
            if (ob->disabled > 0) continue;
            const int oo = o - o_begin;
            const uint32_t bit = 1u << (oo < 31 ? oo : 31);
            const float ux = __builtin_fmaf(ob->pose_inv[2], cz, __builtin_fmaf(ob->pose_inv[1], cy, __builtin_fmaf(ob->pose_inv[0], cx, ob->pose_inv[3]))) - ob->lo[0];
            const float uy = __builtin_fmaf(ob->pose_inv[6], cz, __builtin_fmaf(ob->pose_inv[5], cy, __builtin_fmaf(ob->pose_inv[4], cx, ob->pose_inv[7]))) - ob->lo[1];
            const float uz = __builtin_fmaf(ob->pose_inv[10], cz, __builtin_fmaf(ob->pose_inv[9], cy, __builtin_fmaf(ob->pose_inv[8], cx, ob->pose_inv[11]))) - ob->lo[2];
            const float rbc[3] = {ob->rb_c[0], ob->rb_c[1], ob->rb_c[2]}, rbh[3] = {ob->rb_h[0], ob->rb_h[1], ob->rb_h[2]};
            const bool near = rbox_near(ux, uy, uz, rad, rbc, rbh, ob->rb_r);
            const bool cullable = ob->epsilon < 1.0f && ob->clearance <= 1.0f;
            if (near || !cullable) m |= bit;
        }
        rowmask[l * MR + ci] = m;
    }
    __syncthreads();
    if (LAT && nloc <= 4 && l_end - l_begin == 1) {
        // Latency mode, one link x at most 4 waypoints per workgroup: the rows fit ONE wave, so the four waves take the scene's
        // OBJECTS side by side (wave w: objects w, w + 4 of each group of 8) instead of one wave walking them one gather round
        // trip (two, with a gradient) after the other — the finger links near the goal, in reach of everything, took 14 us where
        // the other workgroups took 6.  An object's contribution starts from zero and is parked in LDS (objc [8][64][5]); wave 0
        // adds the contributions in object order: the same sequence of float32 additions as the walk, bit for bit.
        const int l = l_begin, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int p = lane & 15, ci = lane >> 4;
        const bool valid = (p < P) && (ci < nloc);
        const int cic = valid ? ci : 0, pc = valid ? p : 0;
        const uint32_t msk = valid ? rowmask[l * MR + cic] : 0u;
        Accum acc{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (__any(msk != 0)) {  // the same in all four waves
            float px, py, pz;
            pose9_apply(lds_pose + ((int64_t)l * PS + cic) * 9, rv.pts(l, pc), px, py, pz);
            for (int og = o_begin; og < o_end; og += 8) {
                for (int o = og + wave; o < og + 8 && o < o_end; o += 4) {
                    const int oo = o - o_begin;
                    const uint32_t bit = 1u << (oo < 31 ? oo : 31);
                    Accum c{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    if (__any((msk & bit) != 0)) {
                        ObjTablePtr ob = as_const(a.objects) + o;
                        if (!(ob->disabled > 0)) {
                            const ObjParams op = load_object(ob);
                            const PairPrep pp = pair_prepare(op, px, py, pz);
                            if ((msk & bit) && !pp.far) pair_exact<true>(op, a.pool + ob->grid_offset, pp.tx, pp.ty, pp.tz, c);
                        }
                    }
                    float* dst = objc + ((o - og) * 64 + lane) * 5;
                    dst[0] = c.pot; dst[1] = c.gx; dst[2] = c.gy; dst[3] = c.gz; dst[4] = c.col;
                }
                __syncthreads();
                if (wave == 0) {
                    for (int o = og; o < og + 8 && o < o_end; ++o) {
                        const float* src = objc + ((o - og) * 64 + lane) * 5;
                        acc.pot += src[0]; acc.gx += src[1]; acc.gy += src[2]; acc.gz += src[3]; acc.col += src[4];
                    }
                }
                __syncthreads();
            }
        }
        if (wave == 0) {
            if (a.wp_soften && l >= 8) {  // cost.py:350-353
                acc.pot *= 0.1f; acc.gx *= 0.1f; acc.gy *= 0.1f; acc.gz *= 0.1f; acc.col = 0.0f;
            }
            if (valid) {
                const int64_t kk = (((int64_t)s * n + c_begin + ci) * 10 + l) * P + p;
                a.wp_pot[kk] = acc.pot;
                a.wp_col[kk] = acc.col;
                a.wp_grad[3 * kk] = acc.gx; a.wp_grad[3 * kk + 1] = acc.gy; a.wp_grad[3 * kk + 2] = acc.gz;
            }
        }
        return;
    }
    const int p = threadIdx.x & 15, r = threadIdx.x >> 4;
    for (int ci0 = 0; ci0 < nloc; ci0 += 16) {
        const int ci = ci0 + r;
        const bool valid = (p < P) && (ci < nloc);
        const int cic = valid ? ci : 0, pc = valid ? p : 0;
#pragma unroll 1
        for (int l = l_begin; l < l_end; ++l) {
            const uint32_t msk = valid ? rowmask[l * MR + cic] : 0u;
            Accum acc{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (__any(msk != 0)) {
                float px, py, pz;
                pose9_apply(lds_pose + ((int64_t)l * PS + cic) * 9, rv.pts(l, pc), px, py, pz);
                for (int o = o_begin; o < o_end; ++o) {
                    const int oo = o - o_begin;
                    const uint32_t bit = 1u << (oo < 31 ? oo : 31);
                    if (!__any((msk & bit) != 0)) continue;
                    ObjTablePtr ob = as_const(a.objects) + o;
                    if (ob->disabled > 0) continue;
                    const ObjParams op = load_object(ob);
                    const PairPrep pp = pair_prepare(op, px, py, pz);
                    if ((msk & bit) && !pp.far) pair_exact<true>(op, a.pool + ob->grid_offset, pp.tx, pp.ty, pp.tz, acc);
                }
            }
            if (a.wp_soften && l >= 8) {  // cost.py:350-353
                acc.pot *= 0.1f; acc.gx *= 0.1f; acc.gy *= 0.1f; acc.gz *= 0.1f; acc.col = 0.0f;
            }
            if (valid) {
                const int64_t kk = (((int64_t)s * n + c_begin + ci) * 10 + l) * P + p;
                a.wp_pot[kk] = acc.pot;
                a.wp_col[kk] = acc.col;
                a.wp_grad[3 * kk] = acc.gx; a.wp_grad[3 * kk + 1] = acc.gy; a.wp_grad[3 * kk + 2] = acc.gz;
            }
        }
    }
}

// Debug aid (make CXXFLAGS+=-DOMGX_GS_CLOCK=1; tools/gs_phase_clock.py): every workgroup of k_goalset_queue stamps the
// 100 MHz realtime clock at its phase boundaries — entry, after each stage of the kinematics, after the row culling, when
// its first / last wave leaves the main loop, exit — and its hardware id.  Never part of the shipped library.
#ifdef OMGX_GS_CLOCK
__device__ unsigned long long g_gs_wg[1 << 16][8];
// two launches that run at once (tools/experiments/gs_two_queue_clock.py) keep their stamps apart when one has an odd number of scenes
#define GS_WG_IDX ((blockIdx.x + (((unsigned)a.S & 1u) << 14)) & 0xffffu)
#define GS_WG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < (1u << 16)) { g_gs_wg[GS_WG_IDX][k] = wall_clock64(); if (k == 0) { g_gs_wg[GS_WG_IDX][5] = ~0ull; g_gs_wg[GS_WG_IDX][6] = 0ull; unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_gs_wg[GS_WG_IDX][7] = ((unsigned long long)xcc << 32) | hw; } } } while (0)
__device__ unsigned long long g_gs_wave[1 << 16][8];  // per wave: [w] when wave w finished its part of the chain / culling stage; [4 + w] when it entered it
#define GS_WAVE_STAMP(k) do { if ((threadIdx.x & 63) == 0 && (k) < 8 && blockIdx.x < (1u << 16)) g_gs_wave[blockIdx.x][k] = wall_clock64(); } while (0)  // (wide workgroups: waves 4.. have no slot)
extern "C" int omgx_debug_gs_wave(unsigned long long* h_out, int n_wg) {
    if (n_wg > (1 << 16)) n_wg = 1 << 16;
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gs_wave), sizeof(unsigned long long) * 8 * n_wg) == hipSuccess ? 0 : -2;
}
// shader-clock cycles and 100 MHz ticks a goal workgroup lived: their ratio is the clock the CUs really ran at under this load
__device__ unsigned long long g_gs_freq[1 << 16][2];
#define GS_FREQ_BEGIN() unsigned long long gs_f0 = 0, gs_f1 = 0; if (threadIdx.x == 0) { gs_f0 = __builtin_readcyclecounter(); gs_f1 = wall_clock64(); }
#define GS_FREQ_END() do { if (threadIdx.x == 0 && blockIdx.x < (1u << 16)) { g_gs_freq[blockIdx.x][0] = __builtin_readcyclecounter() - gs_f0; g_gs_freq[blockIdx.x][1] = wall_clock64() - gs_f1; } } while (0)
extern "C" int omgx_debug_gs_freq(unsigned long long* h_out, int n_wg) {
    if (n_wg > (1 << 16)) n_wg = 1 << 16;
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gs_freq), sizeof(unsigned long long) * 2 * n_wg) == hipSuccess ? 0 : -2;
}
extern "C" int omgx_debug_gs_wg(unsigned long long* h_out, int n_wg) {
    if (n_wg > (1 << 16)) n_wg = 1 << 16;
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gs_wg), sizeof(unsigned long long) * 8 * n_wg) == hipSuccess ? 0 : -2;
}
#else
#define GS_WG_STAMP(k)
#define GS_WAVE_STAMP(k)
#define GS_FREQ_BEGIN()
#define GS_FREQ_END()
#endif

// Measurement aid (EXTRA=-DOMGX_GS_ADD_{S,V,V64,B,N}=1; DESIGN_HISTORY.md appendix A): 16 more instructions of ONE class per far test of
// k_goalset_queue — scalar adds, float32 / float64 multiply-adds on four chains, eight taken branches, s_nops — to read off what an
// instruction of that class costs the launch where it runs.  Never part of the shipped library.
#if defined(OMGX_GS_ADD_S)
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { { uint32_t d0 = (uint32_t)(i_), d1 = (uint32_t)(j_); asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\t" "s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5" : "+s"(d0), "+s"(d1) : : "scc"); } } while (0)
#elif defined(OMGX_GS_ADD_V)
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { { float d0 = (a_), d1 = (b_), d2 = (c_), d3 = (d_); asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t" "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t" "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t" "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); } } while (0)
#elif defined(OMGX_GS_ADD_V64)
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { { double d0 = (a_), d1 = (b_), d2 = (c_), d3 = (d_); asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\t" "v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\t" "v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\t" "v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); } } while (0)
#elif defined(OMGX_GS_ADD_B)
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { asm volatile("s_branch 1f\n\ts_nop 0\n\t1:\n\ts_branch 2f\n\ts_nop 0\n\t2:\n\ts_branch 3f\n\ts_nop 0\n\t3:\n\ts_branch 4f\n\ts_nop 0\n\t4:\n\t" "s_branch 5f\n\ts_nop 0\n\t5:\n\ts_branch 6f\n\ts_nop 0\n\t6:\n\ts_branch 7f\n\ts_nop 0\n\t7:\n\ts_branch 8f\n\ts_nop 0\n\t8:"); } while (0)
#elif defined(OMGX_GS_ADD_N)
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0"); } while (0)
#else
#define GS_MARGINAL_COST_PROBE(a_, b_, c_, d_, i_, j_) do { } while (0)
#endif

// Debug aid (EXTRA=-DOMGX_GS_COUNT=1; tools/gs_block_counts.py): how often each block of k_goalset_queue's main loop runs,
// counted per wave.  Never part of the shipped library.
#ifdef OMGX_GS_COUNT
__device__ unsigned long long g_gs_count[16];
#define GS_COUNT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_gs_count[k], 1ull); } while (0)
#define GS_COUNT_N(k, n) do { const unsigned long long n_ = (unsigned long long)(n); if ((threadIdx.x & 63) == 0) atomicAdd(&g_gs_count[k], n_); } while (0)  /* n may hold a ballot: evaluated by all lanes */
extern "C" int omgx_debug_gs_counts(unsigned long long* h_out, int reset) {
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gs_count), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gs_count), z, sizeof(z)) != hipSuccess) return -2; }
    return 0;
}
#else
#define GS_COUNT(k)
#define GS_COUNT_N(k, n)
#endif


#include "omg_goalset_kin.h"
#include "omg_goalset_queue.h"

// =================================================================================================
// C ABI
// =================================================================================================
static thread_local char g_err[256] = "";

int omgx_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return OMGX_ERR_LAUNCH;
}
extern "C" const char* omgx_last_error(void) { return g_err; }

// ---- optional per-launch timing of the goal-set / layer kernel with HIP events ------------------
// A bench aid (bench.py, tools/): events are attached to the dispatch itself (hipExtLaunchKernelGGL), so they bracket
// exactly the kernel.  One recorder per process, guarded by a mutex; the events belong to the device that was current
// when timing was enabled and launches on another device are not recorded.
#define OMGX_TIMING_CAP 4096
static std::mutex g_timing_mu;
static bool g_timing = false;
static int g_timing_dev = -1;
static int g_timing_stride = 1, g_timing_seen = 0;  // every g_timing_stride-th eligible launch is bracketed
static int g_timing_n = 0;
static hipEvent_t g_ev[OMGX_TIMING_CAP][2];
static bool g_ev_made[OMGX_TIMING_CAP];
static int g_ev_kind[OMGX_TIMING_CAP];  // 0 = goal-set launch (with or without the trajectory layer), 1 = layer-only launch

extern "C" int omgx_timing_enable(int32_t on) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_timing = on != 0;
    g_timing_stride = on > 1 ? on : 1;
    g_timing_seen = 0;
    g_timing_n = 0;
    if (g_timing && hipGetDevice(&g_timing_dev) != hipSuccess) { g_timing = false; return OMGX_ERR_LAUNCH; }
    return OMGX_OK;
}

// Waits for the recorded launches and writes their durations in milliseconds; returns the count.
extern "C" int omgx_timing_collect(float* h_ms, int32_t* h_kind, int32_t cap) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    int n = g_timing_n < cap ? g_timing_n : cap;
    for (int i = 0; i < n; ++i) {
        if (h_kind) h_kind[i] = g_ev_kind[i];
        hipError_t e = hipEventSynchronize(g_ev[i][1]);
        if (e != hipSuccess) return omgx_set_error("hipEventSynchronize", e);
        e = hipEventElapsedTime(&h_ms[i], g_ev[i][0], g_ev[i][1]);
        if (e != hipSuccess) return omgx_set_error("hipEventElapsedTime", e);
    }
    g_timing_n = 0;
    return n;
}

// Reserves an event pair for a launch of `kind` (nullptr, nullptr when this launch is not recorded).
static void timing_events(int kind, hipEvent_t* ev0, hipEvent_t* ev1) {
    *ev0 = *ev1 = nullptr;
    if (!g_timing) return;  // unlocked fast path: the flag only changes between benchmark phases
    std::lock_guard<std::mutex> lock(g_timing_mu);
    int dev = -1;
    if (!g_timing || g_timing_n >= OMGX_TIMING_CAP || hipGetDevice(&dev) != hipSuccess || dev != g_timing_dev) return;
    if (g_timing_seen++ % g_timing_stride != 0) return;  // sampled: an event pair costs ~6 us of stream time per launch
    const int i = g_timing_n;
    if (!g_ev_made[i]) {
        if (hipEventCreate(&g_ev[i][0]) != hipSuccess || hipEventCreate(&g_ev[i][1]) != hipSuccess) return;
        g_ev_made[i] = true;
    }
    g_ev_kind[i] = kind;
    *ev0 = g_ev[i][0]; *ev1 = g_ev[i][1];
    ++g_timing_n;
}
extern "C" int omgx_abi_version(void) { return 11; }  // 11: omgx_plan_persistent (K iterations of all scenes in one launch); 10: kinematics pre-pass (k_goalset_kin) behind the `workspace` argument of omgx_goalset_cost / _cost_layer, new trailing `workspace` of _cost_layer_parts / _cost_layer_tiled; 9: omgx_goalset_schedule_ordered (longest first inside an XCD); 8: omgx_goalset_cost_layer_parts, omgx_goalset_schedule_parts (a goal's tiles over several workgroups of the batch kernel); 2: `active` masks; 3: ragged goal sets (goal_count, eta); 4: goal schedule + work; 5: 184-byte object records (influence region = rounded box); 6: omgx_goalset_cost_layer_tiled, omgx_learner_params.cost_parts; 7: pose tables (omgx_pose_table, pointer fields at the end of both parameter blocks, layer_poses)
// One call for "copy these bytes back and wait": hipMemcpyAsync (device -> pinned host) + hipStreamSynchronize on the caller's stream
// — the last step of a planner iteration through the drop-in classes (device_loop.DeviceLoop), where two framework calls cost the
// host more than the copy itself.
extern "C" int omgx_download_sync(void* h_dst, const void* src, int64_t nbytes, void* stream) {
    if (nbytes < 0 || (nbytes > 0 && (!h_dst || !src))) return OMGX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = nbytes > 0 ? hipMemcpyAsync(h_dst, src, (size_t)nbytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    if (e != hipSuccess) return omgx_set_error("hipMemcpyAsync", e);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return omgx_set_error("hipStreamSynchronize", e);
    return OMGX_OK;
}
extern "C" int32_t omgx_device_cu_count(void) {  // compute units of the current device (a plain attribute query: no property table)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
extern "C" int omgx_device_arch(char* h_buf, int32_t h_len) {
    if (!h_buf || h_len <= 0) return OMGX_ERR_INVALID;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return omgx_set_error("hipGetDevice", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return omgx_set_error("hipGetDeviceProperties", e);
    strncpy(h_buf, prop.gcnArchName, h_len - 1);
    h_buf[h_len - 1] = 0;
    return OMGX_OK;
}

extern "C" int omgx_sdf_loss_forward(const float* pose_init, const float* sdf_grids, const float* sdf_limits,
                                     const float* points, const float* epsilons, const float* padding_scales,
                                     const float* clearances, const float* disables, int64_t num_points,
                                     int32_t num_objects, float* potentials, float* potential_grads, float* collides,
                                     void* stream) {
    if (num_points < 0 || num_objects < 0) return OMGX_ERR_INVALID;
    if (num_points == 0) return OMGX_OK;
    if (!points || !potentials || !potential_grads || !collides) return OMGX_ERR_INVALID;
    if (num_objects > 0 && (!pose_init || !sdf_grids || !sdf_limits || !epsilons || !padding_scales || !clearances || !disables))
        return OMGX_ERR_INVALID;
    RawObjects R{pose_init, sdf_grids, sdf_limits, epsilons, padding_scales, clearances, disables};
    const int64_t blocks = (num_points + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(k_sdf_loss, dim3(grid), dim3(256), 0, (hipStream_t)stream, R, points, num_points, num_objects,
                       potentials, potential_grads, collides);
    OMGX_CHECK_LAUNCH("k_sdf_loss");
    return OMGX_OK;
}

// =================================================================================================
// (6) k_point_cloud_sdf — nearest-point distance grid (PointEnv.compute_sdf_from_points, omg/core.py:426-457)
// =================================================================================================
// One thread per grid node; the point cloud streams through LDS in tiles of 1024 points (24 KiB), every lane
// reading the same point at a time (LDS broadcast).  fp64 like the reference's cKDTree query.
#define PCS_TILE 1024
__global__ __launch_bounds__(256) void k_point_cloud_sdf(const double* __restrict__ pts, int N, double ox, double oy, double oz,
                                                          double res, int dx, int dy, int dz, float* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double tile[PCS_TILE * 3];
    const int64_t total = (int64_t)dx * dy * dz;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t idc = id < total ? id : total - 1;
    const int k = (int)(idc % dz), j = (int)((idc / dz) % dy), i = (int)(idc / ((int64_t)dz * dy));
    // np.arange(start, stop, step) fills start, start + step, then start + i * delta with delta = (start + step) - start
    // (numpy's DOUBLE_fill) — not start + i * step: the two differ in the last bit for most nodes
    auto node = [](double start, double step, int i) {
        const double second = start + step, delta = second - start;
        return i == 0 ? start : (i == 1 ? second : start + (double)i * delta);
    };
    const double x = node(ox, res, i), y = node(oy, res, j), z = node(oz, res, k);
    double best = 1.0e300;
    for (int t0 = 0; t0 < N; t0 += PCS_TILE) {
        const int cnt = min(PCS_TILE, N - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += blockDim.x) tile[e] = pts[(int64_t)t0 * 3 + e];
        __syncthreads();
        for (int q = 0; q < cnt; ++q) {
            const double a = tile[3 * q] - x, b = tile[3 * q + 1] - y, c = tile[3 * q + 2] - z;
            const double d2 = (a * a + b * b) + c * c;
            best = d2 < best ? d2 : best;
        }
    }
    if (id < total) out[id] = (float)sqrt(best);
}

extern "C" int omgx_point_cloud_sdf(const double* points, int32_t num_points, const double* h_origin, double resolution,
                                    const int32_t* h_dims, float* out, void* stream) {
    if (!points || !h_origin || !h_dims || !out || num_points < 1 || !(resolution > 0.0)) return OMGX_ERR_INVALID;
    if (h_dims[0] < 1 || h_dims[1] < 1 || h_dims[2] < 1) return OMGX_ERR_INVALID;
    const int64_t total = (int64_t)h_dims[0] * h_dims[1] * h_dims[2];
    if (total > (int64_t)1 << 31) return OMGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_point_cloud_sdf, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points,
                       num_points, h_origin[0], h_origin[1], h_origin[2], resolution, h_dims[0], h_dims[1], h_dims[2], out);
    OMGX_CHECK_LAUNCH("k_point_cloud_sdf");
    return OMGX_OK;
}

// ---- workspace sizing -------------------------------------------------------------------------
// 16 configurations per workgroup: one row per configuration, more workgroups for small batches
static inline int chunk_configs_fk_sdf(int C) { return C < 16 ? C : 16; }

extern "C" int64_t omgx_fk_sdf_workspace_bytes(int32_t num_scenes, int32_t configs_per_scene, int32_t n_points) {
    if (num_scenes <= 0 || configs_per_scene <= 0 || n_points <= 0) return 0;
    // link poses [S][NCH][10][CH][12] double; NCH*CH <= C + 63 for either chunking (16 configurations, or
    // arc_length <= 64 configurations per chunk), plus one start pose set [10][12] per scene
    const int64_t rows = (int64_t)configs_per_scene + 64;
    return (int64_t)num_scenes * (rows * 10 * 12 + 120) * (int64_t)sizeof(double);
}

extern "C" int64_t omgx_goalset_workspace_bytes(int32_t num_scenes, int32_t num_goals, int32_t n_remaining, int32_t n_points) {
    if (num_scenes <= 0 || num_goals <= 0 || n_remaining <= 0 || n_points <= 0) return 0;
    (void)n_points;
    // the kinematics pre-pass (omg_goalset_kin.h): per goal 10 x 9 x (n + 1) pose doubles + 10 x n mask words
    return gk_workspace_bytes((int64_t)num_scenes * num_goals, n_remaining);
}

#ifndef GS_LAYER_FOLLOW_MIN
#define GS_LAYER_FOLLOW_MIN 16  // waypoints per trajectory-layer piece at least, when the pieces follow the goal window (launch_goalset)
#endif
#ifndef GS_RANGE_MIN_WINDOW
#define GS_RANGE_MIN_WINDOW (1 << 20)  // windows beyond this many configurations: two parts of a goal are waypoint ranges — never, by default (launch_goalset)
#endif
static int g_range_min = -1;  // -1: not read yet (OMGX_GS_RANGE_MIN, omgx_debug_set_range)
// test / experiment hook (tests/test_gpu_parts.py, tools/experiments/ab_range.sh; not part of the ABI); negative: the built-in rule
extern "C" void omgx_debug_set_range(int min_window) { g_range_min = min_window >= 0 ? min_window : GS_RANGE_MIN_WINDOW; }
#ifndef GS_WIDE8_MAX_ITEMS
#define GS_WIDE8_MAX_ITEMS 0    // (set by measurement: DESIGN.md section 4.5)
#endif
#ifndef GS_WIDE6_MAX_ITEMS
#define GS_WIDE6_MAX_ITEMS 0
#endif
#ifndef GS_WIDE6_LONG_MAX_ITEMS
#define GS_WIDE6_LONG_MAX_ITEMS 0  // windows beyond 32 waypoints
#endif
// How a k_goalset_queue launch is cut into workgroups (ChunkArgs: NP, layer_*, spread).  The default is the batch layout.
struct GsTiling {
    int goal_parts = 1;  // workgroups per goal (1, 2, 4 or 8) at most: omgx_goalset_parts picks the count for a window
    int layer_lg = 5;    // link groups of the trajectory layer (1, 2, 5 or 10)
    int layer_cb = 0;    // waypoints per layer workgroup; 0: all
    int spread = 0;      // latency mode: workgroups in plain (scene, item) order over all XCDs instead of a scene per XCD
    double* wp_pose_out = nullptr;  // the waypoints' poses for the step that follows (ChunkArgs)
    void* kin_ws = nullptr;         // omgx_goalset_workspace_bytes of scratch: the goals' kinematics run as a launch of their own (omg_goalset_kin.h)
};

// Workgroups per goal for a window of n_remaining configurations: the largest power of two <= max_parts that still leaves every
// workgroup at least 4 tiles (one per wave) of the ceil(n / 4) x 5.
static inline int gs_parts(int n_remaining, int max_parts) {
    const int ntiles = ((n_remaining + 3) / 4) * 5;
    int np = 1;
    while (np * 2 <= max_parts && ntiles / (np * 2) >= 4) np *= 2;
    return np;
}

// Waves per goal workgroup of a batch launch with `items` whole goals and a window of n_remaining configurations: GQ_WAVES, or 6 / 8 for the
// WIDE instantiations (three / two workgroups per CU).  EXPERIMENT knobs (read once): OMGX_GS_WIDE=0 never wide; OMGX_GS_WIDE8_MAX /
// OMGX_GS_WIDE6_MAX = the largest launch (goal workgroups) that runs on eight / six waves.
static int g_wide_on = -1;  // -1: not read yet
static int64_t g_wide_max8 = GS_WIDE8_MAX_ITEMS, g_wide_max6 = GS_WIDE6_MAX_ITEMS, g_wide_long6 = GS_WIDE6_LONG_MAX_ITEMS;
// Waves per goal workgroup of a batch launch of `items` whole goals with a window of n_remaining configurations (PS / MR / P: the launch's
// LDS layout).  The rule (round 6, measured: DESIGN.md section 4.5):
//   * a window whose four-wave layout needs more than 53 248 B of LDS — plans of 57 .. 64 waypoints (15-16 points per link; the launch's poses are the trajectory layer's: all n) — admits TWO workgroups per CU whatever their
//     waves: eight waves each (16 per CU instead of 8; 50 / 100 scenes x 64 goals x 64 waypoints: 0.287 -> 0.240 / 0.500 -> 0.399 ms per step);
//   * everything else on four: up to 56 waypoints the wide workgroups cost a workgroup per CU (stamps: 2.0 six-wave workgroups resident per CU
//     against 2.85 four-wave ones at 50 waypoints: 52 / 56 waypoints +6 .. +10 %), and at 30 waypoints they pay only in launches of a few hundred goals
//     under a layout rule of their own (2 - 8 scenes x 64 goals -4 .. -5 %, 13 x 128 +13 %): thresholds 0, kept as EXPERIMENT knobs —
//     OMGX_GS_WIDE=0 never wide; OMGX_GS_WIDE8_MAX / OMGX_GS_WIDE6_MAX / OMGX_GS_WIDE6_LONG_MAX = the largest launch (goal workgroups) on
//     eight / six waves (windows up to 32 waypoints) / six waves (longer windows), read once; omgx_debug_set_wide.
static inline int gs_wide_waves(int64_t items, int n_remaining, int PS, int MR, int P) {
    if (g_wide_on < 0) {
        const char* e = getenv("OMGX_GS_WIDE");
        const char* e8 = getenv("OMGX_GS_WIDE8_MAX");
        const char* e6 = getenv("OMGX_GS_WIDE6_MAX");
        const char* el = getenv("OMGX_GS_WIDE6_LONG_MAX");
        if (el) g_wide_long6 = (int64_t)atoll(el);
        if (e8) g_wide_max8 = (int64_t)atoll(e8);
        if (e6) g_wide_max6 = (int64_t)atoll(e6);
        g_wide_on = e ? (atoi(e) != 0) : 1;
    }
    if (!g_wide_on || GQ_WAVES != 4) return GQ_WAVES;
    if (GqLayout(PS, MR, P, 4, false, GQ_WAVES).total > 53248 && GqLayout(PS, MR, P, 4, false, 8).total <= 64 * 1024) return 8;
    if (n_remaining > 32) {  // (EXPERIMENT: OMGX_GS_WIDE_LONG_W = 6 | 8 waves for the long windows the knob admits)
        static const int lw = [] { const char* e = getenv("OMGX_GS_WIDE_LONG_W"); return (e && atoi(e) == 8) ? 8 : 6; }();
        return items <= g_wide_long6 ? lw : GQ_WAVES;
    }
    if (items <= g_wide_max8) return 8;
    if (items <= g_wide_max6) return 6;
    return GQ_WAVES;
}
// which instantiation the calling thread's process launched last (test hook: the rules above are the library's business, a test can still hold
// them): waves per goal workgroup | 0x100 waypoint ranges | 0x200 split goals | 0x400 latency mode | 0x800 behind the kinematics pre-pass
static int g_last_goalset_variant = 0;
extern "C" int omgx_debug_last_goalset_variant(void) { return g_last_goalset_variant; }
// test / experiment hook (tests/test_gpu_round6.py, tools/experiments/ab_wide.py; not part of the ABI): the rule's two thresholds; negative = keep
extern "C" void omgx_debug_set_wide(int on, long long max8, long long max6, long long long6) {
    (void)gs_wide_waves(0, 1, 2, 1, 1);  // the environment first, so that it does not overwrite what is set here
    g_wide_on = on != 0;
    if (max8 >= 0) g_wide_max8 = max8; else if (max8 < -1) g_wide_max8 = GS_WIDE8_MAX_ITEMS;  // (-1: keep, below: the built-in rule)
    if (max6 >= 0) g_wide_max6 = max6; else if (max6 < -1) g_wide_max6 = GS_WIDE6_MAX_ITEMS;
    if (long6 >= 0) g_wide_long6 = long6; else if (long6 < -1) g_wide_long6 = GS_WIDE6_LONG_MAX_ITEMS;
}

// dynamic LDS and grid of a k_goalset_queue launch (goal workgroups and / or trajectory-layer workgroups)
static int launch_goalset(ChunkArgs& ca, int timing_kind, hipStream_t st, const GsTiling& tl = GsTiling()) {
    const int scene_groups = (ca.S + 7) / 8;
    const bool layer = ca.wp_traj != nullptr;
    if (tl.goal_parts < 1 || tl.goal_parts > 8 || tl.layer_lg < 1 || tl.layer_lg > 10 || 10 % tl.layer_lg != 0 || tl.layer_cb < 0) return OMGX_ERR_INVALID;
    ca.spread = tl.spread != 0;
    ca.NG = ca.NCH;
    ca.NP = ca.NG > 0 ? gs_parts(ca.CH, tl.goal_parts) : 1;
    ca.NCH = ca.NG * ca.NP;
    const bool split = !ca.spread && ca.NP > 1;  // the batch kernel with a goal's tiles dealt over NP workgroups (omgx_goalset_cost_layer_parts)
    if (ca.spread && (ca.schedule || ca.work)) return OMGX_ERR_UNSUPPORTED;  // a dispatch schedule belongs to the batch layout: a scene per XCD
    ca.wp_pose_out = layer ? tl.wp_pose_out : nullptr;
    ca.layer_lg = tl.layer_lg;
    // RANGES (round 6; omg_goalset_queue.h: RANGE, k_goalset_range): two parts of a LONG window are waypoint ranges, not dealt tiles — each
    // holds only its own configurations' poses (50 waypoints: 29 KB of LDS instead of 50, five workgroups per CU instead of three) and
    // no kinematics run twice.  The second range — the goal's end, where the objects are — is the shorter one and a multiple of four
    // waypoints (its blocks then hold four rows each).  The trajectory layer's pieces are cut in two as well, or their poses would
    // set the launch's LDS.  MEASURED AND LEFT OFF (DESIGN_HISTORY.md appendix A, round 6): a part pays a workgroup's whole latency L0 for half a
    // goal's work — 100 x 64 x 50 waypoints 0.306 against 0.317 ms per step with whole goals, config 5's shape 0.185 against 0.164 with
    // dealt tiles (the goal's end is the heavy range).  OMGX_GS_RANGE_MIN=<n> / omgx_debug_set_range(n): ranges for windows beyond n.
    if (g_range_min < 0) { const char* e = getenv("OMGX_GS_RANGE_MIN"); g_range_min = e ? atoi(e) : GS_RANGE_MIN_WINDOW; }
    const bool range = split && ca.NP == 2 && ca.CH > g_range_min && tl.kin_ws == nullptr && ca.traj_start != nullptr;
    ca.range_h = range ? ca.CH - 4 * (ca.CH / 8) : 0;
    int layer_cb_req = tl.layer_cb;
    if (range && layer && layer_cb_req == 0) layer_cb_req = (ca.wp_n + 1) / 2;
    {   // The trajectory layer's pieces cut to the goal WINDOW's size (round 6): a piece holds the poses of all its waypoints, so five pieces of
        // all n waypoints set the launch's LDS by the plan's LENGTH — in a 50-waypoint plan every goal-set launch asked for 47-50 KB, three
        // workgroups per CU, however short its window had become.  Cut to max(window + 1, 16) waypoints the launch's LDS follows the window and
        // the later launches hold four or five: plan of 100 x 64 x 50 waypoints 14.35 -> 12.64 ms, config 5's shape 10.91 -> 10.23, 50 x 64 x 64
        // 12.18 -> 11.02, 100 x 64 x 41 11.58 -> 11.18 (tools/experiments/ab_layer_follow.sh).  Any split of the layer gives the same bits.
        // Plans up to 32 waypoints hold five workgroups per CU anyway.  OMGX_GS_LAYER_FOLLOW=0 (experiments): the five whole pieces.
        static const int follow = [] { const char* e = getenv("OMGX_GS_LAYER_FOLLOW"); return e ? atoi(e) : GS_LAYER_FOLLOW_MIN; }();
        if (follow > 0 && layer && layer_cb_req == 0 && !ca.spread && ca.NG > 0 && !range && ca.wp_n > 32) {
            const int want = ca.CH + 1 > follow ? ca.CH + 1 : follow;
            if (want < ca.wp_n) layer_cb_req = want;
        }
        // SMALL launches (up to 1 024 goal workgroups: everything is resident at once and the launch lasts as long as its longest workgroup —
        // late in a plan that is a layer piece of 2 links x all waypoints, 26 us, beside goals of 13): ten link groups x blocks of max(window
        // + 1, 16) waypoints.  Plan of 13 x 128 4.53 -> 4.18 ms, 16 x 64 4.25 -> 4.04, 25 x 64 4.65 -> 4.19, 8 x 64 4.06 -> 3.62; the pinned step of
        // 16 x 64 0.0698 -> 0.0679, 13 x 128 unchanged (tools/experiments/ab_layer_small.sh).  OMGX_GS_LAYER_SMALL=0 (experiments): five whole pieces.
        static const int small = [] { const char* e = getenv("OMGX_GS_LAYER_SMALL"); return e ? atoi(e) : GS_LAYER_FOLLOW_MIN; }();
        if (small > 0 && layer && layer_cb_req == 0 && !ca.spread && ca.NG > 0 && !range && tl.layer_lg == 5 && (int64_t)ca.S * ca.NCH <= 1024) {
            ca.layer_lg = 10;
            const int want = ca.CH + 1 > small ? ca.CH + 1 : small;
            if (want < ca.wp_n) layer_cb_req = want;
        }
    }
    ca.layer_cb = layer ? ((layer_cb_req > 0 && layer_cb_req < ca.wp_n) ? layer_cb_req : ca.wp_n) : 1;
    ca.layer_nb = layer ? (ca.wp_n + ca.layer_cb - 1) / ca.layer_cb : 1;
    ca.layer_parts = ca.layer_lg * ca.layer_nb;
    ca.PS = ca.CH + 1; ca.MR = ca.CH; ca.LPW = 10;
    if (range) { ca.PS = ca.range_h + 1; ca.MR = ca.range_h; }  // (range_h >= CH - range_h)
    if (layer) { if (ca.layer_cb > ca.PS) ca.PS = ca.layer_cb; if (ca.layer_cb > ca.MR) ca.MR = ca.layer_cb; }
    const int64_t per_scene = (int64_t)ca.NCH + (layer ? ca.layer_parts : 0);
    const int64_t grid = ca.spread ? (int64_t)ca.S * per_scene
                                   : (ca.schedule ? (int64_t)ca.sched_len : (int64_t)scene_groups * ca.NCH * 8) +
                                         (layer ? (int64_t)scene_groups * ca.layer_parts * 8 : 0);
    if (grid > 0x7fffffff) return OMGX_ERR_UNSUPPORTED;
    if (grid == 0) return OMGX_OK;
    const bool pre = tl.kin_ws != nullptr && ca.NG > 0 && ca.traj_start != nullptr;
    // WIDE workgroups (round 6; omg_goalset_queue.h, template parameter W): a launch of few enough goal workgroups to be resident all at
    // once lasts as long as its heaviest goal — more waves draw that goal's tiles.  Whole goals with their own kinematics only.
    const int wide = (!ca.spread && !split && !pre && ca.NG > 0) ? gs_wide_waves((int64_t)ca.S * ca.NG, ca.CH, ca.PS, ca.MR, ca.P) : GQ_WAVES;
    ca.tbl_n = gq_choose_tbl_n(ca.PS, ca.MR, ca.P, ca.spread, wide);
    const size_t lds = (size_t)GqLayout(ca.PS, ca.MR, ca.P, ca.tbl_n, ca.spread, wide).total;
    if (lds > 64 * 1024) {
        // the batch layout stays below (61 KB at 64 waypoints x 16 points); the latency-mode kernel adds the chain constants and the
        // layer's per-object contributions (73 KB at 64 waypoints) and opts in
        if (!ca.spread) return OMGX_ERR_UNSUPPORTED;
        int rc = allow_big_lds<10>(k_goalset_queue<2, false, true>, "hipFuncSetAttribute(k_goalset_queue)");
        if (rc != OMGX_OK) return rc;
        rc = allow_big_lds<11>(k_goalset_queue<2, false, true, false, true>, "hipFuncSetAttribute(k_goalset_queue)");
        if (rc != OMGX_OK) return rc;
    }
    ca.pre_poses = nullptr; ca.pre_masks = nullptr;
    if (pre) {  // the goals' kinematics and row masks as a launch of their own (omg_goalset_kin.h), one lane per (goal, configuration)
        const int64_t goals = (int64_t)ca.S * ca.NG;
        ca.pre_poses = reinterpret_cast<double*>(tl.kin_ws);
        ca.pre_masks = reinterpret_cast<uint32_t*>(ca.pre_poses + goals * gk_pose_doubles(ca.CH));
        const int ncfg = ca.CH + 1, gpw = ncfg <= 64 ? 64 / ncfg : 1;
        const int64_t waves = (int64_t)ca.S * ((ca.NG + gpw - 1) / gpw);
        hipLaunchKernelGGL(k_goalset_kin, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, ca);
        OMGX_CHECK_LAUNCH("k_goalset_kin");
    }
    hipEvent_t ev0, ev1;
    timing_events(timing_kind, &ev0, &ev1);
    const unsigned g = (unsigned)grid;
    uint32_t l32 = (uint32_t)lds;
#ifdef OMGX_GS_CLOCK  // measurement build only (tools/gs_phase_clock.py): fewer workgroups per CU by asking for more LDS than the layout needs
    if (const char* e = getenv("OMGX_GS_LDS_MIN")) { const uint32_t v = (uint32_t)atoi(e); if (v > l32 && v <= 64 * 1024) l32 = v; }
#endif
#define GQ_GO(STAMP, LAT, SPLIT, PRE) do { if (ev0) hipExtLaunchKernelGGL((k_goalset_queue<2, STAMP, LAT, SPLIT, PRE>), dim3(g), dim3((LAT) ? 256 : GQ_NT), l32, st, ev0, ev1, 0, ca); \
                                            else hipLaunchKernelGGL((k_goalset_queue<2, STAMP, LAT, SPLIT, PRE>), dim3(g), dim3((LAT) ? 256 : GQ_NT), l32, st, ca); } while (0)
#define GQ_GO_W(STAMP, WW) do { if (ev0) hipExtLaunchKernelGGL((k_goalset_queue<2, STAMP, false, false, false, WW>), dim3(g), dim3(64 * WW), l32, st, ev0, ev1, 0, ca); \
                                 else hipLaunchKernelGGL((k_goalset_queue<2, STAMP, false, false, false, WW>), dim3(g), dim3(64 * WW), l32, st, ca); } while (0)
    if (range) {
        if (ca.work) { if (ev0) hipExtLaunchKernelGGL((k_goalset_range<2, true>), dim3(g), dim3(GQ_NT), l32, st, ev0, ev1, 0, ca); else hipLaunchKernelGGL((k_goalset_range<2, true>), dim3(g), dim3(GQ_NT), l32, st, ca); }
        else { if (ev0) hipExtLaunchKernelGGL((k_goalset_range<2, false>), dim3(g), dim3(GQ_NT), l32, st, ev0, ev1, 0, ca); else hipLaunchKernelGGL((k_goalset_range<2, false>), dim3(g), dim3(GQ_NT), l32, st, ca); }
    } else
    if (wide == 6) { if (ca.work) GQ_GO_W(true, 6); else GQ_GO_W(false, 6); }
    else if (wide == 8) { if (ca.work) GQ_GO_W(true, 8); else GQ_GO_W(false, 8); }
    else
    if (ca.spread) { if (pre) GQ_GO(false, true, false, true); else GQ_GO(false, true, false, false); }
    else if (split && ca.work) { if (pre) GQ_GO(true, false, true, true); else GQ_GO(true, false, true, false); }
    else if (split) { if (pre) GQ_GO(false, false, true, true); else GQ_GO(false, false, true, false); }
    else if (ca.work) { if (pre) GQ_GO(true, false, false, true); else GQ_GO(true, false, false, false); }
    else { if (pre) GQ_GO(false, false, false, true); else GQ_GO(false, false, false, false); }
#undef GQ_GO
#undef GQ_GO_W
    g_last_goalset_variant = (range ? 0x100 : 0) | (split ? 0x200 : 0) | (ca.spread ? 0x400 : 0) | (pre ? 0x800 : 0) | (range ? GQ_WAVES : wide);
    OMGX_CHECK_LAUNCH("k_goalset_queue");
    return OMGX_OK;
}

// k_sdf_chunks over poses in the workspace (omgx_fk_sdf beyond a trajectory-sized layer) or with its own kinematics
// (goal-set batch with per-point potentials)
static int launch_chunks(ChunkArgs& ca, hipStream_t st) {
    const int scene_groups = (ca.S + 7) / 8;
    // Small batches would leave most CUs idle with one workgroup per (scene, chunk): split the 10 links over workgroups
    // until ~4 workgroups per CU exist.  Per-chunk reductions need all links in one workgroup.
    int lpw = 10;
    if (!ca.chunk_cost && !ca.chunk_col) {
        const int64_t wgs = (int64_t)scene_groups * ca.NCH * 8;
        if (wgs * 5 <= 1024) lpw = 2; else if (wgs * 2 <= 1024) lpw = 5;
    }
    ca.LPW = lpw;
    const int64_t grid = (int64_t)scene_groups * ca.NCH * 8 * (10 / lpw);
    if (grid > 0x7fffffff) return OMGX_ERR_UNSUPPORTED;
    const size_t mask_bytes = (size_t)10 * ca.CH * sizeof(uint32_t);
    const int lb = lpw % 2 == 0 ? 2 : 1;  // links per batch (2 measured best); it must divide the links of a workgroup
    if (ca.traj_start) {  // goal-set batch with potentials: the workgroup's own kinematics, (CH + 1) x 10 poses in LDS
        const size_t lds = (size_t)(ca.CH + 1) * 120 * sizeof(double) + mask_bytes;
        hipLaunchKernelGGL((k_sdf_chunks<false, 2, true>), dim3((unsigned)grid), dim3(256), lds, st, ca);
    } else if (ca.grad) {
        if (lb == 2) hipLaunchKernelGGL((k_sdf_chunks<true, 2, false>), dim3((unsigned)grid), dim3(256), mask_bytes, st, ca);
        else hipLaunchKernelGGL((k_sdf_chunks<true, 1, false>), dim3((unsigned)grid), dim3(256), mask_bytes, st, ca);
    } else {
        if (lb == 2) hipLaunchKernelGGL((k_sdf_chunks<false, 2, false>), dim3((unsigned)grid), dim3(256), mask_bytes, st, ca);
        else hipLaunchKernelGGL((k_sdf_chunks<false, 1, false>), dim3((unsigned)grid), dim3(256), mask_bytes, st, ca);
    }
    OMGX_CHECK_LAUNCH("k_sdf_chunks");
    return OMGX_OK;
}

extern "C" int omgx_fk_sdf(const double* robot, int32_t n_points, const omgx_object* objects, const int32_t* scene_begin,
                           const float* sdf_pool, const double* joints, int32_t num_scenes, int32_t configs_per_scene,
                           int32_t soften_fingers, int32_t arc_length, const double* arc_start, double time_interval,
                           float* potentials, float* grads, float* collides, void* workspace, void* stream) {
    if (num_scenes < 0 || configs_per_scene < 0) return OMGX_ERR_INVALID;
    if (num_scenes == 0 || configs_per_scene == 0) return OMGX_OK;
    if (!robot || !objects || !scene_begin || !joints || !workspace) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS) return OMGX_ERR_UNSUPPORTED;
    const bool arc = arc_length > 0;
    if (arc) {
        if (!arc_start || !(time_interval > 0.0)) return OMGX_ERR_INVALID;
        if (arc_length > OMGX_MAX_WAYPOINTS || configs_per_scene % arc_length != 0) return OMGX_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!arc && potentials && grads && collides && configs_per_scene <= OMGX_MAX_WAYPOINTS) {
        // A trajectory-sized layer (the optimiser's input): the goal-set kernel's layer workgroups alone — kinematics in
        // LDS, 5 workgroups per scene, one launch instead of k_fk_poses + k_sdf_chunks.  Same arithmetic.
        ChunkArgs ca{};
        ca.robot = robot; ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool;
        ca.S = num_scenes; ca.P = n_points; ca.NCH = 0; ca.CH = 0; ca.C = 0;
        ca.wp_traj = joints; ca.wp_n = configs_per_scene; ca.wp_soften = soften_fingers != 0;
        ca.wp_pot = potentials; ca.wp_grad = grads; ca.wp_col = collides;
        return launch_goalset(ca, 1, st);
    }
    const int CH = arc ? arc_length : chunk_configs_fk_sdf(configs_per_scene);
    const int NCH = (configs_per_scene + CH - 1) / CH;
    double* ws = (double*)workspace;
    double* ws_start = ws + (int64_t)num_scenes * NCH * 10 * CH * 12;
    FkArgs fa{};
    fa.robot = robot; fa.P = n_points; fa.mode = arc ? 2 : 0; fa.joints = joints; fa.traj_start = arc_start; fa.ts_stride = 9;
    fa.S = num_scenes; fa.C = configs_per_scene; fa.CH = CH; fa.ws = ws; fa.ws_start = ws_start;
    const int64_t total = (int64_t)num_scenes * (configs_per_scene + (arc ? 1 : 0));
    hipLaunchKernelGGL(k_fk_poses, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, fa);
    OMGX_CHECK_LAUNCH("k_fk_poses");
    ChunkArgs ca{};
    ca.robot = robot; ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool; ca.ws = ws; ca.ws_start = ws_start;
    ca.S = num_scenes; ca.C = configs_per_scene; ca.CH = CH; ca.NCH = NCH; ca.P = n_points; ca.soften = soften_fingers != 0;
    ca.arc = arc ? 1 : 0; ca.inv_dt = arc ? (float)(1.0 / time_interval) : 0.0f;
    ca.pot = potentials; ca.grad = grads; ca.col = collides;
    return launch_chunks(ca, st);
}

// (2b) forward kinematics with joint info, one lane per configuration
__global__ __launch_bounds__(64) void k_forward_kinematics(const double* __restrict__ robot, int P,
                                                            const double* __restrict__ joints, int64_t B,
                                                            double* __restrict__ poses, double* __restrict__ origins,
                                                            double* __restrict__ axes) {
#pragma clang fp contract(fast)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const RobotView rv(robot, P);
    double q[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) q[d] = joints[9 * b + d];
    fk_chain(rv, q, [&](int l, const Pose& pose) {
        const double* co = robot + OMGX_ROBOT_CENTER_OFFSET + 16 * l;  // output_pose @ center_offset (robot_pykdl.py:203-204)
        double* out = poses + (b * 10 + l) * 16;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c)
                out[4 * r + c] = pose.R[3 * r] * co[c] + pose.R[3 * r + 1] * co[4 + c] + pose.R[3 * r + 2] * co[8 + c] +
                                 (c == 3 ? pose.t[r] : 0.0);
        }
        out[12] = 0.0; out[13] = 0.0; out[14] = 0.0; out[15] = 1.0;
        if (origins && axes) {
            const double* ax = rv.ax(l);
            const double* og = rv.og(l);
            for (int r = 0; r < 3; ++r) {
                const double w = pose.R[3 * r] * ax[0] + pose.R[3 * r + 1] * ax[1] + pose.R[3 * r + 2] * ax[2];
                const double t = pose.R[3 * r] * og[0] + pose.R[3 * r + 1] * og[1] + pose.R[3 * r + 2] * og[2] + pose.t[r];
                axes[(b * 10 + l) * 3 + r] = w;
                origins[(b * 10 + l) * 3 + r] = w + t;  // `_joint_origin` is loaded from `_joint_axis` (robot_pykdl.py:104)
            }
        }
    });
}

extern "C" int omgx_forward_kinematics(const double* robot, int32_t n_points, const double* joints, int64_t num_configs,
                                       double* link_poses, double* joint_origins, double* joint_axes, void* stream) {
    if (num_configs < 0) return OMGX_ERR_INVALID;
    if (num_configs == 0) return OMGX_OK;
    if (!robot || !joints || !link_poses) return OMGX_ERR_INVALID;
    if ((joint_origins == nullptr) != (joint_axes == nullptr)) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS) return OMGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_forward_kinematics, dim3((unsigned)((num_configs + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       robot, n_points, joints, num_configs, link_poses, joint_origins, joint_axes);
    OMGX_CHECK_LAUNCH("k_forward_kinematics");
    return OMGX_OK;
}

// (2c) link poses of N configurations as the step keeps them ([10][12] doubles each): one lane per (configuration, pose row), the
// same fk_joint_sincos / fk_chain_row as inside k_chomp_optimize and the learner's workgroup
__global__ __launch_bounds__(192) void k_pose_table(const double* __restrict__ robot, int P, const double* __restrict__ configs,
                                                    int64_t N, double* __restrict__ poses) {
    const int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x / 3;  // 64 configurations per workgroup
    const int r = threadIdx.x % 3;
    if (e >= N) return;
    const RobotView rv(robot, P);
    const double* q = configs + 9 * e;
    double sc[14];
#pragma unroll
    for (int i = 0; i < 7; ++i) fk_joint_sincos(q[i], sc[2 * i], sc[2 * i + 1]);
    double* out = poses + e * 120;
    fk_chain_row(rv, r, sc, q[7], q[8], [&](int l, double r0, double r1, double r2, double tr) {
        double* dst = out + 12 * l + 3 * r;
        dst[0] = r0; dst[1] = r1; dst[2] = r2;
        dst[9 - 2 * r] = tr;  // element 9 + r of the pose
    });
}

extern "C" int omgx_pose_table(const double* robot, int32_t n_points, const double* configs, int64_t num_configs, double* poses,
                               void* stream) {
    if (num_configs < 0) return OMGX_ERR_INVALID;
    if (num_configs == 0) return OMGX_OK;
    if (!robot || !configs || !poses) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS) return OMGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_pose_table, dim3((unsigned)((num_configs + 63) / 64)), dim3(192), 0, (hipStream_t)stream, robot, n_points,
                       configs, num_configs, poses);
    OMGX_CHECK_LAUNCH("k_pose_table");
    return OMGX_OK;
}

static int goalset_cost_impl(const double* robot, int32_t n_points, const omgx_object* objects, const int32_t* scene_begin,
                             const float* sdf_pool, const double* traj_start, int64_t traj_start_stride, const double* goals,
                             int32_t num_scenes, int32_t num_goals, int32_t n_remaining, double time_interval,
                             int32_t soften_fingers, float* goal_cost, float* potentials, float* collides, void* workspace,
                             const double* layer_traj, int32_t layer_n, int32_t layer_soften, float* layer_pot, float* layer_grad,
                             float* layer_col, const int32_t* active, const int32_t* goal_count, const int32_t* schedule,
                             int32_t schedule_len, uint32_t* work, void* stream, const GsTiling& tiling = GsTiling()) {
    if (num_scenes < 0 || num_goals < 0) return OMGX_ERR_INVALID;
    if (num_scenes == 0 || (num_goals == 0 && !layer_traj)) return OMGX_OK;
    if (!robot || !objects || !scene_begin) return OMGX_ERR_INVALID;
    if (num_goals > 0 && (!traj_start || !goals || !goal_cost)) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS || n_remaining < 1 || n_remaining > OMGX_MAX_WAYPOINTS)
        return OMGX_ERR_UNSUPPORTED;
    if (!(time_interval > 0.0) || (num_goals > 0 && traj_start_stride < 9)) return OMGX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int n = num_goals > 0 ? n_remaining : 0, C = num_goals * n;
    if (layer_traj) {  // the trajectory layer rides on k_goalset_queue (cost-only batch)
        if (!layer_pot || !layer_grad || !layer_col) return OMGX_ERR_INVALID;
        if (layer_n < 1 || layer_n > OMGX_MAX_WAYPOINTS) return OMGX_ERR_UNSUPPORTED;
        if (potentials) return OMGX_ERR_UNSUPPORTED;
    }
    if ((active || goal_count || schedule || work) && potentials) return OMGX_ERR_UNSUPPORTED;  // masks and schedules live in k_goalset_queue
    ChunkArgs ca{};
    ca.robot = robot; ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool;
    ca.S = num_scenes; ca.C = C; ca.CH = n; ca.NCH = num_goals; ca.P = n_points; ca.soften = soften_fingers != 0;
    ca.arc = 1; ca.inv_dt = (float)(1.0 / time_interval);
    ca.pot = potentials; ca.grad = nullptr; ca.col = nullptr; ca.chunk_cost = goal_cost; ca.chunk_col = collides;
    ca.traj_start = traj_start; ca.ts_stride = traj_start_stride; ca.goals = goals;  // every workgroup runs its own kinematics
    if (layer_traj) {
        ca.wp_traj = layer_traj; ca.wp_n = layer_n; ca.wp_soften = layer_soften != 0;
        ca.wp_pot = layer_pot; ca.wp_grad = layer_grad; ca.wp_col = layer_col;
    }
    if (schedule && (schedule_len < 8 || schedule_len % 8 != 0)) return OMGX_ERR_INVALID;
    ca.active = active; ca.goal_count = goal_count; ca.schedule = schedule; ca.sched_len = schedule ? schedule_len : 0; ca.work = work;
    GsTiling tl = tiling;
    tl.kin_ws = workspace;  // non-null: the goals' kinematics as a launch of their own (ABI 10)
    return potentials ? launch_chunks(ca, st) : launch_goalset(ca, 0, st, tl);
}

extern "C" int omgx_goalset_cost(const double* robot, int32_t n_points, const omgx_object* objects,
                                 const int32_t* scene_begin, const float* sdf_pool, const double* traj_start,
                                 int64_t traj_start_stride, const double* goals, int32_t num_scenes, int32_t num_goals,
                                 int32_t n_remaining,
                                 double time_interval, int32_t soften_fingers, float* goal_cost, float* potentials,
                                 float* collides, void* workspace, const int32_t* active, const int32_t* goal_count, void* stream) {
    return goalset_cost_impl(robot, n_points, objects, scene_begin, sdf_pool, traj_start, traj_start_stride, goals, num_scenes,
                             num_goals, n_remaining, time_interval, soften_fingers, goal_cost, potentials, collides, workspace,
                             nullptr, 0, 0, nullptr, nullptr, nullptr, active, goal_count, nullptr, 0, nullptr, stream);
}

// =================================================================================================
// (7) k_goalset_schedule — dispatch order of the goal workgroups from their measured durations (include/omg_hip.h)
// =================================================================================================
// One workgroup of 1024.  Everything is integer arithmetic with fixed tie-breaks (index order), so the schedule is a pure
// function of its inputs.  Per-scene arrays live in LDS; an item's position in the list is found by counting (its scene's rank
// among the scenes, its own rank among its scene's goals) — O(S^2 + S G^2) comparisons, ~0.5 M for 100 x 64.
#define SCH_TPB 1024
#define SCH_LDS_ITEMS 14336  // up to this many items have their weights staged in LDS (56 KB: no opt-in needed); beyond, the loops read global memory
struct SchedArgs {
    const uint32_t* work;
    const int32_t* active;
    const int32_t* goal_count;
    int S, G, slack, slots, staged;
    int32_t* sched;
    int np;  // items per goal (omgx_goalset_schedule_parts): G counts items, item g of a scene belongs to its goal g / np
    int longest_first;  // omgx_goalset_schedule_ordered: inside an XCD the items run longest first across its scenes (staged only)
};

__global__ __launch_bounds__(SCH_TPB) void k_goalset_schedule(SchedArgs a) {
    extern __shared__ unsigned long long sch_lds[];
    const int S = a.S, G = a.G, tid = threadIdx.x;
    unsigned long long* Ws = sch_lds;                                   // [S] scene weight (unclamped)
    unsigned long long* Wc = Ws + S;                                    // [S] scene weight (clamped)
    unsigned long long* offw = Wc + S;                                  // [S] clamped weight of the scenes ranked before
    uint32_t* nval = reinterpret_cast<uint32_t*>(offw + S);             // [S] kept goals
    uint32_t* offp = nval + S;                                          // [S] items of the scenes ranked before
    uint32_t* srank = offp + S;                                         // [S]
    uint32_t* wl = srank + S;                                           // [S*G] weight of a kept item (>= 1), 0 = left out (a.staged)
    unsigned char* xs = reinterpret_cast<unsigned char*>(wl + S * G);   // [S*G] the item's XCD, 255 = left out (a.longest_first)
    __shared__ unsigned long long tot[2];                               // total weight, total clamped weight
    __shared__ uint32_t cnt_all;
    __shared__ uint32_t first[8];    // lowest list position of every piece, clamped cut
    __shared__ uint32_t first_r[8];  // ... cut by the weights as measured
    __shared__ uint32_t cnt_r[8];    // items of every piece under that cut
    // weight of item (s, g), 0 if it is left out: from LDS when staged (the loops below read every weight O(G) times)
    auto item_w = [&](int s, int g) -> uint32_t {
        if (a.staged) return wl[s * G + g];
        if ((a.active && a.active[s] == 0) || (a.goal_count && g / a.np >= a.goal_count[s])) return 0u;
        const uint32_t w = a.work ? a.work[(size_t)s * G + g] : 1u;
        return w ? w : 1u;
    };
    if (tid == 0) { tot[0] = tot[1] = 0ull; cnt_all = 0u; }
    if (tid < 8) { first[tid] = 0xffffffffu; first_r[tid] = 0xffffffffu; cnt_r[tid] = 0u; }
    for (int i = tid; i < a.slots * 8; i += SCH_TPB) a.sched[i] = -1;
    if (a.staged)
        for (int i = tid; i < S * G; i += SCH_TPB) {
            const int s = i / G, g = i - s * G;
            uint32_t w = 0u;
            if (!((a.active && a.active[s] == 0) || (a.goal_count && g / a.np >= a.goal_count[s]))) {
                w = a.work ? a.work[i] : 1u;
                w = w ? w : 1u;
            }
            wl[i] = w;
        }
    __syncthreads();
    for (int s = tid; s < S; s += SCH_TPB) {  // scene weights
        unsigned long long w = 0;
        uint32_t n = 0;
        for (int g = 0; g < G; ++g) {
            const uint32_t wi = item_w(s, g);
            w += wi; n += wi ? 1u : 0u;
        }
        Ws[s] = w; nval[s] = n;
        atomicAdd(&tot[0], w);
        atomicAdd(&cnt_all, n);
    }
    __syncthreads();
    const unsigned long long n_all = cnt_all ? cnt_all : 1u;
    unsigned long long lo = (10ull * tot[0]) / (14ull * n_all);
    if (lo < 1) lo = 1;
    const unsigned long long hi = (unsigned long long)a.slack * lo;
    // clamp(w) = min(max(w, lo), hi) fits 32 bits (w does, and lo <= the largest w): one v_med3_u32 in the inner loops
    const uint32_t lo32 = (uint32_t)lo, hi32 = hi > 0xffffffffull ? 0xffffffffu : (uint32_t)hi;
    auto wclamp = [&](uint32_t w) -> uint32_t { return w < lo32 ? lo32 : (w > hi32 ? hi32 : w); };
    for (int s = tid; s < S; s += SCH_TPB) {  // clamped scene weights, scene ranks (heaviest first, ties by index)
        unsigned long long w = 0;
        for (int g = 0; g < G; ++g) {
            const uint32_t wi = item_w(s, g);
            if (wi) w += wclamp(wi);
        }
        Wc[s] = w;
        atomicAdd(&tot[1], w);
        uint32_t r = 0;
        for (int q = 0; q < S; ++q) r += (Ws[q] > Ws[s] || (Ws[q] == Ws[s] && q < s)) ? 1u : 0u;
        srank[s] = r;
    }
    __syncthreads();
    unsigned long long offr_reg[2] = {0ull, 0ull};  // (S <= OMGX_SCHEDULE_MAX_SCENES = 1792: at most two scenes per thread)
    for (int s = tid, k = 0; s < S; s += SCH_TPB, ++k) {  // what the scenes ranked before this one hold
        unsigned long long w = 0, wr = 0;
        uint32_t n = 0;
        for (int q = 0; q < S; ++q)
            if (srank[q] < srank[s]) { w += Wc[q]; wr += Ws[q]; n += nval[q]; }
        offw[s] = w; offp[s] = n;
        offr_reg[k & 1] = wr;
    }
    __syncthreads();
    unsigned long long* const offr = Wc;  // the scenes' clamped weights have been summed: their array holds the raw prefix from here on
    for (int s = tid, k = 0; s < S; s += SCH_TPB, ++k) offr[s] = offr_reg[k & 1];
    __syncthreads();
    const unsigned long long total_c = tot[1] ? tot[1] : 1ull, total_r = tot[0] ? tot[0] : 1ull;
    // x: the item's piece when the list is cut by CLAMPED work (a piece then never needs more than its slots); xr: when it is cut by
    // the work as measured — used when every piece fits (the usual case: the clamp's bias, light scenes counted for more than they
    // are and heavy ones for less, left the XCD with the heaviest scenes finishing 10-25 % after the one with the lightest)
    auto place = [&](int s, int g, uint32_t w, uint32_t& pos, int& x, int& xr) {
        uint32_t r = 0;
        unsigned long long before = 0, before_r = 0;
#pragma unroll 4
        for (int q = 0; q < G; ++q) {
            const uint32_t wq = item_w(s, q);
            const bool ahead = wq > w || (wq == w && q < g);  // wq = 0 (left out) never ranks before a kept item
            r += ahead ? 1u : 0u;
            before += ahead ? wclamp(wq) : 0u;
            before_r += ahead ? wq : 0u;
        }
        pos = offp[s] + r;
        const unsigned long long c2 = 2ull * (offw[s] + before) + wclamp(w);
        const unsigned long long xx = (8ull * c2) / (2ull * total_c);
        x = xx > 7 ? 7 : (int)xx;
        const unsigned long long r2 = 2ull * (offr[s] + before_r) + w;
        const unsigned long long xq = (8ull * r2) / (2ull * total_r);
        xr = xq > 7 ? 7 : (int)xq;
    };
    // an item's place is computed once and kept in registers for the second pass (up to 4 items per thread: 4096 items;
    // beyond that the second pass computes it again)
    uint32_t kpos[4];
    int kx[4], kxr[4];
    int kept_n = 0;
    for (int i = tid; i < S * G; i += SCH_TPB) {  // first position of every piece
        const int s = i / G, g = i - s * G;
        const uint32_t w = item_w(s, g);
        uint32_t pos = 0; int x = -1, xr = -1;
        if (w) { place(s, g, w, pos, x, xr); atomicMin(&first[x], pos); atomicMin(&first_r[xr], pos); atomicAdd(&cnt_r[xr], 1u); }
        if (kept_n < 4) { kpos[kept_n] = pos; kx[kept_n] = x; kxr[kept_n] = xr; }
        ++kept_n;
        if (a.longest_first) xs[i] = (unsigned char)(w ? (x | (xr << 4)) : 255);
    }
    __syncthreads();
    bool use_raw = true;  // every piece of the raw cut fits its slots
#pragma unroll
    for (int x = 0; x < 8; ++x) use_raw = use_raw && cnt_r[x] <= (uint32_t)a.slots;
    if (a.longest_first) {
        // The XCD of every item as above (whole scenes per XCD, equal work); its place inside the XCD by weight alone — longest first
        // across the XCD's scenes, ties by index.  For launches of a round or two of the chip's workgroup slots, whose span is set by
        // what starts last (tools/experiments/ab_schedule_order.py; with many rounds the scene-major order keeps a scene's volumes in L2 and wins).
        for (int i = tid; i < S * G; i += SCH_TPB) {
            const uint32_t w = wl[i];
            if (!w) continue;
            const int sh = use_raw ? 4 : 0;
            const unsigned char x = (xs[i] >> sh) & 7;
            uint32_t r = 0;
#pragma unroll 4
            for (int j = 0; j < S * G; ++j) r += (wl[j] != 0u && ((xs[j] >> sh) & 7) == x && (wl[j] > w || (wl[j] == w && j < i))) ? 1u : 0u;
            if ((int)r < a.slots) a.sched[(size_t)r * 8 + x] = i;
        }
        return;
    }
    int k = 0;
    for (int i = tid; i < S * G; i += SCH_TPB, ++k) {
        const int s = i / G, g = i - s * G;
        uint32_t pos; int x, xr;
        if (k < 4) { pos = kpos[k]; x = kx[k]; xr = kxr[k]; if (x < 0) continue; }
        else { const uint32_t w = item_w(s, g); if (!w) continue; place(s, g, w, pos, x, xr); }
        if (use_raw) x = xr;
        const uint32_t r = pos - (use_raw ? first_r[x] : first[x]);
        if ((int)r < a.slots) a.sched[(size_t)r * 8 + x] = i;  // always true: checked for the raw cut, bounded by the clamp otherwise (include/omg_hip.h)
    }
}

extern "C" int32_t omgx_goalset_schedule_len(int32_t num_scenes, int32_t num_goals, int32_t slack) {
    if (num_scenes <= 0 || num_goals <= 0 || slack < 1) return 0;
    const int64_t n = (int64_t)num_scenes * num_goals;
    return (int32_t)(((slack * n + 7) / 8 + 2) * 8);
}

static int goalset_schedule_impl(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                 int32_t num_goals, int32_t parts, int32_t slack, int32_t* schedule, void* stream, int32_t order = 0);

extern "C" int omgx_goalset_schedule(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                     int32_t num_goals, int32_t slack, int32_t* schedule, void* stream) {
    return goalset_schedule_impl(work, active, goal_count, num_scenes, num_goals, 1, slack, schedule, stream);
}

extern "C" int omgx_goalset_schedule_parts(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                           int32_t num_goals, int32_t parts, int32_t slack, int32_t* schedule, void* stream) {
    if (parts != 1 && parts != 2 && parts != 4 && parts != 8) return OMGX_ERR_INVALID;
    return goalset_schedule_impl(work, active, goal_count, num_scenes, num_goals, parts, slack, schedule, stream);
}

extern "C" int omgx_goalset_schedule_ordered(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                             int32_t num_goals, int32_t parts, int32_t slack, int32_t order, int32_t* schedule, void* stream) {
    if (parts != 1 && parts != 2 && parts != 4 && parts != 8) return OMGX_ERR_INVALID;
    if (order != OMGX_SCHEDULE_SCENE_MAJOR && order != OMGX_SCHEDULE_LONGEST_FIRST) return OMGX_ERR_INVALID;
    return goalset_schedule_impl(work, active, goal_count, num_scenes, num_goals, parts, slack, schedule, stream, order);
}

// items per scene = goals x parts; everything below counts items
static int goalset_schedule_impl(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                 int32_t num_goals_, int32_t parts, int32_t slack, int32_t* schedule, void* stream, int32_t order) {
    if (num_scenes <= 0 || num_goals_ <= 0 || slack < 1 || !schedule) return OMGX_ERR_INVALID;
    const int64_t items64 = (int64_t)num_goals_ * parts;
    if (items64 > 65536) return OMGX_ERR_UNSUPPORTED;
    const int32_t num_goals = (int32_t)items64;
    // per-scene arrays in dynamic LDS: 36 bytes per scene; 1792 scenes = 63 KB, below the 64 KB a launch gets without opting in
    if ((int64_t)num_scenes * num_goals > 65536 || num_scenes > OMGX_SCHEDULE_MAX_SCENES) return OMGX_ERR_UNSUPPORTED;
    const int items = num_scenes * num_goals;
    const int staged = (items <= SCH_LDS_ITEMS && num_scenes <= 128) ? 1 : 0;  // keeps the launch below 64 KB of dynamic LDS
    // longest first inside an XCD: O(items^2) comparisons in one workgroup and a byte per item in LDS — for the launches it is meant for
    // (a round or two of the chip's slots); larger ones keep the scene-major order, which is the better one there anyway
    const int longest_first = (order == OMGX_SCHEDULE_LONGEST_FIRST && staged && items <= OMGX_SCHEDULE_LONGEST_FIRST_MAX_ITEMS) ? 1 : 0;
    SchedArgs a{work, active, goal_count, num_scenes, num_goals, slack, omgx_goalset_schedule_len(num_scenes, num_goals, slack) / 8, staged, schedule, parts, longest_first};
    const size_t lds = (size_t)num_scenes * (3 * sizeof(unsigned long long) + 3 * sizeof(uint32_t)) + (staged ? (size_t)items * sizeof(uint32_t) : 0) +
                       (longest_first ? (size_t)items : 0);
    hipLaunchKernelGGL(k_goalset_schedule, dim3(1), dim3(SCH_TPB), lds, (hipStream_t)stream, a);
    OMGX_CHECK_LAUNCH("k_goalset_schedule");
    return OMGX_OK;
}

extern "C" int omgx_goalset_cost_layer(const double* robot, int32_t n_points, const omgx_object* objects,
                                       const int32_t* scene_begin, const float* sdf_pool, const double* traj_start,
                                       int64_t traj_start_stride, const double* goals, int32_t num_scenes, int32_t num_goals,
                                       int32_t n_remaining, double time_interval, int32_t soften_fingers, float* goal_cost,
                                       float* collides, void* workspace, const double* traj, int32_t n_waypoints,
                                       int32_t layer_soften_fingers, float* layer_potentials, float* layer_grads,
                                       float* layer_collides, const int32_t* active, const int32_t* goal_count,
                                       const int32_t* schedule, int32_t schedule_len, uint32_t* work, void* stream) {
    if (!traj) return OMGX_ERR_INVALID;
    return goalset_cost_impl(robot, n_points, objects, scene_begin, sdf_pool, traj_start, traj_start_stride, goals, num_scenes,
                             num_goals, n_remaining, time_interval, soften_fingers, goal_cost, nullptr, collides, workspace, traj,
                             n_waypoints, layer_soften_fingers, layer_potentials, layer_grads, layer_collides, active, goal_count, schedule,
                             schedule_len, work, stream);
}

extern "C" int omgx_goalset_cost_layer_parts(const double* robot, int32_t n_points, const omgx_object* objects,
                                             const int32_t* scene_begin, const float* sdf_pool, const double* traj_start,
                                             int64_t traj_start_stride, const double* goals, int32_t num_scenes, int32_t num_goals,
                                             int32_t n_remaining, double time_interval, int32_t soften_fingers, float* goal_cost,
                                             float* collides, const double* traj, int32_t n_waypoints,
                                             int32_t layer_soften_fingers, float* layer_potentials, float* layer_grads,
                                             float* layer_collides, const int32_t* active, const int32_t* goal_count,
                                             const int32_t* schedule, int32_t schedule_len, uint32_t* work, int32_t goal_parts,
                                             double* layer_poses, void* workspace, void* stream) {
    if (!traj) return OMGX_ERR_INVALID;
    GsTiling tl;
    tl.goal_parts = goal_parts;
    tl.wp_pose_out = layer_poses;
    return goalset_cost_impl(robot, n_points, objects, scene_begin, sdf_pool, traj_start, traj_start_stride, goals, num_scenes,
                             num_goals, n_remaining, time_interval, soften_fingers, goal_cost, nullptr, collides, workspace, traj,
                             n_waypoints, layer_soften_fingers, layer_potentials, layer_grads, layer_collides, active, goal_count, schedule,
                             schedule_len, work, stream, tl);
}

extern "C" int32_t omgx_goalset_parts(int32_t n_remaining, int32_t goal_parts) {
    if (n_remaining < 1 || goal_parts < 1 || goal_parts > 8) return 0;
    return gs_parts(n_remaining, goal_parts);
}

extern "C" int omgx_goalset_cost_layer_tiled(const double* robot, int32_t n_points, const omgx_object* objects,
                                             const int32_t* scene_begin, const float* sdf_pool, const double* traj_start,
                                             int64_t traj_start_stride, const double* goals, int32_t num_scenes, int32_t num_goals,
                                             int32_t n_remaining, double time_interval, int32_t soften_fingers, float* goal_cost,
                                             float* collides, const double* traj, int32_t n_waypoints, int32_t layer_soften_fingers,
                                             float* layer_potentials, float* layer_grads, float* layer_collides,
                                             const int32_t* active, const int32_t* goal_count, int32_t goal_parts,
                                             int32_t layer_link_groups, int32_t layer_config_block, int32_t spread, double* layer_poses,
                                             void* workspace, void* stream) {
    if (!traj && num_goals <= 0) return OMGX_ERR_INVALID;
    GsTiling tl;
    tl.goal_parts = goal_parts; tl.layer_lg = layer_link_groups; tl.layer_cb = layer_config_block; tl.spread = spread;
    tl.wp_pose_out = traj ? layer_poses : nullptr;
    return goalset_cost_impl(robot, n_points, objects, scene_begin, sdf_pool, traj_start, traj_start_stride, goals, num_scenes,
                             num_goals, num_goals > 0 ? n_remaining : 1, time_interval, soften_fingers, goal_cost, nullptr, collides, workspace, traj,
                             n_waypoints, layer_soften_fingers, layer_potentials, layer_grads, layer_collides, active, goal_count, nullptr,
                             0, nullptr, stream, tl);
}

// =================================================================================================
// (9) omgx_plan_persistent — K planner iterations for all scenes in one launch (omg_persist.h)
// =================================================================================================
#include "omg_chomp_body.h"   // float64 code, contract(fast) from here on: nothing float32 may follow but what was parsed above
#include "omg_persist.h"
#pragma clang fp contract(off)

static inline int64_t persist_ring_cap(int32_t S) { return ((int64_t)S + 63) & ~63ll; }

extern "C" int64_t omgx_plan_persistent_workspace_bytes(int32_t num_scenes, int32_t n_waypoints) {
    if (num_scenes <= 0 || n_waypoints <= 0) return 0;
    // ring [cap] u64 | claim words: 8 x PQ_SLOTS lines of 128 bytes | control: a line | statistics: a line | arrivals [S] u32 (padded to 128 bytes) |
    // CU roles [4096] u32 | per-XCD counters: 8 lines | update rings [8][cap] u64 | their {tail | head}: 8 lines | gradient rows [S][n][10][8] f64
    return persist_ring_cap(num_scenes) * 8 + 8 * PQ_SLOTS * 128 + 128 + 128 + (((int64_t)num_scenes * 4 + 127) & ~127ll) + 4096 * 4 + 8 * 128 +
           8 * persist_ring_cap(num_scenes) * 8 + 8 * 128 + (int64_t)num_scenes * n_waypoints * 80 * 8;
}

extern "C" int omgx_plan_persistent_status(const void* workspace, int32_t num_scenes, int32_t* h_status /* [4]: failure code, scenes finished, scenes planned, activations made */, void* stream) {
    if (!workspace || !h_status || num_scenes <= 0) return OMGX_ERR_INVALID;
    const char* base = reinterpret_cast<const char*>(workspace) + persist_ring_cap(num_scenes) * 8 + 8 * PQ_SLOTS * 128;
    uint32_t ctl[8];
    hipError_t e = hipMemcpyAsync(ctl, base, sizeof(ctl), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return omgx_set_error("omgx_plan_persistent_status", e);
    h_status[0] = (int32_t)ctl[3]; h_status[1] = (int32_t)ctl[2]; h_status[2] = (int32_t)ctl[4]; h_status[3] = (int32_t)ctl[0];
    return OMGX_OK;
}
// Measurement builds (-DOMGX_PERSIST_STATS; tools/experiments/persist_stats.py): h_stats[8] <- claim spins, item ticks, update ticks, items,
// updates, claim ticks (ticks of 10 ns, summed over workgroups).  Not part of the ABI.
extern "C" int omgx_debug_persist_stats(const void* workspace, int32_t num_scenes, unsigned long long* h_stats, void* stream) {
    const char* base = reinterpret_cast<const char*>(workspace) + persist_ring_cap(num_scenes) * 8 + 8 * PQ_SLOTS * 128 + 128;
    hipError_t e = hipMemcpyAsync(h_stats, base, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? OMGX_OK : omgx_set_error("omgx_debug_persist_stats", e);
}

extern "C" int omgx_plan_persistent(const double* robot, int32_t n_points, const omgx_object* objects, const int32_t* scene_begin,
                                    const float* sdf_pool, const double* goals, int32_t num_scenes, int32_t num_goals,
                                    double time_interval, int32_t soften_fingers, float* goal_cost, float* collides, double* traj,
                                    int32_t n_waypoints, int32_t layer_soften_fingers, float* layer_potentials, float* layer_grads,
                                    float* layer_collides, double* layer_poses, int32_t* active, const int32_t* goal_count,
                                    const omgx_learner_params* h_learner, const double* goal_set, const double* reach,
                                    double* learner_state, int32_t* goal_idx, double* cost_vector, const double* eta,
                                    const omgx_chomp_params* h_params, const double* start, double* end, double* goal,
                                    double* goal_point, double* grad, double* cost_traj, double* info,
                                    const omgx_plan_iter* h_iters, const omgx_plan_iter* d_iters, int32_t num_iters,
                                    void* workspace, int64_t workspace_bytes, int32_t max_workgroups, int32_t update_cus, void* stream) {
    using namespace omg_persist;
    if (num_scenes < 0 || num_iters < 0 || !h_learner || !h_params || !h_iters || !d_iters) return OMGX_ERR_INVALID;
    if (num_scenes == 0 || num_iters == 0) return OMGX_OK;
    if (!robot || !objects || !scene_begin || !traj || !layer_potentials || !layer_grads || !layer_collides || !layer_poses || !start ||
        !end || !goal || !goal_point || !grad || !cost_traj || !info || !workspace)
        return OMGX_ERR_INVALID;
    if (num_scenes > 65535 || num_iters > 65535) return OMGX_ERR_UNSUPPORTED;  // an activation word holds 16 bits of each
    if (n_points < 1 || n_points > OMGX_MAX_POINTS || n_waypoints < 1 || n_waypoints > OMGX_MAX_WAYPOINTS) return OMGX_ERR_UNSUPPORTED;
    if (!(time_interval > 0.0)) return OMGX_ERR_INVALID;
    if (workspace_bytes < omgx_plan_persistent_workspace_bytes(num_scenes, n_waypoints)) return OMGX_ERR_INVALID;
    const omgx_chomp_params& cp = *h_params;
    if (cp.n_waypoints != n_waypoints || cp.n_points != n_points || cp.constraint_num < 1 || cp.constraint_num > OMGX_MAX_CONSTRAINTS ||
        cp.constraint_num > cp.n_waypoints || cp.top_k < 0)
        return OMGX_ERR_INVALID;
    // the step runs inside a goal workgroup's footprint: every pose it needs is handed over (omgx_chomp_params / omgx_learner_params, ABI 7)
    if (!cp.start_poses || !cp.end_poses) return OMGX_ERR_INVALID;
    bool any_select = false;
    int min_start = n_waypoints;
    for (int k = 0; k < num_iters; ++k) {
        const omgx_plan_iter& r = h_iters[k];
        if (r.mode < 0 || r.mode > 1 || r.start_idx < 0 || r.start_idx >= n_waypoints || r.do_update < 0 || r.do_update > 2) return OMGX_ERR_INVALID;
        if (r.mode) { any_select = true; if (r.start_idx < min_start) min_start = r.start_idx; }
    }
    omg_learner::LearnerArgs la{};
    if (any_select) {
        if (num_goals < 1 || !goals || !goal_cost || !goal_set || !learner_state || !goal_idx) return OMGX_ERR_INVALID;
        if (h_learner->alg == OMGX_ALG_PROJ) return OMGX_ERR_UNSUPPORTED;  // no goal-set batch to ride on: the per-iteration launches serve it
        if (!h_learner->goal_pose_table || !h_learner->end_poses_out) return OMGX_ERR_INVALID;
        if (h_learner->cost_parts > 1 || h_learner->n_waypoints != n_waypoints || h_learner->constraint_num != cp.constraint_num) return OMGX_ERR_INVALID;
        const int rc = omg_learner::make_args(h_learner, traj, goal_set, reach, goal_cost, learner_state, num_scenes, goal_idx, end, goal,
                                              goal_point, cost_vector, nullptr, goal_count, eta, la);
        if (rc != OMGX_OK) return rc;
        if (h_learner->num_goals != num_goals) return OMGX_ERR_INVALID;
    }
    PersistArgs pa{};
    ChunkArgs& ca = pa.ca;
    ca.robot = robot; ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool;
    ca.S = num_scenes; ca.NCH = num_goals; ca.NG = num_goals; ca.NP = 1; ca.P = n_points; ca.soften = soften_fingers != 0;
    ca.arc = 1; ca.inv_dt = (float)(1.0 / time_interval);
    ca.chunk_cost = goal_cost; ca.chunk_col = collides;
    ca.ts_stride = (int64_t)n_waypoints * 9; ca.goals = goals;
    ca.wp_traj = traj; ca.wp_n = n_waypoints; ca.wp_soften = layer_soften_fingers != 0;
    ca.wp_pot = layer_potentials; ca.wp_grad = layer_grads; ca.wp_col = layer_collides; ca.wp_pose_out = layer_poses;
    ca.goal_count = goal_count; ca.LPW = 10;
    ca.layer_lg = 5; ca.layer_nb = 1; ca.layer_cb = n_waypoints; ca.layer_parts = 5;
    pa.la = la;
    ChompArgs& ch = pa.ch;
    ch.robot = robot; ch.prm = cp; ch.traj = traj; ch.start = start; ch.end = end; ch.goal = goal; ch.goal_point = goal_point;
    ch.pot = layer_potentials; ch.pgrad = layer_grads; ch.col = layer_collides; ch.grad = grad; ch.cost_traj = cost_traj; ch.info = info;
    ch.prm.waypoint_poses = layer_poses;  // the layer items of this launch leave them there
    pa.iters = d_iters; pa.num_iters = num_iters; pa.G = num_goals; pa.active = active;
    char* w = reinterpret_cast<char*>(workspace);
    pa.cap = (int)persist_ring_cap(num_scenes);
    pa.ring = reinterpret_cast<unsigned long long*>(w); w += (size_t)pa.cap * 8;
    pa.xw = reinterpret_cast<unsigned long long*>(w); w += 8 * PQ_SLOTS * 128;
    pa.ctl = reinterpret_cast<uint32_t*>(w); w += 128;
    pa.stats = reinterpret_cast<unsigned long long*>(w); w += 128;
    pa.arrive = reinterpret_cast<uint32_t*>(w); w += ((size_t)num_scenes * 4 + 127) & ~(size_t)127;
    pa.cu_role = reinterpret_cast<uint32_t*>(w); w += 4096 * 4;
    pa.xcd_cus = reinterpret_cast<uint32_t*>(w); w += 8 * 128;
    pa.uq = reinterpret_cast<unsigned long long*>(w); w += (size_t)8 * pa.cap * 8;
    pa.uq_ht = reinterpret_cast<unsigned long long*>(w); w += 8 * 128;
    ch.light_scratch = reinterpret_cast<double*>(w);
    // dynamic LDS: the goal workgroup's layout for the longest window of the plan, or what the learner / the light step carve from the same block
    const int CHmax = any_select ? n_waypoints - min_start : 1;
    const int PS = CHmax + 1 > n_waypoints ? CHmax + 1 : n_waypoints, MR = CHmax > n_waypoints ? CHmax : n_waypoints;
    size_t lds = (size_t)GqLayout(PS, MR, n_points, gq_choose_tbl_n(PS, MR, n_points)).total;
    const size_t light = chomp_light_lds_bytes(n_waypoints, n_points), learner = (size_t)(5 * OMGX_MAX_GOALS + 5 * 128 + 2) * sizeof(double);
    if (lds < light) lds = light;
    if (lds < learner) lds = learner;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 64 * 1024) return OMGX_ERR_UNSUPPORTED;
    pa.lds_bytes = (uint32_t)lds;
    int cus = omgx_device_cu_count();
    if (cus <= 0) cus = 256;
    int64_t grid = (int64_t)cus * GQ_WG_PER_CU;
    const int64_t items = (int64_t)num_scenes * (5 + (any_select ? num_goals : 0));
    if (grid > items) grid = items;
    if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
    grid = (grid + 7) & ~7ll;
    // dedicated update CUs per XCD (omg_persist.h): only when the launch fills the chip (every CU then holds workgroups of this launch) and
    // there are scenes enough to keep them busy; < 0: this rule, else the caller's number (0: updates run where the scene's last item ran)
    pa.update_cus = update_cus >= 0 ? (update_cus > 8 ? 8 : update_cus) : ((grid >= (int64_t)cus * GQ_WG_PER_CU && num_scenes >= 32 && any_select) ? 2 : 0);
    // NEVER in a launch that does not fill the chip, whatever the caller asks for: the first CUs of an XCD to report become update CUs, and
    // with a workgroup or two per XCD those are ALL of its CUs — nobody would run an item (fuzz campaign r06final: one scene with one to
    // five goals and update_cus = 1 waited 2 s for requests that could not come, failure code 5).  And never more than a quarter of an XCD.
    if (grid < (int64_t)cus * GQ_WG_PER_CU) pa.update_cus = 0;
    if (pa.update_cus > cus / 32) pa.update_cus = cus / 32;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_persist_init, dim3(1), dim3(256), 0, st, pa);
    OMGX_CHECK_LAUNCH("k_persist_init");
    hipEvent_t ev0, ev1;
    timing_events(2, &ev0, &ev1);
    if (ev0) hipExtLaunchKernelGGL((k_plan_persistent<2>), dim3((unsigned)grid), dim3(GQ_NT), (uint32_t)lds, st, ev0, ev1, 0, pa);
    else hipLaunchKernelGGL((k_plan_persistent<2>), dim3((unsigned)grid), dim3(GQ_NT), (uint32_t)lds, st, pa);
    OMGX_CHECK_LAUNCH("k_plan_persistent");
    return OMGX_OK;
}
