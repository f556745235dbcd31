// omg_goalset_kin.h — k_goalset_kin: the kinematics and the row culling of the goal-set batch as a launch of its own (ABI 10).
// Included by omg_kernels.hip after ChunkArgs.
//
// Why.  A goal workgroup of k_goalset_queue spends 40 % of its life in a prologue — (sin, cos), joint matrices, the kinematic
// chain, row culling (omg/util.py:261-290 + robot_pykdl.py:148-215 ahead of omg/cost.py:192-286) — that is a chain of dependent
// float64 steps: one instruction per ~40 cycles per wave, with 32 KB of LDS and five wave slots per SIMD held meanwhile
// (DESIGN.md section 4.1).  Here the same arithmetic runs where it has the parallelism it lacks there: ONE LANE PER
// (goal, configuration).  A lane carries all three rows of its configuration's chain in registers — 12 independent multiply-add
// chains per joint instead of one — and tests every link's bounding ball against the scene's influence regions as soon as the
// link's translation exists; no LDS, no barrier, nothing shared between lanes.  The poses go to a workspace in HBM as
// [goal][link][component][configuration] (a store instruction writes consecutive doubles), the row masks as [goal][link][row];
// k_goalset_queue<..., PRE = true> starts from there with one trip to memory.
//
// Same bits: joint(), fk_joint_sincos, fk_bentry, fk_dot3, fk_chain_tail are the expressions of the fused prologue (explicit
// fma, contraction off), so every pose double and every mask word equals what the workgroup would have computed itself
// (tests/test_gpu_prepass.py compares the launches' outputs bit for bit).
#pragma once

// workspace of one launch: poses [S * NG][10][9][CH + 1] doubles, then masks [S * NG][10][CH] uint32
__host__ __device__ static inline int64_t gk_pose_doubles(int CH) { return (int64_t)90 * (CH + 1); }
static inline int64_t gk_workspace_bytes(int64_t goals, int CH) {
    return ((goals * (gk_pose_doubles(CH) * 8 + (int64_t)10 * CH * 4)) + 15) & ~(int64_t)15;
}

__global__ __launch_bounds__(256) void k_goalset_kin(ChunkArgs a) {
    const int CH = a.CH, ncfg = CH + 1, NG = a.NG;
    const int gpw = ncfg <= 64 ? 64 / ncfg : 1;  // goals per wave: the lanes of a wave belong to ONE scene (object records stay scalar)
    const int ngrp = (NG + gpw - 1) / gpw;       // waves per scene
    const int wv = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int s = wv / ngrp, grp = wv - s * ngrp;
    if (s >= a.S) return;
    if (a.active && as_const(a.active)[s] == 0) return;
    const int lane = threadIdx.x & 63;
    const int gl = lane / ncfg;
    const int goal = grp * gpw + gl;
    const int ngoals = a.goal_count ? (as_const(a.goal_count)[s] < NG ? as_const(a.goal_count)[s] : NG) : NG;
    const bool owner = gl < gpw && goal < ngoals;  // else: a lane without a goal computes on goal 0 and stores nothing
    const int goalc = owner ? goal : 0;
    const int o_begin = as_const(a.scene_begin)[s], o_end = as_const(a.scene_begin)[s + 1];
    const RobotViewS rv(a.robot, a.P);
    const double* q0 = a.traj_start + a.ts_stride * (int64_t)s;
    const double* qg = a.goals + ((int64_t)s * NG + goalc) * 9;
    const int64_t gi = (int64_t)s * NG + goalc;
    double* const pw = a.pre_poses + gi * gk_pose_doubles(CH);
    uint32_t* const mw = a.pre_masks + gi * (int64_t)(10 * CH);
    const double step = 1.0 / (double)(CH + 1);
    double qs[9], qd[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) { qs[d] = q0[d]; qd[d] = qg[d] - qs[d]; }

    for (int cfg = lane - gl * ncfg; cfg < ncfg; cfg += 64) {  // a second pass only beyond 64 configurations
        // cfg 0 = start itself, cfg i = start + (i * step) * (goal - start): util.py:261-290 as in k_goalset_queue
        auto joint = [&](int d) { return cfg == 0 ? qs[d] : qs[d] + ((double)cfg * step) * qd[d]; };
        const bool st = owner;
        auto put = [&](int l, int k, double v) { if (st) pw[(int64_t)(l * 9 + k) * ncfg + cfg] = v; };
        double A[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}}, T[3] = {0.0, 0.0, 0.0};
        float cx[10], cy[10], cz[10];  // the centres of the links' bounding balls as the culling reads them (cull_row: link_ball_center)
#pragma unroll  // unrolled: the link index of every store and of cx / cy / cz is a constant (registers, not scratch)
        for (int i = 0; i < 7; ++i) {
            double sn, cs;
            fk_joint_sincos(joint(i), sn, cs);
            const auto uvw = rv.uvw(i);
            const auto tp = rv.tp(i);
            double B[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) B[k] = fk_bentry(cs, sn, uvw[k], uvw[9 + k], uvw[18 + k]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // fk_chain_row_B's step for row r
                const double n0 = fk_dot3(A[r][0], A[r][1], A[r][2], B[0], B[3], B[6]);
                const double n1 = fk_dot3(A[r][0], A[r][1], A[r][2], B[1], B[4], B[7]);
                const double n2 = fk_dot3(A[r][0], A[r][1], A[r][2], B[2], B[5], B[8]);
                T[r] = fk_dot3(A[r][0], A[r][1], A[r][2], tp[0], tp[1], tp[2]) + T[r];
                A[r][0] = n0; A[r][1] = n1; A[r][2] = n2;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) { put(i, 3 * r, A[r][0]); put(i, 3 * r + 1, A[r][1]); put(i, 3 * r + 2, A[r][2]); }
            put(i, 6, T[0]); put(i, 7, T[1]); put(i, 8, T[2]);
            link_ball_center(A[0], A[1], T, rv.ball(i), cx[i], cy[i], cz[i]);
        }
        const double q7 = joint(7), q8 = joint(8);
        double tr_[3][2][3], tt_[3][3];  // hand and fingers: [link - 7][row 0 | 1][3], translations [link - 7][3]
#pragma unroll
        for (int r = 0; r < 3; ++r)
            fk_chain_tail(rv, A[r][0], A[r][1], A[r][2], T[r], q7, q8, [&](int l, double r0, double r1, double r2, double tr) {
                if (r < 2) { put(l, 3 * r, r0); put(l, 3 * r + 1, r1); put(l, 3 * r + 2, r2); tr_[l - 7][r][0] = r0; tr_[l - 7][r][1] = r1; tr_[l - 7][r][2] = r2; }
                put(l, 6 + r, tr);
                tt_[l - 7][r] = tr;
            });
#pragma unroll
        for (int l = 7; l < 10; ++l) link_ball_center(tr_[l - 7][0], tr_[l - 7][1], tt_[l - 7], rv.ball(l), cx[l], cy[l], cz[l]);
        // Row masks: cull_row of omg_goalset_queue.h with the OBJECT in the outer loop — its record comes through the scalar cache
        // once (two dependent trips: `disabled`, then the fields) and serves the ten links; a mask's bits do not depend on the order
        // they are set in.
        uint32_t m[10];
        float rad[10];
#pragma unroll
        for (int l = 0; l < 10; ++l) { m[l] = 0u; rad[l] = (float)rv.ball(l)[3] + 1.0e-4f; }
        for (int o = o_begin; o < o_end; ++o) {
            ObjTablePtr ob = as_const(a.objects) + o;
            if (ob->disabled > 0) continue;
            const int oo = o - o_begin;
            const uint32_t bit = 1u << (oo < 31 ? oo : 31);
            const float rbc[3] = {ob->rb_c[0], ob->rb_c[1], ob->rb_c[2]}, rbh[3] = {ob->rb_h[0], ob->rb_h[1], ob->rb_h[2]};
            const bool cullable = ob->epsilon < 1.0f && ob->clearance <= 1.0f;
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                const float ux = __builtin_fmaf(ob->pose_inv[2], cz[l], __builtin_fmaf(ob->pose_inv[1], cy[l], __builtin_fmaf(ob->pose_inv[0], cx[l], ob->pose_inv[3]))) - ob->lo[0];
                const float uy = __builtin_fmaf(ob->pose_inv[6], cz[l], __builtin_fmaf(ob->pose_inv[5], cy[l], __builtin_fmaf(ob->pose_inv[4], cx[l], ob->pose_inv[7]))) - ob->lo[1];
                const float uz = __builtin_fmaf(ob->pose_inv[10], cz[l], __builtin_fmaf(ob->pose_inv[9], cy[l], __builtin_fmaf(ob->pose_inv[8], cx[l], ob->pose_inv[11]))) - ob->lo[2];
                const bool near = rbox_near(ux, uy, uz, rad[l], rbc, rbh, ob->rb_r);
                if (near || !cullable) m[l] |= bit;
            }
        }
        if (st && cfg > 0) {
#pragma unroll
            for (int l = 0; l < 10; ++l) mw[l * CH + cfg - 1] = m[l];
        }
    }
}
