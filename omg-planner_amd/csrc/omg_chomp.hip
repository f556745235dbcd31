// omg_chomp.hip — k_chomp_optimize: one Optimizer.optimize step per trajectory, one workgroup each.
//
// Replaces (all float64, like the reference's numpy code):
//   Cost.forward_kinematics_obstacle   omg/cost.py:112-190   FK + joint frames, x / v / a by finite differences
//   Cost.compute_point_jacobian        omg/cost.py:92-110
//   Cost.functional_grad               omg/cost.py:24-43
//   Cost.compute_collision_loss        omg/cost.py:362-423   top-k "last write wins" branch AND the clean branch
//   Cost.compute_smooth_loss           omg/cost.py:425-449
//   Cost.compute_total_loss            omg/cost.py:451-532
//   Optimizer.check_joint_limit        omg/optimizer.py:166-174
//   Optimizer.goal_set_projection      omg/optimizer.py:88-113
//   Trajectory.update                  omg/core.py:43-51
//   Optimizer.handle_joint_limit       omg/optimizer.py:148-164
//
// The SDF potentials/gradients of the n waypoint configurations come from omgx_fk_sdf (float32,
// [n][10][P] reference layout).  Everything else lives in LDS for the duration of the step:
//   link poses of start, n waypoints, end (double [n+2][10][12]; joint axes/origins are re-derived from them on
//   use: axis = R_link ax', origin = R_link og' + t_link), per-(waypoint,link) gradient slots ([n][10][8]), the
//   trajectory and its gradient ([n][9]).  ~147 KB at the 64-waypoint limit.
//
// A = D^T D is tridiagonal (-1,2,-1)/dt^2 with last diagonal 1/dt^2 (goal-set, free end) or 2/dt^2
// (fixed end) — omg/config.py:208-220, util.py:165-178 — so A^-1 has the closed forms
//   free end : Ainv[i][k] = dt^2 (min(i,k)+1)
//   fixed end: Ainv[i][k] = dt^2 (min(i,k)+1)(n-max(i,k))/(n+1)
// and the goal-set projector M = Ainv C^T (C Ainv C^T)^-1 (optimizer.py:107) with C = [0 I_c] is
//   M[i][0] = (i+1)/(n-c+1) for i < n-c,  M[n-c+q][q] = 1, zero elsewhere
// (inverse of a min(x_r,x_q) kernel is tridiagonal).  No matrix is stored or inverted on device.
#include <hip/hip_runtime.h>
#include <atomic>

#include <stdint.h>

// chomp_scene and what it is made of (this translation unit holds float64 code only: the header lets the compiler fuse multiply-adds)
#include "omg_chomp_body.h"

// MI: see chomp_scene — MI_SMALL serves up to 32 waypoints (every configuration of BASELINE.json but the 50-waypoint one) with half the
// unrolled prefetch / count loops and 20 fewer VGPRs, MI_FULL up to OMGX_MAX_WAYPOINTS; same results.
#define MI_SMALL 10
#define MI_FULL ((OMGX_MAX_WAYPOINTS * 160 + CH_TPB - 1) / CH_TPB)
template <int MI>
__global__ __launch_bounds__(CH_TPB) void k_chomp_optimize(ChompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    chomp_scene<MI>(a, smem, blockIdx.x);
}

// The same pair with the goal update in its own workgroup: workgroups [0, S) run the learner of scene b and publish
// (ticket << 8) | chosen goal in goal_flags[b]; workgroups [S, 2S) run the optimiser step of scene b - S, whose goal-independent two thirds
// (FK, top-k, per-point costs, winners' gradients) overlap the learner; they wait for the flag only before the part that
// uses the goal.  Learner workgroups come first in the grid, so a waiting workgroup's producer has always been
// dispatched already (no deadlock); the wait is bounded anyway.
template <int MI>
__global__ __launch_bounds__(CH_TPB) void k_update_optimize_split(omg_learner::LearnerArgs la, ChompArgs a, uint32_t* goal_flags,
                                                                  uint32_t ticket, uint32_t publish /* == ticket (test hook: see omgx_debug_drop_ticket) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef OMGX_UPD_PRIO
    __builtin_amdgcn_s_setprio(OMGX_UPD_PRIO);  // experiment build (DESIGN_HISTORY appendix A): the update's waves ahead of the goal workgroups that share their CU
#endif
    const int S = la.S;
    if ((int)blockIdx.x < S) {
#ifdef OMGX_PHASE_TIMING
        if (blockIdx.x == 0 && threadIdx.x == 0) g_chomp_phase[26] = __builtin_readcyclecounter();
#endif
        double* shl = reinterpret_cast<double*>(smem);
        int* const sh_idx = reinterpret_cast<int*>(shl + 5 * OMGX_MAX_GOALS + 5 * 128);  // the chosen goal, for the whole workgroup
        // learner_scene publishes (ticket << 8) | goal index in the scene's word the moment the index is known: the step's workgroup
        // needs nothing else from here (it reads the goal's configuration, rows and poses from the tables itself: chomp_scene)
        omg_learner::learner_scene(la, blockIdx.x, reinterpret_cast<double (*)[OMGX_MAX_GOALS]>(shl),
                                   reinterpret_cast<double (*)[128]>(shl + 5 * OMGX_MAX_GOALS), sh_idx, goal_flags + blockIdx.x, publish);
        // ... what the launches to come read: `end_poses_out` follows the goal (k_chomp_optimize takes its end pose from it)
        const bool scene_on = !(a.active && a.active[blockIdx.x] == 0);
        if (scene_on && la.prm.goal_pose_table && la.prm.end_poses_out) {
            __syncthreads();
            const double* src = la.prm.goal_pose_table + ((size_t)blockIdx.x * la.prm.num_goals + *sh_idx) * 120;
            if (threadIdx.x < 120) la.prm.end_poses_out[(size_t)blockIdx.x * 120 + threadIdx.x] = src[threadIdx.x];
        }
#ifdef OMGX_PHASE_TIMING
        if (blockIdx.x == 0 && threadIdx.x == 0) g_chomp_phase[27] = __builtin_readcyclecounter();
#endif
        return;
    }
    const int s = (int)blockIdx.x - S;
    chomp_scene<MI>(a, smem, s, goal_flags + s, ticket, &la);
}

// Learner.update_goal followed by Optimizer.optimize for the same scene in one workgroup (planner.py:612-621 calls them
// back to back): one launch and no stream round trip between the goal choice and the step that uses it.  The
// learner's LDS (5 x 256 + 5 x 128 doubles) borrows the front of the dynamic region, which chomp_scene initialises
// itself after the barrier.
template <int MI>
__global__ __launch_bounds__(CH_TPB) void k_update_optimize(omg_learner::LearnerArgs la, ChompArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* shl = reinterpret_cast<double*>(smem);
    omg_learner::learner_scene(la, blockIdx.x, reinterpret_cast<double (*)[OMGX_MAX_GOALS]>(shl),
                               reinterpret_cast<double (*)[128]>(shl + 5 * OMGX_MAX_GOALS));
    __syncthreads();  // the goal written by wave 0 (global memory) is visible to the whole workgroup
    if (la.prm.goal_pose_table && la.prm.end_poses_out && !(la.active && la.active[blockIdx.x] == 0)) {  // keep the end pose current
        const double* src = la.prm.goal_pose_table + ((size_t)blockIdx.x * la.prm.num_goals + la.goal_idx[blockIdx.x]) * 120;
        if (threadIdx.x < 120) la.prm.end_poses_out[(size_t)blockIdx.x * 120 + threadIdx.x] = src[threadIdx.x];
        __syncthreads();
    }
    chomp_scene<MI>(a, smem, blockIdx.x);
}

#ifdef OMGX_PHASE_TIMING
extern "C" int omgx_debug_learner_phase_times(unsigned long long* h_out, int n) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(omg_learner::g_learner_phase), sizeof(unsigned long long) * (n < 16 ? n : 16));
}
extern "C" int omgx_debug_chomp_phase_times(unsigned long long* h_out, int n) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_chomp_phase), sizeof(unsigned long long) * (n < 48 ? n : 48));
}
#endif

static bool fits_small(int n) { return n * 160 <= MI_SMALL * CH_TPB; }
static size_t host_lds_bytes(int n, int P) {
    size_t d = (size_t)(n + 2) * 120 + 60 + (size_t)n * 80 + (size_t)n * 10 + (size_t)n * 9 * 6 + (n + 1) + 30 * P + 64 + 246;
    size_t i = (size_t)n * 10 + 512 + (n * 160 + 31) / 32 + 16 + (size_t)n * 10;
    return d * 8 + i * 4;
}
static const size_t kLdsLimit = 160 * 1024;  // gfx950: 160 KB per workgroup
static bool host_pot_in_lds(int n, int P) { return host_lds_bytes(n, P) + (size_t)n * 160 * 4 <= kLdsLimit; }

extern "C" int64_t omgx_chomp_aux_doubles(int32_t n) { return n < 1 ? 0 : (int64_t)n * 9 + (int64_t)n * 10 + (int64_t)n * 9 + n + 1; }

// Argument checks + ChompArgs of omgx_chomp_optimize (shared with omgx_goal_update_optimize).
static int chomp_make_args(const double* robot, const omgx_chomp_params* h_params, double* traj, const double* start,
                           const double* end, const double* goal, const double* goal_point, const float* potentials,
                           const float* grads, const float* collides, const int32_t* active, int32_t num_scenes, double* grad,
                           double* cost_traj, double* info, double* aux, ChompArgs& a, size_t& lds) {
    if (!h_params || num_scenes < 0) return OMGX_ERR_INVALID;
    if (!robot || !traj || !start || !end || !goal || !goal_point || !potentials || !grads || !collides || !grad ||
        !cost_traj || !info)
        return OMGX_ERR_INVALID;
    const omgx_chomp_params& p = *h_params;
    if (p.n_waypoints < 1 || p.n_waypoints > OMGX_MAX_WAYPOINTS || p.n_points < 1 || p.n_points > OMGX_MAX_POINTS ||
        p.constraint_num < 1 || p.constraint_num > OMGX_MAX_CONSTRAINTS || p.constraint_num > p.n_waypoints)
        return OMGX_ERR_UNSUPPORTED;
    if (!(p.time_interval > 0.0) || p.top_k < 0 || p.do_update < 0 || p.do_update > 2) return OMGX_ERR_INVALID;
    a = ChompArgs{};
    a.robot = robot; a.prm = p; a.traj = traj; a.start = start; a.end = end; a.goal = goal; a.goal_point = goal_point;
    a.pot = potentials; a.pgrad = grads; a.col = collides; a.active = active; a.grad = grad; a.cost_traj = cost_traj;
    a.info = info; a.aux = aux;
    a.pot_in_lds = host_pot_in_lds(p.n_waypoints, p.n_points) ? 1 : 0;
    lds = host_lds_bytes(p.n_waypoints, p.n_points) + (a.pot_in_lds ? (size_t)p.n_waypoints * 160 * 4 : 0);
    return OMGX_OK;
}

extern "C" int omgx_chomp_optimize(const double* robot, const omgx_chomp_params* h_params, double* traj,
                                   const double* start, const double* end, const double* goal,
                                   const double* goal_point, const float* potentials, const float* grads,
                                   const float* collides, int32_t* active, int32_t num_scenes, double* grad,
                                   double* cost_traj, double* info, double* aux, int32_t stop_on_terminate, void* stream) {
    if (h_params && num_scenes == 0) return OMGX_OK;
    ChompArgs a;
    size_t lds = 0;
    int rc = chomp_make_args(robot, h_params, traj, start, end, goal, goal_point, potentials, grads, collides, active, num_scenes,
                             grad, cost_traj, info, aux, a, lds);
    if (rc != OMGX_OK) return rc;
    if (stop_on_terminate) {
        if (!active) return OMGX_ERR_INVALID;
        a.deactivate = active;
    }
    if (fits_small(a.prm.n_waypoints)) {
        if ((rc = allow_big_lds<0>(k_chomp_optimize<MI_SMALL>, "hipFuncSetAttribute(k_chomp_optimize)")) != OMGX_OK) return rc;
        hipLaunchKernelGGL(k_chomp_optimize<MI_SMALL>, dim3(num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, a);
    } else {
        if ((rc = allow_big_lds<1>(k_chomp_optimize<MI_FULL>, "hipFuncSetAttribute(k_chomp_optimize)")) != OMGX_OK) return rc;
        hipLaunchKernelGGL(k_chomp_optimize<MI_FULL>, dim3(num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, a);
    }
    OMGX_CHECK_LAUNCH("k_chomp_optimize");
    return OMGX_OK;
}

// Test hook for the bounded wait of k_update_optimize_split: while on, the learner workgroups publish a WRONG ticket, so every step
// workgroup runs into its 2 s bound and reports a NaN cost instead of hanging the device (tests/test_gpu_round3.py).  Not part of the ABI.
static std::atomic<int> g_drop_ticket{0};
extern "C" int omgx_debug_drop_ticket(int32_t on) { g_drop_ticket.store(on ? 1 : 0, std::memory_order_relaxed); return OMGX_OK; }

extern "C" int omgx_goal_update_optimize(const omgx_learner_params* h_learner, const double* goal_set, const double* reach,
                                         const float* goal_cost, double* learner_state, int32_t* goal_idx, double* cost_vector,
                                         const double* robot, const omgx_chomp_params* h_params, double* traj,
                                         const double* start, double* end, double* goal, double* goal_point,
                                         const float* potentials, const float* grads, const float* collides,
                                         int32_t* active, int32_t num_scenes, double* grad, double* cost_traj, double* info,
                                         double* aux, int32_t* scene_flags, int32_t ticket, int32_t stop_on_terminate, const int32_t* goal_count,
                                         const double* eta, void* stream) {
    if (h_learner && h_params && num_scenes == 0) return OMGX_OK;
    omg_learner::LearnerArgs la;
    int rc = omg_learner::make_args(h_learner, traj, goal_set, reach, goal_cost, learner_state, num_scenes, goal_idx, end, goal,
                                    goal_point, cost_vector, active, goal_count, eta, la);
    if (rc != OMGX_OK) return rc;
    ChompArgs a;
    size_t lds = 0;
    rc = chomp_make_args(robot, h_params, traj, start, end, goal, goal_point, potentials, grads, collides, active, num_scenes, grad,
                         cost_traj, info, aux, a, lds);
    if (rc != OMGX_OK) return rc;
    if (h_learner->n_waypoints != h_params->n_waypoints || h_learner->constraint_num != h_params->constraint_num) return OMGX_ERR_INVALID;
    if (stop_on_terminate) {
        if (!active) return OMGX_ERR_INVALID;
        a.deactivate = active;
    }
    const size_t learner_lds = (size_t)(5 * OMGX_MAX_GOALS + 5 * 128 + 2) * sizeof(double);  // + the chosen goal's index (k_update_optimize_split)
    if (lds < learner_lds) lds = learner_lds;
    const bool small = fits_small(a.prm.n_waypoints);
    if (scene_flags) {
        if (ticket < 1 || ticket >= (1 << 24)) return OMGX_ERR_INVALID;  // the word is (ticket << 8) | goal index
        const uint32_t publish = g_drop_ticket.load(std::memory_order_relaxed) ? (uint32_t)ticket ^ 0x400000u : (uint32_t)ticket;
        if (small) {
            if ((rc = allow_big_lds<2>(k_update_optimize_split<MI_SMALL>, "hipFuncSetAttribute(k_update_optimize_split)")) != OMGX_OK) return rc;
            hipLaunchKernelGGL(k_update_optimize_split<MI_SMALL>, dim3(2 * num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, la, a,
                               reinterpret_cast<uint32_t*>(scene_flags), (uint32_t)ticket, publish);
        } else {
            if ((rc = allow_big_lds<3>(k_update_optimize_split<MI_FULL>, "hipFuncSetAttribute(k_update_optimize_split)")) != OMGX_OK) return rc;
            hipLaunchKernelGGL(k_update_optimize_split<MI_FULL>, dim3(2 * num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, la, a,
                               reinterpret_cast<uint32_t*>(scene_flags), (uint32_t)ticket, publish);
        }
        OMGX_CHECK_LAUNCH("k_update_optimize_split");
        return OMGX_OK;
    }
    if (small) {
        if ((rc = allow_big_lds<4>(k_update_optimize<MI_SMALL>, "hipFuncSetAttribute(k_update_optimize)")) != OMGX_OK) return rc;
        hipLaunchKernelGGL(k_update_optimize<MI_SMALL>, dim3(num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, la, a);
    } else {
        if ((rc = allow_big_lds<5>(k_update_optimize<MI_FULL>, "hipFuncSetAttribute(k_update_optimize)")) != OMGX_OK) return rc;
        hipLaunchKernelGGL(k_update_optimize<MI_FULL>, dim3(num_scenes), dim3(CH_TPB), lds, (hipStream_t)stream, la, a);
    }
    OMGX_CHECK_LAUNCH("k_update_optimize");
    return OMGX_OK;
}
