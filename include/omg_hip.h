/*
 * omg_hip.h — C ABI of libomg_hip.so, the MI355X (gfx950) CHOMP trajectory-update engine.
 *
 * Drop-in boundary for the hot path of liruiw/OMG-Planner.  Every entry point
 *   - takes plain device pointers + sizes (no torch types),
 *   - is asynchronous on the hipStream_t passed as `stream` (void*, NULL = default stream),
 *   - returns OMGX_OK or a negative error code and NEVER exits the process
 *     (the reference does `exit(-1)` on a launch error: layers/sdf_matching_loss_kernel.cu:241-246).
 *
 * All pointers are DEVICE pointers unless the name starts with `h_`.
 * Reference citations are file:line into liruiw/OMG-Planner.
 */
#ifndef OMG_HIP_H
#define OMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Error codes
 * ------------------------------------------------------------------------------------------- */
#define OMGX_OK 0
#define OMGX_ERR_INVALID (-1)     /* null pointer / bad size / unsupported combination of sizes */
#define OMGX_ERR_LAUNCH (-2)      /* HIP reported a launch error; text via omgx_last_error()      */
#define OMGX_ERR_UNSUPPORTED (-3) /* a size beyond what this build supports (see limits below)    */

/* Compile-time limits of this build. */
#define OMGX_NUM_LINKS 10        /* Panda: 7 arm links + hand + 2 fingers (omg/core.py:166-190)   */
#define OMGX_NUM_DOF 9           /* 7 arm joints + 2 finger joints                                 */
#define OMGX_MAX_POINTS 16       /* collision points per link (cfg.collision_point_num = 15)       */
#define OMGX_MAX_WAYPOINTS 64    /* cfg.timesteps (30 default, 50 in the PyBullet drivers)         */
#define OMGX_MAX_CONSTRAINTS 8   /* cfg.reach_tail_length (5) or 1; must be <= n_waypoints          */
#define OMGX_INFO_STRIDE 16      /* doubles per trajectory in the info record                      */

/* ---------------------------------------------------------------------------------------------
 * Robot constants blob (double, device).  Raw tables of ycb_render/robotPose/robot_p3.pkl in the
 * order used by robot_pykdl.py:148-215, followed by the sampled collision points and joint limits.
 *   [  0,160)  pose_0        [10][4][4]
 *   [160,320)  tip2joint     [10][4][4]
 *   [320,480)  center_offset [10][4][4]
 *   [480,510)  joint_axis    [10][3]
 *   [510,519)  joint_lower_limit [9]   (omg/core.py:157-164, already padded by soft_joint_limit_padding)
 *   [519,528)  joint_upper_limit [9]
 *   [528,528+30P)  collision_points [10][P][3]  (Robot.collision_points, omg/core.py:166-190)
 * followed, at D = 528+30P, by constants DERIVED from the tables above (they only re-associate the FK
 * products so the device does 3x3 work instead of 4x4; PandaModel.blob() in omg-planner_amd/robot.py
 * is the reference implementation of the derivation):
 *   D+0    UVW [7][3][3][3]  rotation of pose_0[i].Rz(q).Rx(off_i).N_i  =  cos(q) U_i + sin(q) V_i + W_i,
 *                            off = (0,-pi,pi,pi,-pi,pi,pi), N_0 = I, N_i = diag(1,-1,-1) (robot_pykdl.py:167-176)
 *   D+189  TP  [7][3]        pose_0[i][:3,3]
 *   D+210  H, LF, RF [3][12] rows of pose_0[7], pose_0[8], pose_0[9]
 *   D+246  PTS [10][P][3]    center_offset[l] applied to collision_points[l][p]
 *   D+246+30P AX [10][3]     tip2joint[l][:3,:3] . joint_axis[l]
 *   D+276+30P OG [10][3]     tip2joint[l][:3,3]
 *   D+306+30P RAD [10]       max_p |PTS[l][p]|: bounding-sphere radius of a link's points about its frame origin
 *   D+316+30P BALL [10][4]   (c_x, c_y, c_z, r): a ball around PTS[l] itself — centre c in the link frame, r >= max_p |PTS[l][p] - c|
 *                            (ABI 10).  The row-level culling tests it (centre R c + t per configuration) instead of (t, RAD).
 * Total length 528 + 60P + 356 doubles.
 * ------------------------------------------------------------------------------------------- */
#define OMGX_ROBOT_POSE0 0
#define OMGX_ROBOT_TIP2JOINT 160
#define OMGX_ROBOT_CENTER_OFFSET 320
#define OMGX_ROBOT_JOINT_AXIS 480
#define OMGX_ROBOT_LOWER 510
#define OMGX_ROBOT_UPPER 519
#define OMGX_ROBOT_POINTS 528

/* ---------------------------------------------------------------------------------------------
 * Scene object table.  One 184-byte record per obstacle/target object; replaces the five per-call
 * host->device copies of Cost.compute_obstacle_cost_layer (omg/cost.py:303-335) and the
 * pad-to-max `sdf_torch[O,X,Y,Z]` + `sdf_limits[O,10]` contract of Env.combine_sdfs
 * (omg/core.py:366-411).  `grid_offset` lets grids live ragged in one float pool; the reference's
 * padded layout is the special case grid_offset = o*X*Y*Z, dim = (X,Y,Z), hi = stretched max.
 * ------------------------------------------------------------------------------------------- */
typedef struct omgx_object {
    float pose_inv[12];  /* rows 0..2 of se3_inverse(obj.pose_mat), row-major [3][4] (omg/util.py:129-135) */
    float lo[3];         /* sdf_limits[0:3]  min coords                                              */
    float hi[3];         /* sdf_limits[3:6]  (stretched) max coords                                  */
    int32_t dim[3];      /* sdf_limits[6:9]  grid dims, x-major storage x*dy*dz + y*dz + z           */
    float delta;         /* sdf_limits[9]    voxel size                                              */
    float epsilon;       /* cfg.epsilon / cfg.target_epsilon                                         */
    float padding_scale; /* 1, or 0.5 for the table when the target is attached                     */
    float clearance;     /* cfg.clearance / cfg.target_clearance                                     */
    int32_t disabled;    /* 1 = skip (name == "floor" or in cfg.disable_collision_set)               */
    int64_t grid_offset; /* element offset of this object's grid inside the sdf pool                 */
    double inv_extent[3]; /* derived: 1.0 / (double)((float)hi[a] - (float)lo[a])                     */
    float rb_c[3];       /* derived: the INFLUENCE REGION of the object, a rounded box in offset coordinates t = R p + t - lo:     */
    float rb_h[3];       /*   sum_k max(|t_k - rb_c[k]| - rb_h[k], 0)^2 <= rb_r2.  Outside it a lookup adds nothing (value > epsilon */
    float rb_r;          /*   and >= clearance), so the kernels skip the pair.  Default (scenes.finish_records): the grid as a plain  */
    float rb_r2;         /*   box, [-1.5 voxels, extent + 1.5 voxels], rb_r = 0; scenes.tighten_far_boxes() fits it to the lookups   */
                         /*   that can matter (a ball around a sphere-like object, ...); rb_r2 < 0: nothing can, every pair is skipped */
    double inv_delta;    /* derived: 1.0 / (double)delta                                              */
    float inv_2eps;      /* derived: 1.0f / (2.0f * epsilon)   (float32 arithmetic, .cu:167)           */
    float inv_eps;       /* derived: 1.0f / epsilon            (float32 arithmetic, .cu:168)           */
} omgx_object; /* sizeof == 184; derived fields: scenes.finish_records() is the reference derivation */

/* ---------------------------------------------------------------------------------------------
 * CHOMP parameters for one optimiser step (a frozen snapshot of the reference's global mutable
 * `cfg`, omg/config.py:30-131, after Optimizer.update() has applied the schedules,
 * omg/optimizer.py:59-80).
 * ------------------------------------------------------------------------------------------- */
typedef struct omgx_chomp_params {
    int32_t n_waypoints;              /* cfg.timesteps                                              */
    int32_t n_points;                 /* P = cfg.collision_point_num                                */
    int32_t top_k;                    /* cfg.top_k_collision; 0 = clean sum branch (cost.py:380-388)*/
    int32_t consider_finger;          /* cfg.consider_finger                                        */
    int32_t goal_set_proj;            /* cfg.goal_set_proj                                          */
    int32_t constraint_num;           /* reach_tail_length if cfg.use_standoff else 1               */
    int32_t use_standoff;             /* cfg.use_standoff (only for info["standoff_idx"])           */
    int32_t uncheck_finger_collision; /* cfg.uncheck_finger_collision (-1 softens fingers)          */
    int32_t joint_limit_max_steps;    /* cfg.joint_limit_max_steps                                  */
    int32_t allow_collision_point;    /* cfg.allow_collision_point                                  */
    int32_t pre_terminate;            /* cfg.pre_terminate                                          */
    int32_t do_update;                /* 0 = info_only; 1 = always update (force_update=True);
                                         2 = update unless info["terminate"] (optimizer.py:124-125)   */
    double time_interval;             /* cfg.time_interval                                          */
    double obstacle_weight;           /* cfg.obstacle_weight                                        */
    double smoothness_weight;         /* cfg.smoothness_weight                                      */
    double step_size;                 /* cfg.step_size                                              */
    double clip_grad_scale;           /* cfg.clip_grad_scale                                        */
    double terminate_smooth_loss;     /* cfg.terminate_smooth_loss                                  */
    double link_smooth_weight[OMGX_NUM_DOF]; /* cfg.link_smooth_weight                              */
    /* Optional link poses the step would otherwise compute itself (ABI 7; NULL = compute).  Pose layout: omgx_pose_table.  The
     * caller vouches that they belong to the configurations the step sees (same kinematics code: same bits, nothing else changes). */
    const double* waypoint_poses; /* [S][n][10][12] device: poses of the CURRENT trajectory, as omgx_goalset_cost_layer_tiled's
                                     layer workgroups leave them in `layer_poses`                                              */
    const double* start_poses;    /* [S][10][12] device: omgx_pose_table(start)                                                */
    const double* end_poses;      /* [S][10][12] device: poses of `end` (the goal); omgx_goal_update_optimize's learner keeps it
                                     current when omgx_learner_params.end_poses_out points at the same buffer                  */
} omgx_chomp_params;

/* info record layout (doubles), one per trajectory: the numeric keys of Cost.compute_total_loss's
 * `info` dict (omg/cost.py:509-530) + Optimizer.check_joint_limit (omg/optimizer.py:166-174). */
#define OMGX_INFO_COST 0
#define OMGX_INFO_OBS 1
#define OMGX_INFO_SMOOTH 2
#define OMGX_INFO_WEIGHTED_OBS 3
#define OMGX_INFO_WEIGHTED_SMOOTH 4
#define OMGX_INFO_WEIGHTED_OBS_GRAD 5
#define OMGX_INFO_WEIGHTED_SMOOTH_GRAD 6
#define OMGX_INFO_GRAD 7
#define OMGX_INFO_COLLIDE 8
#define OMGX_INFO_REACH 9
#define OMGX_INFO_TERMINATE 10
#define OMGX_INFO_FAILURE_TERMINATE 11
#define OMGX_INFO_EXECUTE 12
#define OMGX_INFO_STANDOFF_IDX 13
#define OMGX_INFO_VIOLATE_LIMIT 14
#define OMGX_INFO_LIMIT_STEPS 15 /* build-only: joint-limit projection iterations actually used     */

/* ---------------------------------------------------------------------------------------------
 * (1) omgx_sdf_loss_forward
 * Replaces  omg_cuda.sdf_loss_forward  — layers/omg_layers.cpp:24-49 (binding),
 *           sdf_loss_cuda_forward       — layers/sdf_matching_loss_kernel.cu:204-262 (4 launches + 2 syncs),
 *           SDFdistanceForward / sum_gradients — .cu:96-181 / 185-195.
 * Same eight inputs in the same order, float32 contiguous; outputs are the three reduced tensors
 * the binding returns: potentials[N], potential_grads[N,3], collides[N].  One fused launch, the
 * [N,O,*] intermediates and the atomicAdd reduction do not exist; objects are summed in index order.
 * ------------------------------------------------------------------------------------------- */
int omgx_sdf_loss_forward(const float* pose_init,       /* [O,4,4] inverse object poses            */
                          const float* sdf_grids,       /* [O,X,Y,Z]                               */
                          const float* sdf_limits,      /* [O,10]                                  */
                          const float* points,          /* [N,3]                                   */
                          const float* epsilons,        /* [O]                                     */
                          const float* padding_scales,  /* [O]                                     */
                          const float* clearances,      /* [O]                                     */
                          const float* disables,        /* [O]                                     */
                          int64_t num_points, int32_t num_objects,
                          float* potentials,            /* [N]   out                               */
                          float* potential_grads,       /* [N,3] out                               */
                          float* collides,              /* [N]   out                               */
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * (2) omgx_fk_sdf
 * Replaces the FK -> points -> SDF-layer chain of Cost.batch_obstacle_cost (omg/cost.py:192-232,
 * arc_length <= 0) and of Cost.forward_kinematics_obstacle (omg/cost.py:124-143), i.e.
 * robot_kinematics.forward_kinematics_parallel (robot_pykdl.py:148-215) + Cost.forward_points
 * (cost.py:60-72) + Cost.compute_obstacle_cost_layer (cost.py:288-360), for S scenes x C robot
 * configurations per scene, entirely on device.
 *   joints      [S,C,9]  double, radians (9-dof; wrap_values' degree round trip is applied inside)
 *   objects     table of omgx_object; scene s owns objects [scene_begin[s], scene_begin[s+1])
 *   soften_fingers != 0  <=> uncheck_finger_collision == -1 (cost.py:350-353)
 *   arc_length > 0: Cost.batch_obstacle_cost's arc-length branch (cost.py:235-275): the C configurations
 *     of a scene are groups of `arc_length` consecutive waypoints; potentials are multiplied by
 *     ||(x_i - x_{i-1}) / time_interval|| (float32), the first waypoint of every group differencing
 *     against FK(arc_start[s]) (arc_start [S,9]).  C must be a multiple of arc_length <= 64.
 * Outputs (float32): potentials [S,C,10,P], grads [S,C,10,P,3], collides [S,C,10,P]; any may be NULL.
 *   workspace   device scratch of omgx_fk_sdf_workspace_bytes(S, C, P) bytes (FK-produced link poses)
 * ------------------------------------------------------------------------------------------- */
int64_t omgx_fk_sdf_workspace_bytes(int32_t num_scenes, int32_t configs_per_scene, int32_t n_points);
int omgx_fk_sdf(const double* robot, int32_t n_points,
                const omgx_object* objects, const int32_t* scene_begin, const float* sdf_pool,
                const double* joints, int32_t num_scenes, int32_t configs_per_scene,
                int32_t soften_fingers, int32_t arc_length, const double* arc_start, double time_interval,
                float* potentials, float* grads, float* collides, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (2b) omgx_forward_kinematics
 * Replaces robot_kinematics.forward_kinematics_parallel(..., return_joint_info=True)
 * (ycb_render/robotPose/robot_pykdl.py:148-215) as used by Cost.forward_poses (omg/cost.py:45-58).
 *   joints [B,9] double radians  ->  link_poses [B,10,4,4] (center_offset applied), joint_origins [B,10,3],
 *   joint_axes [B,10,3] (double; origins/axes may be NULL).  joint_origins reproduces the reference's
 *   `_joint_origin := _joint_axis` quirk (robot_pykdl.py:104): origin = R axis_local + t.
 * ------------------------------------------------------------------------------------------- */
int omgx_forward_kinematics(const double* robot, int32_t n_points, const double* joints, int64_t num_configs,
                            double* link_poses, double* joint_origins, double* joint_axes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (2c) omgx_pose_table (ABI 7)
 * Link poses of N configurations in the layout the optimiser step keeps them in: per configuration 10 links x 12 doubles —
 * rotation row-major [9], translation [3], BEFORE center_offset (robot_pykdl output_pose; omgx_forward_kinematics applies it).
 * Same kinematics code as inside the step / learner kernels, so a tabulated pose carries the bits they would compute:
 * omgx_chomp_params.start_poses / end_poses, omgx_learner_params.goal_pose_table.
 * ------------------------------------------------------------------------------------------- */
int omgx_pose_table(const double* robot, int32_t n_points, const double* configs, int64_t num_configs, double* poses, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (3) omgx_goalset_cost
 * Replaces Learner.cost_vector's device work (omg/online_learner.py:104-148):
 * multi_interpolate_waypoints(..., "linear") (omg/util.py:261-290) -> Cost.batch_obstacle_cost with
 * arc_length = n_remaining (omg/cost.py:192-286, incl. get_derivative_torch, config.py:162-187)
 * -> sum over (link, point) and over the n_remaining waypoints, for S scenes x G goals.
 *   traj_start row s at traj_start + s * traj_start_stride doubles: the waypoint the interpolation starts from
 *              (traj.data[start]); stride 9 for a dense [S,9] array, n*9 to point into a [S,n,9] trajectory tensor
 *   goals      [S,G,9]
 *   goal_cost  [S,G]  float32 out: sum_i sum_link sum_pt potential * ||velocity||
 *   potentials [S,G,n_remaining,10,P] float32 out, optional (NULL to skip) — the weighted
 *              potentials batch_obstacle_cost returns
 *   collides   [S,G] float32 out, optional: number of (config, link, point, object) collisions
 *   workspace  NULL: every goal workgroup runs the kinematics of its own configurations (one launch).  Non-NULL (ABI 10), device
 *              scratch of omgx_goalset_workspace_bytes(S, G, n_remaining, P) bytes: the KINEMATICS PRE-PASS — the goals' link
 *              poses and row masks are computed by a launch of their own (k_goalset_kin: one lane per (goal, configuration),
 *              no LDS, no barrier) into the workspace ([S*G][10][9][n_remaining+1] doubles, then [S*G][10][n_remaining] uint32),
 *              and the goal workgroups start from there with one trip to memory.  Same arithmetic, same bits; the workspace
 *              must not be shared by launches that may run at once (different streams).  Ignored with `potentials`.
 *   active, goal_count  optional [S] int32 (ABI 4), as in omgx_goalset_cost_layer below: scenes with active[s] == 0 and goals
 *              >= goal_count[s] are neither read nor written.  Only without `potentials` (else OMGX_ERR_UNSUPPORTED).
 * ------------------------------------------------------------------------------------------- */
int64_t omgx_goalset_workspace_bytes(int32_t num_scenes, int32_t num_goals, int32_t n_remaining, int32_t n_points);
int omgx_goalset_cost(const double* robot, int32_t n_points,
                      const omgx_object* objects, const int32_t* scene_begin, const float* sdf_pool,
                      const double* traj_start, int64_t traj_start_stride, const double* goals,
                      int32_t num_scenes, int32_t num_goals, int32_t n_remaining,
                      double time_interval, int32_t soften_fingers,
                      float* goal_cost, float* potentials, float* collides, void* workspace,
                      const int32_t* active, const int32_t* goal_count, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (3b) omgx_goalset_cost_layer
 * omgx_goalset_cost (cost-only: no per-point potentials) and, in the SAME launch, omgx_fk_sdf of the current
 * trajectories traj [S,n_waypoints,9] (arc_length off): one extra workgroup per scene writes
 *   layer_potentials [S,n,10,P], layer_grads [S,n,10,P,3], layer_collides [S,n,10,P] float32
 * — the inputs of omgx_chomp_optimize.  One planner iteration (omg/planner.py:612-621) is then two launches on one
 * stream: this one and omgx_goal_update_optimize.  Results are identical to the two separate calls
 * (goal_cost: same float32 summation order; layer outputs: bit-identical).
 *   active  optional [S] int32 (NULL = all): scenes with active[s] == 0 are skipped — the reference leaves a scene's
 *           loop once it terminates (omg/planner.py:626) — and their outputs keep their previous contents.
 *   goal_count  optional [S] int32 (NULL = num_goals everywhere): scene s has only goal_count[s] goals, the rest of its
 *           rows in `goals` / `goal_cost` is padding that is neither read nor written (ragged goal sets in one batch).
 *   schedule  optional [schedule_len] int32, schedule_len a multiple of 8 (NULL = scene-major order, scene s on XCD s % 8):
 *           the launch has schedule_len goal workgroups, workgroup k (it runs on XCD k % 8) works on goal schedule[k] % num_goals
 *           of scene schedule[k] / num_goals, or on nothing when schedule[k] < 0.  Every (scene, goal) that is to be
 *           evaluated must appear exactly once; any such list gives the same results.  Keep a scene's goals on one or two
 *           XCDs (k % 8): its SDF volumes then stay in that XCD's L2 — spread over all eight the launch takes 1.7x as long.
 *           omgx_goalset_schedule (section 7) derives an order from `work` that gives every XCD the same measured
 *           work, heaviest scenes first (ABI 4).
 *   work    optional [S*num_goals] uint32: receives how long each goal's workgroup ran, in 10 ns ticks (0 = skipped).
 * How many waves a goal's workgroup has is the library's business (four; eight for plans of 57 - 64 waypoints, where the LDS admits
 * two workgroups per CU either way — round 6): a goal's sum is exact and its tiles are drawn from one list, the results do not depend on it.
 * ------------------------------------------------------------------------------------------- */
int omgx_goalset_cost_layer(const double* robot, int32_t n_points,
                            const omgx_object* objects, const int32_t* scene_begin, const float* sdf_pool,
                            const double* traj_start, int64_t traj_start_stride, const double* goals,
                            int32_t num_scenes, int32_t num_goals, int32_t n_remaining,
                            double time_interval, int32_t soften_fingers,
                            float* goal_cost, float* collides, void* workspace,
                            const double* traj, int32_t n_waypoints, int32_t layer_soften_fingers,
                            float* layer_potentials, float* layer_grads, float* layer_collides,
                            const int32_t* active, const int32_t* goal_count,
                            const int32_t* schedule, int32_t schedule_len, uint32_t* work, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (3c) omgx_goalset_cost_layer_tiled — the same launch cut into more, smaller workgroups: LATENCY mode (ABI 6)
 * For one or a few scenes (BASELINE configs 1-2: one scene x 64 goals, omg/planner.py:600-653 one plan at a time) the batch
 * layout of (3b) — one workgroup per goal, five per trajectory layer, a scene per XCD — leaves most of the chip idle behind a
 * few long workgroups.  This entry point runs the same arithmetic per (point, object) pair with
 *   goal_parts          1, 2, 4 or 8: a goal's TILES (4 waypoints x 2 links) are dealt over NP = omgx_goalset_parts(n_remaining,
 *                       goal_parts) workgroups — the largest power of two <= goal_parts that leaves each at least 4 of the
 *                       ceil(n_remaining / 4) x 5 tiles; part p takes the tiles t with t % NP == p, so the heavy ones (last waypoints,
 *                       hand links) spread evenly — each running the kinematics of all configurations (the chain's latency does not
 *                       depend on their number).  goal_cost / collides are then [S][G][NP] PARTIAL sums and the goal's cost is
 *                       their float32 sum in part order — omgx_goal_update(_optimize) adds them when
 *                       omgx_learner_params.cost_parts = NP.  A goal's cost differs from (3b)'s by the rounding of a float32 sum
 *                       taken in another order (~1e-7 relative); everything else is bit-identical.  (Without `spread`, more than one
 *                       part runs the batch kernel with split goals: (3d).)
 *   layer_link_groups   1, 2, 5 or 10 and
 *   layer_config_block  b >= 0: the trajectory layer of a scene is computed by layer_link_groups x ceil(n_waypoints / b)
 *                       workgroups (10 / layer_link_groups links x b waypoints each; 0 = all waypoints).  Layer outputs do not
 *                       depend on the split (every element is computed on its own).
 *   spread              non-zero: the latency-mode kernel — workgroups in plain (scene, item) order over all XCDs instead of a scene
 *                       per XCD, the kinematic chain's constants staged in LDS (a workgroup alone on a cold CU otherwise pays a
 *                       scalar-cache miss per joint).
 * layer_poses  optional [S][n_waypoints][10][12] (ABI 7): the layer workgroups of link group 0 leave the waypoint configurations'
 *              link poses there (layout: omgx_pose_table) for the step that follows (omgx_chomp_params.waypoint_poses).
 * num_goals = 0 (goals, traj_start, goal_cost NULL): only the trajectory layer (what omgx_fk_sdf computes for the step);
 * traj = NULL: only the goal-set batch (what omgx_goalset_cost computes, as partial sums).
 * No dispatch schedule / work counters here (they order whole goals, a scene per XCD).
 * ------------------------------------------------------------------------------------------- */
int32_t omgx_goalset_parts(int32_t n_remaining, int32_t goal_parts);
int omgx_goalset_cost_layer_tiled(const double* robot, int32_t n_points,
                                  const omgx_object* objects, const int32_t* scene_begin, const float* sdf_pool,
                                  const double* traj_start, int64_t traj_start_stride, const double* goals,
                                  int32_t num_scenes, int32_t num_goals, int32_t n_remaining,
                                  double time_interval, int32_t soften_fingers,
                                  float* goal_cost, float* collides,
                                  const double* traj, int32_t n_waypoints, int32_t layer_soften_fingers,
                                  float* layer_potentials, float* layer_grads, float* layer_collides,
                                  const int32_t* active, const int32_t* goal_count,
                                  int32_t goal_parts, int32_t layer_link_groups, int32_t layer_config_block,
                                  int32_t spread, double* layer_poses, void* workspace /* ABI 10: as in (3), may be NULL */,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * (3d) omgx_goalset_cost_layer_parts — (3b) with a goal's tiles dealt over several workgroups of the BATCH kernel (ABI 8)
 * Mid-size batches — a launch of about half a round to a few rounds of the chip's 1280 workgroup slots: one GPU's share of
 * BASELINE config 4 on 8 GPUs (13 scenes x 128 goals), 25 x 64 — are bound by the latency of one goal workgroup, not by the
 * chip's capacity.  goal_parts (1, 2, 4, 8) deals a goal's tiles over NP = omgx_goalset_parts(n_remaining, goal_parts)
 * workgroups exactly as (3c) does — part p takes the tiles t with t % NP == p and runs the kinematics of all configurations —
 * but keeps the batch layout: a scene's workgroups on one XCD, five workgroups per CU, the dispatch schedule.
 *   goal_cost / collides  [S][G][NP] PARTIAL sums (omgx_learner_params.cost_parts = NP adds them in part order): a goal's
 *                         cost differs from (3b)'s by the rounding of a float32 sum taken in another order; the layer outputs
 *                         are (3b)'s bit for bit.
 *   schedule / work       over the S * G * NP items (scene, goal, part): schedule[k] = (scene * G + goal) * NP + part;
 *                         omgx_goalset_schedule_parts builds it (goal_count still counts GOALS).
 *   layer_poses           optional, as in (3c).
 * goal_parts = 1 is (3b).
 * ------------------------------------------------------------------------------------------- */
int omgx_goalset_cost_layer_parts(const double* robot, int32_t n_points,
                                  const omgx_object* objects, const int32_t* scene_begin, const float* sdf_pool,
                                  const double* traj_start, int64_t traj_start_stride, const double* goals,
                                  int32_t num_scenes, int32_t num_goals, int32_t n_remaining,
                                  double time_interval, int32_t soften_fingers,
                                  float* goal_cost, float* collides,
                                  const double* traj, int32_t n_waypoints, int32_t layer_soften_fingers,
                                  float* layer_potentials, float* layer_grads, float* layer_collides,
                                  const int32_t* active, const int32_t* goal_count,
                                  const int32_t* schedule, int32_t schedule_len, uint32_t* work,
                                  int32_t goal_parts, double* layer_poses, void* workspace /* ABI 10: as in (3), may be NULL */,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * (4) omgx_chomp_optimize
 * Replaces one Optimizer.optimize step (omg/optimizer.py:115-135) for S independent trajectories:
 * Cost.compute_total_loss (omg/cost.py:451-532) = compute_smooth_loss (425-449) +
 * compute_collision_loss (362-423, both the top-k quirk branch and the clean branch) on top of
 * forward_kinematics_obstacle's Jacobians / velocities / accelerations (cost.py:112-190, 92-110,
 * 24-43), then check_joint_limit (optimizer.py:166-174), goal_set_projection (88-113) or the plain
 * -eta*Ainv*g step (132), Trajectory.update (omg/core.py:43-51) and handle_joint_limit (148-164).
 * The SDF potentials/gradients of the S*n waypoint configurations must have been produced by
 * omgx_fk_sdf on the same stream (configs_per_scene = n_waypoints).
 *   traj   [S,n,9] double, in/out (updated in place when params.do_update)
 *   start  [S,9], end [S,9] double   (traj.start, traj.end)
 *   goal   [S,c,9] double  chosen goal rows (reach_grasps[goal_idx] or goal_set[goal_idx])
 *   goal_point [S,9] double  traj.goal_set[traj.goal_idx], only for info["reach"] (cost.py:483-487)
 *   potentials [S,n,10,P], grads [S,n,10,P,3], collides [S,n,10,P] float32 from omgx_fk_sdf
 *   active [S] int32, optional: 0 = leave this trajectory untouched (terminated), NULL = all active
 * Outputs: grad [S,n,9] double (info["gradient"]), cost_traj [S,n] double, info [S,16] double.
 *   aux    optional [S, omgx_chomp_aux_doubles(n)] double: the un-weighted pieces the reference's
 *          compute_collision_loss / compute_smooth_loss return — obs_grad [n,9] | obs_cost [n,10] |
 *          smooth_grad [n,9] | smooth_loss [n+1]   (NULL to skip)
 *   stop_on_terminate  non-zero: a scene whose info says `terminate` after this step gets active[s] = 0 (see omgx_goal_update_optimize)
 * ------------------------------------------------------------------------------------------- */
int64_t omgx_chomp_aux_doubles(int32_t n_waypoints);
int omgx_chomp_optimize(const double* robot, const omgx_chomp_params* h_params,
                        double* traj, const double* start, const double* end, const double* goal,
                        const double* goal_point,
                        const float* potentials, const float* grads, const float* collides,
                        int32_t* active, int32_t num_scenes,
                        double* grad, double* cost_traj, double* info, double* aux,
                        int32_t stop_on_terminate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (5) omgx_goal_update
 * Replaces Learner.update_goal (omg/online_learner.py:237-249) for S scenes: the host arithmetic of
 * Learner.cost_vector after the obstacle batch (online_learner.py:145-160: per-goal sum is omgx_goalset_cost's
 * output; smoothness proxy ||diff(traj_start - goal_set, axis=-1)||^2 — adjacent JOINT columns, sic —,
 * weights, optional normalisation), update_goal_dist (162-235: FTL, FTC, Exp, MD = mirror descent over 5 experts
 * with the Bregman projection `bp` + `find_zero`, 16-58, or Proj) and the argmax / goal gather (243-246,
 * optimizer.py:93-99).
 *   traj        [S,n,9] double; traj_start = traj[:, start_idx]; Proj uses traj[:, n-1]
 *   goal_set    [S,G,9] double  (traj.goal_set)
 *   reach       [S,G,c,9] double or NULL (target_obj.reach_grasps when use_standoff)
 *   goal_cost   [S,G] float32 from omgx_goalset_cost (ignored for Proj)
 *   state       [S, omgx_learner_state_doubles(G)] double, in/out:
 *               sum_costs [G] | p [G] | experts_p [5][G] | q [5] | experts_costs [5]
 *               (the caller initialises it as Learner.__init__ does, online_learner.py:78-92: zeros | 1/G | 1/G | 1/5 | zeros)
 * Outputs: goal_idx [S] int32, end [S,9] (traj.end), goal_rows [S,c,9] (chosen goal rows for the projection),
 *          goal_point [S,9] (goal_set[goal_idx]), cost_vector [S,G] double (optional, NULL to skip).
 * G <= OMGX_MAX_GOALS.  active: optional [S] int32 (NULL = all); scenes with 0 keep goal, outputs and state untouched
 * (omgx_goal_update_optimize applies its `active` to the goal update as well as to the step).
 * goal_count / eta: optional [S] int32 / [S] double for ragged goal sets — scene s uses its first goal_count[s] goals (all
 * arrays keep the padded stride num_goals; a fresh state holds 1 / goal_count[s] there and 0 in the padding) and its own
 * eta[s] = sqrt(log(goal_count[s] + 1) / optim_steps); every sum, norm and constant (delta = 1 / (4 G + 1)) is taken over
 * the scene's own goals, so the scene computes exactly what it would compute alone.
 * ------------------------------------------------------------------------------------------- */
#define OMGX_MAX_GOALS 256
#define OMGX_SCHEDULE_MAX_SCENES 1792  /* omgx_goalset_schedule: per-scene arrays of 36 bytes in < 64 KB of LDS */
#define OMGX_ALG_FTL 0
#define OMGX_ALG_FTC 1
#define OMGX_ALG_EXP 2
#define OMGX_ALG_MD 3
#define OMGX_ALG_PROJ 4
typedef struct omgx_learner_params {
    int32_t alg;             /* OMGX_ALG_*  (cfg.ol_alg)                                            */
    int32_t num_goals;       /* G                                                                   */
    int32_t n_waypoints;     /* cfg.timesteps                                                       */
    int32_t start_idx;       /* min(int(t / optim_steps * timesteps), timesteps - 1), online_learner.py:109-110 */
    int32_t constraint_num;  /* c rows of goal_rows: reach_tail_length if use_standoff else 1        */
    int32_t use_standoff;    /* goal_rows from `reach` instead of goal_set                           */
    int32_t normalize_cost;  /* cfg.normalize_cost                                                  */
    int32_t cost_parts;      /* goal_cost holds this many partial sums per goal, [S][G][cost_parts] (omgx_goalset_cost_layer_tiled); 0 or 1: one */
    double base_obstacle_weight; /* cfg.base_obstacle_weight                                        */
    double smooth_weight;        /* cfg.smoothness_base_weight * cfg.dist_eps                       */
    double eta;                  /* sqrt(log(G + 1) / optim_steps), online_learner.py:80            */
    /* Optional (ABI 7; NULL = compute / skip): the goal configurations' link poses, tabulated once per plan (the goals are fixed),
     * so that omgx_goal_update_optimize copies the chosen goal's poses instead of running its kinematics */
    const double* goal_pose_table; /* [S][G][10][12] device: omgx_pose_table(goal_set)                                        */
    double* end_poses_out;         /* [S][10][12] device: receives the chosen goal's poses (for later steps' end_poses)       */
} omgx_learner_params;
int64_t omgx_learner_state_doubles(int32_t num_goals);
int omgx_goal_update(const omgx_learner_params* h_params, const double* traj, const double* goal_set, const double* reach,
                     const float* goal_cost, double* state, int32_t num_scenes,
                     int32_t* goal_idx, double* end, double* goal_rows, double* goal_point, double* cost_vector,
                     const int32_t* active, const int32_t* goal_count, const double* eta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (5b) omgx_goal_update_optimize
 * omgx_goal_update immediately followed by omgx_chomp_optimize for the same scenes in ONE launch — the pair
 * `learner.update_goal(); optim.optimize(traj, force_update=True)` of the planner loop (omg/planner.py:612-621).
 * Same arguments, same arithmetic and bit-identical results as the two calls in that order: `end`, `goal` (the
 * learner's goal_rows) and `goal_point` are written by the goal update and consumed by the step inside the kernel.
 * The two parameter blocks must agree on n_waypoints and constraint_num.
 *   scene_flags  optional [S] int32 device scratch, zeroed once by the caller.  When given, the goal update and the
 *                goal-independent part of the step (FK, top-k, per-point costs, most gradients) run in different
 *                workgroups of the launch and meet through the scene's word: the learner's workgroup stores
 *                (ticket << 8) | chosen goal there (ABI 9; one relaxed store, the moment the goal is known), the step's
 *                workgroup waits for the ticket and reads the goal's configuration, rows and poses from goal_set / reach /
 *                goal_pose_table itself — `end`, `goal`, `goal_point` are still written, for the launches that follow.  `ticket`
 *                in [1, 2^24) and different from every ticket still stored there (use 1, 2, 3, ... per call, wrapping long
 *                before 2^24).  NULL: one workgroup per scene does both in turn.
 *   stop_on_terminate  non-zero: a scene whose info says `terminate` after this step gets active[s] = 0 (active must be
 *                given), i.e. it leaves the planner loop like `if self.info[-1]["terminate"] and t > 0: break`
 *                (omg/planner.py:626) — later launches that take the mask skip it.
 * ------------------------------------------------------------------------------------------- */
int omgx_goal_update_optimize(const omgx_learner_params* h_learner, const double* goal_set, const double* reach,
                              const float* goal_cost, double* learner_state, int32_t* goal_idx, double* cost_vector,
                              const double* robot, const omgx_chomp_params* h_params, double* traj,
                              const double* start, double* end, double* goal, double* goal_point,
                              const float* potentials, const float* grads, const float* collides,
                              int32_t* active, int32_t num_scenes,
                              double* grad, double* cost_traj, double* info, double* aux,
                              int32_t* scene_flags, int32_t ticket, int32_t stop_on_terminate,
                              const int32_t* goal_count, const double* eta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (6) omgx_point_cloud_sdf
 * Replaces the distance query of PointEnv.compute_sdf_from_points (omg/core.py:426-457): the workspace grid
 * np.arange(lo[a], hi[a], resolution) per axis ("ij" meshgrid), value = Euclidean distance from each grid node to
 * the nearest of the N perceived points (scipy cKDTree.query, k=1, p=2) — unsigned, float64 arithmetic, stored as
 * the float32 grid SignedDensityField.data_torch holds (omg/sdf_tools.py:31), x-major like every other grid.
 *   points [N,3] double;  origin[3] = bounds_min - margin;  dims[3] = len(np.arange(...)) per axis;
 *   node i of an axis sits where np.arange puts it: origin, origin + resolution, then origin + i * ((origin + resolution) - origin)
 *   (numpy's fill, which is not origin + i * resolution in the last bit).   out [X,Y,Z] float32.
 * ------------------------------------------------------------------------------------------- */
int omgx_point_cloud_sdf(const double* points, int32_t num_points, const double* h_origin, double resolution,
                         const int32_t* h_dims, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (7) omgx_goalset_schedule — dispatch order for omgx_goalset_cost_layer from the durations it recorded in `work` (ABI 4)
 * No counterpart in the reference (it plans one scene at a time); this is the engine's load balancing across the 8 XCDs.
 * The kept (scene, goal) items — scene active, goal < goal_count[scene] — are laid out scene by scene, scenes by decreasing
 * total work, each scene's goals longest first, and this list is cut into 8 contiguous pieces of equal work;
 * piece x becomes the workgroups k = 8 r + x, r = 0, 1, ... of the launch.  A scene's workgroups thus stay on one XCD (two
 * where a cut falls inside it: its SDF volumes stay in those L2s), every XCD gets the same work.  The cut uses the weights
 * as measured whenever every piece then fits its slots (slack times the even share); otherwise weights clamped to
 * [L, slack L], L = mean / 1.4, with which no piece can need more.
 *   work      [S*G] uint32 (0 counts as 1), or NULL: all items weigh the same
 *   active, goal_count   optional [S] int32 as above
 *   schedule  [omgx_goalset_schedule_len(S, G, slack)] int32 out (unused slots: -1)
 * Exact (integer) definition, restated in tests/test_gpu_schedule.py: w = work, 0 -> 1; T = sum of w over the kept items, N their
 * number; L = max(1, 10 T / (14 N)), wc = min(max(w, L), slack L); scenes ordered by decreasing sum of w (ties: lower index
 * first), a scene's goals by decreasing w (ties: lower goal first); an item with c = the wc of all items before it in that
 * list belongs to piece x = min(7, 8 (2 c + wc) / (2 Tc)), Tc = sum of wc — and, with cr = the w of all items before it, to piece
 * xr = min(7, 8 (2 cr + w) / (2 T)) of the raw cut, which replaces x for ALL items if no piece of the raw cut holds more than
 * omgx_goalset_schedule_len / 8 items (ABI 10, second half of round 5: the clamp counts light scenes for more and heavy ones for less
 * than they are); with p its position in the list and first(x) the lowest position of piece x:
 * schedule[8 (p - first(x)) + x] = scene * G + goal.
 * One workgroup; S * G <= 65536 and S <= OMGX_SCHEDULE_MAX_SCENES (1792: its per-scene arrays stay below 64 KB of LDS), else
 * OMGX_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
int32_t omgx_goalset_schedule_len(int32_t num_scenes, int32_t num_goals, int32_t slack);
int omgx_goalset_schedule(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                          int32_t num_goals, int32_t slack, int32_t* schedule, void* stream);

int omgx_goalset_schedule_parts(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                int32_t num_goals, int32_t parts, int32_t slack, int32_t* schedule, void* stream);

/* ABI 9: the order INSIDE an XCD.  OMGX_SCHEDULE_SCENE_MAJOR is the list above.  OMGX_SCHEDULE_LONGEST_FIRST keeps every item on
 * the XCD the list above gives it (whole scenes per XCD, equal work) and runs an XCD's items by decreasing w across its scenes
 * (ties: lower item index first): schedule[8 r + x] = item, r = the number of items of piece x that run before it.  For launches of
 * a round or two of the chip's workgroup slots (13 scenes x 128 goals, 25 x 64), whose span is set by what starts LAST; a launch
 * of many rounds is faster scene by scene (a scene's volumes stay in L2: +16 % longest first at 100 x 64).  Above
 * OMGX_SCHEDULE_LONGEST_FIRST_MAX_ITEMS items (S * G * parts) the call falls back to the scene-major order. */
#define OMGX_SCHEDULE_SCENE_MAJOR 0
#define OMGX_SCHEDULE_LONGEST_FIRST 1
#define OMGX_SCHEDULE_LONGEST_FIRST_MAX_ITEMS 8192
int omgx_goalset_schedule_ordered(const uint32_t* work, const int32_t* active, const int32_t* goal_count, int32_t num_scenes,
                                  int32_t num_goals, int32_t parts, int32_t slack, int32_t order, int32_t* schedule, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (8) Scenes that change while resident in HBM (ABI 8)
 * The reference rebuilds the per-object parameters on every call (Cost.compute_obstacle_cost_layer, omg/cost.py:296-335) and
 * a perception frame replaces an obstacle's volume (PointEnv.compute_sdf_from_points, omg/core.py:426-457).  With the object
 * table and the SDF pool on the device a change is:
 *   a new pose    12 floats at offset 0 of the object's record (rows of se3_inverse(pose_mat), float32) — a 48-byte copy; the
 *                 influence region lives in object coordinates and survives;
 *   a new volume  the grid (already on the device, e.g. written by omgx_point_cloud_sdf straight into the pool), then
 *     omgx_object_set_grid       lo / hi / dim / delta / grid_offset of the record and what is derived from them (inv_extent,
 *                                inv_delta) + the LOOSE influence region (the grid with 1.5 voxels of slack), one 1-thread launch;
 *     omgx_fit_influence_region  the fitted rounded box (rb_c, rb_h, rb_r, rb_r2) of the lookups that can add anything, computed
 *                                ON THE DEVICE from the volume: the algorithm of scenes.influence_rbox + tighten_far_boxes
 *                                (omg-planner_amd/scenes.py is the specification; float64, the same expressions), six small
 *                                launches on `stream`, no host pass over the voxels, no synchronisation.  epsilon / clearance =
 *                                the record's (host copies).  A record the kernels do not cull with (epsilon >= 1 or
 *                                clearance > 1) or a grid without interior keeps the loose region.
 *   scratch  device memory of omgx_region_scratch_bytes(dims) bytes, owned by the caller until the launches have run.
 * Everything is ordered on `stream`: a plan enqueued behind these calls sees the new scene.
 * ------------------------------------------------------------------------------------------- */
int64_t omgx_region_scratch_bytes(int32_t dx, int32_t dy, int32_t dz);
int omgx_object_set_grid(omgx_object* object, const float* h_lo, const float* h_hi, const int32_t* h_dims, float delta,
                         int64_t grid_offset, void* stream);
int omgx_fit_influence_region(omgx_object* object, const float* grid, const int32_t* h_dims, const float* h_lo, const float* h_hi,
                              float epsilon, float clearance, void* scratch, void* stream);

/* ABI 10: the same fit for a whole object table in seven launches (first build of a batch without a host pass over the voxels,
 * omg/core.py:366-411 Env.combine_sdfs).  The records must already hold their grid fields (lo, hi, dim, grid_offset, epsilon,
 * clearance: what omgx_object_set_grid / scenes.pack_table write) and the pool their volumes.
 *   fit_list      [n_fit] int32 (device): one object index per DISTINCT (volume, thresholds) to fit; only records the kernels cull
 *                 for (epsilon < 1, clearance <= 1, positive extents, dims >= 2) belong here
 *   need_offsets  [n_fit] int64 (device): byte offset of each entry's window flags inside the scratch's flag area — a running sum
 *                 of dx*dy*dz over the list; need_bytes = its total
 *   max_voxels    the largest dx*dy*dz in the list (sizes the launches)
 *   copy_src      [num_objects] int32 (device) or NULL: object o takes the region of object copy_src[o] (-1: keeps its own) — the
 *                 records that share a fitted entry's volume and thresholds
 *   scratch       omgx_regions_scratch_bytes(n_fit, need_bytes) bytes
 * Every fitted record equals what omgx_fit_influence_region writes for it (= scenes.tighten_far_boxes), field for field. */
int64_t omgx_regions_scratch_bytes(int32_t n_fit, int64_t need_bytes);
/* hashes [num_objects][2] uint64 (device) <- a 128-bit content hash of every object's volume (dims x float bits): equal volumes
 * give equal hashes, so a caller can fit volumes that occur several times once (DeviceScenes.fit_all). */
int omgx_volume_hashes(const omgx_object* objects, int32_t num_objects, const float* pool, uint64_t* hashes, void* stream);
int omgx_fit_influence_regions(omgx_object* objects, int32_t num_objects, const float* pool, const int32_t* fit_list,
                               const int64_t* need_offsets, int32_t n_fit, int64_t max_voxels, const int32_t* copy_src,
                               void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (9) omgx_plan_persistent — K iterations of the planner loop for all scenes in ONE launch (ABI 11)
 * Replaces the loop body of Planner.plan (omg/planner.py:612-630) — Learner.update_goal (omg/online_learner.py:237-249) followed by
 * Optimizer.optimize(traj, force_update=True) (omg/optimizer.py:115-135), num_iters times — for S independent scenes: what
 * num_iters x (omgx_goalset_cost_layer + omgx_goal_update_optimize) compute, bit for bit, scheduled by the one dependency the
 * algorithm has (iteration t + 1 of scene s needs iteration t of scene s; omg/core.py:869-885 plans scenes independently)
 * instead of by launch boundaries.  Resident workgroups claim work items — a goal of the goal-set batch, a piece of a scene's
 * trajectory layer — the last item of a scene's iteration to finish runs the scene's learner and step and activates its next
 * iteration (csrc/omg_persist.h).  Batch layout (whole goals), poses handed over: omgx_chomp_params.start_poses / end_poses,
 * omgx_learner_params.goal_pose_table / end_poses_out and layer_poses are REQUIRED.
 *   iteration k   omgx_plan_iter: mode 1 = goal-selecting (goal-set batch over the window that starts at start_idx + layer, then
 *                 learner + step), 0 = goal fixed (layer + step: the plan's last cfg.extra_smooth_steps iterations);
 *                 obstacle_weight / smoothness_weight / step_size = Optimizer.update's schedule for that iteration
 *                 (omg/optimizer.py:59-80); stop_on_terminate: a scene whose step reports info["terminate"] leaves the loop
 *                 (planner.py:626; active[s] <- 0); do_update as in omgx_chomp_params.
 *   h_iters       [num_iters] host (checked here), d_iters: the same bytes in device memory (read by the kernel)
 *   active        [S] int32 or NULL: scenes with 0 are not planned
 *   workspace     omgx_plan_persistent_workspace_bytes(S, n) bytes of device memory: queue state (re-initialised by every call) and
 *                 the step's scratch; one workspace per launch in flight
 *   max_workgroups  0: five per compute unit (what is resident at 30 waypoints); tests pass small numbers
 *   update_cus    compute units per XCD whose workgroups take no items and serve the scenes' updates instead (the update is the
 *                 scene's critical path and latency-bound: beside four goal workgroups it runs three times as long); < 0: the library's
 *                 rule (2 when the launch fills the chip and holds >= 32 scenes, else 0), 0: every update runs where the scene's last item ran
 * Everything else as in omgx_goalset_cost_layer_parts (goal_parts = 1) and omgx_goal_update_optimize.  Asynchronous; errors inside
 * the launch (a bounded wait that ran out) are reported by omgx_plan_persistent_status after the stream has been synchronised:
 * h_status[0] != 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct omgx_plan_iter {
    int32_t mode;              /* 1: Learner.update_goal + Optimizer.optimize, 0: Optimizer.optimize with the goal fixed */
    int32_t start_idx;         /* Learner's window start of this iteration (online_learner.py:109-110); mode 0: ignored    */
    int32_t stop_on_terminate; /* planner.py:626                                                                           */
    int32_t do_update;         /* omgx_chomp_params.do_update of this iteration                                            */
    double obstacle_weight, smoothness_weight, step_size; /* cfg.* after Optimizer.update for this iteration              */
} omgx_plan_iter;
int64_t omgx_plan_persistent_workspace_bytes(int32_t num_scenes, int32_t n_waypoints);
int omgx_plan_persistent(const double* robot, int32_t n_points, const omgx_object* objects, const int32_t* scene_begin,
                         const float* sdf_pool, const double* goals, int32_t num_scenes, int32_t num_goals,
                         double time_interval, int32_t soften_fingers, float* goal_cost, float* collides, double* traj,
                         int32_t n_waypoints, int32_t layer_soften_fingers, float* layer_potentials, float* layer_grads,
                         float* layer_collides, double* layer_poses, int32_t* active, const int32_t* goal_count,
                         const omgx_learner_params* h_learner, const double* goal_set, const double* reach,
                         double* learner_state, int32_t* goal_idx, double* cost_vector, const double* eta,
                         const omgx_chomp_params* h_params, const double* start, double* end, double* goal,
                         double* goal_point, double* grad, double* cost_traj, double* info,
                         const omgx_plan_iter* h_iters, const omgx_plan_iter* d_iters, int32_t num_iters,
                         void* workspace, int64_t workspace_bytes, int32_t max_workgroups, int32_t update_cus, void* stream);
/* h_status[4] <- {failure code (0 = none), scenes finished, scenes planned, activations made} of the last launch on this workspace;
 * synchronises `stream`. */
int omgx_plan_persistent_status(const void* workspace, int32_t num_scenes, int32_t* h_status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics
 * ------------------------------------------------------------------------------------------- */
const char* omgx_last_error(void); /* thread-local text of the last OMGX_ERR_LAUNCH               */
/* Measurement hook (bench.py): while enabled (on = 1: every launch; on = N > 1: every N-th launch — an event pair costs
 * about 6 us of stream time), launches of the SDF kernel behind omgx_fk_sdf / omgx_goalset_cost are bracketed by HIP
 * events on their own stream.  omgx_timing_collect waits for them,
 * writes the per-launch durations (ms) to h_ms[0..cap) and the variant to h_kind (0 = potentials only, i.e. the
 * goal-set batch; 1 = with gradients, i.e. the waypoint batch; may be NULL) and returns how many;
 * not for graph capture.
 * Threads: every other entry point may be called concurrently from several host threads (on different streams); the
 * library's only process-wide state beside this hook is set-once (kernel attributes per device); there are no
 * environment switches.  The timing hook is a process-wide recorder for ONE device, guarded by a mutex: enable, launch, collect;
 * launches on another device than the one current at omgx_timing_enable are not recorded.  h_kind: 0 = goal-set launch
 * (with or without the trajectory layer), 1 = layer-only launch (omgx_fk_sdf on a trajectory-sized batch). */
int omgx_timing_enable(int32_t on);
int omgx_timing_collect(float* h_ms, int32_t* h_kind, int32_t cap);
int omgx_abi_version(void);        /* bumps when a signature or struct layout changes              */
int omgx_device_arch(char* h_buf, int32_t h_len); /* writes gcnArchName of the current device      */
int32_t omgx_device_cu_count(void);               /* compute units of the current device (< 0: error)  */
/* h_dst (pinned host memory) <- nbytes from device memory `src`, ordered behind everything enqueued on `stream`, and wait for it:
 * hipMemcpyAsync + hipStreamSynchronize in one call (the closing step of a planner iteration through the drop-in classes). */
int omgx_download_sync(void* h_dst, const void* src, int64_t nbytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMG_HIP_H */
