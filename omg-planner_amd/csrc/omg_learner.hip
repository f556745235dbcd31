// omg_learner.hip — k_goal_update: Learner.update_goal for S scenes, one workgroup of 5 waves per scene
// (device code: omg_learner_body.h).
#include "omg_learner_body.h"

namespace {

__global__ __launch_bounds__(320) void k_goal_update(omg_learner::LearnerArgs a) {
    __shared__ double sh_pn[5][OMGX_MAX_GOALS];
    __shared__ double sh_tab[5][128];
    omg_learner::learner_scene(a, blockIdx.x, sh_pn, sh_tab);
}

}  // namespace

extern "C" int64_t omgx_learner_state_doubles(int32_t num_goals) { return num_goals < 1 ? 0 : 7 * (int64_t)num_goals + 10; }

extern "C" int omgx_goal_update(const omgx_learner_params* h_params, const double* traj, const double* goal_set,
                                const double* reach, const float* goal_cost, double* state, int32_t num_scenes,
                                int32_t* goal_idx, double* end, double* goal_rows, double* goal_point, double* cost_vector,
                                const int32_t* active, const int32_t* goal_count, const double* eta, void* stream) {
    if (h_params && num_scenes == 0) return OMGX_OK;
    omg_learner::LearnerArgs a;
    const int rc = omg_learner::make_args(h_params, traj, goal_set, reach, goal_cost, state, num_scenes, goal_idx, end, goal_rows,
                                          goal_point, cost_vector, active, goal_count, eta, a);
    if (rc != OMGX_OK) return rc;
    hipLaunchKernelGGL(k_goal_update, dim3(num_scenes), dim3(320), 0, (hipStream_t)stream, a);
    OMGX_CHECK_LAUNCH("k_goal_update");
    return OMGX_OK;
}
