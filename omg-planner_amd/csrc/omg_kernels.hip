// omg_kernels.hip — gfx950 kernels + the C ABI of include/omg_hip.h (part 1: SDF / FK / goal-set).
//
// Kernels in this file
//   k_sdf_loss        points x objects SDF potential/gradient/collides, one launch (API 1)
//   k_fk_points       Panda FK, one lane per robot configuration -> float32 collision points
//   k_sdf_chunks      the same SDF evaluation over FK-produced points, per (scene, chunk) workgroup,
//                     optional arc-length weighting + per-chunk reduction (APIs 2 and 3)
//
// Hot-path data layout in HBM:
//   objects[]  128-byte omgx_object records, wave-uniform reads -> SGPRs via scalar loads
//   sdf pool   float32 grids, x-major, z fastest: a trilinear row (z0-1..z0+2) is one 16-byte load
//   points ws  [scene][chunk][link][config-in-chunk][point][3] float32: consecutive lanes are the P
//              points of one link at consecutive waypoints -> spatially coherent gathers, coalesced
//              point reads
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

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

__global__ __launch_bounds__(256) void k_sdf_loss(RawObjects R, const float* __restrict__ points, int64_t N, int O,
                                                   float* __restrict__ pot, float* __restrict__ grad,
                                                   float* __restrict__ col) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        Accum acc{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int o = 0; o < O; ++o) {  // uniform: parameters travel through scalar loads
            if (R.dis[o] > 0.0f) continue;
            ObjParams op;
#pragma unroll
            for (int k = 0; k < 12; ++k) op.T[k] = R.pose_init[16 * o + k];
            const float* L = R.sdf_limits + 10 * o;
#pragma unroll
            for (int k = 0; k < 3; ++k) { op.lo[k] = L[k]; op.hi[k] = L[3 + k]; op.dim[k] = (int)L[6 + k]; }
            op.delta = L[9]; op.eps = R.eps[o]; op.pad = R.pad[o]; op.clr = R.clr[o];
            const int64_t off = (int64_t)o * op.dim[0] * op.dim[1] * op.dim[2];  // .cu:147
            sdf_pair<true>(op, R.sdf_grids + off, px, py, pz, acc);
        }
        pot[i] = acc.pot;
        col[i] = acc.col;
        grad[3 * i] = acc.gx; grad[3 * i + 1] = acc.gy; grad[3 * i + 2] = acc.gz;
    }
}

// =================================================================================================
// (2) k_fk_points — one lane per configuration
// =================================================================================================
// Config sources:
//   mode 0: joints[S][C][9] given                                   (omgx_fk_sdf, chomp waypoints)
//   mode 1: joints interpolated start + (i+1)/(n+1) (goal - start)  (omgx_goalset_cost; util.py:261-290 "linear")
//           plus one extra "config n" per (scene) = traj_start itself, written to ws_start
struct FkArgs {
    const double* robot;
    int P;
    int mode;
    const double* joints;      // mode 0: [S][C][9]
    const double* traj_start;  // mode 1: [S][9]
    const double* goals;       // mode 1: [S][G][9]
    int S, C;                  // C configs per scene (mode 1: C = G * n)
    int n;                     // mode 1: waypoints per goal
    int CH;                    // configs per chunk
    float* ws;                 // [S][NCH][10][CH][P][3]
    float* ws_start;           // mode 1: [S][10][P][3]
};

__global__ __launch_bounds__(64) void k_fk_points(FkArgs a) {
    const RobotView rv(a.robot, a.P);
    const int per_scene = a.C + (a.mode == 1 ? 1 : 0);
    const int64_t total = (int64_t)a.S * per_scene;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int s = (int)(id / per_scene), c = (int)(id % per_scene);
    double q[9];
    bool is_start = false;
    if (a.mode == 0) {
        const double* src = a.joints + ((int64_t)s * a.C + c) * 9;
#pragma unroll
        for (int d = 0; d < 9; ++d) q[d] = src[d];
    } else {
        const double* q0 = a.traj_start + 9 * (int64_t)s;
        if (c == a.C) {
            is_start = true;
#pragma unroll
            for (int d = 0; d < 9; ++d) q[d] = q0[d];
        } else {
            const int g = c / a.n, i = c % a.n;
            const double* qg = a.goals + ((int64_t)s * (a.C / a.n) + g) * 9;
            const double t = (double)(i + 1) / (double)(a.n + 1);
#pragma unroll
            for (int d = 0; d < 9; ++d) q[d] = q0[d] + t * (qg[d] - q0[d]);
        }
    }
    const int P = a.P, CH = a.CH;
    const int NCH = (a.C + CH - 1) / CH;
    const int chunk = c / CH, ci = c % CH;
    fk_chain(rv, q, [&](int l, const Pose& pose) {
        float* dst = is_start ? a.ws_start + ((int64_t)s * 10 + l) * P * 3
                              : a.ws + ((((int64_t)s * NCH + chunk) * 10 + l) * CH + ci) * (int64_t)P * 3;
        for (int p = 0; p < P; ++p) {
            double x, y, z;
            pose_apply(pose, rv.pts(l, p), x, y, z);
            dst[3 * p] = (float)x; dst[3 * p + 1] = (float)y; dst[3 * p + 2] = (float)z;  // .cuda().float(), cost.py:136,218
        }
    });
}

// =================================================================================================
// (3) k_sdf_chunks — SDF over FK points, one workgroup per (scene, chunk)
// =================================================================================================
struct ChunkArgs {
    const omgx_object* objects;
    const int32_t* scene_begin;
    const float* pool;
    const float* ws;        // [S][NCH][10][CH][P][3]
    const float* ws_start;  // [S][10][P][3] or null
    int S, C, CH, NCH, P;
    int soften;             // uncheck_finger_collision == -1 (cost.py:350-353)
    int arc;                // weight potentials by ||(x_i - x_{i-1}) / dt|| (cost.py:235-275)
    float inv_dt;
    float* pot;             // [S][C][10][P] or null
    float* grad;            // [S][C][10][P][3] or null
    float* col;             // [S][C][10][P] or null
    float* chunk_cost;      // [S][NCH] or null: sum of (weighted) potentials of the chunk
    float* chunk_col;       // [S][NCH] or null: sum of collides of the chunk
};

template <bool WANT_GRAD>
__global__ __launch_bounds__(256) void k_sdf_chunks(ChunkArgs a) {
    __shared__ float red[2][4];
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed; used for L2 affinity only).
    // All chunks of a scene go to the same XCD so the scene's SDF volumes stay in one 4 MiB L2.
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int s = (j / a.NCH) * 8 + xcd, chunk = j % a.NCH;
    if (s >= a.S) return;
    const int o_begin = a.scene_begin[s], o_end = a.scene_begin[s + 1];
    const int P = a.P, CH = a.CH;
    const int nvalid = min(CH, a.C - chunk * CH);  // configs in this (possibly last, partial) chunk
    const int items = 10 * CH * P;
    const float* base = a.ws + ((int64_t)s * a.NCH + chunk) * (int64_t)items * 3;
    float tsum = 0.0f, tcol = 0.0f;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int p = it % P, ci = (it / P) % CH, l = it / (P * CH);
        if (ci >= nvalid) continue;
        const float px = base[3 * it], py = base[3 * it + 1], pz = base[3 * it + 2];
        Accum acc = sdf_point<WANT_GRAD>(a.objects, o_begin, o_end, a.pool, px, py, pz);
        if (a.soften && l >= 8) { acc.pot *= 0.1f; acc.gx *= 0.1f; acc.gy *= 0.1f; acc.gz *= 0.1f; acc.col = 0.0f; }
        if (a.arc) {
            const float* prev = ci > 0 ? base + 3 * (it - P) : a.ws_start + (((int64_t)s * 10 + l) * P + p) * 3;
            const float vx = (px - prev[0]) * a.inv_dt, vy = (py - prev[1]) * a.inv_dt, vz = (pz - prev[2]) * a.inv_dt;
            acc.pot = acc.pot * sqrtf(vx * vx + vy * vy + vz * vz);
        }
        const int64_t k = (((int64_t)s * a.C + chunk * CH + ci) * 10 + l) * P + p;
        if (a.pot) a.pot[k] = acc.pot;
        if (a.col) a.col[k] = acc.col;
        if (WANT_GRAD) { a.grad[3 * k] = acc.gx; a.grad[3 * k + 1] = acc.gy; a.grad[3 * k + 2] = acc.gz; }
        tsum += acc.pot;
        tcol += acc.col;
    }
    if (a.chunk_cost || a.chunk_col) {  // fixed-order block reduction: lanes -> waves -> thread 0
        const float ws_ = wave_sum(tsum), wc_ = wave_sum(tcol);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = ws_; red[1][w] = wc_; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t k = (int64_t)s * a.NCH + chunk;
            if (a.chunk_cost) a.chunk_cost[k] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            if (a.chunk_col) a.chunk_col[k] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        }
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
static thread_local char g_err[256] = "";

int omgx_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return OMGX_ERR_LAUNCH;
}
extern "C" const char* omgx_last_error(void) { return g_err; }

// ---- optional per-launch timing of the dominant kernel (k_sdf_chunks) with HIP events ----------
// bench.py enables it around its timed region; events are recorded on the launch stream.
#define OMGX_TIMING_CAP 4096
static bool g_timing = false;
static int g_timing_n = 0;
static hipEvent_t g_ev[OMGX_TIMING_CAP][2];
static bool g_ev_made[OMGX_TIMING_CAP];

extern "C" int omgx_timing_enable(int32_t on) {
    g_timing = on != 0;
    g_timing_n = 0;
    return OMGX_OK;
}

// Waits for the recorded launches and writes their durations in milliseconds; returns the count.
extern "C" int omgx_timing_collect(float* h_ms, int32_t cap) {
    int n = g_timing_n < cap ? g_timing_n : cap;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventSynchronize(g_ev[i][1]);
        if (e != hipSuccess) return omgx_set_error("hipEventSynchronize", e);
        e = hipEventElapsedTime(&h_ms[i], g_ev[i][0], g_ev[i][1]);
        if (e != hipSuccess) return omgx_set_error("hipEventElapsedTime", e);
    }
    g_timing_n = 0;
    return n;
}

static inline int timing_slot() {
    if (!g_timing || g_timing_n >= OMGX_TIMING_CAP) return -1;
    const int i = g_timing_n;
    if (!g_ev_made[i]) {
        if (hipEventCreate(&g_ev[i][0]) != hipSuccess || hipEventCreate(&g_ev[i][1]) != hipSuccess) return -1;
        g_ev_made[i] = true;
    }
    return i;
}
extern "C" int omgx_abi_version(void) { return 1; }
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

// ---- workspace sizing -------------------------------------------------------------------------
static inline int chunk_configs_fk_sdf(int C) { return C < 32 ? C : 32; }

extern "C" int64_t omgx_fk_sdf_workspace_bytes(int32_t num_scenes, int32_t configs_per_scene, int32_t n_points) {
    if (num_scenes <= 0 || configs_per_scene <= 0 || n_points <= 0) return 0;
    const int CH = chunk_configs_fk_sdf(configs_per_scene);
    const int64_t NCH = (configs_per_scene + CH - 1) / CH;
    return (int64_t)num_scenes * NCH * 10 * CH * n_points * 3 * sizeof(float);
}

extern "C" int64_t omgx_goalset_workspace_bytes(int32_t num_scenes, int32_t num_goals, int32_t n_remaining, int32_t n_points) {
    if (num_scenes <= 0 || num_goals <= 0 || n_remaining <= 0 || n_points <= 0) return 0;
    const int64_t pts = (int64_t)num_scenes * num_goals * 10 * n_remaining * n_points * 3 * sizeof(float);
    const int64_t start = (int64_t)num_scenes * 10 * n_points * 3 * sizeof(float);
    return pts + start;
}

static int launch_chunks(const ChunkArgs& ca, hipStream_t st) {
    const int scene_groups = (ca.S + 7) / 8;
    const int64_t grid = (int64_t)scene_groups * ca.NCH * 8;
    if (grid > 0x7fffffff) return OMGX_ERR_UNSUPPORTED;
    const int slot = timing_slot();
    if (slot >= 0) (void)hipEventRecord(g_ev[slot][0], st);
    if (ca.grad)
        hipLaunchKernelGGL(k_sdf_chunks<true>, dim3((unsigned)grid), dim3(256), 0, st, ca);
    else
        hipLaunchKernelGGL(k_sdf_chunks<false>, dim3((unsigned)grid), dim3(256), 0, st, ca);
    if (slot >= 0) { (void)hipEventRecord(g_ev[slot][1], st); ++g_timing_n; }
    OMGX_CHECK_LAUNCH("k_sdf_chunks");
    return OMGX_OK;
}

extern "C" int omgx_fk_sdf(const double* robot, int32_t n_points, const omgx_object* objects, const int32_t* scene_begin,
                           const float* sdf_pool, const double* joints, int32_t num_scenes, int32_t configs_per_scene,
                           int32_t soften_fingers, float* potentials, float* grads, float* collides, void* workspace,
                           void* stream) {
    if (num_scenes < 0 || configs_per_scene < 0) return OMGX_ERR_INVALID;
    if (num_scenes == 0 || configs_per_scene == 0) return OMGX_OK;
    if (!robot || !objects || !scene_begin || !joints || !workspace) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS) return OMGX_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int CH = chunk_configs_fk_sdf(configs_per_scene);
    const int NCH = (configs_per_scene + CH - 1) / CH;
    FkArgs fa{};
    fa.robot = robot; fa.P = n_points; fa.mode = 0; fa.joints = joints; fa.S = num_scenes; fa.C = configs_per_scene;
    fa.CH = CH; fa.ws = (float*)workspace;
    const int64_t total = (int64_t)num_scenes * configs_per_scene;
    hipLaunchKernelGGL(k_fk_points, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, fa);
    OMGX_CHECK_LAUNCH("k_fk_points");
    ChunkArgs ca{};
    ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool; ca.ws = (const float*)workspace;
    ca.S = num_scenes; ca.C = configs_per_scene; ca.CH = CH; ca.NCH = NCH; ca.P = n_points; ca.soften = soften_fingers != 0;
    ca.pot = potentials; ca.grad = grads; ca.col = collides;
    return launch_chunks(ca, st);
}

extern "C" int omgx_goalset_cost(const double* robot, int32_t n_points, const omgx_object* objects,
                                 const int32_t* scene_begin, const float* sdf_pool, const double* traj_start,
                                 const double* goals, int32_t num_scenes, int32_t num_goals, int32_t n_remaining,
                                 double time_interval, int32_t soften_fingers, float* goal_cost, float* potentials,
                                 float* collides, void* workspace, void* stream) {
    if (num_scenes < 0 || num_goals < 0) return OMGX_ERR_INVALID;
    if (num_scenes == 0 || num_goals == 0) return OMGX_OK;
    if (!robot || !objects || !scene_begin || !traj_start || !goals || !goal_cost || !workspace) return OMGX_ERR_INVALID;
    if (n_points < 1 || n_points > OMGX_MAX_POINTS || n_remaining < 1 || n_remaining > OMGX_MAX_WAYPOINTS)
        return OMGX_ERR_UNSUPPORTED;
    if (!(time_interval > 0.0)) return OMGX_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int n = n_remaining, C = num_goals * n;
    float* ws = (float*)workspace;
    float* ws_start = ws + (int64_t)num_scenes * num_goals * 10 * n * n_points * 3;
    FkArgs fa{};
    fa.robot = robot; fa.P = n_points; fa.mode = 1; fa.traj_start = traj_start; fa.goals = goals; fa.S = num_scenes;
    fa.C = C; fa.n = n; fa.CH = n; fa.ws = ws; fa.ws_start = ws_start;
    const int64_t total = (int64_t)num_scenes * (C + 1);
    hipLaunchKernelGGL(k_fk_points, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, fa);
    OMGX_CHECK_LAUNCH("k_fk_points");
    ChunkArgs ca{};
    ca.objects = objects; ca.scene_begin = scene_begin; ca.pool = sdf_pool; ca.ws = ws; ca.ws_start = ws_start;
    ca.S = num_scenes; ca.C = C; ca.CH = n; ca.NCH = num_goals; ca.P = n_points; ca.soften = soften_fingers != 0;
    ca.arc = 1; ca.inv_dt = (float)(1.0 / time_interval);
    ca.pot = potentials; ca.grad = nullptr; ca.col = nullptr; ca.chunk_cost = goal_cost; ca.chunk_col = collides;
    return launch_chunks(ca, st);
}
