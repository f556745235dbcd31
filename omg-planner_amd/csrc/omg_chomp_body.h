// omg_chomp_body.h — device code of ONE Optimizer.optimize step for one trajectory (chomp_scene), shared by the update kernels of
// omg_chomp.hip (k_chomp_optimize, k_update_optimize, k_update_optimize_split: 512 threads, everything staged in LDS) and by the
// persistent planner kernel of omg_persist.h (256 threads inside a goal-set workgroup's footprint: LIGHT).  See omg_chomp.hip for what it
// replaces in the reference.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "omg_device.h"
#include "omg_host.h"
#include "omg_learner_body.h"

using namespace omg;

// float64 code only (the float32 SDF arithmetic that must stay bit-identical to the oracle lives in omg_kernels.hip and its headers): the
// compiler may fuse multiply-adds from here on.  A translation unit that goes on with float32 code restores `contract(off)` after the include.
#pragma clang fp contract(fast)

#define CH_TPB 512
#define CH_WAVES (CH_TPB / 64)

// Debug aid (make CXXFLAGS+=-DOMGX_PHASE_TIMING): workgroup 0 stamps the shader clock at the phase boundaries;
// tools/phase_timing.py reads them through omgx_debug_phase_times.  Not part of the ABI, compiled out by default.
#ifdef OMGX_PHASE_TIMING
__device__ unsigned long long g_chomp_phase[48];
#define PHASE_MARK(i) do { if (s == 0 && threadIdx.x == 0) g_chomp_phase[i] = __builtin_readcyclecounter(); } while (0)
#define PHASE_MARK_T(i, t) do { if (s == 0 && threadIdx.x == (t)) g_chomp_phase[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE_MARK_T(i, t) do { } while (0)
#define PHASE_MARK(i) do { } while (0)
#endif

namespace {

struct ChompArgs {
    const double* robot;
    omgx_chomp_params prm;
    double* traj;              // [S][n][9] in/out
    const double* start;       // [S][9]
    const double* end;         // [S][9]
    const double* goal;        // [S][c][9]
    const double* goal_point;  // [S][9]
    const float* pot;          // [S][n][10][P]
    const float* pgrad;        // [S][n][10][P][3]
    const float* col;          // [S][n][10][P]
    const int32_t* active;     // [S] or null
    int32_t* deactivate;       // == active (writable) when a scene that terminates is to leave the loop (planner.py:626), else null
    double* grad;              // [S][n][9]
    double* cost_traj;         // [S][n]
    double* info;              // [S][16]
    double* aux;               // [S][aux_doubles(n)] or null
    int pot_in_lds;            // host decision: the item-ordered copy of the potentials fits beside the rest
    double* light_scratch;     // LIGHT (omg_persist.h): [S][n][10][8] doubles in global memory where the winners' gradient rows are parked (L.gl)
};

// wrap_joint(l+1) (omg/util.py:213-220): k-th joint index (into the 10-joint tables) of link l; count via njoints().
__device__ __forceinline__ int njoints(int l) { return l < 7 ? l + 1 : (l == 7 ? 7 : 8); }
__device__ __forceinline__ int joint_of(int l, int k) { return k < 7 ? k : l; /* k==7: finger joint 8 or 9 == link index */ }
// wrap_index(l+1) (omg/util.py:205-210): trajectory column of slot k of link l.
__device__ __forceinline__ int column_of(int l, int k) { return k < 7 ? k : l - 1; /* 8->7, 9->8 */ }

__device__ __forceinline__ uint32_t float_key(float f) {  // order-preserving float -> uint
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct Lds {
    double* pose;    // [n+2][10][12]
    double* jconst;  // [10][6] joint axis ax' and origin og' in their link frames (robot blob AX, OG)
    double* gl;      // [n][10][8]
    double* gcost;   // [n][10]
    double* xi;      // [n][9]
    double* g;       // [n][9]  total gradient
    double* og;      // [n][9]  obstacle gradient -> clipped weighted obstacle gradient
    double* sg;      // [n][9]  smoothness gradient -> weighted
    double* tv;      // [n][9]
    double* tvs;     // [n][9]
    double* sml;     // [n+1]   smoothness loss
    double* pts;     // [10][P][3] centred collision points
    double* red;     // [64] scratch for reductions / scalars
    int* gwin;       // [n][10] winner point of the group or -1
    uint32_t* hist;  // [2][256]
    uint32_t* tie;   // [(n*10*16+31)/32] tie bit mask
    int* iscr;       // [16] int scalars
    int* wlist;      // [n*10] groups that have a winner (phase 3)
    float* potl;     // [n*160] this trajectory's potentials by item (0 in the padding lanes), or null when LDS is short
    double* fkc;     // [246] kinematic-chain constants of the robot blob (UVW, TP, H, LF, RF)
    // LIGHT: the link poses stay where the launches before left them (global memory, L2-resident) instead of being copied into `pose`
    const double* gp_start;  // [10][12] the start configuration's
    const double* gp_wp;     // [n][10][12] the waypoints' (left by the layer workgroups)
    const double* gp_end;    // [10][12] the end configuration's (the chosen goal's row of the goal pose table)
    int ncfg;
};

// pose (12 doubles) of link l at configuration cfg (0 = start, 1..n = waypoints, n + 1 = end)
template <bool LIGHT>
__device__ __forceinline__ const double* pose_of(const Lds& L, int cfg, int l) {
    if constexpr (LIGHT) {
        const double* base = cfg == 0 ? L.gp_start : (cfg == L.ncfg - 1 ? L.gp_end : L.gp_wp + (size_t)(cfg - 1) * 120);
        return base + 12 * l;
    } else return L.pose + ((size_t)cfg * 10 + l) * 12;
}

template <bool LIGHT = false>
__device__ __forceinline__ Lds carve(unsigned char* base, int n, int P, bool pot_in_lds) {
    Lds L;
    double* d = reinterpret_cast<double*>(base);
    L.pose = LIGHT ? nullptr : d; if (!LIGHT) d += (size_t)(n + 2) * 120;
    L.jconst = d; d += 60;
    L.gl = LIGHT ? nullptr : d; if (!LIGHT) d += (size_t)n * 80;  // (LIGHT: ChompArgs::light_scratch)
    L.gcost = d; d += (size_t)n * 10;
    L.xi = d; d += (size_t)n * 9;
    L.g = d; d += (size_t)n * 9;
    L.og = d; d += (size_t)n * 9;
    L.sg = d; d += (size_t)n * 9;
    L.tv = d; d += (size_t)n * 9;
    L.tvs = d; d += (size_t)n * 9;
    L.sml = d; d += (n + 1);
    L.pts = d; d += 30 * P;
    L.red = d; d += 64;
    L.fkc = LIGHT ? nullptr : d; if (!LIGHT) d += 246;
    L.gp_start = L.gp_wp = L.gp_end = nullptr; L.ncfg = n + 2;
    int* ip = reinterpret_cast<int*>(d);
    L.gwin = ip; ip += n * 10;
    L.hist = reinterpret_cast<uint32_t*>(ip); ip += 512;  // two histograms of 256 bins (the radix select's passes take turns)
    L.tie = reinterpret_cast<uint32_t*>(ip); ip += (n * 160 + 31) / 32;
    L.iscr = ip; ip += 16;
    L.wlist = ip; ip += n * 10;
    L.potl = (!LIGHT && pot_in_lds) ? reinterpret_cast<float*>(ip) : nullptr;
    return L;
}

// bytes carve<true> hands out (host and device)
__host__ __device__ static inline size_t chomp_light_lds_bytes(int n, int P) {
    const size_t d = 60 + (size_t)n * 10 + (size_t)n * 9 * 6 + (n + 1) + 30 * (size_t)P + 64;
    const size_t i = (size_t)n * 10 + 512 + (n * 160 + 31) / 32 + 16 + (size_t)n * 10;
    return d * 8 + i * 4;
}

// x = out_l(cfg) . pts'(l,p)
template <bool LIGHT = false>
__device__ __forceinline__ void point_at(const Lds& L, int cfg, int l, int p, int P, double* x) {
    const double* A = pose_of<LIGHT>(L, cfg, l);
    const double* q = L.pts + 3 * (l * P + p);
    x[0] = A[0] * q[0] + A[1] * q[1] + A[2] * q[2] + A[9];
    x[1] = A[3] * q[0] + A[4] * q[1] + A[5] * q[2] + A[10];
    x[2] = A[6] * q[0] + A[7] * q[1] + A[8] * q[2] + A[11];
}

// Cost.functional_grad for one point (cost.py:24-43): returns c*||v||, g = ||v|| P grad_c - c P a / (||v||^2 + 1e-8)
__device__ __forceinline__ double functional_g(const double* v, const double* a, double c, const double* dc, double* g) {
    const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double nv[3], Pa[3], Pg[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) nv[r] = v[r] / (vn + 1e-8);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double pa = 0.0, pg = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double m = (r == q ? 1.0 : 0.0) - nv[r] * nv[q];
            pa += m * a[q];
            pg += m * dc[q];
        }
        Pa[r] = pa; Pg[r] = pg;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) g[r] = vn * Pg[r] - c * (Pa[r] / (vn * vn + 1e-8));
    return c * vn;
}

// J_k . g for the k-th joint of link l at waypoint i (cost.py:92-110)
template <bool LIGHT = false>
__device__ __forceinline__ double jacobian_dot(const Lds& L, int i, int l, int k, const double* x, const double* g) {
    const int j = joint_of(l, k);
    // joint frame = out_j . tip2joint_j (robot_pykdl.py:190-201): axis = R_j ax'_j, origin = R_j og'_j + t_j
    const double* A = pose_of<LIGHT>(L, i + 1, j);
    const double* c = L.jconst + 6 * j;
    const double ax[3] = {A[0] * c[0] + A[1] * c[1] + A[2] * c[2], A[3] * c[0] + A[4] * c[1] + A[5] * c[2],
                          A[6] * c[0] + A[7] * c[1] + A[8] * c[2]};
    if (l >= 8 && k == 7) return ax[0] * g[0] + ax[1] * g[1] + ax[2] * g[2];  // "prsimatic" finger joint
    const double o[3] = {A[0] * c[3] + A[1] * c[4] + A[2] * c[5] + A[9], A[3] * c[3] + A[4] * c[4] + A[5] * c[5] + A[10],
                         A[6] * c[3] + A[7] * c[4] + A[8] * c[5] + A[11]};
    const double d0 = x[0] - o[0], d1 = x[1] - o[1], d2 = x[2] - o[2];
    const double J0 = ax[1] * d2 - ax[2] * d1, J1 = ax[2] * d0 - ax[0] * d2, J2 = ax[0] * d1 - ax[1] * d0;
    return J0 * g[0] + J1 * g[1] + J2 * g[2];
}

// v, a of point (i,l,p) by finite differences along the waypoint axis (config.py:134-159); cfg index = i+1
template <bool LIGHT = false>
__device__ __forceinline__ void point_kinematics(const Lds& L, int i, int l, int p, int P, double dt, double* x, double* v, double* a) {
    double xm[3], xp[3];
    point_at<LIGHT>(L, i + 1, l, p, P, x);
    point_at<LIGHT>(L, i, l, p, P, xm);
    point_at<LIGHT>(L, i + 2, l, p, P, xp);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        v[r] = (x[r] - xm[r]) / dt;
        a[r] = (xm[r] - 2.0 * x[r] + xp[r]) / (dt * dt);
    }
}

// Sum of arr[0..count) by wave 0 in a fixed order; result broadcast through L.red[slot].  Call from ALL threads.
__device__ __forceinline__ double block_sum(const Lds& L, const double* arr, int count, int slot) {
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int k = threadIdx.x; k < count; k += 64) s += arr[k];
        s = wave_allsum(s);
        if (threadIdx.x == 0) L.red[slot] = s;
    }
    __syncthreads();
    return L.red[slot];
}

__device__ __forceinline__ double ainv_entry(int i, int k, int n, bool free_end, double dt2) {
    const int lo = i < k ? i : k, hi = i < k ? k : i;
    return free_end ? dt2 * (double)(lo + 1) : dt2 * (double)(lo + 1) * (double)(n - hi) / (double)(n + 1);
}

// out[i][d] = sum_k Ainv[i][k] in[k][d]   (cfg.Ainv.dot(.), optimizer.py:109,132,156) with the closed-form entries factored:
//   sum_k Ainv[i][k] x_k = c [ u_i sum_(k<=i) (k + 1) x_k  +  (i + 1) sum_(k>i) w_k x_k ]
//   free end : u_i = 1,     w_k = 1,     c = dt^2              (Ainv[i][k] = dt^2 (min + 1))
//   fixed end: u_i = n - i, w_k = n - k, c = dt^2 / (n + 1)    (Ainv[i][k] = dt^2 (min + 1)(n - max) / (n + 1))
// One thread per element, its two sums in ascending k: per term one conversion and one multiply-add instead of the ~10 instructions
// of the entry's closed form (min / max, two conversions, a division for the fixed end), and the LDS reads of a column are independent
// of the arithmetic (4.0 K -> of the step's 64 K cycles at 30 waypoints).  Another association of the same sum than the dense product:
// ~1e-16 relative (fixture tolerances: 1e-9).  (Running sums per column — O(n) — were tried first: 18 serial lanes, each step a dependent
// LDS round trip: 6.2 K cycles, slower than the dense product.)
__device__ __forceinline__ void apply_ainv(const double* __restrict__ in, double* __restrict__ out, int n, bool free_end, double dt2) {
    const double c = free_end ? dt2 : dt2 / (double)(n + 1);
    for (int e = threadIdx.x; e < n * 9; e += blockDim.x) {
        const int i = e / 9, d = e - 9 * i;
        const double* col = in + d;
        double pre = 0.0, suf = 0.0;
#pragma unroll 8
        for (int k = 0; k <= i; ++k) pre += (double)(k + 1) * col[9 * k];
        if (free_end) {
#pragma unroll 8
            for (int k = i + 1; k < n; ++k) suf += col[9 * k];
        } else {
#pragma unroll 8
            for (int k = i + 1; k < n; ++k) suf += (double)(n - k) * col[9 * k];
        }
        out[e] = c * ((free_end ? 1.0 : (double)(n - i)) * pre + (double)(i + 1) * suf);
    }
}

}  // namespace

// Optimizer.optimize for scene s; called by all NT threads of a workgroup, smem = the dynamic LDS (host_lds_bytes).
// LIGHT (NT = 256, omg_persist.h): the footprint of a goal-set workgroup — poses read from global memory where the layer workgroups and
// the pose tables left them (waypoint_poses / start_poses / end_pose: all three REQUIRED), the winners' gradient rows parked in
// ChompArgs::light_scratch, no staged potentials, no kinematics here.  Same arithmetic per element in the same order: same bits.
// wait_goal != nullptr (k_update_optimize_split): the goal (end, goal rows, goal point) is being written by ANOTHER
// workgroup; everything that does not need it — FK of start and waypoints, top-k, per-point costs, the winners'
// gradients of all but the last waypoint — runs first, then thread 0 waits for *wait_goal == ticket.
template <int MAXIT, int NT = CH_TPB, bool LIGHT = false>  // MAXIT: items (16-lane groups of potentials) per thread the prefetch loops are unrolled for: ceil(n * 160 / NT) <= MAXIT
__device__ __forceinline__ void chomp_scene(const ChompArgs& a, unsigned char* smem, const int s, const uint32_t* wait_goal = nullptr,
                                            const uint32_t ticket = 0, const omg_learner::LearnerArgs* la = nullptr,
                                            const double* light_end_pose = nullptr /* LIGHT: [10][12] poses of the end configuration */) {
    static_assert(NT == CH_TPB || LIGHT, "the staged form is laid out for CH_TPB threads");
    static_assert(!LIGHT || NT == 256, "the light form is laid out for four waves");
    constexpr int LIMC_T0 = NT >= 512 ? 256 : 232;  // first of the 18 threads that bring the joint limits into LDS (behind the goal block's 128 .. 209)
    // The scene's `active` word is REQUESTED here and tested in front of the first barrier, behind the requests of everything else the
    // workgroup starts from: tested at once it is a trip to memory of its own (~2 us after a launch boundary) ahead of all the others.
    // Until then nothing is written but LDS.
    const int scene_active = a.active ? a.active[s] : 1;
    const omgx_chomp_params& prm = a.prm;
    const int n = prm.n_waypoints, P = prm.n_points, c = prm.constraint_num;
    const double dt = prm.time_interval, dt2 = dt * dt;
    const bool free_end = prm.goal_set_proj != 0;
    Lds L = carve<LIGHT>(smem, n, P, a.pot_in_lds != 0);
    if constexpr (LIGHT) {
        L.gl = a.light_scratch + (size_t)s * n * 80;
        L.gp_start = prm.start_poses + (size_t)s * 120;
        L.gp_wp = prm.waypoint_poses + (size_t)s * n * 120;
        L.gp_end = light_end_pose;
    }
    const RobotView rv(a.robot, P);
    const int tid = threadIdx.x;
    const int total = n * 10 * P;             // reference flat size of potentials [n][10][P]
    const int nitems = n * 160;               // 16-lane groups: item = (i*10 + l)*16 + p
    const float* pot = a.pot + (size_t)s * total;
    const float* pgrad = a.pgrad + (size_t)s * total * 3;
    const float* col = a.col + (size_t)s * total;
    double* traj = a.traj + (size_t)s * n * 9;
    const double* start = a.start + 9 * (size_t)s;
    const double* end = a.end + 9 * (size_t)s;  // (with a ticket: re-pointed at the chosen goal's row of the goal set once it is known)

    // ---------------------------------------------------------------- phase 0: loads + FK
    PHASE_MARK(0);
    // Everything the workgroup starts from is REQUESTED before anything is stored: written as copy loops, every loop was a trip to
    // memory of its own (load, wait, LDS store — six in a row, 8 K of the step's 55 K cycles: tools/phase_timing.py and the ISA).
    // One element per thread and array (two of the trajectory beyond 56 waypoints).
    // (LIGHT: four waves, nothing staged but the trajectory, the collision points and the joint constants — plain loops further down)
    static_assert(LIGHT || (CH_TPB >= 480 && 2 * CH_TPB >= OMGX_MAX_WAYPOINTS * 9), "phase 0 keeps one element per thread of the points / constants and two of the trajectory");
    const double r_xi0 = (!LIGHT && tid < n * 9) ? traj[tid] : 0.0, r_xi1 = (!LIGHT && tid + CH_TPB < n * 9) ? traj[tid + CH_TPB] : 0.0;
    const double r_pts = (!LIGHT && tid < 30 * P) ? rv.pts(0, 0)[tid] : 0.0;
    const int jc_j = tid < 60 ? tid / 6 : 0, jc_k = tid < 60 ? tid % 6 : 0;
    const int jc_off = jc_k < 3 ? 3 * jc_j + jc_k : 30 + 3 * jc_j + jc_k - 3;  // ax | og are neighbours in the robot blob: ONE load behind a
    const double r_jc = rv.ax(0)[jc_off];                                       // selected offset (two loads in a branch wait for each other)
    const double r_fkc = (!LIGHT && tid < 246) ? rv.uvw(0)[tid] : 0.0;  // chain constants: LDS reads instead of scalar loads per joint
    // This thread's potentials / collision flags: all loads are issued back to back (one memory latency instead of
    // one per item); first use is after the FK.  Items it = r * CH_TPB + tid, item = (i*10 + l)*16 + p.
    float pv[MAXIT], cv[MAXIT];
    if (L.potl) {
#pragma unroll
        for (int r = 0; r < MAXIT; ++r) {
            const int it = r * CH_TPB + tid, p = it & 15, grp = it >> 4;
            const bool valid = it < nitems && p < P;
            const int f = valid ? grp * P + p : 0;  // padding lanes read element 0 and are masked at the use sites:
            pv[r] = pot[f];                         // a select here would make the wave wait for the load right away
            cv[r] = col[f];
        }
    }
    // Poses the caller hands over (omgx_chomp_params: the layer launch's waypoint poses, the tabulated start / end poses) are copied
    // HERE, with the loads above: one trip to memory for everything.  What is not handed over is computed below: same code either
    // way, same bits.
    const int ncfg = n + 2;
    const double* wp = prm.waypoint_poses ? prm.waypoint_poses + (size_t)s * n * 120 : nullptr;
    const double* sp = prm.start_poses ? prm.start_poses + (size_t)s * 120 : nullptr;
    const double* ep = LIGHT ? light_end_pose : ((!wait_goal && prm.end_poses) ? prm.end_poses + (size_t)s * 120 : nullptr);
    constexpr int PB = 8;  // 8 x 512 doubles = 34 waypoints' poses per round
    double pvv[PB];
    if (!LIGHT && wp) {
#pragma unroll
        for (int r = 0; r < PB; ++r) {
            const int e = r * CH_TPB + tid;
            pvv[r] = wp[e < n * 120 ? e : 0];
        }
    }
    const double r_sp = (!LIGHT && sp && tid < 120) ? sp[tid] : 0.0, r_ep = (!LIGHT && ep && tid < 120) ? ep[tid] : 0.0;
    // With a ticket: the goal the scene HAD (the learner of this launch may be writing the word: either value will do) — its
    // configuration, rows and poses are fetched SPECULATIVELY below, and kept if the ticket names the same goal (it usually does)
    const int spec_goal = (wait_goal && la) ? min(max(la->goal_idx[s], 0), la->prm.num_goals - 1) : -1;
    PHASE_MARK_T(30, 0);
    // ---- the stores
    if constexpr (LIGHT) {
        for (int e = tid; e < n * 9; e += NT) L.xi[e] = traj[e];
        for (int e = tid; e < 30 * P; e += NT) L.pts[e] = rv.pts(0, 0)[e];
    } else {
        if (tid < n * 9) L.xi[tid] = r_xi0;
        PHASE_MARK_T(31, 0);
        if (tid + CH_TPB < n * 9) L.xi[tid + CH_TPB] = r_xi1;
        if (tid < 30 * P) L.pts[tid] = r_pts;
        if (tid < 246) L.fkc[tid] = r_fkc;
    }
    if (tid < 60) L.jconst[tid] = r_jc;
    for (int e = tid; e < 256; e += blockDim.x) L.hist[e] = 0;
    for (int e = tid; e < (nitems + 31) / 32; e += blockDim.x) L.tie[e] = 0;
    if (tid == 0) { L.iscr[2] = 0; L.red[7] = 0.0; L.red[50] = 0.0; L.red[51] = 0.0; L.red[53] = 0.0; L.red[54] = 0.0; L.red[55] = 0.0; }
    if (!LIGHT && wp) {
#pragma unroll
        for (int r = 0; r < PB; ++r) {
            const int e = r * CH_TPB + tid;
            if (e < n * 120) L.pose[120 + e] = pvv[r];
        }
        for (int e = PB * CH_TPB + tid; e < n * 120; e += CH_TPB) L.pose[120 + e] = wp[e];  // beyond 34 waypoints
    }
    if (!LIGHT && sp && tid < 120) L.pose[tid] = r_sp;
    if (!LIGHT && ep && tid < 120) L.pose[(size_t)(ncfg - 1) * 120 + tid] = r_ep;
    if (scene_active == 0) return;  // workgroup-uniform, before any barrier and any global write
    __syncthreads();
    PHASE_MARK_T(16, CH_TPB - 128);
    // FK of start, waypoints, end (cost.py:124-165) in two stages (omg_device.h: fk_chain_row); the (sin, cos) table
    // borrows L.gl, which is first written in phase 2/3.
    double* sc = L.gl;  // [ncfg][7][2]
    auto config_of = [&](int cfg) { return cfg == 0 ? start : (cfg == ncfg - 1 ? end : L.xi + 9 * (cfg - 1)); };
    const RobotView rvl(a.robot, P, L.fkc);
    auto fk_configs = [&](int c_begin, int c_end, double* tab) {  // tab [c_end - c_begin][7][2]; two barriers inside: call from all threads
        const int nc = c_end - c_begin;
        for (int t = tid; t < nc * 7; t += blockDim.x) {
            const int cfg = c_begin + t / 7, i = t % 7;
            double sn, cs;
            fk_joint_sincos(config_of(cfg)[i], sn, cs);
            tab[2 * t] = sn; tab[2 * t + 1] = cs;
        }
        __syncthreads();
        for (int t = tid; t < nc * 3; t += blockDim.x) {
            const int cfg = c_begin + t / 3, r = t % 3;
            const double* q = config_of(cfg);
            double* dst0 = L.pose + (size_t)cfg * 120 + 3 * r;
            fk_chain_row(rvl, r, tab + 14 * (cfg - c_begin), q[7], q[8], [&](int l, double r0, double r1, double r2, double tr) {
                double* dst = dst0 + 12 * l;
                dst[0] = r0; dst[1] = r1; dst[2] = r2;
                dst[9 - 2 * r] = tr;  // element 9 + r of the pose
            });
        }
        __syncthreads();
    };
    // What was not handed over (copied above, in front of the barrier).  The end configuration is the goal: later, if it is not known yet.
    if constexpr (!LIGHT) {  // (LIGHT: every pose is handed over — no kinematics, no chain constants in LDS)
        const bool need_end = !wait_goal && !ep;
        if (!wp) fk_configs(sp ? 1 : 0, (need_end ? ncfg : ncfg - 1), sc);  // start (unless given) + waypoints (+ end)
        else {
            if (!sp) fk_configs(0, 1, sc);
            if (need_end) fk_configs(ncfg - 1, ncfg, sc);
        }
    }
    PHASE_MARK_T(17, CH_TPB - 128);
    PHASE_MARK_T(20, 0);
    double colsum = 0.0;
    if (L.potl) {
#pragma unroll
        for (int r = 0; r < MAXIT; ++r) {  // same order as the per-item loop; padding adds +0.0
            const int it = r * CH_TPB + tid;
            const bool valid = it < nitems && (it & 15) < P;
            asm volatile("" : "+v"(pv[r]), "+v"(cv[r]));  // first use stays here: no wait for the loads before the FK
            pv[r] = valid ? pv[r] : 0.0f;
            colsum += valid ? (double)cv[r] : 0.0;
        }
    }
    // ---------------------------------------------------------------- phase 1: top-k threshold (cost.py:392-398)
    PHASE_MARK(1);
    // Radix select of the K-th largest key over the `total` potentials, up to 4 passes of 8 bits.
    const int K = prm.top_k;
    const bool topk_mode = K > 0;
    uint32_t tau = 0;      // key of the K-th largest potential; keys > tau are selected outright
    int tie_take = 0;      // how many keys == tau are selected (those with the highest flat index)
    bool tau_is_zero = false;
    if (topk_mode && K < total) {
        // Most potentials are exactly 0 (points out of every object's reach).  They are the smallest keys and add nothing to cost or
        // gradient: only non-zero keys are counted, and if no more than K of them exist every one is selected (the common case with
        // K = 1000) — known after the FIRST histogram, whose bins add up to their number.
        // Two histograms take turns: while wave 0 scans the one just filled, the other
        // waves clear the one the next pass fills — two barriers per pass instead of four, and no counting pass in front.
        const uint32_t key0 = float_key(0.0f);
        uint32_t* const hbuf[2] = {L.hist /* cleared in phase 0 */, L.hist + 256};
        uint32_t prefix = 0, mask = 0;
        int want = K;  // rank (from the top) still to locate inside the current prefix bucket
        int nz = 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            uint32_t* const H = hbuf[pass & 1];
            if (L.potl) {  // the keys are in registers already (padding lanes hold 0.0f = key0: never counted)
#pragma unroll
                for (int r = 0; r < MAXIT; ++r) {
                    const uint32_t key = float_key(pv[r]);
                    if (key > key0 && (key & mask) == prefix) atomicAdd(&H[(key >> shift) & 255u], 1u);
                }
            } else {
                for (int f = tid; f < total; f += blockDim.x) {
                    const uint32_t key = float_key(pot[f]);
                    if (key > key0 && (key & mask) == prefix) atomicAdd(&H[(key >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            if (tid < 64) {  // wave 0: locate the bin holding the `want`-th largest key (suffix sums over 256 bins)
                const uint32_t h0 = H[4 * tid], h1 = H[4 * tid + 1], h2 = H[4 * tid + 2], h3 = H[4 * tid + 3];
                const int minel = (int)(h0 + h1 + h2 + h3);
                const int incl = wave_suffix_sum_i32(minel);  // inclusive suffix sum over lanes tid..63
                if (pass == 0 && tid == 0) L.iscr[2] = incl;  // all non-zero keys
                const int above = incl - minel;
                if (above < want && want <= incl) {  // exactly one lane (none when there are fewer than `want` keys: first pass only)
                    int acc = above, b = 4 * tid + 3;
                    const uint32_t hh[4] = {h0, h1, h2, h3};
                    for (; b > 4 * tid; --b) {
                        if (acc + (int)hh[b - 4 * tid] >= want) break;
                        acc += (int)hh[b - 4 * tid];
                    }
                    L.iscr[0] = b;
                    L.iscr[1] = want - acc;
                }
            } else {
                uint32_t* const Hn = hbuf[(pass + 1) & 1];
                for (int e = tid - 64; e < 256; e += blockDim.x - 64) Hn[e] = 0;
            }
            __syncthreads();
            if (pass == 0) {
                nz = L.iscr[2];
                if (nz <= K) break;  // workgroup-uniform
            }
            prefix |= (uint32_t)L.iscr[0] << shift;
            mask |= 255u << shift;
            want = L.iscr[1];
        }
        if (nz <= K) {
            tau = key0;
            tau_is_zero = true;
            tie_take = K - nz;
        } else {
            tau = prefix;
            tie_take = want;  // >= 1
            tau_is_zero = false;
        }
    }
    PHASE_MARK_T(18, CH_TPB - 128);
    PHASE_MARK_T(21, 0);
    if (L.potl) {
#pragma unroll
        for (int r = 0; r < MAXIT; ++r) {
            const int it = r * CH_TPB + tid;
            if (it < nitems) L.potl[it] = pv[r];
        }
    }
    PHASE_MARK_T(22, 0);
    __syncthreads();  // FK results visible
    PHASE_MARK_T(19, CH_TPB - 128);
    // Ties at a non-zero threshold: the reference keeps whichever numpy's unstable argsort placed last;
    // this build (like the oracle) defines it as the highest flat indices.  Mark them in a bit mask.
    if (topk_mode && K < total && !tau_is_zero) {
        if (L.potl) {  // from the registers; a non-zero threshold never matches a padding lane's 0.0f
#pragma unroll
            for (int r = 0; r < MAXIT; ++r) {
                const int it = r * CH_TPB + tid, f = (it >> 4) * P + (it & 15);
                if (float_key(pv[r]) == tau) atomicOr(&L.tie[f >> 5], 1u << (f & 31));
            }
        } else {
            for (int f = tid; f < total; f += blockDim.x)
                if (float_key(pot[f]) == tau) atomicOr(&L.tie[f >> 5], 1u << (f & 31));
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- phase 2: per point
    PHASE_MARK(2);
    const int mlinks = topk_mode ? (prm.consider_finger ? 10 : 8) : 10;  // cost.py:401-404
    const int i_defer = wait_goal ? n - 1 : n;  // waypoints >= i_defer need the end pose (acceleration): second pass
    double* const goalc = reinterpret_cast<double*>(L.hist);  // [9 + c * 9] goal point | goal rows; the histogram is dead after phase 1 (c <= 8: 81 of its 128 doubles)
    double* const limc = goalc + 96;                          // [18] joint limits (lower | upper): LDS reads in the totals and the limit loop instead of global loads
    auto fetch_goal = [&](const int gi) {  // goal point | goal rows, joint limits, the end configuration's poses (if tabulated) -> LDS; no barrier inside
        const int GS_ = la->prm.num_goals;
        const double* const grow = la->goal_set + ((size_t)s * GS_ + gi) * 9;
        if (tid >= 128 && tid < 128 + 9 + c * 9) {
            const int e = tid - 128;
            goalc[e] = e < 9 ? grow[e] : (la->prm.use_standoff ? la->reach[((size_t)s * GS_ + gi) * c * 9 + (e - 9)] : grow[(e - 9) % 9]);
        }
        if (tid >= LIMC_T0 && tid < LIMC_T0 + 18) limc[tid - LIMC_T0] = tid < LIMC_T0 + 9 ? rv.lower()[tid - LIMC_T0] : rv.upper()[tid - LIMC_T0 - 9];
        if constexpr (LIGHT) L.gp_end = la->prm.goal_pose_table + ((size_t)s * GS_ + gi) * 120;  // (LIGHT needs the table)
        else if (la->prm.goal_pose_table && tid < 120) L.pose[(size_t)(ncfg - 1) * 120 + tid] = la->prm.goal_pose_table[((size_t)s * GS_ + gi) * 120 + tid];
    };
    if (spec_goal >= 0) fetch_goal(spec_goal);  // (the histogram these words shared is dead; the end pose's slot has no reader before the ticket)
    auto phase2_item = [&](const int it, const bool second_pass) {
        const bool inb = it < nitems;
        const int p = it & 15, grp = it >> 4, l = grp % 10, i = grp / 10;
        const bool valid = inb && p < P;
        const int f = valid ? (i * 10 + l) * P + p : 0;
        float cf;
        if (L.potl) cf = inb ? L.potl[it] : 0.0f;
        else {
            cf = valid ? pot[f] : 0.0f;
            if (valid && !second_pass) colsum += (double)col[f];
        }
        double contrib = 0.0;
        if (topk_mode) {
            bool sel = false;
            if (valid && l < mlinks && cf != 0.0f) {  // zero potentials add nothing to cost or gradient
                if (K >= total) sel = true;
                else {
                    const uint32_t key = float_key(cf);
                    if (key > tau) sel = true;
                    else if (key == tau) {  // rank among ties counted from the highest index
                        int above = 0;
                        const int w0 = f >> 5;
                        above += __popc(L.tie[w0] >> (f & 31)) - 1;
                        for (int w = w0 + 1; w < (total + 31) / 32; ++w) above += __popc(L.tie[w]);
                        sel = above < tie_take;
                    }
                }
            }
            if (!__any(sel)) {  // nothing selected in this wave's four groups (the common case): empty groups
                if (inb && p == 0) { L.gwin[grp] = -1; L.gcost[grp] = 0.0; }
                return;
            }
            double vn = 0.0;
            if (sel) {
                double x[3], xm[3];
                point_at<LIGHT>(L, i + 1, l, p, P, x);
                point_at<LIGHT>(L, i, l, p, P, xm);
                const double v0 = (x[0] - xm[0]) / dt, v1 = (x[1] - xm[1]) / dt, v2 = (x[2] - xm[2]) / dt;
                vn = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
                contrib = (double)cf * vn;
            }
            // group winner = selected point with the largest potential, ties -> largest p
            // ("last write wins" of the buffered fancy-index +=, cost.py:421)
            float bv = sel ? cf : -1.0f;
            int bp = sel ? p : -1;
#define OMG_ARGMAX_STEP(CTRL)                                                   \
    {                                                                           \
        const float ov = dpp_f32<CTRL>(bv);                                     \
        const int op = dpp_i32<CTRL>(bp);                                       \
        if (ov > bv || (ov == bv && op > bp)) { bv = ov; bp = op; }             \
    }
            OMG_ARGMAX_STEP(0xB1) OMG_ARGMAX_STEP(0x4E) OMG_ARGMAX_STEP(0x141) OMG_ARGMAX_STEP(0x140)
#undef OMG_ARGMAX_STEP
            contrib = row16_allsum(contrib);
            if (inb && p == 0) { L.gwin[grp] = bp; L.gcost[grp] = contrib; }
        } else {
            // clean branch (cost.py:380-388): every point of every link contributes J.g
            double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (valid && (second_pass || i < i_defer)) {
                double x[3], v[3], acc[3], g[3];
                point_kinematics<LIGHT>(L, i, l, p, P, dt, x, v, acc);
                const double dc[3] = {(double)pgrad[3 * f], (double)pgrad[3 * f + 1], (double)pgrad[3 * f + 2]};
                contrib = functional_g(v, acc, (double)cf, dc, g);
                const int nk = njoints(l);
                for (int k = 0; k < nk; ++k) out[k] = jacobian_dot<LIGHT>(L, i, l, k, x, g);
            }
            contrib = row16_allsum(contrib);
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] = row16_allsum(out[k]);
            if (inb && p == 0) {
                L.gcost[grp] = contrib;
#pragma unroll
                for (int k = 0; k < 8; ++k) L.gl[(size_t)grp * 8 + k] = out[k];
            }
        }
    };
    for (int it0 = 0; it0 < nitems; it0 += blockDim.x) phase2_item(it0 + tid, false);
    {   // collide.sum() over the layer output (cost.py:187): per-thread partials -> wave sums (block sum in phase 5)
        const double wsum = wave_allsum(colsum);
        if ((tid & 63) == 0) L.red[56 + (tid >> 6)] = wsum;
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 3: winners' gradients (top-k branch)
    PHASE_MARK(3);
    // Few of the n x 10 (waypoint, link) groups have a winner, and a winner's J_k . g for its up to 8 joints are independent: the
    // groups with a winner are compacted into L.wlist (order irrelevant: every (group, k) element is computed on its own) and
    // then served by 8 lanes each, lane k -> joint slot k, instead of one lane walking the 8 slots of its group (a chain of
    // ~200 dependent LDS reads: 13 K of the step's 68 K cycles).  Same arithmetic per element, same bits.
    auto winners_gradients = [&](const int g_begin, const int g_end) {  // two barriers inside: call from all threads
        if ((g_end - g_begin) * 8 <= (int)blockDim.x) {
            // a range that fits the workgroup at 8 lanes per group (the last waypoint's ten groups, after the goal has arrived): no
            // compaction — the same element by the same arithmetic, without the counter and its two barriers
            const int grp = g_begin + (tid >> 3), k = tid & 7;
            if (grp < g_end) {
                const int p = L.gwin[grp];
                double val = 0.0;
                if (p >= 0) {
                    const int l = grp % 10, i = grp / 10;
                    const int f = (i * 10 + l) * P + p;
                    double x[3], v[3], acc[3], g[3];
                    point_kinematics<LIGHT>(L, i, l, p, P, dt, x, v, acc);
                    const double dc[3] = {(double)pgrad[3 * f], (double)pgrad[3 * f + 1], (double)pgrad[3 * f + 2]};
                    functional_g(v, acc, (double)(L.potl ? L.potl[(grp << 4) + p] : pot[f]), dc, g);
                    val = k < njoints(l) ? jacobian_dot<LIGHT>(L, i, l, k, x, g) : 0.0;
                }
                L.gl[(size_t)grp * 8 + k] = val;
            }
            return;
        }
        if (tid == 0) L.iscr[3] = 0;
        __syncthreads();
        for (int grp = g_begin + tid; grp < g_end; grp += blockDim.x) {
            if (L.gwin[grp] >= 0) L.wlist[atomicAdd(&L.iscr[3], 1)] = grp;
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) L.gl[(size_t)grp * 8 + k] = 0.0;
            }
        }
        __syncthreads();
        const int cnt = L.iscr[3];
        for (int idx = tid >> 3; idx < cnt; idx += blockDim.x >> 3) {
            const int grp = L.wlist[idx], k = tid & 7;
            const int l = grp % 10, i = grp / 10;
            const int p = L.gwin[grp];
            const int f = (i * 10 + l) * P + p;
            double x[3], v[3], acc[3], g[3];
            point_kinematics<LIGHT>(L, i, l, p, P, dt, x, v, acc);
            const double dc[3] = {(double)pgrad[3 * f], (double)pgrad[3 * f + 1], (double)pgrad[3 * f + 2]};
            functional_g(v, acc, (double)(L.potl ? L.potl[(grp << 4) + p] : pot[f]), dc, g);
            L.gl[(size_t)grp * 8 + k] = k < njoints(l) ? jacobian_dot<LIGHT>(L, i, l, k, x, g) : 0.0;
        }
    };
    if (topk_mode) {
        winners_gradients(0, i_defer * 10);
        __syncthreads();
    }
    PHASE_MARK_T(9, 0);
    const double* w = prm.link_smooth_weight;
    // (t0, nt): the calling threads' index and number — the whole workgroup, or a group of its waves beside another group's work
    auto obstacle_rows = [&](const int e_begin, const int e_end, const int t0, const int nt) {  // obstacle gradient [n][9] from the groups' J.g
        for (int e = e_begin + t0; e < e_end; e += nt) {
            const int i = e / 9, d = e % 9;
            // all ten reads first, then the additions in ascending link order (= the reference's += order) behind selects: with a branch
            // per link the loop was read, wait, add ten times over
            double gv[10];
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                int k = -1;
                if (d < 7) { if (d < njoints(l)) k = d; }
                else if (l == d + 1) k = 7;  // column 7 <- link 8, column 8 <- link 9
                gv[l] = L.gl[((size_t)i * 10 + l) * 8 + (k >= 0 ? k : 0)];
            }
            double sgrad = 0.0;
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                int k = -1;
                if (d < 7) { if (d < njoints(l)) k = d; }
                else if (l == d + 1) k = 7;
                const double next = sgrad + gv[l];
                sgrad = (l < mlinks && k >= 0) ? next : sgrad;
            }
            L.og[e] = sgrad;
        }
    };
    auto smooth_elements = [&](const int t0, const int nt) {  // needs the end configuration only for a fixed end (not goal-set mode)
        for (int e = t0; e < n * 9; e += nt) {
            const int i = e / 9, d = e % 9;
            // compute_smooth_loss gradient: A xi + D1^T ed (cost.py:447-448)
            const double xc = L.xi[e];
            const double xm = i > 0 ? L.xi[e - 9] : start[d];
            double sm;
            if (i < n - 1) sm = (2.0 * xc - xm - L.xi[e + 9]) / dt2;
            else sm = free_end ? (xc - xm) / dt2 : (2.0 * xc - xm - end[d]) / dt2;
            L.sg[e] = sm * w[d];
        }
        // smoothness loss rows 0..n (cost.py:430-445).  The 9 weighted velocities of a row (two divisions each) are computed by 9
        // lanes — one lane per row walked them in turn: 6 K of the step's 68 K cycles — and parked in L.tv (+ the first row of
        // L.tvs behind it: both are free until phase 5); the row's lane then adds their squares in joint order, as before.
        double* const evs = L.tv;  // [(n + 1)][9]
        for (int e = t0; e < (n + 1) * 9; e += nt) {
            const int i = e / 9, d = e % 9;
            double vel;
            if (i == 0) vel = L.xi[d] / dt + (-1.0 * start[d] / dt);
            else if (i < n) vel = (L.xi[i * 9 + d] - L.xi[(i - 1) * 9 + d]) / dt;
            else vel = free_end ? 0.0 : (-L.xi[(n - 1) * 9 + d] / dt + end[d] / dt);
            evs[e] = vel * w[d];
        }
    };
    auto smooth_rows = [&]() {  // behind a barrier after smooth_elements
        const double* const evs = L.tv;
        for (int i = tid; i <= n; i += blockDim.x) {
            double s2 = 0.0;
            for (int d = 0; d < 9; ++d) {
                const double ev = evs[i * 9 + d];
                s2 += ev * ev;
            }
            const double nrm = sqrt(s2);
            L.sml[i] = 0.5 * nrm * nrm;
        }
    };
    auto smooth_terms = [&]() {
        smooth_elements(tid, blockDim.x);
        __syncthreads();
        smooth_rows();
    };
    if (wait_goal) {
        // everything that does not involve the goal, before waiting for it.  The [n][9] loops keep n * 9 / 64 waves busy (5 of 8 at 30
        // waypoints): here the obstacle rows on three waves and the smoothness elements (two loops with a division or two per element:
        // the longer half) on the other five at the same time
        if (free_end) {
            constexpr int SPLIT_AT = 3 * 64;
            if (tid < SPLIT_AT) obstacle_rows(0, i_defer * 9, tid, SPLIT_AT);
            else smooth_elements(tid - SPLIT_AT, NT - SPLIT_AT);
            PHASE_MARK_T(10, 0);
            __syncthreads();
            smooth_rows();
        } else {
            obstacle_rows(0, i_defer * 9, tid, blockDim.x);
            PHASE_MARK_T(10, 0);
        }
        PHASE_MARK_T(11, 0);
        // ------------------------------------------------------------ the goal: wait for the learner's workgroup
        PHASE_MARK_T(23, 0);
        // The rendezvous word carries the chosen goal: (ticket << 8) | index, stored relaxed by the learner's workgroup the moment
        // the index is known.  Everything else of the goal — goal point, goal rows, the end configuration and its poses — is read
        // HERE from the goal set / standoff / pose tables, which no launch in flight writes: no acquire, no cache invalidation, and
        // the learner's own stores of the goal (for the launches to come) are off the iteration's critical path.
        if (tid == 0) {
            uint32_t word = __hip_atomic_load(wait_goal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((word >> 8) != ticket) {
                const long long t0 = wall_clock64();  // 100 MHz
                do {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > 200000000LL) {  // 2 s: the producer is gone; fail loudly instead of hanging the device
                        L.red[53] = 1.0;  // the wait ran out: the thread that writes `info` reports NaN
                        word = 0u;        // (goal 0: the arithmetic below stays in bounds)
                        break;
                    }
                    word = __hip_atomic_load(wait_goal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((word >> 8) != ticket);
            }
            static_assert(OMGX_MAX_GOALS <= 256, "the rendezvous word is (ticket << 8) | goal index: the index must fit 8 bits");
            L.iscr[3] = (int)(word & 0xffu);
        }
        __syncthreads();
        PHASE_MARK_T(24, 0);
        const int GS_ = la->prm.num_goals;
        const int gi = min(L.iscr[3], GS_ - 1);
        end = la->goal_set + ((size_t)s * GS_ + gi) * 9;  // the chosen goal's configuration
        if (gi != spec_goal) {  // (workgroup-uniform) the goal changed in this iteration: fetch again
            fetch_goal(gi);
            if (!LIGHT && la->prm.goal_pose_table) __syncthreads();
        }
        if constexpr (!LIGHT)
            if (!la->prm.goal_pose_table) fk_configs(ncfg - 1, ncfg, L.red + 8);  // no table: the end configuration's kinematics here (red[8..21] is free scratch)
        PHASE_MARK_T(28, 0);
        if (topk_mode) {
            winners_gradients(i_defer * 10, n * 10);
        } else {
            for (int it0 = i_defer * 160; it0 < nitems; it0 += blockDim.x) phase2_item(it0 + tid, true);
        }
        PHASE_MARK_T(29, 0);
        __syncthreads();
        PHASE_MARK_T(25, 0);
    }

    // The top-k branch's link sums — each link's cost summed over the waypoints, in waypoint order (cost.py:416): a chain of n dependent
    // additions on ten lanes, 3.4 K cycles at 30 waypoints — run in THREE pieces on a wave that has no element of the [n][9] loops
    // (up to 42 waypoints), beside the other waves' deferred rows, weighting and block sums: the whole workgroup used to wait for them
    // at a barrier.  Same additions in the same order.
    constexpr int LSW = NT / 64 - 2;  // the wave (512 threads: 6)
    double ls_cl = 0.0;
    int ls_any = 0;
    const int ls_piece = n >= 24 ? 8 * (n / 24) : n;  // whole batches of 8 waypoints for the first two pieces (30 waypoints: 8 + 8 + 14)
    auto link_sum_piece = [&](const int i_begin, const int i_end_) {
        const int i_end = i_end_ < n ? i_end_ : n;
        if (!(topk_mode && (tid >> 6) == LSW && (tid & 63) < 10)) return;
        const int ln = tid & 63;
        for (int i0 = i_begin; i0 < i_end; i0 += 8) {  // the LDS reads of 8 waypoints at once, then the additions behind selects
            double gv[8];
            int gw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j < i_end ? i0 + j : i_end - 1;
                gv[j] = L.gcost[i * 10 + ln]; gw[j] = L.gwin[i * 10 + ln];
            }
            if (i0 + 8 <= i_end) {  // a whole batch: the bare chain of additions
#pragma unroll
                for (int j = 0; j < 8; ++j) { ls_cl += gv[j]; ls_any |= gw[j] >= 0 ? 1 : 0; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = i0 + j < i_end;
                    const double next = ls_cl + gv[j];
                    ls_cl = in ? next : ls_cl;
                    ls_any |= (in && gw[j] >= 0) ? 1 : 0;
                }
            }
        }
    };

    // ---------------------------------------------------------------- phase 4: obstacle gradient [n][9], smoothness
    PHASE_MARK(4);
    link_sum_piece(0, ls_piece);
    if (!wait_goal && tid >= 128 && tid < 128 + 9 + c * 9) {  // (with a ticket: loaded right behind it, above)
        const int e = tid - 128;
        goalc[e] = e < 9 ? a.goal_point[9 * (size_t)s + e] : a.goal[(size_t)s * c * 9 + (e - 9)];
    }
    if (!wait_goal && tid >= LIMC_T0 && tid < LIMC_T0 + 18) limc[tid - LIMC_T0] = tid < LIMC_T0 + 9 ? rv.lower()[tid - LIMC_T0] : rv.upper()[tid - LIMC_T0 - 9];
    if (wait_goal) {  // only what the goal-dependent passes above produced is still missing
        obstacle_rows(i_defer * 9, n * 9, tid, blockDim.x);
        if (!free_end) smooth_terms();
    } else {
        obstacle_rows(0, n * 9, tid, blockDim.x);
        smooth_terms();
    }
    __syncthreads();
    if (a.aux) {  // un-weighted pieces: obs_grad | obs_cost | smooth_grad | smooth_loss
        double* ax = a.aux + (size_t)s * (n * 9 + n * 10 + n * 9 + n + 1);
        for (int e = tid; e < n * 9; e += blockDim.x) { ax[e] = L.og[e]; ax[n * 19 + e] = L.sg[e]; }
        for (int i = tid; i <= n; i += blockDim.x) ax[n * 28 + i] = L.sml[i];
    }

    // ---------------------------------------------------------------- phase 5: totals (cost.py:464-530)
    PHASE_MARK(5);
    link_sum_piece(ls_piece, 2 * ls_piece);
    for (int e = tid; e < n * 9; e += blockDim.x) {
        double og = prm.obstacle_weight * L.og[e];
        og = fmin(fmax(og, -prm.clip_grad_scale), prm.clip_grad_scale);
        const double sgw = prm.smoothness_weight * L.sg[e];
        const double gt = og + sgw;
        L.og[e] = og * og; L.sg[e] = sgw * sgw; L.g[e] = gt; L.tv[e] = gt * gt;
        a.grad[(size_t)s * n * 9 + e] = gt;
    }
    // check_joint_limit (optimizer.py:166-174): flags only when a low AND a high violation exist
    const double* lower = limc;
    const double* upper = limc + 9;
    {
        bool lowv = false, highv = false;
        for (int e = tid; e < n * 9; e += blockDim.x) {
            const int d = e % 9;
            lowv = lowv || (L.xi[e] < lower[d] - 5e-3);
            highv = highv || (L.xi[e] > upper[d] + 5e-3);
        }
        if (lowv) L.red[50] = 1.0;   // benign same-value races; both were zeroed in phase 0
        if (highv) L.red[51] = 1.0;
    }
    PHASE_MARK_T(12, 0);
    __syncthreads();
    PHASE_MARK_T(37, 0);
    // The independent block sums run on different waves at once, each in wave_allsum's fixed order.
    {
        const int wv = tid >> 6, ln = tid & 63;
        const double* arr = wv == 0 ? L.sml : (wv == 1 ? L.og : (wv == 2 ? L.sg : (wv == 3 ? L.tv : L.gcost)));
        const int count = wv == 0 ? n + 1 : (wv <= 3 ? n * 9 : n * 10);
        if constexpr (NT >= 512) {
            if (wv <= 3 || (wv == 4 && !topk_mode)) {  // (wave 4: the clean branch's sum of all group costs)
                double acc = 0.0;
                for (int k = ln; k < count; k += 64) acc += arr[k];
                acc = wave_allsum(acc);
                if (ln == 0) L.red[wv == 4 ? 0 : wv + 1] = acc;  // slots: 1 smooth, 2 |w og|^2, 3 |w sg|^2, 4 |g|^2, 0 obstacle (clean branch)
            } else if (wv == LSW && topk_mode && ln < 10) {
                PHASE_MARK_T(38, 384);
                link_sum_piece(2 * ls_piece, n);
                L.red[32 + ln] = (ln < mlinks && ls_any) ? ls_cl : 0.0;  // each link's summed cost is broadcast to every waypoint (cost.py:416)
                PHASE_MARK_T(39, 384);
            }
        } else {  // four waves: each sums its array (one wave per array, 64 lanes: the same order of additions), then wave 0 / wave LSW go on
            {
                double acc = 0.0;
                for (int k = ln; k < count; k += 64) acc += arr[k];
                acc = wave_allsum(acc);
                if (ln == 0) L.red[wv + 1] = acc;
            }
            if (wv == 0 && !topk_mode) {
                double acc = 0.0;
                for (int k = ln; k < n * 10; k += 64) acc += L.gcost[k];
                acc = wave_allsum(acc);
                if (ln == 0) L.red[0] = acc;
            }
            if (wv == LSW && topk_mode && ln < 10) {
                link_sum_piece(2 * ls_piece, n);
                L.red[32 + ln] = (ln < mlinks && ls_any) ? ls_cl : 0.0;
            }
        }
        PHASE_MARK_T(32, 0); PHASE_MARK_T(33, 64); PHASE_MARK_T(34, 128); PHASE_MARK_T(35, 192); PHASE_MARK_T(36, 384);
    }
    // What the sums feed — cost_traj, info — is written BEHIND the projected step's `A^-1 g`, which needs none of them (one barrier less
    // on the way to the new trajectory).
    auto outputs_of_the_sums = [&]() {  // behind a barrier after the sums
        double obs_sum;
        if (topk_mode) {
            double per_wp = 0.0;
            for (int l = 0; l < 10; ++l) per_wp += L.red[32 + l];
            obs_sum = per_wp * (double)n;
            // (the weighted obstacle term rounded on its own, then ONE fused multiply-add: written out, because which of the two products
            // of `a b + c d` the compiler fuses depends on the code around it — the four-wave build chose the other one, 1 ulp)
            const double w_obs_wp = prm.obstacle_weight * per_wp;
            for (int i = tid; i < n; i += blockDim.x)
                a.cost_traj[(size_t)s * n + i] = __builtin_fma(prm.smoothness_weight, L.sml[i], w_obs_wp);
            if (a.aux) {
                double* oc = a.aux + (size_t)s * (n * 9 + n * 10 + n * 9 + n + 1) + n * 9;
                for (int e = tid; e < n * 10; e += blockDim.x) oc[e] = L.red[32 + e % 10];
            }
        } else {
            obs_sum = L.red[0];
            for (int i = tid; i < n; i += blockDim.x) {
                double r = 0.0;
                for (int l = 0; l < 10; ++l) r += L.gcost[i * 10 + l];
                const double w_obs_r = prm.obstacle_weight * r;
                a.cost_traj[(size_t)s * n + i] = __builtin_fma(prm.smoothness_weight, L.sml[i], w_obs_r);
            }
            if (a.aux) {
                double* oc = a.aux + (size_t)s * (n * 9 + n * 10 + n * 9 + n + 1) + n * 9;
                for (int e = tid; e < n * 10; e += blockDim.x) oc[e] = L.gcost[e];
            }
        }
        const double smooth_sum = L.red[1], n_og = L.red[2], n_sg = L.red[3], n_g = L.red[4];
        double collide = 0.0;
        for (int wv = 0; wv < NT / 64; ++wv) collide += L.red[56 + wv];

        // The scalars of `info` (four square roots, a norm, 16 stores: 2.6 K cycles of one thread) are written by a thread of the LAST wave,
        // which has no element of the [n][9] loops below: the other waves go on to the projected step meanwhile (they used to wait for
        // thread 0 at the next barrier).  Thread 0's bounded wait reports a failure through L.red[53] (written before the barriers above).
        if (tid == NT - 64) {
            const bool wait_failed = L.red[53] > 0.0;
            double goal_dist = 0.0;
            if (prm.goal_set_proj) {
                const double* gp = goalc;
                for (int d = 0; d < 9; ++d) { const double e = L.xi[(n - 1) * 9 + d] - gp[d]; goal_dist += e * e; }
                goal_dist = sqrt(goal_dist);
            }
            const bool violate = (L.red[50] > 0.0) && (L.red[51] > 0.0);
            const double w_obs = prm.obstacle_weight * obs_sum, w_sm = prm.smoothness_weight * smooth_sum;
            bool terminate = (collide <= prm.allow_collision_point) && prm.pre_terminate && (goal_dist < 0.01) &&
                             (smooth_sum < prm.terminate_smooth_loss);
            terminate = terminate && !violate;
            const bool failure = (collide >= prm.allow_collision_point * 10) || (smooth_sum >= prm.terminate_smooth_loss * 2.5);
            const bool execute = (collide <= prm.allow_collision_point) && (smooth_sum < prm.terminate_smooth_loss);
            double* info = a.info + (size_t)s * OMGX_INFO_STRIDE;
            info[OMGX_INFO_COST] = wait_failed ? __builtin_nan("") : w_obs + w_sm;  // a goal that never arrived must not look like a result
            info[OMGX_INFO_OBS] = obs_sum;
            info[OMGX_INFO_SMOOTH] = smooth_sum;
            info[OMGX_INFO_WEIGHTED_OBS] = w_obs;
            info[OMGX_INFO_WEIGHTED_SMOOTH] = w_sm;
            info[OMGX_INFO_WEIGHTED_OBS_GRAD] = sqrt(n_og);
            info[OMGX_INFO_WEIGHTED_SMOOTH_GRAD] = sqrt(n_sg);
            info[OMGX_INFO_GRAD] = sqrt(n_g);
            info[OMGX_INFO_COLLIDE] = collide;
            info[OMGX_INFO_REACH] = goal_dist;
            info[OMGX_INFO_TERMINATE] = terminate ? 1.0 : 0.0;
            if (a.deactivate && terminate) a.deactivate[s] = 0;  // read again only by later launches on this stream
            info[OMGX_INFO_FAILURE_TERMINATE] = failure ? 1.0 : 0.0;
            info[OMGX_INFO_EXECUTE] = execute ? 1.0 : 0.0;
            info[OMGX_INFO_STANDOFF_IDX] = prm.use_standoff ? (double)(n - c) : (double)(n - 1);
            info[OMGX_INFO_VIOLATE_LIMIT] = violate ? 1.0 : 0.0;
            info[OMGX_INFO_LIMIT_STEPS] = 0.0;
            L.red[52] = terminate ? 1.0 : 0.0;
        }
    };
    if (prm.do_update != 1) {  // evaluation only, or a step that depends on `terminate`: the old order
        __syncthreads();
        PHASE_MARK_T(13, 0);
        outputs_of_the_sums();
        PHASE_MARK_T(14, 0);
        if (!prm.do_update) return;
        __syncthreads();
        if (L.red[52] > 0.0) return;  // Optimizer.optimize without force_update: a terminated trajectory is left alone
    }

    // ---------------------------------------------------------------- phase 6: covariant (projected) step
    PHASE_MARK(6);
    // (no barrier here: L.g is complete since the barrier before the block sums, L.tvs has no reader left, and `info`'s thread reads
    // L.xi / L.red, which change only behind the barriers below)
    apply_ainv(L.g, L.tvs, n, free_end, dt2);  // Ag = Ainv g
    __syncthreads();
    PHASE_MARK_T(15, 0);
    if (prm.do_update == 1) outputs_of_the_sums();  // (their inputs: L.red, L.xi's last row, goalc — all unchanged until the barriers below)
    const double eta = prm.step_size;
    const double* goal = goalc + 9;
    // The new trajectory stays in REGISTERS (one or two elements per thread): handle_joint_limit's first pass — is any joint out of
    // range? — runs on them, and in the usual case (none is) they go straight to global memory: no staging copy in LDS and two barriers
    // less behind the step.  Only a violation brings them into L.xi for the projection loop below.
    constexpr int XR = (OMGX_MAX_WAYPOINTS * 9 + NT - 1) / NT;  // elements per thread (512 threads: 2)
    double xnew[XR] = {};
    int out_first = 0;
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int e = tid + r * NT;
        if (e >= n * 9) continue;
        const int i = e / 9, d = e % 9;
        double upd;
        if (!free_end) {
            upd = -eta * L.tvs[e];  // optimizer.py:132
        } else {  // goal_set_projection, optimizer.py:101-112 with the closed-form M
            const int first = n - c;
            if (i >= first) {
                const int q = i - first;
                const double b = L.xi[e] - goal[q * 9 + d];
                upd = -eta * L.tvs[e] + eta * L.tvs[e] - b;
            } else {
                const double m = (double)(i + 1) / (double)(first + 1);
                const double b0 = L.xi[first * 9 + d] - goal[d];
                upd = -eta * L.tvs[e] + eta * (m * L.tvs[first * 9 + d]) - m * b0;
            }
        }
        // Trajectory.update (core.py:43-51): fingers frozen unless consider_finger, then clamped to [0, 0.04]
        double xv = L.xi[e];
        if (d < 7 || prm.consider_finger) xv += upd;
        if (d >= 7) xv = fmin(fmax(xv, 0.0), 0.04);
        xnew[r] = xv;  // (L.xi still feeds neighbours' b0 reads and info's goal distance)
        const double t = (xv < lower[d] ? lower[d] - xv : 0.0) + (xv > upper[d] ? upper[d] - xv : 0.0);  // compute_traj_v
        out_first |= (t != 0.0) ? 1 : 0;  // (NaN counts)
    }

    // ---------------------------------------------------------------- phase 7: handle_joint_limit (optimizer.py:148-164)
    PHASE_MARK(7);
    if (out_first) L.red[7] = 1.0;  // benign same-value race; cleared in phase 0
    __syncthreads();
    if (!(L.red[7] > 0.0)) {  // workgroup-uniform: every joint inside its limits — norm 0, no projection step (the usual case)
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + r * NT;
            if (e < n * 9) {
                if constexpr (LIGHT) __hip_atomic_store(traj + e, xnew[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read by OTHER workgroups of the same launch (sc1 loads)
                else traj[e] = xnew[r];
            }
        }
        PHASE_MARK(8);
        return;  // (info's LIMIT_STEPS already holds 0)
    }
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int e = tid + r * NT;
        if (e < n * 9) L.xi[e] = xnew[r];
    }
    int cnt = 0;
    for (;;) {
        __syncthreads();
        int out_of_range = 0;
        for (int e = tid; e < n * 9; e += blockDim.x) {  // compute_traj_v
            const int d = e % 9;
            const double x = L.xi[e];
            const double t = (x < lower[d] ? lower[d] - x : 0.0) + (x > upper[d] ? upper[d] - x : 0.0);
            L.tv[e] = t;
            L.g[e] = t * t;
            out_of_range |= (t != 0.0) ? 1 : 0;  // (NaN counts: its sum must be formed)
        }
        // the usual case — every joint inside its limits — needs no sum: all terms are +0.0 and so is their sum, in any order; one
        // barrier behind a vote instead of block_sum's two barriers around a single wave's walk over the array
        // (the vote goes through an LDS word per parity of the pass — __syncthreads_or would bring static LDS, and this kernel
        // asks for all of the CU's as dynamic: thread 0 clears the other parity's word for the next pass behind the barrier)
        double* const vote = L.red + 54 + (cnt & 1);
        if (out_of_range) *vote = 1.0;  // benign same-value race
        __syncthreads();
        const bool any_out = *vote > 0.0;
        if (tid == 0) L.red[54 + ((cnt + 1) & 1)] = 0.0;
        const double nrm2 = any_out ? block_sum(L, L.g, n * 9, 5) : 0.0;
        if (!(sqrt(nrm2) > 1e-2) || cnt >= prm.joint_limit_max_steps) break;
        apply_ainv(L.tv, L.tvs, n, free_end, dt2);
        __syncthreads();
        if (tid < 64) {  // np.abs(traj_v).argmax(): first maximum in flat order
            double best = -__builtin_inf();  // numpy order: first occurrence, NaN wins (omg::np_arg_better)
            int bi = 0x7fffffff;
            for (int e = tid; e < n * 9; e += 64) {
                const double v = fabs(L.tv[e]);
                if (np_arg_better<false>(v, e, best, bi)) { best = v; bi = e; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (np_arg_better<false>(ov, oi, best, bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) L.red[6] = best / (fabs(L.tvs[bi]) + 1e-8);  // safe_div
        }
        __syncthreads();
        const double scale = L.red[6];
        for (int e = tid; e < n * 9; e += blockDim.x) L.xi[e] += scale * L.tvs[e];
        ++cnt;
    }
    for (int e = tid; e < n * 9; e += blockDim.x) {
        if constexpr (LIGHT) __hip_atomic_store(traj + e, L.xi[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else traj[e] = L.xi[e];
    }
    if (tid == NT - 64) a.info[(size_t)s * OMGX_INFO_STRIDE + OMGX_INFO_LIMIT_STEPS] = (double)cnt;  // the thread that wrote the other scalars (and a 0 here): program order
    PHASE_MARK(8);
}

