// omg_learner_body.h — device code of Learner.update_goal for ONE scene (shared by k_goal_update, omg_learner.hip,
// and the fused k_update_optimize, omg_chomp.hip).
//
//
// Replaces (float64, like the reference's numpy code):
//   Learner.cost_vector tail        omg/online_learner.py:145-160  (weights, smoothness proxy, normalisation)
//   Learner.update_goal_dist        :162-235  FTL / FTC / Exp / MD (mirror descent over 5 experts) / Proj
//   bp + find_zero                  :16-58    Bregman projection onto the simplex by nested bisection
//   Learner.update_goal             :237-249  argmax of the goal distribution, traj.end
//   chosen goal rows                omg/optimizer.py:93-99
//
// Lane j of a wave owns goals j, j+64, j+128, j+192 (G <= 256); every sum is a 64-lane butterfly, so all
// lanes hold identical values and the bisection's control flow is wave-uniform.  The workgroup has 5 waves:
// for MD each wave runs the Bregman projection of one expert (they are independent; only the mixture update
// that follows is sequential in the expert index), handing its result to wave 0 through LDS.  The other
// rules use wave 0 only.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "omg_device.h"
#include "omg_host.h"

namespace omg_learner {
#pragma clang fp contract(fast)


#define NPL 4  // goals per lane

// Debug aid (EXTRA=-DOMGX_PHASE_TIMING, tools/phase_timing.py): scene 0 stamps the shader clock; wave 4 runs the slowest expert of MD.
#ifdef OMGX_PHASE_TIMING
static __device__ unsigned long long g_learner_phase[16];
#define LPHASE(i, w) do { if (s == 0 && lane == 0 && wave == (w)) g_learner_phase[i] = __builtin_readcyclecounter(); } while (0)
#define LCOUNT(i, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 4) g_learner_phase[i] = (unsigned long long)(v); } while (0)
#else
#define LPHASE(i, w) do { } while (0)
#define LCOUNT(i, v) do { } while (0)
#endif

struct LearnerArgs {
    omgx_learner_params prm;
    const double* traj;
    const double* goal_set;
    const double* reach;
    const float* goal_cost;
    double* state;
    int S;
    int32_t* goal_idx;
    double* end;
    double* goal_rows;
    double* goal_point;
    double* cost_vector;
    const int32_t* active;  // [S] or null: scenes with 0 keep their goal and state (the reference has left its loop, planner.py:626)
    const int32_t* goal_count;  // [S] or null: scene s has goal_count[s] <= num_goals goals (arrays stay padded to num_goals)
    const double* eta;          // [S] or null: per-scene eta = sqrt(log(goal_count + 1) / optim_steps), else prm.eta
};

__device__ __forceinline__ double lane_bcast(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum(double v) { return omg::wave_allsum(v); }  // identical result in every lane
__device__ __forceinline__ double wmax(double v) { return omg::wave_allmax(v); }
// arg-extreme in numpy's order (first occurrence, NaN wins; omg::np_arg_better).  Lanes start from the neutral element
// (+-inf, INT_MAX), which loses every tie against a real entry.
// (value, index) pairs with distinct indices are totally ordered by np_arg_better, so the winner does not depend on the shape of the
// reduction: four DPP exchanges inside the rows of 16 lanes and the four rows' winners through lane reads — the six rounds of
// __shfl_xor this replaces were 18 trips through the LDS crossbar on the iteration's critical path.  Call from all 64 lanes.
template <bool MIN>
__device__ __forceinline__ int warg(double v, int i) {
#define OMG_WARG_STEP(CTRL)                                                       \
    {                                                                             \
        const double ov = omg::dpp_f64<CTRL>(v);                                  \
        const int oi = omg::dpp_i32<CTRL>(i);                                     \
        if (omg::np_arg_better<MIN>(ov, oi, v, i)) { v = ov; i = oi; }            \
    }
    OMG_WARG_STEP(0xB1) OMG_WARG_STEP(0x4E) OMG_WARG_STEP(0x141) OMG_WARG_STEP(0x140)
#undef OMG_WARG_STEP
    double bv = omg::readlane_f64(v, 0);
    int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ov = omg::readlane_f64(v, r);
        const int oi = __builtin_amdgcn_readlane(i, r);
        if (omg::np_arg_better<MIN>(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    return bi;
}
#define OMG_ARG_NEUTRAL_MIN __builtin_inf()
#define OMG_ARG_NEUTRAL_MAX (-__builtin_inf())

// bp (online_learner.py:32-58) with find_zero (:16-29) inlined.  x, v per-lane slices; w = 1, delta = 1/(4G+1).
//
// find_zero bisects f(L) = sum_j shiftx_j exp(L + z_j) - target, z = alpha - v.  With zmax = max_j z_j and
// C = sum_j shiftx_j exp(z_j - zmax) (no overflow: every term <= shiftx_j, one term == shiftx_j), f(L) = target (exp(d) - 1)
// for d = (L + zmax) - log(target / C): the sign of f is the sign of d and |f| < err is log1p(-err/target) < d < log1p(err/target).
// A bisection step is therefore two compares and an add on L's own dyadic sequence (the reference's x -= s sign(y); s /= 2) —
// no exponentials, no table, and nothing that can overflow for un-normalised costs (v ~ 1e4: the termwise exp(L + z_j) of
// the reference stays finite near the root, a factored exp(L) * sum exp(z_j) would be inf * 0).  exp(L + zmax) at the
// end is (target / C) exp(d) with |d| <~ 1e-6 (third-order series; the full exp only if the bisection did not converge).
// Decisions can differ from the reference's floating-point f only when |f| is within ~1e-15 of 0 or of err.
template <int NPLT = NPL>  // goals per lane: NPLT serves G <= 256; a build for fewer goals holds fewer registers (same bits: padded lanes add exact zeros)
__device__ void bregman_projection(const double* x, const double* v, double delta, int G, int lane, double* y) {
    const int max_iter = 100;
    const double err = 1e-6;
    double alpha[NPLT] = {}, shiftx[NPLT], lds[NPLT], ezs[NPLT];
    const double target = 1.0 + delta * (double)G;
    const double dlo = log1p(-err / target), dhi = log1p(err / target), ltarget = log(target);
    double vmax = -__builtin_inf();
#pragma unroll
    for (int j = 0; j < NPLT; ++j) {
        const bool ok = lane + 64 * j < G;
        shiftx[j] = x[j] + delta;
        lds[j] = ok ? log(delta / shiftx[j]) : 0.0;
        ezs[j] = 0.0;
        if (ok) vmax = fmax(vmax, 1.0 + v[j]);
    }
    const double x1 = wmax(vmax);
    const double L0 = (0.0 + x1) / 2.0, s0 = (x1 - 0.0) / 4.0, rx1 = 1.0 / x1;
    // zmax = max_j (alpha_j - v_j) of the pass to come is reduced TOGETHER with the norm that decides whether there is one: two
    // independent butterflies in one basic block interleave, one after the other they are two dependent chains per pass.
    double zmax;
    {
        double zm = -__builtin_inf();
#pragma unroll
        for (int j = 0; j < NPLT; ++j)
            if (lane + 64 * j < G) zm = fmax(zm, alpha[j] - v[j]);
        zmax = wmax(zm);
    }
    for (int it = 0; it < max_iter; ++it) {
        double partC = 0.0;
#pragma unroll
        for (int j = 0; j < NPLT; ++j)
            if (lane + 64 * j < G) { ezs[j] = exp((alpha[j] - v[j]) - zmax); partC += shiftx[j] * ezs[j]; }
        const double Csum = wsum(partC);
        const double lstar = ltarget - log(Csum);  // L + zmax at the root
        // find_zero's sequence L_0 = x1 / 2, L_(k+1) = L_k -+ x1 / 2^(k+2) visits, at step k, the midpoint of the dyadic interval of
        // length x1 / 2^k that holds the root Lr = lstar - zmax: L_k = x1 (2 floor(Lr / x1 2^k) + 1) / 2^(k+1) (floor clamped
        // to [0, 2^k - 1]: a root outside [0, x1] sends the sequence to that end).  Lane k evaluates step k and the first lane
        // inside the tolerance wins: ~20 instructions instead of ~20 dependent steps of 13.  L_k is rounded once here and k times in
        // the sequential sums: a few ulps apart, and a step decides differently only if |d| is below the rounding of L + zmax,
        // where it has converged anyway.  Whatever does not fit — no x1 > 0, no step below 52 inside the tolerance — takes the
        // sequential loop below.
        double L = L0, d = (L0 + zmax) - lstar;
        bool converged = false;
        {
            const double r = (lstar - zmax) * rx1;
            const double pk = ldexp(1.0, lane);
            const double qf = fmin(fmax(floor(r * pk), 0.0), pk - 1.0);
            const double Lk = ldexp(x1 * (2.0 * qf + 1.0), -(lane + 1));
            const double dk = (Lk + zmax) - lstar;
            const unsigned long long hit = __ballot(lane < 52 && dk > dlo && dk < dhi);
            if (hit != 0ull && x1 > 0.0) {
                const int k = __builtin_ctzll(hit);
                L = lane_bcast(Lk, k);
                d = lane_bcast(dk, k);
                converged = true;
            }
        }
        if (!converged) {
            double sstep = s0;
            for (int k = 0; k < max_iter; ++k) {
                if (d > dlo && d < dhi) { converged = true; break; }
                if (d > 0) L -= sstep;
                else if (d < 0) L += sstep;
                else L = d;  // NaN: x -= s * np.sign(nan) poisons x in the reference too
                sstep /= 2.0;
                d = (L + zmax) - lstar;
            }
        }
        // exp(L + zmax) = exp(lstar) exp(d) = (target / Csum) exp(d)
        const double ed = converged ? 1.0 + d * (1.0 + d * (0.5 + d * (1.0 / 6.0))) : exp(d);
        const double EL = (target / Csum) * ed;
        double nrm = 0.0, ap[NPLT];
#pragma unroll
        for (int j = 0; j < NPLT; ++j) {
            const bool ok = lane + 64 * j < G;
            y[j] = ok ? shiftx[j] * (EL * ezs[j]) - delta : 0.0;  // shiftx exp(L + alpha - v) - delta
            ap[j] = ok ? fmax(0.0, v[j] - L + lds[j]) : 0.0;
            nrm += (alpha[j] - ap[j]) * (alpha[j] - ap[j]);
        }
        LCOUNT(8, it + 1);
        double zm_next = -__builtin_inf();
#pragma unroll
        for (int j = 0; j < NPLT; ++j)
            if (lane + 64 * j < G) zm_next = fmax(zm_next, ap[j] - v[j]);
        const double nrm_all = wsum(nrm), zmax_next = wmax(zm_next);
        // sqrt(x) < 1e-6 (np.linalg.norm(alpha - alpha_prime) < err) <=> x < 0x1.19799812dea10p-40, the smallest double whose
        // correctly rounded root reaches 1e-6 (sqrt is monotone; NaN fails both)
        if (nrm_all < 0x1.19799812dea10p-40) break;
#pragma unroll
        for (int j = 0; j < NPLT; ++j) alpha[j] = ap[j];
        zmax = zmax_next;
    }
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < NPLT; ++j) { y[j] = fmax(y[j], 0.0); part += y[j]; }
    const double sy = wsum(part);
#pragma unroll
    for (int j = 0; j < NPLT; ++j) y[j] /= sy;
}


// All threads of the workgroup call this (>= 5 waves for MD); sh_pn [5][OMGX_MAX_GOALS] and sh_tab [5][128] are LDS.
// sh_idx (LDS, optional): receives the chosen goal's index — valid for the workgroup after its next barrier (active scenes only).
// flag / publish (optional, k_update_optimize_split): the scene's rendezvous word receives (publish << 8) | index the moment the index
// is known — one relaxed store: whoever waits for it reads nothing but the index and tables that no launch in flight writes.
// FOUR: a workgroup of FOUR waves (the persistent planner kernel, omg_persist.h).  MD's five experts then share four waves — wave 0 the
// sharpest expert 4 (7-8 projection passes), wave 1 expert 3, wave 2 experts 2 and 0, wave 3 expert 1 and then the mixture — every
// expert's projection is still one wave's work from the same inputs: same bits.
template <bool FOUR = false, int NPLT = NPL>  // NPLT: goals per lane this instantiation serves (G <= 64 NPLT)
__device__ __forceinline__ void learner_scene(const LearnerArgs& a, int s, double (*sh_pn)[OMGX_MAX_GOALS], double (*sh_tab)[128], int* sh_idx = nullptr,
                                              uint32_t* flag = nullptr, uint32_t publish = 0u) {
    // The scene's `active` word is REQUESTED here and tested where the first write would happen, behind the requests of the cost vector's
    // inputs (and the experts' distributions): tested at once it is a trip to memory of its own — ~2 us after a launch boundary, on
    // the critical path of every iteration (goal costs -> learner -> the step's tail) — ahead of all the others.
    const int scene_active = a.active ? a.active[s] : 1;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LPHASE(0, 4);
    if (a.prm.alg != OMGX_ALG_MD && wave > 0) return;  // no barrier on these paths
    const omgx_learner_params& prm = a.prm;
    // GS: padded goal count (array strides); G: this scene's own count — every loop, sum and constant below uses G, so a scene
    // in a ragged batch computes exactly what it would compute alone (padded lanes contribute exact zeros)
    const int GS = prm.num_goals, n = prm.n_waypoints, c = prm.constraint_num;
    const int G = a.goal_count ? min(max(a.goal_count[s], 1), GS) : GS;
    // (not `a.eta ? a.eta[s] : prm.eta`: the compiler makes that a select between two ADDRESSES — one of them a stack copy of prm.eta
    // read through the flat aperture; the empty asm keeps the loaded value a value)
    double eta = prm.eta;
    if (a.eta) {
        double es = a.eta[s];
        asm volatile("" : "+v"(es));
        eta = es;
    }
    double* st = a.state + (int64_t)s * (7 * (int64_t)GS + 10);
    double *sum_costs = st, *p = st + GS, *experts_p = st + 2 * GS, *q = st + 7 * GS, *ecost = st + 7 * GS + 5;
    const double* gs = a.goal_set + (int64_t)s * GS * 9;
    // MD: this wave's expert distribution (last iteration's), requested with everything else (padded lanes: masked at the use)
    // (FOUR: the wave's experts ex0 [, ex1]; else expert = wave)
    const int ex0 = FOUR ? (wave == 0 ? 4 : (wave == 1 ? 3 : (wave == 2 ? 2 : 1))) : (wave < 5 ? wave : -1);
    const int ex1 = (FOUR && wave == 2) ? 0 : -1;
    double epw_pre[NPLT] = {};
    double epw_pre1[FOUR ? NPLT : 1] = {};
    if (prm.alg == OMGX_ALG_MD && ex0 >= 0) {
#pragma unroll
        for (int j = 0; j < NPLT; ++j) {
            const int g = lane + 64 * j;
            if (g < GS) epw_pre[j] = experts_p[(int64_t)ex0 * GS + g];
            if constexpr (FOUR) if (ex1 >= 0 && g < GS) epw_pre1[j] = experts_p[(int64_t)ex1 * GS + g];
        }
    }
    // ... and the mixture's state (wave 0, lanes 0-4: expert k's weight and last cost)
    // The wave that runs the mixture update and everything behind it: NOT one that shares a SIMD with the sharpest expert's wave 4
    // (waves w and w + 4 do: on wave 0 the mixture's divisions and exponentials slowed that expert's passes by what they hid) — wave 5,
    // beside expert 1's wave, which finishes early; workgroups of five waves (k_goal_update) keep wave 0.
    const int mixw = FOUR ? (prm.alg == OMGX_ALG_MD ? 3 : 0) : ((prm.alg == OMGX_ALG_MD && blockDim.x >= 384) ? 5 : 0);
    const int kk_pre = lane < 5 ? lane : 0;
    double q_pre = 0.0, ecost_pre = 0.0;
    if (prm.alg == OMGX_ALG_MD && wave == mixw) { q_pre = q[kk_pre]; ecost_pre = ecost[kk_pre]; }
    int idx = 0;
    if (prm.alg == OMGX_ALG_PROJ) {  // :196-206
        if (scene_active == 0) return;
        const double* last = a.traj + ((int64_t)s * n + n - 1) * 9;
        double best = OMG_ARG_NEUTRAL_MIN;
        int bi = 0x7fffffff;
        for (int j = 0; j < NPLT; ++j) {
            const int g = lane + 64 * j;
            if (g < G) {
                double d2 = 0.0;
                for (int d = 0; d < 9; ++d) { const double e = last[d] - gs[g * 9 + d]; d2 += e * e; }
                const double dist = sqrt(d2);
                if (omg::np_arg_better<true>(dist, g, best, bi)) { best = dist; bi = g; }
            }
        }
        idx = warg<true>(best, bi);
        for (int j = 0; j < NPLT; ++j) { const int g = lane + 64 * j; if (g < G) p[g] = (g == idx) ? 1.0 : 0.0; }
    } else {
        const double* ts = a.traj + ((int64_t)s * n + prm.start_idx) * 9;
        double cv[NPLT];
        double part = 0.0;
        for (int j = 0; j < NPLT; ++j) {
            const int g = lane + 64 * j;
            cv[j] = 0.0;
            if (g < G) {
                double s2 = 0.0;
                for (int d = 0; d + 1 < 9; ++d) {  // np.diff(traj_start - goal_set, axis=-1): adjacent joint columns
                    const double e = (ts[d + 1] - gs[g * 9 + d + 1]) - (ts[d] - gs[g * 9 + d]);
                    s2 += e * e;
                }
                const double nr = sqrt(s2);
                // the goal's cost: one float32, or (latency mode) the cost_parts partial sums of its parts added in part order
                const int NPc = prm.cost_parts > 1 ? prm.cost_parts : 1;
                const float* gc = a.goal_cost + ((int64_t)s * GS + g) * NPc;
                float gcost = gc[0];
                for (int k = 1; k < NPc; ++k) gcost += gc[k];
                const float wc = (float)prm.base_obstacle_weight * gcost;  // float32 product
                cv[j] = (double)wc + prm.smooth_weight * (nr * nr);
                part += cv[j] * cv[j];
            }
        }
        if (prm.normalize_cost) {
            const double nn = sqrt(wsum(part));
            for (int j = 0; j < NPLT; ++j) cv[j] /= nn;
        }
        if (scene_active == 0) return;  // nothing has been written yet (all waves take the same branch: no barrier is skipped by some)
        if (a.cost_vector && wave == 0)
            for (int j = 0; j < NPLT; ++j) { const int g = lane + 64 * j; if (g < G) a.cost_vector[(int64_t)s * GS + g] = cv[j]; }

        if (prm.alg == OMGX_ALG_FTL || prm.alg == OMGX_ALG_FTC) {  // :175-189
            double best = OMG_ARG_NEUTRAL_MIN;
            int bi = 0x7fffffff;
            for (int j = 0; j < NPLT; ++j) {
                const int g = lane + 64 * j;
                if (g < G) {
                    double key = cv[j];
                    if (prm.alg == OMGX_ALG_FTL) { key = sum_costs[g] + cv[j]; sum_costs[g] = key; }
                    if (omg::np_arg_better<true>(key, g, best, bi)) { best = key; bi = g; }
                }
            }
            idx = warg<true>(best, bi);
            for (int j = 0; j < NPLT; ++j) { const int g = lane + 64 * j; if (g < G) p[g] = (g == idx) ? 1.0 : 0.0; }
        } else if (prm.alg == OMGX_ALG_EXP) {  // :208-217
            double sc[NPLT], pn[NPLT], tot = 0.0;
            for (int j = 0; j < NPLT; ++j) {
                const int g = lane + 64 * j;
                sc[j] = 0.0;
                if (g < G) { sc[j] = sum_costs[g] + cv[j]; sum_costs[g] = sc[j]; tot += sc[j]; }
            }
            tot = wsum(tot);
            double ps = 0.0;
            for (int j = 0; j < NPLT; ++j) {
                const int g = lane + 64 * j;
                pn[j] = 0.0;
                if (g < G) { pn[j] = exp(-eta * cv[j]) * p[g] * 0.999 + (sc[j] / (tot + 1e-8)) * 0.001; ps += pn[j]; }
            }
            ps = wsum(ps);
            double best = OMG_ARG_NEUTRAL_MAX;
            int bi = 0x7fffffff;
            for (int j = 0; j < NPLT; ++j) {
                const int g = lane + 64 * j;
                if (g < G) { const double v = pn[j] / (ps + 1e-8); p[g] = v; if (omg::np_arg_better<false>(v, g, best, bi)) { best = v; bi = g; } }
            }
            idx = warg<false>(best, bi);
        } else {  // MD, :219-235
            const double pw[5] = {0.25, 0.5, 1.0, 4.0, 16.0};  // eta * 2**[-2,-1,0,2,4], :82
            const double delta = 1.0 / (4.0 * (double)G + 1.0);
            // An expert's wave announces its result through a word in LDS (release / acquire at workgroup scope) instead of a
            // barrier: the mixture update below walks the experts in order and needs expert i only in its pass i, so wave 0 runs
            // its first four passes — a dependent division each — while the sharpest expert (7-8 projection passes) is still busy.
            int* const expert_done = reinterpret_cast<int*>(&sh_tab[0][8]);  // [5]
            if (threadIdx.x < 5) expert_done[threadIdx.x] = 0;
            __syncthreads();
            for (int q = 0; q < (FOUR ? 2 : 1); ++q) {  // waves 0..4: Bregman projection of their own expert (reads the OLD experts_p, like the reference)
                const int ex = q == 0 ? ex0 : ex1;
                if (ex < 0) break;
                double v[NPLT], epw[NPLT], pn[NPLT];
                for (int j = 0; j < NPLT; ++j) {
                    const int g = lane + 64 * j;
                    v[j] = eta * pw[ex] * cv[j];
                    if constexpr (FOUR) epw[j] = g < G ? (q == 0 ? epw_pre[j] : epw_pre1[j]) : 0.0;
                    else epw[j] = g < G ? epw_pre[j] : 0.0;
                }
                LPHASE(1, 4);
                bregman_projection<NPLT>(epw, v, delta, G, lane, pn);
                LPHASE(2, 4);
                // this expert's cost (:229-230) on its own wave: sum_g cv pn + |pn - old p|; the step table is dead by now
                double part2 = 0.0;
                for (int j = 0; j < NPLT; ++j) {
                    const int g = lane + 64 * j;
                    if (g < G) { sh_pn[ex][g] = pn[j]; part2 += cv[j] * pn[j] + fabs(pn[j] - epw[j]); }
                }
                const double ecw = wsum(part2);
                if (lane == 0) {
                    sh_tab[ex][0] = ecw;
                    __hip_atomic_store(expert_done + ex, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (a wave's LDS stores execute in order)
                }
            }
            LPHASE(3, 4);
            if (wave != mixw) return;
            LPHASE(4, mixw);
            // The mixture update sits INSIDE the expert loop (:231-235): after expert i, q_k *= exp(-cost_k) for ALL k with
            // the costs as they stand (new for k <= i, last iteration's for k > i), then q is normalised.  All ten
            // exponentials are known up front.  The mixture
            // p = sum_k q_k p_k is overwritten in every pass of the reference loop, so only the last one is formed.
            // Lane k < 5 carries expert k through the loop (its q_k, both of its exponentials), the normalising sum is formed
            // from lane broadcasts in the reference's order: one division per pass instead of five.
            double qv[5], ec[5], ep[5][NPLT];
            {
                const int kk = kk_pre;
                const double e_old = exp(-1.0 * ecost_pre);
                double e_new = 0.0;  // lane kk: exp(-cost of expert kk), known from pass kk on
                double ql = q_pre;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    while (__hip_atomic_load(expert_done + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
                    ec[i] = sh_tab[i][0];
                    for (int j = 0; j < NPLT; ++j) { const int g = lane + 64 * j; ep[i][j] = g < G ? sh_pn[i][g] : 0.0; }
                    const double en = exp(-1.0 * ec[i]);  // the same number in every lane; lane i keeps it
                    e_new = kk == i ? en : e_new;
                    ql = ql * (kk <= i ? e_new : e_old);
                    double qs = 0.0;
                    for (int k = 0; k < 5; ++k) qs += lane_bcast(ql, k);
                    ql /= qs;
                }
                for (int k = 0; k < 5; ++k) qv[k] = lane_bcast(ql, k);
            }
            double pm[NPLT], ps = 0.0;
            for (int j = 0; j < NPLT; ++j) {
                double m = 0.0;
                for (int k = 0; k < 5; ++k) m += ep[k][j] * qv[k];
                pm[j] = (lane + 64 * j < G) ? m : 0.0;
                ps += pm[j];
            }
            ps = wsum(ps);
            for (int j = 0; j < NPLT; ++j) pm[j] /= ps;
            double best = OMG_ARG_NEUTRAL_MAX;
            int bi = 0x7fffffff;
            for (int j = 0; j < NPLT; ++j) {
                const int g = lane + 64 * j;
                if (g < G) {
                    p[g] = pm[j];
                    for (int i = 0; i < 5; ++i) experts_p[(int64_t)i * GS + g] = ep[i][j];
                    if (omg::np_arg_better<false>(pm[j], g, best, bi)) { best = pm[j]; bi = g; }
                }
            }
            if (lane == 0)
                for (int i = 0; i < 5; ++i) { q[i] = qv[i]; ecost[i] = ec[i]; }
            idx = warg<false>(best, bi);
        }
    }
    LPHASE(5, mixw);
    // traj.end / goal rows (online_learner.py:243-245, optimizer.py:93-99)
    if (lane == 0) {
        static_assert(OMGX_MAX_GOALS <= 256, "the rendezvous word is (ticket << 8) | goal index: the index must fit 8 bits");
        if (flag) __hip_atomic_store(flag, (publish << 8) | (uint32_t)idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.goal_idx[s] = idx;
        if (sh_idx) *sh_idx = idx;
    }
    if (lane < 9) {
        const double v = gs[idx * 9 + lane];
        a.end[s * 9 + lane] = v;
        a.goal_point[s * 9 + lane] = v;
    }
    for (int e = lane; e < c * 9; e += 64)
        a.goal_rows[(int64_t)s * c * 9 + e] =
            prm.use_standoff ? a.reach[((int64_t)s * GS + idx) * c * 9 + e] : gs[idx * 9 + e % 9];
}


// Host side: argument checks of omgx_goal_update (shared with omgx_goal_update_optimize).
static inline int make_args(const omgx_learner_params* h_params, const double* traj, const double* goal_set, const double* reach,
                            const float* goal_cost, double* state, int32_t num_scenes, int32_t* goal_idx, double* end,
                            double* goal_rows, double* goal_point, double* cost_vector, const int32_t* active, const int32_t* goal_count,
                            const double* eta, LearnerArgs& a) {
    if (!h_params || num_scenes < 0) return OMGX_ERR_INVALID;
    const omgx_learner_params& p = *h_params;
    if (!traj || !goal_set || !state || !goal_idx || !end || !goal_rows || !goal_point) return OMGX_ERR_INVALID;
    if (p.alg < OMGX_ALG_FTL || p.alg > OMGX_ALG_PROJ) return OMGX_ERR_INVALID;
    if (p.alg != OMGX_ALG_PROJ && !goal_cost) return OMGX_ERR_INVALID;
    if (p.use_standoff && !reach) return OMGX_ERR_INVALID;
    if (p.num_goals < 1 || p.num_goals > OMGX_MAX_GOALS || p.n_waypoints < 1 || p.constraint_num < 1 ||
        p.constraint_num > OMGX_MAX_CONSTRAINTS)
        return OMGX_ERR_UNSUPPORTED;
    if (p.start_idx < 0 || p.start_idx >= p.n_waypoints) return OMGX_ERR_INVALID;
    a = LearnerArgs{};
    a.prm = p; a.traj = traj; a.goal_set = goal_set; a.reach = reach; a.goal_cost = goal_cost; a.state = state; a.S = num_scenes;
    a.goal_idx = goal_idx; a.end = end; a.goal_rows = goal_rows; a.goal_point = goal_point; a.cost_vector = cost_vector;
    a.active = active; a.goal_count = goal_count; a.eta = eta;
    return OMGX_OK;
}

}  // namespace omg_learner
