// omg_persist.h — k_plan_persistent: K iterations of the planner loop (omg/planner.py:612-630) for every scene of a batch in ONE launch.
// Included at the end of omg_kernels.hip (needs ChunkArgs, gq_item, launch helpers) behind omg_chomp_body.h.
//
// Why.  Launched per iteration, a part of the batch runs "goal-set launch, then update launch" on a stream, and the dispatcher shares
// the chip fairly between the parts' queues: their goal-set launches end together, their update launches (two latency-bound
// workgroups per scene) then run on an empty chip — 13 % of the step — and every launch pays its ramp and
// tail.  But the ONLY dependency in the algorithm is per scene: iteration t + 1 of scene s needs the trajectory that iteration t of
// scene s left, nothing of any other scene (omg/core.py:869-885: scenes are independent plans).  This kernel schedules by that
// dependency alone:
//
//   * work ITEMS of (scene s, iteration t): 5 trajectory-layer pieces (2 links x all waypoints each) and one item per goal of the
//     goal set (Learner.cost_vector's obstacle batch) — the same code as a workgroup of k_goalset_queue (gq_item);
//   * the scene's LAST item to finish (one agent-scope counter per scene; nobody waits for anybody) runs Learner.update_goal and
//     Optimizer.optimize for the scene in place — learner_scene<FOUR> + chomp_scene<LIGHT>, inside the goal workgroup's footprint of
//     LDS and registers — and then ACTIVATES (s, t + 1): its items become claimable;
//   * resident workgroups (5 per CU) claim items until every scene has run its K iterations or has left the loop (planner.py:626).
//
// The chip never drains between iterations: while one scene's learner and step run on ONE workgroup slot, the other scenes' items
// fill the other 1 279.
//
// Queue.  An activation is a 64-bit word {tag = position + 1 | iteration | scene} in a ring (one producer per slot, written with ONE sc1
// store: MI355X_MICROARCH.md, granules).  An XCD drains ONE activation at a time — a scene's volumes then sit in one L2 (spread over all
// eight a launch takes 1.7x as long) — through its claim word {position + 1 | next item}: a claim is one returning atomic add.  The
// workgroup that finds its XCD's activation exhausted locks the word, takes the next activation off the ring's head (or finds the plan
// finished) and installs it; its neighbours sleep on the word meanwhile.  Which XCD runs which activation is first come, first served:
// load balance at the granularity of one scene-iteration, no schedule, no measuring launch.
//
// Visibility between workgroups (cdna_hip_programming.md, guideline 16; nothing depends on placement or dispatch order):
//   trajectory       step workgroup: sc1 stores, drained        ->  item workgroups: sc1 loads (8-byte agent-scope atomics on both sides)
//   goal cost / count  goal item: sc1 stores, drained, then the arrival       ->  last arriver: acquire fence, plain loads
//   layer outputs, poses  layer item: plain stores, release fence, then the arrival  ->  last arriver: acquire fence, plain loads
//   learner state, goal, info  step workgroup: plain stores, release fence, then the activation  ->  the scene's next step workgroup: acquire fence
// Every spin is bounded (2 s) and reports through the control block; the host zeroes / initialises the queue before every launch.
#pragma once

typedef const OMG_CONST_AS omgx_plan_iter* IterTablePtr;  // the iteration table is never written by the kernel: scalar loads
__device__ __forceinline__ IterTablePtr as_const(const omgx_plan_iter* p) { return (IterTablePtr)(uintptr_t)p; }

namespace omg_persist {

#ifdef OMGX_PERSIST_STATS
#define PQ_STAT(i, v) __hip_atomic_fetch_add(pa.stats + (i), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define PQ_STAT(i, v) do { } while (0)
#endif
// claim word: [63:48] scene | [47:32] iteration | [31] LOCK | [30] VALID | [29:24] install count | [23:0] next item
#define PQ_LOCK 0x80000000ull
#define PQ_VALID 0x40000000ull
#define PQ_KMASK 0xffffffull
#define PQ_DONE 0xffffffffffffffffull
#define PQ_SLOTS 3  // claim words per XCD

struct PersistArgs {
    ChunkArgs ca;                     // a goal-set + layer launch in the batch layout; CH / PS / MR / tbl_n / traj_start follow the iteration
    omg_learner::LearnerArgs la;      // prm.start_idx follows the iteration
    ChompArgs ch;                     // prm.obstacle_weight / smoothness_weight / step_size / do_update follow the iteration
    const omgx_plan_iter* iters;      // [num_iters] device
    int num_iters;
    int G;                            // goals per scene (padded)
    int32_t* active;                  // [S] or null: scenes with 0 are not planned; a scene that terminates under stop_on_terminate gets 0
    unsigned long long* ring;         // [cap] activations
    unsigned long long* xw;           // [8][PQ_SLOTS][16] claim words, one 128-byte line each (word 0 of the line)
    uint32_t* ctl;                    // [32] (a line of its own) [8..9] the ring's {tail | head} as one 64-bit word, [0] activations made, [1] -, [2] scenes finished, [3] failure code, [4] scenes in the plan
    unsigned long long* stats;        // [16] (a line of its own; -DOMGX_PERSIST_STATS) [0] claim spins, [1] item ticks, [2] update ticks, [3] items, [4] updates, [5] claim ticks
    uint32_t* arrive;                 // [S] items of the scene's current iteration that have finished
    int cap;
    uint32_t lds_bytes;               // dynamic LDS of the launch
    // dedicated update CUs (update_cus per XCD; 0: the scene's last item runs its update in place)
    int update_cus;
    uint32_t* cu_role;                // [4096] by hardware CU key: 0 unknown, 3 being decided, 1 item CU, 2 update CU
    uint32_t* xcd_cus;                // [8][32] (a line each) [0] CUs registered on the XCD, [1] update workgroups registered
    unsigned long long* uq;           // [8][cap] update requests per XCD: {tag = position + 1 | iteration | scene}
    unsigned long long* uq_ht;        // [8][16] (a line each) {tail | head} of the XCD's update ring
};

__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a read as a read-modify-write atomic: performed where the word lives, whatever this XCD's L2 still holds of the line
__device__ __forceinline__ uint32_t rmw_u32(uint32_t* p) { return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long rmw_u64(unsigned long long* p) { return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (Re-)initialise the queue for a launch: every active scene activated for iteration 0, in scene order; one workgroup.
__global__ __launch_bounds__(256) void k_persist_init(PersistArgs pa) {
    for (int i = threadIdx.x; i < pa.ca.S; i += 256) pa.arrive[i] = 0u;
    if (threadIdx.x < 8 * PQ_SLOTS) pa.xw[16 * threadIdx.x] = 0ull;
    if (threadIdx.x < 16) pa.stats[threadIdx.x] = 0ull;
    for (int i = threadIdx.x; i < 4096; i += 256) pa.cu_role[i] = 0u;
    for (int i = threadIdx.x; i < 8 * 32; i += 256) pa.xcd_cus[i] = 0u;
    if (threadIdx.x < 8) pa.uq_ht[16 * threadIdx.x] = 0ull;
    // every polled word starts from zero in EVERY launch: a ring slot still holding last launch's word would pass for this launch's (the
    // tags count positions from the same start), in the window between a producer's ticket and its store
    for (int i = threadIdx.x; i < pa.cap; i += 256) pa.ring[i] = 0ull;
    for (int i = threadIdx.x; i < 8 * pa.cap; i += 256) pa.uq[i] = 0ull;
    __syncthreads();
    __syncthreads();
    if (threadIdx.x == 0) {  // (serial: the order of the ring is the scenes' order; S <= 65535)
        uint32_t n = 0;
        for (int s = 0; s < pa.ca.S; ++s)
            if (!pa.active || pa.active[s] != 0) { pa.ring[n] = ((unsigned long long)(n + 1) << 32) | (unsigned long long)(uint32_t)s; ++n; }
        pa.ctl[0] = n; pa.ctl[1] = 0u; pa.ctl[2] = 0u; pa.ctl[3] = 0u; pa.ctl[4] = n;
        *reinterpret_cast<unsigned long long*>(pa.ctl + 8) = (unsigned long long)n << 32;  // {tail | head} of the ring
    }
}

// What an activation's item k is: the first layer_parts items are the trajectory layer's pieces, then one per goal.
struct Item { int s, t, k, nitems, mode; };

typedef __attribute__((address_space(3))) unsigned char* LdsBytes;

// a value every lane holds, as a scalar (function arguments arrive in vector registers)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// a kernel argument block (or a part of it) read through the constant address space: scalar loads, field by field
template <class T>
__device__ __forceinline__ T load_const(unsigned long long addr) {
    static_assert(sizeof(T) % 4 == 0, "dwords");
    T out;
    typedef uint32_t __attribute__((may_alias)) u32_any;  // (the dwords ARE the object's fields, of whatever type: no type-based reordering)
    const OMG_CONST_AS u32_any* src = (const OMG_CONST_AS u32_any*)(uintptr_t)addr;
    u32_any* dst = reinterpret_cast<u32_any*>(&out);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return out;
}

// One work item (omg_goalset_queue.h: gq_item) behind a CALL: the item's code gets the register allocation it has in k_goalset_queue,
// whatever the kernel around it keeps alive.  k < 5: piece k of the scene's trajectory layer, else goal k - 5; start_idx: the learner's
// window of the scene's iteration.
template <int LB, int ROLE>  // ROLE 1: a trajectory-layer piece, 2: a goal
__device__ __forceinline__ void persist_item_impl(const unsigned long long pa_addr_in, LdsBytes lds, const int s_in, const int k_in, const int start_in) {
    const unsigned long long pa_addr = uniform_u64(pa_addr_in);
    const int s = __builtin_amdgcn_readfirstlane(s_in), k = __builtin_amdgcn_readfirstlane(k_in), start_idx = __builtin_amdgcn_readfirstlane(start_in);
    ChunkArgs ca = load_const<ChunkArgs>(pa_addr + offsetof(PersistArgs, ca));
    const uint32_t lds_bytes = load_const<uint32_t>(pa_addr + offsetof(PersistArgs, lds_bytes));
    const int G = load_const<int>(pa_addr + offsetof(PersistArgs, G));
    const int n = ca.wp_n;
    // (readfirstlane: provably wave-uniform for the scalar copies the item's main loop makes of them)
    const int CH = __builtin_amdgcn_readfirstlane(n - start_idx);
    ca.CH = CH; ca.C = G * CH;
    ca.PS = __builtin_amdgcn_readfirstlane(CH + 1 > n ? CH + 1 : n); ca.MR = __builtin_amdgcn_readfirstlane(CH > n ? CH : n);
    {   // exact-path records staged: whatever fits the launch's LDS beside this window's poses (any number gives the same bits)
        const int base = GqLayout(ca.PS, ca.MR, ca.P, 0).total;
        int tn = ((int)lds_bytes - base) / 64;
        ca.tbl_n = __builtin_amdgcn_readfirstlane(tn < 0 ? 0 : (tn > GQ_TBL_MAX ? GQ_TBL_MAX : tn));
    }
    ca.traj_start = ca.wp_traj + (int64_t)start_idx * 9;  // row s at + s * ts_stride (= n * 9)
    const bool is_layer = k < 5;
    double* const lds_pose = reinterpret_cast<double*>((unsigned char*)lds);
    gq_item<LB, false, false, false, false, true, ROLE>(ca, lds_pose, s, is_layer, is_layer ? k : 0, is_layer ? 0 : k - 5, 1);
}

#ifndef OMGX_PERSIST_GOAL_ATTR
#define OMGX_PERSIST_GOAL_ATTR __attribute__((noinline))
#endif
template <int LB>
__device__ OMGX_PERSIST_GOAL_ATTR void persist_goal_item(const unsigned long long pa_addr, LdsBytes lds, const int s, const int k, const int start_idx) {
    persist_item_impl<LB, 2>(pa_addr, lds, s, k, start_idx);
}
template <int LB>
__device__ __attribute__((noinline)) void persist_layer_item(const unsigned long long pa_addr, LdsBytes lds, const int s, const int k, const int start_idx) {
    persist_item_impl<LB, 1>(pa_addr, lds, s, k, start_idx);
}

// The scene's update behind CALLS too, in two functions with their own register allocations: the learner (its arrays hold NPLT goals per
// lane: a build per goal count class instead of one for 256 goals) and the step.  Both start from the workgroup's dynamic LDS.
__device__ __forceinline__ omgx_plan_iter load_iter(const PersistArgs& pa, int t) {
    omgx_plan_iter rec;
    IterTablePtr r = as_const(pa.iters) + t;
    rec.mode = r->mode; rec.start_idx = r->start_idx; rec.stop_on_terminate = r->stop_on_terminate; rec.do_update = r->do_update;
    rec.obstacle_weight = r->obstacle_weight; rec.smoothness_weight = r->smoothness_weight; rec.step_size = r->step_size;
    return rec;
}

// Learner.update_goal for scene s (iteration t's window); leaves the chosen goal's index in the LDS word behind the learner's tables
// and its link poses in end_poses_out.
template <int NPLT>
__device__ __attribute__((noinline)) void persist_learner(const unsigned long long pa_addr_in, LdsBytes lds, const int s_in, const int t_in) {
    const unsigned long long pa_addr = uniform_u64(pa_addr_in);
    const int s = __builtin_amdgcn_readfirstlane(s_in), t = __builtin_amdgcn_readfirstlane(t_in);
    omg_learner::LearnerArgs la = load_const<omg_learner::LearnerArgs>(pa_addr + offsetof(PersistArgs, la));
    const omgx_plan_iter* iters = load_const<const omgx_plan_iter*>(pa_addr + offsetof(PersistArgs, iters));
    la.prm.start_idx = (as_const(iters) + t)->start_idx;
    la.active = nullptr;  // only scenes in the loop are ever activated
    double* shl = reinterpret_cast<double*>((unsigned char*)lds);
    int* const sh_idx = reinterpret_cast<int*>(shl + 5 * OMGX_MAX_GOALS + 5 * 128);
    omg_learner::learner_scene<true, NPLT>(la, s, reinterpret_cast<double (*)[OMGX_MAX_GOALS]>(shl), reinterpret_cast<double (*)[128]>(shl + 5 * OMGX_MAX_GOALS), sh_idx);
    __syncthreads();  // the goal (global memory, this CU) and its index (LDS) are the workgroup's
    const int gi = *sh_idx;
    const double* src = la.prm.goal_pose_table + ((size_t)s * la.prm.num_goals + gi) * 120;
    if (threadIdx.x < 120) la.prm.end_poses_out[(size_t)s * 120 + threadIdx.x] = src[threadIdx.x];  // (for the fixed-goal iterations and later launches)
}

// Optimizer.optimize for scene s with iteration t's schedule; gi >= 0: the goal the learner has just chosen (its poses come from the
// goal pose table), else the scene's current end pose.
__device__ __attribute__((noinline)) void persist_step(const unsigned long long pa_addr_in, LdsBytes lds, const int s_in, const int t_in, const int gi_in) {
    const unsigned long long pa_addr = uniform_u64(pa_addr_in);
    const int s = __builtin_amdgcn_readfirstlane(s_in), t = __builtin_amdgcn_readfirstlane(t_in), gi = __builtin_amdgcn_readfirstlane(gi_in);
    ChompArgs ch = load_const<ChompArgs>(pa_addr + offsetof(PersistArgs, ch));
    const omgx_plan_iter* iters = load_const<const omgx_plan_iter*>(pa_addr + offsetof(PersistArgs, iters));
    int32_t* const active = load_const<int32_t*>(pa_addr + offsetof(PersistArgs, active));
    IterTablePtr r = as_const(iters) + t;
    ch.prm.obstacle_weight = r->obstacle_weight; ch.prm.smoothness_weight = r->smoothness_weight; ch.prm.step_size = r->step_size;
    ch.prm.do_update = r->do_update;
    ch.active = nullptr;
    ch.deactivate = (r->stop_on_terminate && active) ? active : nullptr;
    const double* end_pose = ch.prm.end_poses + (size_t)s * 120;
    if (gi >= 0) {
        const double* table = load_const<const double*>(pa_addr + offsetof(PersistArgs, la) + offsetof(omg_learner::LearnerArgs, prm) + offsetof(omgx_learner_params, goal_pose_table));
        const int G = load_const<int>(pa_addr + offsetof(PersistArgs, G));
        end_pose = table + ((size_t)s * G + gi) * 120;
    }
    chomp_scene<1, GQ_NT, true>(ch, (unsigned char*)lds, s, nullptr, 0u, nullptr, end_pose);
}

// Learner.update_goal + Optimizer.optimize of scene s at iteration t, then the scene's next activation (or its end); called by the whole
// workgroup that finished the scene's last item.
__device__ __forceinline__ void persist_update(const PersistArgs& pa, const unsigned long long ka, LdsBytes lds, const int s, const int t) {
    const int tid = (int)threadIdx.x;
    IterTablePtr rec = as_const(pa.iters) + t;
    const int mode = rec->mode, stop = rec->stop_on_terminate;
    // The update is the scene's critical path (its next iteration waits for it) and latency-bound (chains of dependent float64
    // instructions, barriers): beside four goal workgroups on its CU, served oldest-first, it took 190 us against 55 us alone, and
    // with every scene spending most of its cycle here the chip ran out of items (measured, 100 x 64).  Its waves go first.
#ifndef OMGX_PERSIST_UPDATE_PRIO
#define OMGX_PERSIST_UPDATE_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(OMGX_PERSIST_UPDATE_PRIO);
    if (tid == 0) {
        __hip_atomic_store(pa.arrive + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody touches it before the next activation
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 forgets what other workgroups have rewritten
    }
    __syncthreads();
    int gi = -1;
    if (mode) {
#ifdef OMGX_PERSIST_NPL4  // experiment build: one learner for every goal count
        persist_learner<4>(ka, lds, s, t);
#else
        if (pa.G <= 64) persist_learner<1>(ka, lds, s, t);
        else if (pa.G <= 128) persist_learner<2>(ka, lds, s, t);
        else persist_learner<4>(ka, lds, s, t);
#endif
        gi = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(reinterpret_cast<const double*>((unsigned char*)lds) + 5 * OMGX_MAX_GOALS + 5 * 128));
        __syncthreads();  // chomp_scene reuses the learner's LDS
    }
    persist_step(ka, lds, s, t, gi);
    // ---- everything the step left is visible at agent scope, then the scene goes on (or has finished)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool more = t + 1 < pa.num_iters;
        if (more && stop && pa.active) more = __hip_atomic_load(pa.active + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (more) {
            // the slot's word first, then the tail: whoever sees the tail moved finds the word (the tag is checked anyway)
            __hip_atomic_fetch_add(pa.ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (activations made, for the status)
            const uint32_t p = (uint32_t)(__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(pa.ctl + 8), 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
            __hip_atomic_store(pa.ring + p % (uint32_t)pa.cap, ((unsigned long long)(p + 1u) << 32) | ((unsigned long long)(uint32_t)(t + 1) << 16) | (unsigned long long)(uint32_t)s,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else __hip_atomic_fetch_add(pa.ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
}


template <int LB>
__global__ __launch_bounds__(GQ_NT, GQ_WG_PER_CU) void k_plan_persistent(PersistArgs pa) {
    extern __shared__ __attribute__((aligned(16))) double lds_pose[];
    const int tid = (int)threadIdx.x;
    // where this workgroup runs: XCC_ID (hardware register 20) and the CU's coordinates inside it (HW_ID bits 8-15: CU, SH, SE) —
    // used for cache affinity and for the roles below, never for correctness
    const int xcd = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u);
    const uint32_t cu_key = ((uint32_t)xcd << 8) | ((__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) >> 8) & 0xffu);
    unsigned long long* const xw0 = pa.xw + 16 * PQ_SLOTS * xcd;  // this XCD's claim words, one 128-byte line each
    int my_slot = (int)((blockIdx.x >> 3) % PQ_SLOTS);                  // where this workgroup looks first (tid 0 only)
    int* const bc = reinterpret_cast<int*>(lds_pose);  // [8] broadcast words at the front of the LDS (dead between items)
    const int LP = 5;                                   // layer pieces per scene (2 links each)
    const unsigned long long ka = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
    unsigned long long* const uht = pa.uq_ht + 16 * xcd;

    // ---------------------------------------------------------------------------------------------------- roles
    // DEDICATED UPDATE CUs.  The update (learner + step) is the scene's critical path and latency-bound; on a CU it shares with four
    // goal workgroups it took 157-190 us against 55 us alone, every scene spent most of its cycle in it, and the chip ran out of items
    // (workgroups idle a third of their time: measured, 100 x 64).  So the first pa.update_cus CUs of every XCD to report take no
    // items: their workgroups serve the XCD's update requests, which the scenes' last items post.  Elected at run time from the hardware's
    // own CU numbers (no assumption about which CUs exist or where the dispatcher puts a workgroup).
    int role = 1;
    if (pa.update_cus > 0) {
        if (tid == 0) {
            uint32_t r = rmw_u32(pa.cu_role + cu_key);
            if (r == 0u) {
                uint32_t expect = 0u;
                if (__hip_atomic_compare_exchange_strong(pa.cu_role + cu_key, &expect, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    const uint32_t idx = __hip_atomic_fetch_add(pa.xcd_cus + 32 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    r = idx < (uint32_t)pa.update_cus ? 2u : 1u;
                    __hip_atomic_store(pa.cu_role + cu_key, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else r = expect;
            }
            for (int spin = 0; r == 3u && spin < 100000; ++spin) { __builtin_amdgcn_s_sleep(1); r = rmw_u32(pa.cu_role + cu_key); }
            if (r != 2u) r = 1u;
            if (r == 2u) __hip_atomic_fetch_add(pa.xcd_cus + 32 * xcd + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bc[0] = (int)r;
        }
        __syncthreads();
        role = __builtin_amdgcn_readfirstlane(bc[0]);
        __syncthreads();
    }
    if (role == 2) {
        // ---- an update workgroup: serve this XCD's update requests until the plan is finished
        for (;;) {
            if (tid == 0) {
                int got_s = -1, got_t = 0;
                const long long t_begin = wall_clock64();
                unsigned long long q = rmw_u64(uht);
                for (int idle = 1;;) {
                    const uint32_t h = (uint32_t)q, tl = (uint32_t)(q >> 32);
                    if ((int32_t)(tl - h) > 0) {
                        if (!__hip_atomic_compare_exchange_strong(uht, &q, q + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                        unsigned long long e = rmw_u64(pa.uq + (size_t)xcd * pa.cap + h % (uint32_t)pa.cap);
                        while ((uint32_t)(e >> 32) != h + 1u && wall_clock64() - t_begin < 200000000LL) { __builtin_amdgcn_s_sleep(1); e = rmw_u64(pa.uq + (size_t)xcd * pa.cap + h % (uint32_t)pa.cap); }
                        if ((uint32_t)(e >> 32) != h + 1u) { __hip_atomic_store(pa.ctl + 3, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        got_s = (int)(e & 0xffffu); got_t = (int)((e >> 16) & 0xffffu);
                        break;
                    }
                    if (rmw_u32(pa.ctl + 2) >= rmw_u32(pa.ctl + 4) || rmw_u32(pa.ctl + 3) != 0u) break;
                    for (int z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(2);
                    idle = idle < 8 ? idle * 2 : 8;
                    if (wall_clock64() - t_begin > 200000000LL) { __hip_atomic_store(pa.ctl + 3, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    q = rmw_u64(uht);
                }
                bc[0] = got_s; bc[1] = got_t;
            }
            __syncthreads();
            const int us = __builtin_amdgcn_readfirstlane(bc[0]), ut = __builtin_amdgcn_readfirstlane(bc[1]);
            __syncthreads();
            if (us < 0) return;
#ifdef OMGX_PERSIST_STATS
            const long long t_upd = wall_clock64();
#endif
            persist_update(pa, ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, us, ut);
#ifdef OMGX_PERSIST_STATS
            if (tid == 0) { PQ_STAT(2, wall_clock64() - t_upd); PQ_STAT(4, 1); }
#endif
            __syncthreads();
        }
    }

    for (;;) {
        // ------------------------------------------------------------------------------------------------ claim an item (one lane)
        // A claim word says WHICH activation it hands out — {scene | iteration | LOCK | VALID | install count | next item} — so a claim is
        // one returning atomic add and needs nothing else from memory.  (The ring slot of an activation that has been taken off the ring
        // may be rewritten while its items are still being claimed: only the installer reads the ring.)  An XCD has PQ_SLOTS such words:
        // while one is being refilled — four dependent round trips to words the whole chip shares — its workgroups claim from the
        // others.  Everything another workgroup may have written is read with a read-modify-write atomic (performed where the word
        // lives, never served from a stale cache line).
        if (tid == 0) {
            int got_s = -1, got_t = 0, got_k = 0, got_n = 0, got_mode = 0;
            const long long t_begin = wall_clock64();
            unsigned spins = 0; (void)spins;
            int nap = 1;
            auto items_of = [&](int s, int t, int& mode) {
                mode = (as_const(pa.iters) + t)->mode;
                const int gs = pa.ca.goal_count ? as_const(pa.ca.goal_count)[s] : pa.G;
                return mode ? LP + gs : LP;
            };
            bool done = false;
            while (!done && got_s < 0) {
                ++spins;
                int free_slot = -1;
                unsigned long long free_word = 0ull;
                for (int a = 0; a < PQ_SLOTS && got_s < 0 && !done; ++a) {
                    const int sl = (my_slot + a) % PQ_SLOTS;
                    unsigned long long* const xw = xw0 + 16 * sl;
                    unsigned long long w = ld_u64(xw);
                    if ((spins & 31u) == 0u) w = rmw_u64(xw);  // (a stale line must not keep a workgroup asleep)
                    if (w == PQ_DONE) { done = true; break; }
                    if (w & PQ_LOCK) continue;  // its installer is at work
                    int mode;
                    if (!(w & PQ_VALID) || (int)(w & PQ_KMASK) >= items_of((int)(w >> 48), (int)((w >> 32) & 0xffffu), mode)) {
                        if (free_slot < 0) { free_slot = sl; free_word = w; }
                        continue;
                    }
                    const unsigned long long old = __hip_atomic_fetch_add(xw, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == PQ_DONE) { __hip_atomic_store(xw, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); done = true; break; }
                    if ((old & PQ_VALID) && !(old & PQ_LOCK)) {
                        const int s = (int)(old >> 48), t = (int)((old >> 32) & 0xffffu), k = (int)(old & PQ_KMASK);
                        const int n2 = items_of(s, t, mode);
                        if (k < n2) { got_s = s; got_t = t; got_k = k; got_n = n2; got_mode = mode; my_slot = sl; }
                    }
                }
                if (done || got_s >= 0) break;
                if (free_slot < 0) {  // every word is being refilled: back off (a look every 0.05 .. 2 us)
                    for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
                    nap = nap < 32 ? nap * 2 : 32;
                    continue;
                }
                // ---- an exhausted (or never filled) word: become its installer — lock it, unless somebody else has
                unsigned long long* const xw = xw0 + 16 * free_slot;
                unsigned long long cur = free_word;
                if (!__hip_atomic_compare_exchange_strong(xw, &cur, cur | PQ_LOCK, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (cur == PQ_DONE) { done = true; break; }
                    // the counter moved on, or somebody locked it: if it is still the same exhausted activation and unlocked, once more
                    if ((cur >> 24) != (free_word >> 24) || (cur & PQ_LOCK)) continue;
                    if (!__hip_atomic_compare_exchange_strong(xw, &cur, cur | PQ_LOCK, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                }
                // ---- take the next activation off the ring (or find the plan finished): ctl64 = {tail | head}
                unsigned long long* const ht = reinterpret_cast<unsigned long long*>(pa.ctl + 8);
                uint32_t pos = 0u;
                bool have = false, finished = false;
                unsigned long long q = rmw_u64(ht);
                for (int idle = 1; !have && !finished;) {
                    const uint32_t h = (uint32_t)q, tl = (uint32_t)(q >> 32);
                    if ((int32_t)(tl - h) > 0) {
                        if (__hip_atomic_compare_exchange_strong(ht, &q, q + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { pos = h; have = true; }
                        continue;  // (a failed exchange left the word's current value in q)
                    }
                    if (rmw_u32(pa.ctl + 2) >= rmw_u32(pa.ctl + 4) || rmw_u32(pa.ctl + 3) != 0u) { finished = true; break; }
                    // nothing to take right now: give the word back (another slot of this XCD may still hold items) after a short wait
                    for (int z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(2);
                    idle = idle < 16 ? idle * 2 : 16;
                    if (wall_clock64() - t_begin > 200000000LL) { __hip_atomic_store(pa.ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); finished = true; break; }
                    q = rmw_u64(ht);
                }
                if (finished) {
                    for (int a = 0; a < PQ_SLOTS; ++a) __hip_atomic_store(xw0 + 16 * a, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                    break;
                }
                unsigned long long e;
                for (;;) {  // the producer took the slot (tail) before it wrote the word: wait for the tag
                    e = rmw_u64(pa.ring + pos % (uint32_t)pa.cap);
                    if ((uint32_t)(e >> 32) == pos + 1u) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t_begin > 200000000LL) { __hip_atomic_store(pa.ctl + 3, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if ((uint32_t)(e >> 32) != pos + 1u) {
                    for (int a = 0; a < PQ_SLOTS; ++a) __hip_atomic_store(xw0 + 16 * a, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                    break;
                }
                const int s = (int)(e & 0xffffu), t = (int)((e >> 16) & 0xffffu);
                int mode;
                const int n = items_of(s, t, mode);
                // installed with item 0 taken by this workgroup; the install count tells this activation from the one before it
                const unsigned long long seq = ((cur >> 24) + 1ull) & 0x3full;
                __hip_atomic_store(xw, ((unsigned long long)(uint32_t)s << 48) | ((unsigned long long)(uint32_t)t << 32) | PQ_VALID | (seq << 24) | 1ull,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got_s = s; got_t = t; got_k = 0; got_n = n; got_mode = mode; my_slot = free_slot;
            }
            bc[0] = got_s; bc[1] = got_t; bc[2] = got_k; bc[3] = got_n; bc[4] = got_mode;
            PQ_STAT(0, spins); PQ_STAT(5, wall_clock64() - t_begin);
        }
        __syncthreads();
        Item it{bc[0], bc[1], bc[2], bc[3], bc[4]};
        it.s = __builtin_amdgcn_readfirstlane(it.s); it.t = __builtin_amdgcn_readfirstlane(it.t); it.k = __builtin_amdgcn_readfirstlane(it.k);
        it.nitems = __builtin_amdgcn_readfirstlane(it.nitems); it.mode = __builtin_amdgcn_readfirstlane(it.mode);
        __syncthreads();  // the broadcast words are the poses' from here on
        if (it.s < 0) return;
        const int start_idx = (as_const(pa.iters) + it.t)->start_idx;  // the Learner's window of this iteration

        // ------------------------------------------------------------------------------------------------ the item
        {
            const bool is_layer = it.k < LP;
#ifdef OMGX_PERSIST_STATS
            const long long t_item = wall_clock64();
#endif
            if (is_layer) persist_layer_item<LB>(ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.k, start_idx);
            else persist_goal_item<LB>(ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.k, start_idx);
#ifdef OMGX_PERSIST_STATS
            if (tid == 0) { PQ_STAT(1, wall_clock64() - t_item); PQ_STAT(3, 1); }
#endif
            // ---- arrival: what this item wrote is visible at agent scope before the counter moves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (is_layer) {  // plain stores of potentials / gradients / collisions / poses: write the L2's dirty lines back
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2: guideline 16, pitfall 12)
                }
                const uint32_t old = __hip_atomic_fetch_add(pa.arrive + it.s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int last_one = (old + 1u == (uint32_t)it.nitems) ? 1 : 0;
                if (last_one && pa.update_cus > 0 && rmw_u32(pa.xcd_cus + 32 * xcd + 1) > 0u) {
                    // the scene's update goes to this XCD's update workgroups (what the items wrote is visible: every item released before it arrived)
                    const uint32_t p = (uint32_t)(__hip_atomic_fetch_add(uht, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
                    __hip_atomic_store(pa.uq + (size_t)xcd * pa.cap + p % (uint32_t)pa.cap,
                                       ((unsigned long long)(p + 1u) << 32) | ((unsigned long long)(uint32_t)it.t << 16) | (unsigned long long)(uint32_t)it.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last_one = 0;
                }
                bc[0] = last_one;
            }
            __syncthreads();
        }
        const int last = __builtin_amdgcn_readfirstlane(bc[0]);
        __syncthreads();
        if (!last) continue;

        // ------------------------------------------------------------------------------------------------ the scene's update, by its last arriver
        // A CALL, not inlined: the learner and the step need more registers than a goal item has (they spill to scratch at this
        // kernel's 96), and inlined they drag the item's main loop into the same allocation — scratch traffic inside the hot loop.
        // Behind a call the item's code is allocated as in k_goalset_queue, and only the update pays for its own spills.
#ifdef OMGX_PERSIST_STATS
        const long long t_upd = wall_clock64();
#endif
        persist_update(pa, ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.t);
#ifdef OMGX_PERSIST_STATS
        if (tid == 0) { PQ_STAT(2, wall_clock64() - t_upd); PQ_STAT(4, 1); }
#endif
        __syncthreads();
    }
}

}  // namespace omg_persist
