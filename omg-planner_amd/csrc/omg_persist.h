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

#define PQ_LOCK 0x80000000u
#define PQ_DONE 0xffffffffffffffffull

struct PersistArgs {
    ChunkArgs ca;                     // a goal-set + layer launch in the batch layout; CH / PS / MR / tbl_n / traj_start follow the iteration
    omg_learner::LearnerArgs la;      // prm.start_idx follows the iteration
    ChompArgs ch;                     // prm.obstacle_weight / smoothness_weight / step_size / do_update follow the iteration
    const omgx_plan_iter* iters;      // [num_iters] device
    int num_iters;
    int G;                            // goals per scene (padded)
    int32_t* active;                  // [S] or null: scenes with 0 are not planned; a scene that terminates under stop_on_terminate gets 0
    unsigned long long* ring;         // [cap] activations
    unsigned long long* xw;           // [8] per-XCD claim words
    uint32_t* ctl;                    // [0] ring tail, [1] ring head, [2] scenes finished, [3] failure code, [4] scenes in the plan
    uint32_t* arrive;                 // [S] items of the scene's current iteration that have finished
    int cap;
    uint32_t lds_bytes;               // dynamic LDS of the launch
};

__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (Re-)initialise the queue for a launch: every active scene activated for iteration 0, in scene order; one workgroup.
__global__ __launch_bounds__(256) void k_persist_init(PersistArgs pa) {
    for (int i = threadIdx.x; i < pa.ca.S; i += 256) pa.arrive[i] = 0u;
    if (threadIdx.x < 8) pa.xw[threadIdx.x] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {  // (serial: the order of the ring is the scenes' order; S <= 65535)
        uint32_t n = 0;
        for (int s = 0; s < pa.ca.S; ++s)
            if (!pa.active || pa.active[s] != 0) { pa.ring[n] = ((unsigned long long)(n + 1) << 32) | (unsigned long long)(uint32_t)s; ++n; }
        pa.ctl[0] = n; pa.ctl[1] = 0u; pa.ctl[2] = 0u; pa.ctl[3] = 0u; pa.ctl[4] = n;
    }
}

// What an activation's item k is: the first layer_parts items are the trajectory layer's pieces, then one per goal.
struct Item { int s, t, k, nitems, mode; };

typedef __attribute__((address_space(3))) unsigned char* LdsBytes;

// a kernel argument block (or a part of it) read through the constant address space: scalar loads, field by field
template <class T>
__device__ __forceinline__ T load_const(unsigned long long addr) {
    static_assert(sizeof(T) % 4 == 0, "dwords");
    T out;
    const OMG_CONST_AS uint32_t* src = (const OMG_CONST_AS uint32_t*)(uintptr_t)addr;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return out;
}

// One work item (omg_goalset_queue.h: gq_item) behind a CALL: the item's code gets the register allocation it has in k_goalset_queue,
// whatever the kernel around it keeps alive.  k < 5: piece k of the scene's trajectory layer, else goal k - 5; start_idx: the learner's
// window of the scene's iteration.
template <int LB, int ROLE>  // ROLE 1: a trajectory-layer piece, 2: a goal — two functions, two register allocations
__device__ __attribute__((noinline)) void persist_item(const unsigned long long pa_addr, LdsBytes lds, const int s_in, const int k_in, const int start_in) {
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

// Learner.update_goal + Optimizer.optimize of scene s at iteration t, then the scene's next activation (or its end); called by the whole
// workgroup that finished the scene's last item.  pa_addr: the kernel's argument block; lds: the workgroup's dynamic LDS.
__device__ __attribute__((noinline)) void persist_update(const unsigned long long pa_addr, LdsBytes lds, const int s_in, const int t_in) {
    const int tid = (int)threadIdx.x;
    Item it{__builtin_amdgcn_readfirstlane(s_in), __builtin_amdgcn_readfirstlane(t_in), 0, 0, 0};
    const PersistArgs pa = load_const<PersistArgs>(pa_addr);
    omgx_plan_iter rec;
    {
        IterTablePtr r = as_const(pa.iters) + it.t;
        rec.mode = r->mode; rec.start_idx = r->start_idx; rec.stop_on_terminate = r->stop_on_terminate; rec.do_update = r->do_update;
        rec.obstacle_weight = r->obstacle_weight; rec.smoothness_weight = r->smoothness_weight; rec.step_size = r->step_size;
    }
    it.mode = rec.mode;
    double* const lds_pose = reinterpret_cast<double*>((unsigned char*)lds);
    {
        if (tid == 0) {
            __hip_atomic_store(pa.arrive + it.s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody touches it before the next activation
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 forgets what other workgroups have rewritten
        }
        __syncthreads();
        unsigned char* const smem = reinterpret_cast<unsigned char*>(lds_pose);
        const double* end_pose = pa.ch.prm.end_poses + (size_t)it.s * 120;
        if (it.mode) {
            omg_learner::LearnerArgs la = pa.la;
            la.prm.start_idx = rec.start_idx;
            la.active = nullptr;  // only scenes in the loop are ever activated
            double* shl = reinterpret_cast<double*>(smem);
            int* const sh_idx = reinterpret_cast<int*>(shl + 5 * OMGX_MAX_GOALS + 5 * 128);
            omg_learner::learner_scene<true>(la, it.s, reinterpret_cast<double (*)[OMGX_MAX_GOALS]>(shl), reinterpret_cast<double (*)[128]>(shl + 5 * OMGX_MAX_GOALS), sh_idx);
            __syncthreads();  // the goal (global memory, this CU) and its index (LDS) are the workgroup's
            const int gi = *sh_idx;
            const double* src = la.prm.goal_pose_table + ((size_t)it.s * la.prm.num_goals + gi) * 120;
            if (tid < 120) la.prm.end_poses_out[(size_t)it.s * 120 + tid] = src[tid];  // (for the fixed-goal iterations and later launches)
            end_pose = src;
            __syncthreads();  // chomp_scene reuses the learner's LDS
        }
        {
            ChompArgs ch = pa.ch;
            ch.prm.obstacle_weight = rec.obstacle_weight; ch.prm.smoothness_weight = rec.smoothness_weight; ch.prm.step_size = rec.step_size;
            ch.prm.do_update = rec.do_update;
            ch.active = nullptr;
            ch.deactivate = (rec.stop_on_terminate && pa.active) ? pa.active : nullptr;
            chomp_scene<1, GQ_NT, true>(ch, smem, it.s, nullptr, 0u, nullptr, end_pose);
        }
        // ---- everything the step left is visible at agent scope, then the scene goes on (or has finished)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool more = it.t + 1 < pa.num_iters;
            if (more && rec.stop_on_terminate && pa.active) more = __hip_atomic_load(pa.active + it.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (more) {
                const uint32_t p = __hip_atomic_fetch_add(pa.ctl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pa.ring + p % (uint32_t)pa.cap, ((unsigned long long)(p + 1u) << 32) | ((unsigned long long)(uint32_t)(it.t + 1) << 16) | (unsigned long long)(uint32_t)it.s,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else __hip_atomic_fetch_add(pa.ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}


template <int LB>
__global__ __launch_bounds__(GQ_NT, GQ_WG_PER_CU) void k_plan_persistent(PersistArgs pa) {
    extern __shared__ __attribute__((aligned(16))) double lds_pose[];
    const int tid = (int)threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u);  // observed placement (block b on XCD b % 8): used for cache affinity only, never for correctness
    unsigned long long* const xw = pa.xw + xcd;
    int* const bc = reinterpret_cast<int*>(lds_pose);  // [8] broadcast words at the front of the LDS (dead between items)
    const int LP = 5;                                   // layer pieces per scene (2 links each)

    for (;;) {
        // ------------------------------------------------------------------------------------------------ claim an item (one lane)
        if (tid == 0) {
            int got_s = -1, got_t = 0, got_k = 0, got_n = 0, got_mode = 0;
            const long long t_begin = wall_clock64();
            for (;;) {
                unsigned long long w = ld_u64(xw);
                if (w == PQ_DONE) break;
                const uint32_t pos1 = (uint32_t)(w >> 32), kk = (uint32_t)w;
                bool exhausted = pos1 == 0u;
                if (pos1 != 0u && !(kk & PQ_LOCK)) {
                    const unsigned long long old = __hip_atomic_fetch_add(xw, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == PQ_DONE) { __hip_atomic_store(xw, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    const uint32_t opos1 = (uint32_t)(old >> 32), ok = (uint32_t)old;
                    if (opos1 != 0u && !(ok & PQ_LOCK)) {
                        const unsigned long long e = ld_u64(pa.ring + (opos1 - 1u) % (uint32_t)pa.cap);  // installed => published
                        const int s = (int)(e & 0xffffu), t = (int)((e >> 16) & 0xffffu);
                        const int mode = as_const(pa.iters)[t].mode;
                        const int gs = pa.ca.goal_count ? as_const(pa.ca.goal_count)[s] : pa.G;
                        const int n = mode ? LP + gs : LP;
                        if ((int)ok < n) { got_s = s; got_t = t; got_k = (int)ok; got_n = n; got_mode = mode; break; }
                        exhausted = true;
                        w = old + 1ull;
                    } else { __builtin_amdgcn_s_sleep(8); continue; }  // locked (an installer is at work) or emptied meanwhile
                } else if (pos1 != 0u) { __builtin_amdgcn_s_sleep(8); continue; }  // locked: its installer is looking for the next activation
                if (exhausted) {
                    // become the installer: lock the word (whatever its item counter has reached), unless somebody else did
                    unsigned long long cur = ld_u64(xw);
                    if (cur == PQ_DONE) break;
                    if ((uint32_t)(cur >> 32) != (uint32_t)(w >> 32) || ((uint32_t)cur & PQ_LOCK)) { __builtin_amdgcn_s_sleep(4); continue; }
                    const unsigned long long locked = (cur & 0xffffffff00000000ull) | PQ_LOCK;
                    if (!__hip_atomic_compare_exchange_strong(xw, &cur, locked, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                    // ---- take the next activation off the ring (or find the plan finished)
                    uint32_t pos = 0u;
                    bool finished = false;
                    for (;;) {
                        uint32_t h = ld_u32(pa.ctl + 1);
                        const uint32_t tl = ld_u32(pa.ctl + 0);
                        if ((int32_t)(tl - h) > 0) {
                            if (__hip_atomic_compare_exchange_strong(pa.ctl + 1, &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { pos = h; break; }
                            continue;
                        }
                        if (ld_u32(pa.ctl + 2) >= ld_u32(pa.ctl + 4) || ld_u32(pa.ctl + 3) != 0u) { finished = true; break; }
                        __builtin_amdgcn_s_sleep(16);
                        if (wall_clock64() - t_begin > 200000000LL) { __hip_atomic_store(pa.ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); finished = true; break; }
                    }
                    if (finished) { __hip_atomic_store(xw, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    unsigned long long e;
                    for (;;) {  // the producer took the slot (tail) before it wrote the word: wait for the tag
                        e = ld_u64(pa.ring + pos % (uint32_t)pa.cap);
                        if ((uint32_t)(e >> 32) == pos + 1u) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (wall_clock64() - t_begin > 200000000LL) { __hip_atomic_store(pa.ctl + 3, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                    if ((uint32_t)(e >> 32) != pos + 1u) { __hip_atomic_store(xw, PQ_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    const int s = (int)(e & 0xffffu), t = (int)((e >> 16) & 0xffffu);
                    const int mode = as_const(pa.iters)[t].mode;
                    const int gs = pa.ca.goal_count ? as_const(pa.ca.goal_count)[s] : pa.G;
                    // installed with item 0 taken by this workgroup
                    __hip_atomic_store(xw, ((unsigned long long)(pos + 1u) << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got_s = s; got_t = t; got_k = 0; got_n = mode ? LP + gs : LP; got_mode = mode;
                    break;
                }
            }
            bc[0] = got_s; bc[1] = got_t; bc[2] = got_k; bc[3] = got_n; bc[4] = got_mode;
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
            const unsigned long long ka = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
            if (is_layer) persist_item<LB, 1>(ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.k, start_idx);
            else persist_item<LB, 2>(ka, (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.k, start_idx);
            // ---- arrival: what this item wrote is visible at agent scope before the counter moves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (is_layer) {  // plain stores of potentials / gradients / collisions / poses: write the L2's dirty lines back
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2: guideline 16, pitfall 12)
                }
                const uint32_t old = __hip_atomic_fetch_add(pa.arrive + it.s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bc[0] = (old + 1u == (uint32_t)it.nitems) ? 1 : 0;
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
        persist_update(reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr()),
                       (LdsBytes)(__attribute__((address_space(3))) void*)lds_pose, it.s, it.t);
        __syncthreads();
    }
}

}  // namespace omg_persist
