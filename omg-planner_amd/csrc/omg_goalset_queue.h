// omg_goalset_queue.h — k_goalset_queue: the goal-set batch (Learner.cost_vector's Cost.batch_obstacle_cost with arc length,
// omg/online_learner.py:104-160, omg/cost.py:192-286) with a per-wave QUEUE of exact-path work.  Included by omg_kernels.hip
// (needs ChunkArgs, waypoint_layer_block, the GS_* debug macros).
//
// Work split as before: one workgroup of 4 waves per (scene, goal); kinematics of the start + n interpolated
// configurations into LDS; one lane per (link, configuration) row tests the link's bounding ball against every object's
// influence region (row masks); then every wave walks its rows — 4 consecutive waypoints x 16 point lanes, LB links per step —
// and far-tests its points against the objects in reach, record in SGPRs.
//
// What is new is what happens to a (point, object) pair that survives the far test.  It becomes a 5-dword ENTRY — the
// object-space offset, the point's arc-length weight ||x_i - x_(i-1)|| / dt and the object's index — and joins a queue
// that lives in a wave-private LDS ring (slot i = entry i; the object index in a register of lane i) across objects, links
// and waypoints.  Only when 64 entries are
// there does the wave run the exact path (grid coordinates, 8 voxels, trilinear value, hinge) — on 64 live lanes instead
// of the 63 % the per-(link pair, object) batches reached — with the per-object constants read per lane from a 64-byte
// LDS record.  The exact path is also split in two: ISSUE (coordinates, addresses, the four 8-byte gathers) and, a whole
// queue-fill later, CONSUME (interpolation, hinge, weighted sums), so that the L2 latency of the gathers is covered by the
// wave's own far tests instead of by other waves.  Nothing between the two uses vector memory (the robot's collision
// points come from LDS too), so the gathers complete in the background.
//
// A pair's arithmetic is operation for operation that of pair_prepare / pair_exact<false> (omg_device.h), i.e. of
// SDFdistanceForward (layers/sdf_matching_loss_kernel.cu:111-171).  A goal's cost is the float32 sum of pot * weight over
// its pairs in queue order (fixed by the program: deterministic, no atomics) instead of per-point sums over objects times
// the weight: equal up to float32 rounding of a sum of ~300 terms (checked against the oracle at 1e-5).
//
// Two instantiations share this source.  k_goalset_queue<2, STAMP, false> is the batch kernel (one workgroup per goal, five per
// trajectory layer, a scene per XCD, five workgroups per CU).  k_goalset_queue<2, false, true> is the LATENCY-MODE kernel for one or
// a few scenes (omgx_goalset_cost_layer_tiled): workgroups in plain order over all XCDs, a goal's tiles dealt over up to 8 of them,
// layer workgroups of one link x 4 waypoints whose waves take the objects side by side, the chain's constants and the joints'
// matrices in LDS, the object records pulled into the scalar cache at the top, every wave culling the rows of its own tiles; a
// goal's cost is then the sum of its parts' sums.  Everything LAT-specific is behind `if constexpr (LAT)` / constant-folded
// conditions: the batch kernel's code does not change with it (tests/test_kernel_budget.py watches its registers).
#pragma once

#define GQ_TBL_MAX 32    // at most this many objects of a scene have their exact-path constants staged in LDS (others: evaluated on the
                         // spot); how many do is chosen per launch (gq_choose_layout): whatever fits the LDS of the occupancy step the launch is on
#ifndef GQ_WG_PER_CU
#define GQ_WG_PER_CU 5   // 32 KB of LDS per workgroup (30 waypoints), <= 96 VGPRs
#endif
#ifndef GQ_WAVES
#define GQ_WAVES 4       // waves per workgroup.  Other values are EXPERIMENT builds (tools/ab_goalset.py; DESIGN_HISTORY.md appendix A): only the
#endif                   // batch kernel behind the kinematics pre-pass (PRE, no split goals) deals its tiles over GQ_WAVES waves
#define GQ_NT (64 * GQ_WAVES)

struct GqFar {  // what the far test of one object needs (wave-uniform, SGPRs)
    float T[12], lo[3], rc[3], rh[3], rr2, rb_r;
    bool cullable;
    int disabled;
};

// EVERY field in ONE trip through the scalar cache, by three wide loads (the record's dwords 0-15, 22-25, 34-41) and one wait.
// Written field by field the compiler issues nine loads and — whatever ties them together in the source — waits two or three
// times for them (it sinks the fields the `disabled` test does not need behind that test): 50-140 cycles per trip on the
// critical path of every (tile, object) iteration, and the scalar unit's issue slots are as scarce as the vector unit's in this
// kernel (tools/issue_probe.hip: a scalar instruction occupies its SIMD for 4.4 cycles, a float32 VALU instruction for 2).
// `cullable` — epsilon < 1 and clearance <= 1, else a lookup outside the volume (value 1.0) still adds something — is decided on
// the bit patterns: for floats that are not NaN the signed integer order of the patterns is the float order on one side of zero,
// and a negative value (sign bit) is below 1.0 either way; a NaN with its sign bit set would read "cullable", where the hinge
// terms are NaN-free zeros anyway.  No vector compare, no wait for a VALU result on the scalar unit.
typedef uint32_t GqU16 __attribute__((ext_vector_type(16)));
typedef uint32_t GqU8 __attribute__((ext_vector_type(8)));
typedef uint32_t GqU4 __attribute__((ext_vector_type(4)));
static_assert(offsetof(omgx_object, pose_inv) == 0 && offsetof(omgx_object, lo) == 0x30 && offsetof(omgx_object, epsilon) == 0x58 &&
              offsetof(omgx_object, clearance) == 0x60 && offsetof(omgx_object, disabled) == 0x64 && offsetof(omgx_object, rb_c) == 0x88 &&
              offsetof(omgx_object, rb_h) == 0x94 && offsetof(omgx_object, rb_r) == 0xa0 && offsetof(omgx_object, rb_r2) == 0xa4 &&
              sizeof(omgx_object) >= 0xa8, "gq_load_far reads the record by byte offsets");
__device__ __forceinline__ GqFar gq_load_far(ObjTablePtr ob) {
    GqU16 a;
    GqU4 b;
    GqU8 c;
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x58\n\ts_load_dwordx8 %2, %3, 0x88\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"((uint64_t)(uintptr_t)ob));
    GqFar f;
#pragma unroll
    for (int k = 0; k < 12; ++k) f.T[k] = __uint_as_float(a[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { f.lo[k] = __uint_as_float(a[12 + k]); f.rc[k] = __uint_as_float(c[k]); f.rh[k] = __uint_as_float(c[3 + k]); }
    f.rb_r = __uint_as_float(c[6]);
    f.rr2 = __uint_as_float(c[7]);
    f.cullable = ((int32_t)b[0] < 0x3f800000) & ((int32_t)b[2] <= 0x3f800000);
    f.disabled = (int32_t)b[3];
    return f;
}
// u = T (x, y, z, 1) with the rows of T in SGPRs (SE3(pose) * point, .cu:125-133: fma(T2, z, fma(T1, y, fma(T0, x, T3))) per row) as
// nine v_fma_f32 in ONE asm statement, the three rows' chains interleaved.  Left to itself the compiler packs two rows into
// v_pk_fma_f32 and pays two s_mov per packed operand to bring their coefficients side by side — six scalar instructions per far test,
// and a scalar instruction occupies the SIMD for longer than a float32 multiply-add (tools/issue_probe.hip).
__device__ __forceinline__ void se3_apply_s(const float* T, float x, float y, float z, float& ux, float& uy, float& uz) {
    ux = T[3]; uy = T[7]; uz = T[11];
    asm("v_fma_f32 %0, %3, %12, %0\n\tv_fma_f32 %1, %6, %12, %1\n\tv_fma_f32 %2, %9, %12, %2\n\t"
        "v_fma_f32 %0, %4, %13, %0\n\tv_fma_f32 %1, %7, %13, %1\n\tv_fma_f32 %2, %10, %13, %2\n\t"
        "v_fma_f32 %0, %5, %14, %0\n\tv_fma_f32 %1, %8, %14, %1\n\tv_fma_f32 %2, %11, %14, %2"
        : "+v"(ux), "+v"(uy), "+v"(uz)
        : "s"(T[0]), "s"(T[1]), "s"(T[2]), "s"(T[4]), "s"(T[5]), "s"(T[6]), "s"(T[8]), "s"(T[9]), "s"(T[10]), "v"(x), "v"(y), "v"(z));
}
// votes as scalar tests of the lane mask itself (HIP's __any goes through an int per lane: a select and a second compare)
__device__ __forceinline__ unsigned long long wave_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// dynamic LDS behind the poses (bytes, all 16-byte aligned): row masks | exact-path records | collision points | staging
struct GqLayout {
    int mask_off, tile_off, tbl_off, pts_off, stage_off, fkc_off, objc_off, btab_off, total;
    __host__ __device__ GqLayout(int PS, int MR, int P, int tbl_n, bool with_fkc = false, int waves = GQ_WAVES) {
        mask_off = PS * 90 * 8;
        tile_off = mask_off + 10 * MR * 4;  // 3 words behind the row masks: bit (block * 5 + link pair) = the tile has a row in reach of something
        tbl_off = mask_off + ((10 * MR * 4 + 16 + 15) & ~15);
        pts_off = tbl_off + tbl_n * 64;
        stage_off = pts_off + ((10 * P * 3 * 8 + 15) & ~15);
        total = stage_off + waves * 64 * 16;  // a 1 KB ring per wave
        // the (sin, cos) table of the kinematics [PS][7][2] doubles borrows the queues' region (first used in the main loop), so
        // that the records and collision points can be staged while the kinematics run
        const int fk = stage_off + PS * 14 * 8 + 16;  // + the chain waves' progress flags (4 words)
        if (total < fk) total = fk;
        total = (total + 15) & ~15;
        fkc_off = total;  // latency mode: the kinematic chain's 246 constants (RobotViewT::uvw .. rf)
        if (with_fkc) total += 246 * 8;
        objc_off = total;  // latency mode: per-object contributions of the layer's object-parallel evaluation, [8][64][5] floats
        if (with_fkc) total += 8 * 64 * 5 * 4;
        btab_off = total;  // latency mode: the joints' matrices of every configuration, [PS][7][9] doubles (fk_chain_row_B)
        if (with_fkc) total += PS * 7 * 9 * 8;
    }
};

// How many exact-path records to stage: as many as still fit the LDS of the occupancy step the launch is on anyway.  The steps
// are what a CU admits (tools/lds_occupancy_probe.hip: 6 workgroups up to 26 624 B, 5 up to 31 744 B, 4 up to 40 960 B — at
// 32 768 B the occupancy API still says 5 but only 4 become resident).
static inline int gq_choose_tbl_n(int PS, int MR, int P, bool with_fkc = false, int waves = GQ_WAVES) {
    static const int step[] = {26624, 31744, 40960, 53248, 80896, 163840};
    const int base = GqLayout(PS, MR, P, 0, with_fkc, waves).total;
    for (int k = 0; k < 6; ++k)
        if (base + 4 * 64 <= step[k]) {  // at least 4 records
            const int n = (step[k] - base) / 64;
            return n < GQ_TBL_MAX ? n : GQ_TBL_MAX;
        }
    return 0;
}

struct GqTblRec { double rw[3]; int32_t dim[3]; int64_t goffb; float eps, clr, pad, i2e; };
__device__ __forceinline__ GqTblRec gq_tbl_load(const omgx_object* ob) {
    GqTblRec r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.rw[k] = ob->inv_extent[k]; r.dim[k] = ob->dim[k]; }
    r.goffb = ob->grid_offset * 4;
    r.eps = ob->epsilon; r.clr = ob->clearance; r.pad = ob->padding_scale; r.i2e = ob->inv_2eps;
    return r;
}
// record (16 dwords): [0..5] 1 / extent as doubles | [6..8] dims | [9,10] byte offset of the grid in the pool |
//                     [11] eps / 2 | [12] eps | [13] clearance | [14] padding scale | [15] 1 / (2 eps)
__device__ __forceinline__ void gq_tbl_store(uint32_t* e, const GqTblRec& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e[2 * k] = (uint32_t)__double2loint(r.rw[k]); e[2 * k + 1] = (uint32_t)__double2hiint(r.rw[k]);
        e[6 + k] = (uint32_t)r.dim[k];
    }
    e[9] = (uint32_t)(r.goffb & 0xffffffffll); e[10] = (uint32_t)(r.goffb >> 32);
    e[11] = __float_as_uint(0.5f * r.eps);  // exact: (double)(0.5f * eps) == 0.5 * (double)eps
    e[12] = __float_as_uint(r.eps); e[13] = __float_as_uint(r.clr);
    e[14] = __float_as_uint(r.pad); e[15] = __float_as_uint(r.i2e);
}

// LAT: the latency-mode variant (omgx_goalset_cost_layer_tiled with `spread`): workgroups in plain (scene, item) order over all
// XCDs, a goal's tiles dealt over a.NP workgroups, the kinematic chain's constants staged in LDS (a workgroup alone on a cold
// CU pays a scalar-cache miss per joint otherwise: 6.5 us of chain for 9 configurations, measured).  LAT = false compiles to
// exactly the batch kernel.
// SPLIT: the batch kernel with a goal's tiles dealt over a.NP workgroups (omgx_goalset_cost_layer_parts; mid-size batches whose launch
// is a round or two of the chip's workgroup slots: half as long a workgroup, the kinematics run in every part).  Everything else —
// scene per XCD, dispatch schedule over (scene, goal, part) items, five workgroups per CU — is the batch kernel's.
// PRE: the goals' link poses and row masks come from k_goalset_kin's workspace (omg_goalset_kin.h) — the workgroup's prologue is ONE
// trip to memory (poses, masks, collision points, records) instead of the kinematics and the culling; everything from the main loop
// on is the same code on the same LDS contents.
// PERSIST (omg_persist.h: the persistent planner kernel calls this once per work item): what OTHER workgroups of the same launch wrote or
// will read goes through agent-scope accesses — the trajectory is read with sc1 loads (the step's workgroup stored it with sc1 stores), the
// goal's cost and collision count are stored with sc1 stores (their reader is the scene's last workgroup to arrive); everything else is the
// same code.  (MI355X_MICROARCH.md, inter-workgroup visibility: 8-byte agent-scope atomics on both sides need no fence.)
template <bool PERSIST>
__device__ __forceinline__ double gq_ld_traj(const double* p) {
    if constexpr (PERSIST) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool PERSIST>
__device__ __forceinline__ void gq_st_out(float* p, float v) {
    if constexpr (PERSIST) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// One work item of k_goalset_queue — a goal (or one part of it) or one piece of a scene's trajectory layer — by a whole workgroup.
// ROLE (omg_persist.h compiles the two kinds of item as functions of their own): 0 = either, 1 = a trajectory-layer piece, 2 = a goal.
// W (round 6): waves per workgroup.  4 everywhere but in the WIDE instantiations of the batch kernel (whole goals, own kinematics), which
// mid-size launches use: a launch whose workgroups all start at once lasts as long as its heaviest goal, and a goal's tiles are drawn by
// however many waves there are.  Same masks, same tiles, same exact sum: same bits.
// RANGE (round 6; with SPLIT, NP = 2): the two parts of a goal are RANGES of its window — part 0 the waypoints 1 .. a.range_h, part 1 the
// rest — instead of tiles dealt round-robin: each part runs the kinematics of ITS configurations only (+ the one before them, which the
// first waypoint's arc-length weight needs) and holds only their poses.  For long windows that is the difference between three
// workgroups per CU and five (50 waypoints: 50 KB of LDS against 29 KB), and no kinematics run twice.  A part is a goal workgroup of a
// shorter window whose configurations are numbered from cfg_off on: everything behind the kinematics is the whole-goal code, tile bits and
// drawn tile list included.  Rows, masks and terms are those of the whole goal; the two partial sums are exact, their float32 roundings
// are added by the learner like the tile parts' (omgx_learner_params.cost_parts).
template <int LB, bool STAMP, bool LAT, bool SPLIT, bool PRE, bool PERSIST = false, int ROLE = 0, int W = GQ_WAVES, bool RANGE = false>
__device__ __forceinline__ void gq_item(const ChunkArgs& a, double* const lds_pose, const int s, const bool is_layer, const int layer_part, const int chunk, const int NP) {
    const int o_begin = as_const(a.scene_begin)[s], o_end = as_const(a.scene_begin)[s + 1];
    const int P = a.P;
    // LAT: the scene's object records and the links' radii into the scalar cache, asynchronously (consumed after the (sin, cos) stage)
    const bool warming = LAT && o_end > o_begin;
    // A goal's TILES may be dealt over a.NP workgroups (latency mode): chunk = goal * NP + part, part takes the tiles t with
    // t % NP == part — (waypoint block + link pair) % NP for NP = 2, 4: the heavy tiles (last waypoints, hand links) are spread
    // evenly — and runs the kinematics of all configurations itself (the chain's latency does not depend on their number).
    const int goal = NP > 1 ? chunk / NP : chunk;
    const int part = NP > 1 ? chunk - goal * NP : 0;
    static_assert(!RANGE || (SPLIT && !LAT && !PRE && !PERSIST && W == GQ_WAVES), "waypoint ranges: the batch kernel with two parts per goal and their own kinematics");
    const int CHg = a.CH;                                     // the goal's whole window
    const int cfg_off = (RANGE && part > 0) ? a.range_h : 0;  // configuration 0 of this workgroup in the window's numbering
    const int CH = RANGE ? (part > 0 ? CHg - a.range_h : a.range_h) : CHg;
    // The blocks of 4 waypoints are aligned with the END of the window: when CH is no multiple of 4 the short block is the first one
    // (far from the goal, mostly culled as a whole) instead of the last one — the heaviest tiles, which then ran at half their lanes.
    const int blk_shift = (4 - (CH & 3)) & 3;
    const int p = threadIdx.x & 15, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably wave-uniform: tile indices and their address arithmetic stay on the scalar unit
    const int tid = (int)threadIdx.x;
    const RobotViewS rv(a.robot, P);
    const int pstride = a.PS, MR = a.MR;
    static_assert(W == GQ_WAVES || (!LAT && !SPLIT && !PRE && !PERSIST && W >= 4), "wide workgroups: the batch kernel with whole goals and its own kinematics");
    constexpr int NT = 64 * W;  // threads of the workgroup
    const GqLayout L(pstride, MR, P, a.tbl_n, LAT, W);
    char* const lds_bytes = reinterpret_cast<char*>(lds_pose);
    uint32_t* const rowmask = reinterpret_cast<uint32_t*>(lds_bytes + L.mask_off);
    uint32_t* const tilebits = reinterpret_cast<uint32_t*>(lds_bytes + L.tile_off);  // goal workgroups: which tiles have anything in reach
    if (ROLE == 1 || (ROLE == 0 && is_layer)) {
        const int lgi = a.layer_nb > 1 ? layer_part / a.layer_nb : layer_part, cbi = layer_part - lgi * a.layer_nb;
        const int lpg = 10 / a.layer_lg, c_begin = cbi * a.layer_cb;
        const int c_end = c_begin + a.layer_cb < a.wp_n ? c_begin + a.layer_cb : a.wp_n;
        waypoint_layer_block<LAT, PERSIST>(a, s, lgi * lpg, (lgi + 1) * lpg, c_begin, c_end, lds_pose, rowmask, o_begin, o_end, rv,
                                  reinterpret_cast<double*>(lds_bytes + L.fkc_off), warming,
                                  reinterpret_cast<float*>(lds_bytes + L.objc_off), reinterpret_cast<double*>(lds_bytes + L.btab_off));
        GS_WG_STAMP(4);
        return;
    }
    if (a.goal_count && goal >= as_const(a.goal_count)[s]) {  // padding of a ragged goal set
        if (STAMP && tid == 0) a.work[(int64_t)s * a.NCH + chunk] = 0u;
        return;
    }
    const unsigned long long work_t0 = STAMP ? wall_clock64() : 0ull;
    GS_FREQ_BEGIN();
    uint32_t* const tbl = reinterpret_cast<uint32_t*>(lds_bytes + L.tbl_off);      // [a.tbl_n][16]
    double* const pts = reinterpret_cast<double*>(lds_bytes + L.pts_off);          // [10][P][3]
    float* const stage = reinterpret_cast<float*>(lds_bytes + L.stage_off) + wave * 256;  // wave-private [64][4]

    // Row-level culling of row (link l, configuration ci): the link's bounding ball against every object's influence region
    // -> bit mask of the objects in reach (objects >= 31 share the last bit).
    auto cull_row = [&](int l, int ci) {
        const double* A = lds_pose + 9 + ((int64_t)l * pstride + ci) * 9;
        const auto bl = rv.ball(l);  // the ball around the link's own points (robot blob BALL), not the one about its frame origin
        float cx, cy, cz;
        link_ball_center(A, A + 3, A + 6, bl, cx, cy, cz);
        const float rad = (float)bl[3] + 1.0e-4f;
        uint32_t m = 0;
        for (int o = o_begin; o < o_end; ++o) {
            ObjTablePtr ob = as_const(a.objects) + o;
            // the record in ONE trip through the scalar cache (gq_load_far: written field by field the compiler waits for `disabled`,
            // then for epsilon / clearance, then for the rest — two to three dependent trips per object and pass on the prologue's tail)
            const GqFar f = gq_load_far(ob);
            if (f.disabled > 0) continue;
            const int oo = o - o_begin;
            const uint32_t bit = 1u << (oo < 31 ? oo : 31);
            float ux, uy, uz;
            se3_apply_s(f.T, cx, cy, cz, ux, uy, uz);
            const bool near = rbox_near(ux - f.lo[0], uy - f.lo[1], uz - f.lo[2], rad, f.rc, f.rh, f.rb_r);
            if (near || !f.cullable) m |= bit;
        }
        rowmask[l * CH + ci] = m;
        return m;
    };
    // TILE BITS (round 5): 28 of a goal's 40 tiles have no row in reach of anything, and finding that out in the main loop costs two
    // mask reads, an OR and a vote per tile.  The culling leaves one bit per tile instead — bit 4 * block + (pair & 3) of word pair >> 2,
    // behind the masks — and an empty tile is a scalar bit test.  Only where the tiles are dealt plainly and a link pair's rows fit one
    // pass (batch kernel, whole goals, own kinematics, window <= 32: eight blocks); elsewhere every bit is set.
    // Round 6: windows of 33 .. 64 waypoints (BASELINE config 5 plans with 50) have up to 16 blocks: bit 4 (block & 7) + pair of word
    // block >> 3 for the pairs 0-3, bit `block` of word 3 for pair 4 (word 2 stays the tile counter) — and their up to 80 tiles get the
    // drawn list too (two entries per lane).  They used to visit all 65 tiles, one LDS atomic each, votes and mask reads included.
    constexpr bool TILEBITS = !LAT && (!SPLIT || RANGE) && !PRE && LB == 2;
    const bool tb_on = TILEBITS;
    const bool tb_wide = CH > 32;  // which of the two bit layouts
    auto tile_bit_set = [&](int rb, int pr) -> bool {  // after the culling (LDS reads)
        if (!tb_wide) return (tilebits[pr >> 2] >> ((4 * rb + (pr & 3)) & 31)) & 1u;
        return pr < 4 ? ((tilebits[rb >> 3] >> (4 * (rb & 7) + pr)) & 1u) : ((tilebits[3] >> rb) & 1u);
    };
    if (tid < 4) tilebits[tid] = (tb_on || tid >= 2) ? 0u : 0xffffffffu;  // (ordered before the culling by the barrier behind the (sin, cos) stage; word 2: the main loop's tile counter)
    // The chain stage of the kinematics keeps ceil(3 (CH + 1) / 64) waves busy (one lane per (configuration, pose row)) and
    // produces the links' poses in order; the other waves cull the rows of a link as soon as every chain wave has published
    // it (a progress word per chain wave in LDS, release / acquire at workgroup scope): the culling stage disappears behind
    // the chain.  With 64 waypoints all four waves run the chain and the rows are culled afterwards.
    // LAT: the q-th tile of this wave.  The part's tiles (t = part + NP j) are dealt to the four waves heaviest first — the work grows
    // with the waypoint block and towards the hand, i.e. with j — in a serpentine (ranks 0 1 2 3 | 3 2 1 0 | 0 1 ...), so the wave that
    // gets the heaviest tile gets the lightest ones with it; dealt in plain order (j mod 4) the waves with three tiles also held
    // the two heaviest.  < 0: no such tile.
    auto lat_tile = [&](int q) {
        const int ntl = ((CH + 3) >> 2) * (10 / LB);
        const int ntp = ntl > part ? (ntl - part + NP - 1) / NP : 0;  // tiles of this part
        const int r = 4 * q + ((q & 1) ? 3 - wave : wave);             // rank by weight, heaviest = 0
        return r < ntp ? part + NP * (ntp - 1 - r) : -1;
    };
#ifndef OMGX_GS_BTAB_IN_PLACE
#define OMGX_GS_BTAB_IN_PLACE 1  // batch kernel: the joints' matrices tabulated IN the pose array before the chain (see below)
#endif
    // Batch kernel: the chain's lanes are (configuration, pose row) with a configuration's three rows in ONE wave (21 configurations
    // per wave) — joint i's matrix B_i = c U + s V + W of a configuration is tabulated beforehand (one lane per configuration, the
    // joint wave-uniform: its 27 constants are scalar operands) into the 9 doubles that will hold link i's pose of that configuration,
    // and the chain step reads it, multiplies (12 multiply-adds instead of 27 + 12 and 60 dwords of scalar loads per joint on the
    // stage's critical wave) and writes the pose over it: a wave's LDS operations execute in order, so all three row lanes have read
    // the matrix before any of them writes.  No extra LDS; same expressions, same bits (fk_joint_matrix).  (Requesting joint i + 1's
    // matrix before the arithmetic of joint i and dropping the flag's release wait changed nothing: 4.2 us for the chain either way.)
    constexpr bool BTAB = !LAT && OMGX_GS_BTAB_IN_PLACE != 0;
    const int chain_waves = BTAB ? (CH + 1 + 20) / 21 : (3 * (CH + 1) + 63) >> 6;
    // Only while two waves are free and a link pair's rows fit one pass (CH <= 32): with 50 waypoints a single free wave walked the 10
    // links one pass each and the prologue got LONGER (0.49 instead of 0.41 ms per step at 50 waypoints: measured) — there all four
    // waves cull after the chain as before.  LAT: every wave culls the rows of its own tiles after the chain (below).
    const bool cull_beside_chain = !PRE && !LAT && chain_waves <= 2 && CH <= 32;
    if constexpr (PRE) {
        // Everything the main loop starts from, requested before the first LDS store: the goal's poses (workspace layout
        // [link][component][configuration] -> LDS [link][configuration][9]) and row masks, the robot's collision points, the records.
        static_assert(30 * OMGX_MAX_POINTS <= 2 * GQ_NT && 10 * OMGX_MAX_WAYPOINTS <= 3 * GQ_NT && GQ_WAVES >= 4,
                      "the prologue loads the collision points in two passes and the row masks in three, and the final reduction reads four waves' sums");
        const int ncfg = CH + 1;
        const int64_t gi = (int64_t)s * a.NG + goal;
        const double* pw = a.pre_poses + gi * gk_pose_doubles(CH);
        const uint32_t* mw = a.pre_masks + gi * (int64_t)(10 * CH);
        const double pv0 = tid < 30 * P ? rv.g[246 + tid] : 0.0, pv1 = tid + GQ_NT < 30 * P ? rv.g[246 + tid + GQ_NT] : 0.0;
        const bool tbl_lane = tid >= 128 && tid - 128 < a.tbl_n && o_begin + tid - 128 < o_end;
        GqTblRec trec{};
        if (tbl_lane) trec = gq_tbl_load(a.objects + o_begin + (tid - 128));
        if (warming) gq_warm_scalar_cache(a.objects, o_begin, o_end, a.robot + OMGX_ROBOT_POINTS + 30 * P + 316 + 30 * P);
        const int nm = 10 * CH, ne = 90 * ncfg;
        uint32_t mv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) mv[j] = tid + GQ_NT * j < nm ? mw[tid + GQ_NT * j] : 0u;  // 10 x 64 rows at most
        // element e = tid + 256 j of the goal's poses sits at (lk = e / ncfg, c = e % ncfg): one division per thread, then steps
        const int dq = GQ_NT / ncfg, dr = GQ_NT - dq * ncfg;
        int lk = tid / ncfg, c = tid - lk * ncfg;
        for (int e0 = tid; e0 < ne; e0 += GQ_NT * 12) {
            double v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = e0 + GQ_NT * j < ne ? pw[e0 + GQ_NT * j] : 0.0;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int l = lk / 9, k = lk - 9 * l;
                if (e0 + GQ_NT * j < ne) lds_pose[((size_t)l * pstride + c) * 9 + k] = v[j];
                c += dr; lk += dq;
                if (c >= ncfg) { c -= ncfg; ++lk; }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) if (tid + GQ_NT * j < nm) rowmask[tid + GQ_NT * j] = mv[j];
        if (tid < 30 * P) pts[tid] = pv0;
        if (tid + GQ_NT < 30 * P) pts[tid + GQ_NT] = pv1;
        if (tbl_lane) gq_tbl_store(tbl + (tid - 128) * 16, trec);
        GS_WG_STAMP(1);
    } else
    {   // Kinematics of the start + CH interpolated configurations in two stages (omg_device.h: fk_joint_sincos on
        // (configuration, joint) lanes, fk_chain_row on (configuration, pose row) lanes); the (sin, cos) table borrows the
        // region behind the poses, which is first written after the barriers below.
        // cfg 0 = start itself, cfg i+1 = linspace(0,1,n+2)[1:-1][i]: numpy's linspace is i * step with step = fl(1 / (n + 1)),
        // not i / (n + 1) — bit-identical to util.py:261-290 (interp1d)
        const double* q0 = a.traj_start + a.ts_stride * (int64_t)s;
        const double* qg = a.goals + ((int64_t)s * a.NG + goal) * 9;
        const int ncfg = CH + 1;
        double* sc = reinterpret_cast<double*>(lds_bytes + L.stage_off);  // [ncfg][7][2]: the queues' region, first used in the main loop
        int* const progress = reinterpret_cast<int*>(lds_bytes + L.stage_off + pstride * 14 * 8);  // [4] links a chain wave has published
        if (tid < 4) progress[tid] = (tid < chain_waves || (cull_beside_chain && tid == 3)) ? 0 : 99;  // [3]: the culling's pair counter when two waves are free
        // the robot's collision points -> LDS: the loads are issued here and land while the (sin, cos) stage runs
        const double pv0 = tid < 30 * P ? rv.g[246 + tid] : 0.0, pv1 = (tid < 256 && tid + 256 < 30 * P) ? rv.g[246 + tid + 256] : 0.0;  // (30 P <= 480)
        double* const fkc = reinterpret_cast<double*>(lds_bytes + L.fkc_off);  // LAT: the chain's constants, one coalesced load
        const double fkv = (LAT && tid < 246) ? rv.g[tid] : 0.0;
        const bool tbl_lane = tid >= 128 && tid - 128 < a.tbl_n && o_begin + tid - 128 < o_end;  // lanes of wave 2: idle during the chain stage
        GqTblRec trec{};
        if (LAT && tbl_lane) trec = gq_tbl_load(a.objects + o_begin + (tid - 128));  // LAT: requested here, stored after the barrier
        if (warming) gq_warm_scalar_cache(a.objects, o_begin, o_end, a.robot + OMGX_ROBOT_POINTS + 30 * P + 316 + 30 * P);
        auto joint = [&](int cfg_local, int d) {  // (RANGE: this workgroup's configuration cfg_local is the window's cfg_off + cfg_local)
            const int cfg = cfg_local + cfg_off;
            const double q0d = gq_ld_traj<PERSIST>(q0 + d);
            return cfg == 0 ? q0d : q0d + ((double)cfg * (1.0 / (double)(CHg + 1))) * (qg[d] - q0d);
        };
        for (int t = tid; t < ncfg * 7; t += NT) {
            const int cfg = t / 7, i = t - cfg * 7;
            double sn, cs;
#ifdef OMGX_GS_PRO2  // measurement build: the (sin, cos) stage's arithmetic twice — what a prologue instruction costs the step
            {
                double qq = joint(cfg, i);
                asm volatile("" : "+v"(qq));
                fk_joint_sincos(qq, sn, cs);
                asm volatile("" : : "v"(sn), "v"(cs));
            }
#endif
            fk_joint_sincos(joint(cfg, i), sn, cs);
            sc[2 * t] = sn; sc[2 * t + 1] = cs;
        }
        if (LAT && tid < 246) fkc[tid] = fkv;
        __syncthreads();
        GS_WG_STAMP(1);
        if (tid < 30 * P) pts[tid] = pv0;
        if (tid < 256 && tid + 256 < 30 * P) pts[tid + 256] = pv1;
        // ---- exact-path records of the scene's first a.tbl_n objects -> LDS, by lanes of wave 2 (idle during the chain stage)
        if (tbl_lane) {
            if (!LAT) trec = gq_tbl_load(a.objects + o_begin + (tid - 128));
            gq_tbl_store(tbl + (tid - 128) * 16, trec);
        }
        auto run_chain = [&](const auto& view) {
            for (int t = tid; t < ncfg * 3; t += NT) {
                const int cfg = t / 3, rr = t - cfg * 3;
                fk_chain_row(view, rr, sc + 14 * cfg, joint(cfg, 7), joint(cfg, 8), [&](int l, double r0, double r1, double r2, double tr) {
                    double* dst = lds_pose + ((size_t)l * pstride + cfg) * 9;  // rows 0 and 1 of R, then t
                    if (rr < 2) { dst[3 * rr] = r0; dst[3 * rr + 1] = r1; dst[3 * rr + 2] = r2; }
                    dst[6 + rr] = tr;
                    // link l of this wave's configurations is in LDS (a wave's LDS operations execute in order; the release keeps the
                    // compiler from moving the flag ahead of the pose)
                    if (lane == 0) __hip_atomic_store(progress + wave, l + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                });
            }
        };
        if constexpr (LAT) {
            // the joints' matrices first, one lane per (configuration, joint) like the (sin, cos) stage (whose table they read), then
            // the chain over the tabulated matrices: its 7 dependent steps shrink from 30 + 30 to 12 + 12 reads / multiply-adds each
            const RobotView rvl(a.robot, P, fkc);
            double* const btab = reinterpret_cast<double*>(lds_bytes + L.btab_off);
            for (int t = tid; t < ncfg * 7; t += 256) fk_joint_matrix(rvl, t - (t / 7) * 7, sc[2 * t], sc[2 * t + 1], btab + 9 * t);
            __syncthreads();
            for (int t = tid; t < ncfg * 3; t += 256) {
                const int cfg = t / 3, rr = t - cfg * 3;
                fk_chain_row_B(rvl, rr, btab + 63 * cfg, joint(cfg, 7), joint(cfg, 8), [&](int l, double r0, double r1, double r2, double tr) {
                    double* dst = lds_pose + ((size_t)l * pstride + cfg) * 9;  // rows 0 and 1 of R, then t
                    if (rr < 2) { dst[3 * rr] = r0; dst[3 * rr + 1] = r1; dst[3 * rr + 2] = r2; }
                    dst[6 + rr] = tr;
                });
            }
        } else if constexpr (BTAB) {
            for (int j = wave; j < 7; j += W)  // wave-uniform joint: U, V, W of the joint are scalar operands
                for (int cfg = lane; cfg < ncfg; cfg += 64)
                    fk_joint_matrix(rv, j, sc[2 * (cfg * 7 + j)], sc[2 * (cfg * 7 + j) + 1], lds_pose + ((size_t)j * pstride + cfg) * 9);
            __syncthreads();
            GS_WAVE_STAMP(4 + wave);
            if (wave < chain_waves && lane < 63) {
                const int cfg = 21 * wave + lane / 3, rr = lane - 3 * (lane / 3);
                if (cfg < ncfg)
                    fk_chain_row_B(rv, rr, lds_pose + (size_t)cfg * 9, joint(cfg, 7), joint(cfg, 8), [&](int l, double r0, double r1, double r2, double tr) {
                        double* dst = lds_pose + ((size_t)l * pstride + cfg) * 9;  // rows 0 and 1 of R, then t
                        if (rr < 2) { dst[3 * rr] = r0; dst[3 * rr + 1] = r1; dst[3 * rr + 2] = r2; }
                        dst[6 + rr] = tr;
                        if (lane == 0) __hip_atomic_store(progress + wave, l + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }, pstride * 9);
                GS_WAVE_STAMP(6 + wave);  // (instrumented build: when this chain wave finished the chain; overwrites the entry stamps of waves 2, 3)
            }
        } else run_chain(rv);                                          // constants through the scalar cache (warm in a batch)
        if (cull_beside_chain) {
            // The rows of a link PAIR fit one wave (CH <= 32: lanes 0-31 link l, lanes 32-63 link l + 1).  The five pairs are claimed
            // from a counter in LDS, in order: the waves that do not run the chain start at once (a pair is culled as soon as every
            // chain wave has published its later link), the chain waves join when they are done — two waves walking 3 + 2 passes were
            // the longest thing in the prologue (8 us beside a 4 us chain: tools/gs_wave_clock.py).  Who culls a row does not change
            // its mask.
            int* const next_pair = progress + 3;  // (chain_waves <= 2 here: the word is no progress flag)
            for (;;) {
                int pr = 0;
                if (lane == 0) pr = __hip_atomic_fetch_add(next_pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pr = __builtin_amdgcn_readfirstlane(pr);
                if (pr >= 5) break;
                const int l = 2 * pr, need = l + 1;  // the later link of the pair
                for (;;) {  // every lane reads the same words: broadcast
                    int done = 99;
                    for (int w = 0; w < chain_waves; ++w) {
                        const int d = __hip_atomic_load(progress + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        done = d < done ? d : done;
                    }
                    if (done > need) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                uint32_t m = 0u;
                if ((lane & 31) < CH) m = cull_row(l + (lane >> 5), lane & 31);
                if constexpr (TILEBITS) {  // (cull_beside_chain implies CH <= 32: tb_on)
                    const unsigned long long bal = __ballot(m != 0u);
                    unsigned long long x = (unsigned long long)((uint32_t)bal | (uint32_t)(bal >> 32)) << blk_shift;  // per configuration, either link; blocks of 4 from bit 0
                    x |= x >> 1; x |= x >> 2; x &= 0x111111111ull;  // bit 4 b: block b has a row in reach
                    const uint32_t w = (uint32_t)x << (pr & 3);
                    if (lane == 0 && w) __hip_atomic_fetch_or(tilebits + (pr >> 2), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        GS_WAVE_STAMP(wave);
    }
    __syncthreads();
    GS_WG_STAMP(2);
    const double* base = lds_pose + 9;

    static_assert(LB == 2 || !LAT, "the latency-mode culling maps 8 lanes to a tile of 4 waypoints x 2 links");
    if (PRE) {
        // (the masks came with the poses)
    } else if (LAT) {
        // A pass over a link's rows costs the latency of its object loop whatever the number of lanes in it (~1 us for a lone wave:
        // beside the chain, 5 links per culling wave were the longest thing in the prologue).  Here every wave culls exactly the rows
        // its own tiles will read — lane -> (tile slot, link of the pair, waypoint of the block), normally one pass — and goes on
        // to its main loop without a barrier (its own LDS writes, in order).
        for (int q0 = 0; lat_tile(q0) >= 0; q0 += 8) {  // the pass's first tile slot: wave-uniform
            const int t = lat_tile(q0 + (lane >> 3));
            const int rb = t / (10 / LB), l = (t - rb * (10 / LB)) * LB + ((lane >> 2) & 1), ci = rb * 4 + (lane & 3) - blk_shift;
            if (t >= 0 && ci >= 0 && ci < CH) cull_row(l, ci);
        }
    } else if (!cull_beside_chain) {
        for (int row = tid; row < 10 * CH; row += NT) {
            const int l = row / CH, ci = row - l * CH;
            const uint32_t m = cull_row(l, ci);
            if (tb_on && m != 0u) {  // (chain on more than two waves, or a window beyond 32 waypoints)
                const int rb = (ci + blk_shift) >> 2, pr = l >> 1;
                if (!tb_wide) __hip_atomic_fetch_or(tilebits + (l >> 3), 1u << (4 * rb + (pr & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_or(tilebits + (pr < 4 ? (rb >> 3) : 3), 1u << (pr < 4 ? 4 * (rb & 7) + pr : rb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
    }
    GS_WG_STAMP(3);
#ifdef OMGX_GS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif

    // The main loop's wave-uniform inputs as copies of their own (an empty asm the compiler cannot see through): the kernel's
    // arguments arrive as 4- and 8-dword tuples of neighbouring fields, and a tuple that stays live for ONE field is spilled and
    // restored whole — lane moves on the VALU, per tile, per far test, per ring-fill.
    int h_CH, h_P, h_ps, h_ob, h_oe, h_tbl, h_soft;
    float h_idt;
    const omgx_object* h_objects;
    const float* h_pool;
    asm volatile("s_mov_b32 %0, %10\n\ts_mov_b32 %1, %11\n\ts_mov_b32 %2, %12\n\ts_mov_b32 %3, %13\n\ts_mov_b32 %4, %14\n\ts_mov_b32 %5, %15\n\t"
                 "s_mov_b32 %6, %16\n\ts_mov_b32 %7, %17\n\ts_mov_b64 %8, %18\n\ts_mov_b64 %9, %19"
                 : "=&s"(h_CH), "=&s"(h_P), "=&s"(h_ps), "=&s"(h_ob), "=&s"(h_oe), "=&s"(h_tbl), "=&s"(h_soft), "=&s"(h_idt), "=&s"(h_objects), "=&s"(h_pool)
                 : "s"(CH), "s"(P), "s"(pstride), "s"(o_begin), "s"(o_end), "s"(a.tbl_n), "s"(a.soften), "s"(a.inv_dt), "s"(a.objects), "s"(a.pool));
    // ---- the wave's queue.  Pending entries (object-space offset, weight) sit in the wave's LDS ring `stage`, slot i = entry i,
    // their object index | soft << 16 in q_meta of lane i; lanes / slots [0, pending).  f_*: the batch whose gathers are in flight.
    uint32_t q_meta = 0u;
    int pending = 0;
    F2 f_r00{0.0f, 0.0f}, f_r01{0.0f, 0.0f}, f_r10{0.0f, 0.0f}, f_r11{0.0f, 0.0f};
    float f_fx = 0.0f, f_fy = 0.0f, f_fz = 0.0f, f_w = 0.0f;
    uint32_t f_meta = 0u;  // object index | soft << 16 | in_c << 30 | valid << 31
    bool inflight = false;
    // A goal's cost is the sum of pot * weight (a float32 product, as before) over its pairs — accumulated EXACTLY: every term is
    // rounded to a multiple of 2^-36 ((x + C) - C with C = 1.5 * 2^16: the grid is 2^-36 for |x| < 2^15 and a coarser power of two
    // — still multiples of 2^-36 — beyond) and added in float64, where sums of such multiples are exact while their magnitude stays
    // below 2^17 (53 - 36 bits).  In that range the sum does not depend on the order of its terms: whatever the queue order, the
    // tile assignment, the number of workgroups a goal is split over or the dispatch schedule, the workgroup's sum is the same
    // number, rounded to float32 once at the end (the oracle's bar: 1e-5 relative).  RANGE: a goal cost of 131072 or more (table-top
    // scenes: < 10^3; it takes potentials of metres times speeds of km/s) leaves it — the float64 additions then round at 2^-53
    // relative, the sum depends on the order again at that level, and the float32 result can differ in its last bit between two
    // launch layouts about once in 10^9 goals; nothing else changes (tests/test_gpu_prepass.py::test_huge_goal_costs_...).
    // Terms below 2^-37 are dropped: 7e-12 absolute against costs whose float32 resolution is >= 6e-8 relative.
    double tsum = 0.0;
    float tcol = 0.0f;
    auto add_term = [&](float term) { const double C = 98304.0; /* 1.5 * 2^16: (x + C) - C rounds x to a multiple of 2^-36 */ tsum += ((double)term + C) - C; };

    // CONSUME: finish the batch in flight (interpolation, hinge, weighted sums).  f_w is 0 for lanes without an entry.
    auto consume = [&]() {
#ifdef OMGX_GS_NO_EXACT  // measurement build (tools/collect_profiles.sh): the exact path compiled out, what remains is the triage
        inflight = false;
        return;
#endif
        const float tv = trilerp(f_r00.a, f_r00.b, f_r01.a, f_r01.b, f_r10.a, f_r10.b, f_r11.a, f_r11.b, f_fx, f_fy, f_fz);
        const uint4 h = *reinterpret_cast<const uint4*>(tbl + (f_meta & 0xffffu) * 16 + 12);
        const float eps = __uint_as_float(h.x), clr = __uint_as_float(h.y), pad = __uint_as_float(h.z), i2eps = __uint_as_float(h.w);
        const float heps = __uint_as_float(tbl[(f_meta & 0xffffu) * 16 + 11]);
        const bool in_c = (f_meta & (1u << 30)) != 0, counts = (f_meta & (1u << 31)) != 0, soft = (f_meta & (1u << 16)) != 0;
        const float value = in_c ? tv : 1.0f;                                   // .cu:49-50
        const float p_in = (float)(-(double)value + (double)heps);              // .cu:158-160
        const float d = value - eps;
        const float p_band = i2eps * d * d * pad;                               // .cu:165-167
        float pot = value <= 0.0f ? p_in : (value <= eps ? p_band : 0.0f);
        pot = soft ? pot * 0.1f : pot;                                          // cost.py:350-353
        add_term(pot != 0.0f ? pot * f_w : 0.0f);                               // cost.py:260-275
        tcol += (counts && value < clr) ? 1.0f : 0.0f;                          // .cu:150-151
        GS_COUNT_N(11, __popcll(__ballot(f_w != 0.0f ? (pot != 0.0f || value < clr) : false)));  // entries that contribute anything
        inflight = false;
    };

    // ISSUE: grid coordinates, addresses and the four gathers of the `count` queued entries.  Lanes >= count compute on
    // whatever their slot holds; their weight is 0, in_c false (address = the grid's first voxel) and they count nothing.
    auto issue = [&](int count) {
        GS_COUNT(8);
#ifdef OMGX_GS_NO_EXACT
        tsum += (double)count * 1.0e-30;  // keeps the queue bookkeeping alive
        inflight = true;
        return;
#endif
        const bool valid = lane < count;
#ifdef OMGX_GS_SORT  // experiment (verdict item 3, DESIGN_HISTORY.md appendix A): the ring's entries grouped by object before the gathers
        int src = lane;
        uint32_t q_meta_s = q_meta;
        {
            const uint32_t key = q_meta & 0xffffu;
            int pos = lane, base = 0;
            unsigned long long todo = __ballot(valid);
            while (todo) {  // one round per object present in the ring (wave-uniform)
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(todo));
                const unsigned long long m = __ballot(valid && key == k);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (valid && key == k) pos = base + rank;
                base += __popcll(m);
                todo &= ~m;
            }
            src = __builtin_amdgcn_ds_permute(pos * 4, lane);             // lane pos receives the entry of this lane
            q_meta_s = (uint32_t)__builtin_amdgcn_ds_permute(pos * 4, (int)q_meta);
        }
#define q_meta q_meta_s
        const float4 qe = *reinterpret_cast<const float4*>(stage + 4 * src);
#else
        const float4 qe = *reinterpret_cast<const float4*>(stage + 4 * lane);
#endif
        const float q_tx = qe.x, q_ty = qe.y, q_tz = qe.z, q_w = qe.w;
        const uint32_t oo = valid ? (q_meta & 0xffffu) : 0u;
        const uint32_t* rec = tbl + oo * 16;
        const uint4 c0 = *reinterpret_cast<const uint4*>(rec), c1 = *reinterpret_cast<const uint4*>(rec + 4), c2 = *reinterpret_cast<const uint4*>(rec + 8);
        const double rw0 = __hiloint2double((int)c0.y, (int)c0.x), rw1 = __hiloint2double((int)c0.w, (int)c0.z);
        const double rw2 = __hiloint2double((int)c1.y, (int)c1.x);
        const int dx = (int)c1.z, dy = (int)c1.w, dz = (int)c2.x;
        const uint64_t goffb = ((uint64_t)c2.z << 32) | (uint64_t)c2.y;
        const float gx = (float)((double)q_tx * rw0) * (float)dx;  // .cu:137-142 (see pair_exact for the quotient)
        const float gy = (float)((double)q_ty * rw1) * (float)dy;
        const float gz = (float)((double)q_tz * rw2) * (float)dz;
        // axis_of without its range flags: a coordinate beyond +-1e9 saturates the conversion and fails the unsigned
        // comparison below like the oracle's explicit test; NaN converts to 0 and is rejected by the ordered comparisons
        const float sx_ = gx - 0.5f, sy_ = gy - 0.5f, sz_ = gz - 0.5f;
        int ix = (int)sx_, iy = (int)sy_, iz = (int)sz_;
        float fx = sx_ - (float)ix, fy = sy_ - (float)iy, fz = sz_ - (float)iz;
        const bool ex = (sx_ == -1.0f) && (gx > -0.5f), ey = (sy_ == -1.0f) && (gy > -0.5f), ez = (sz_ == -1.0f) && (gz > -0.5f);
        ix = ex ? 0 : ix; iy = ey ? 0 : iy; iz = ez ? 0 : iz;
        fx = ex ? -1.0f : fx; fy = ey ? -1.0f : fy; fz = ez ? -1.0f : fz;
        // (axis_of's g > -1e9 && g <= 1e9; at g == -1e9 exactly the index is negative anyway)
        const bool ordered = (__builtin_fabsf(gx) <= 1.0e9f) && (__builtin_fabsf(gy) <= 1.0e9f) && (__builtin_fabsf(gz) <= 1.0e9f);
        const bool in_c = valid && ordered && (uint32_t)ix < (uint32_t)(dx - 1) && (uint32_t)iy < (uint32_t)(dy - 1) && (uint32_t)iz < (uint32_t)(dz - 1);
#ifdef OMGX_GS_NO_GATHER  // measurement build: every lane reads the grid's first voxels — the exact path's arithmetic without its cache misses
        const uint32_t b = in_c ? (uint32_t)((ix * dy + iy) * dz + iz) & 1u : 0u;
#else
        const uint32_t b = in_c ? (uint32_t)((ix * dy + iy) * dz + iz) : 0u;
#endif
        // GLOBAL address space, said out loud: h_pool went through an asm copy, behind which the compiler no longer knows where it
        // points and falls back to flat_load, which counts on lgkmcnt as well as vmcnt (measured: no difference for this kernel —
        // the waits that follow an issue are about as long as the gathers — but there is no reason to keep the coupling)
        const GlobalBytes g0 = (GlobalBytes)(uintptr_t)h_pool + goffb + (uint64_t)b * 4u;
        const uint32_t syb = (uint32_t)dz * 4u, sxb = (uint32_t)(dy * dz) * 4u;
        f_r00 = global_f2(g0);
        f_r01 = global_f2(g0 + syb);
        f_r10 = global_f2(g0 + sxb);
        f_r11 = global_f2(g0 + sxb + syb);
        f_fx = fx; f_fy = fy; f_fz = fz;
        f_w = valid ? q_w : 0.0f;
        f_meta = (q_meta & 0x1ffffu) | (in_c ? 1u << 30 : 0u) | ((valid && !(q_meta & 0x10000u)) ? 1u << 31 : 0u);
#ifdef OMGX_GS_SORT
#undef q_meta
#endif
        inflight = true;
    };

    // ENQUEUE the lanes with `live` (object oo_soft = index | soft << 16): each writes its entry to the ring slot behind the pending
    // ones, a write and nothing else — a wave's LDS operations execute in program order, so the ISSUE that reads the slots
    // needs no round trip through registers.  A full ring is issued (after the batch in flight has been consumed) and the
    // entries that did not fit start the next one.
    auto enqueue = [&](bool live, unsigned long long bal, float tx, float ty, float tz, float w, uint32_t oo_soft) {  // bal = wave_ballot(live)
        const int n = __popcll(bal);
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        const int room = 64 - pending;
        const int take = n < room ? n : room;
        if (live && rank < room) *reinterpret_cast<float4*>(stage + 4 * (pending + rank)) = make_float4(tx, ty, tz, w);
        q_meta = (lane >= pending && lane < pending + take) ? oo_soft : q_meta;
        pending += take;
        if (pending == 64) {
            if (inflight) consume();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            issue(64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the slots have been read: the rest of this call's entries start the next ring
            if (live && rank >= room) *reinterpret_cast<float4*>(stage + 4 * (rank - room)) = make_float4(tx, ty, tz, w);
            pending = n - take;
            q_meta = lane < pending ? oo_soft : q_meta;
        }
    };

    // The waves' work: TILES of 4 consecutive waypoints x LB links.  The pairs that survive the culling are concentrated at the goal
    // end of the path and on the links near the hand (tests/fuzz/pair_density.py: the last waypoints carry ~20x the work of the
    // first).  Measured and rejected as tile shapes: single waypoints (the rows of a tile then see different objects: +2 %), one
    // link per tile (+5 %).
    const int ntiles = ((h_CH + 3) >> 2) * (10 / LB);
    const int pc3 = 3 * (p < h_P ? p : 0);  // lane part of a collision-point address (doubles)
    GS_COUNT(0);
    // WHO TAKES WHICH TILE (round 5; not in latency mode, where the wave's q-th tile is lat_tile(q)): the waves draw from ONE list — the goal's non-empty tiles, goal
    // end and hand links first — through a counter in LDS.  Dealt statically ((block, pair) -> wave (5 block + pair) % 4) the first
    // wave left the main loop 10.1 us after the prologue and the last 14.7 us (tools/gs_phase_clock.py): a third of the loop's span
    // was three waves waiting for the fourth, with the workgroup's LDS and registers held.  The exact path's work per tile cannot be
    // predicted from the masks (measured: a split by mask hits was 7 % worse than round-robin), so the waves balance themselves.
    // Nothing depends on who computes what: the goal's sum is exact (tsum), the counts are integers.
    // The list: lane i of EVERY wave holds the i-th tile of the order (each wave builds its own copy in its ring, which the queue
    // does not use before the first enqueue; a wave's LDS operations run in program order), a draw is one LDS atomic.
    constexpr bool DYN = !LAT;  // (latency mode: every wave has culled the rows of ITS tiles only)
    // the tiles this workgroup holds, heaviest first: all of the goal's, or — SPLIT — the tiles part, part + NP, ... of it
    constexpr bool DEALT = SPLIT && !RANGE;  // the goal's tiles dealt over the parts (a RANGE part holds all the tiles of its own window)
    const int ncand = DEALT ? (ntiles > part ? (ntiles - part + NP - 1) / NP : 0) : ntiles;
    auto cand = [&](int j) { return DEALT ? part + NP * (ncand - 1 - j) : ntiles - 1 - j; };
    int my_tile = -1, my_tile2 = -1, n_list = 0;  // the list: entry i in lane i of my_tile, entry 64 + i in lane i of my_tile2
    uint32_t* const tile_counter = tilebits + 2;
    constexpr int LIST_MAX = TILEBITS ? 128 : 64;  // (80 tiles at 64 waypoints)
    if constexpr (DYN) {
        if (ncand <= LIST_MAX) {
            int* const lst = reinterpret_cast<int*>(stage);
            int base = 0;
            for (int j0 = 0; j0 < ncand; j0 += 64) {  // (one round up to 64 candidates)
                const int j = j0 + lane;
                const int t = j < ncand ? cand(j) : -1;
                bool ne = t >= 0;
                if constexpr (TILEBITS) {
                    const int rb = (t < 0 ? 0 : t) / (10 / LB), pr = (t < 0 ? 0 : t) - rb * (10 / LB);
                    ne = ne && tile_bit_set(rb, pr);
                }
                const unsigned long long bal = wave_ballot(ne);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (ne) lst[base + rank] = t;
                base += __popcll(bal);
            }
            n_list = base;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            my_tile = lane < n_list ? lst[lane] : -1;
            if constexpr (TILEBITS) my_tile2 = 64 + lane < n_list ? lst[64 + lane] : -1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the list has been read: the ring is the queue's from here on
        } else n_list = ncand;
    }
    int q_static = 0;
    auto next_tile = [&]() -> int {
        if constexpr (DYN) {
            int q = 0;
            if (lane == 0) q = (int)__hip_atomic_fetch_add(tile_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q >= n_list) return -1;
            // (a draw requested one tile ahead, to hide the atomic's trip behind the previous tile, was measured: every wave then holds
            // two of the dozen tiles from the start and the balance is the static deal's again — first / last wave out 10.8 / 15.1 us)
            if (ncand > LIST_MAX) return cand(q);
            if constexpr (TILEBITS) { if (q >= 64) return __builtin_amdgcn_readlane(my_tile2, q - 64); }
            return __builtin_amdgcn_readlane(my_tile, q);
        } else return lat_tile(q_static++);
    };
#pragma unroll 1
    for (int t = next_tile(); t >= 0; t = next_tile()) {  // every lane stays active: invalid items are flagged, not skipped
        const int rb = t / (10 / LB), l0 = (t - rb * (10 / LB)) * LB;
        GS_COUNT(1);
        {
            const int ci = rb * 4 + (lane >> 4) - blk_shift;
            const bool valid = (p < h_P) && (ci >= 0) && (ci < h_CH);
            const int cic = valid ? ci : 0;
            const int cic9 = cic * 9;  // lane part of a pose address (doubles); base = configuration 1, lds_pose = configuration 0
            float px[LB], py[LB], pz[LB], w[LB];
            uint32_t msk[LB], sm[LB];
            bool wdone[LB];
            uint32_t s_any = 0;
            // sm[k]: the OR of the four rows' masks of link k as a SCALAR (the rows sit in lanes 0, 16, 32, 48; lanes without a row hold
            // 0), so that "does any row of this tile / of this link reach object o" is a scalar bit test.  As votes (`__any((msk & bit)
            // != 0)`) these tests cost a v_cmp, a trip of its result to the scalar unit and a branch on it — 37 cycles of the wave's
            // time each (tools/issue_probe.hip), three per (tile, object) iteration — and visited every object of the scene.
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                msk[k] = valid ? (rowmask + (l0 + k) * h_CH)[cic] : 0u;  // uniform part of every address on the scalar unit
                sm[k] = (uint32_t)__builtin_amdgcn_readlane((int)msk[k], 0) | (uint32_t)__builtin_amdgcn_readlane((int)msk[k], 16) |
                        (uint32_t)__builtin_amdgcn_readlane((int)msk[k], 32) | (uint32_t)__builtin_amdgcn_readlane((int)msk[k], 48);
                s_any |= sm[k];
                wdone[k] = false;
                w[k] = 0.0f;
            }
            if (s_any == 0) continue;  // nothing in reach of any row of this tile
            GS_COUNT(2);
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                px[k] = py[k] = pz[k] = 0.0f;
                if (sm[k] != 0)  // a link none of whose four rows reaches anything needs no points (its far tests are skipped too)
                    pose9_apply((base + (l0 + k) * h_ps * 9) + cic9, (pts + 3 * (l0 + k) * h_P) + pc3, px[k], py[k], pz[k]);
            }
            for (int o = h_ob; o < h_oe; ++o) {
                const int oo = o - h_ob;
                const int sh = oo < 31 ? oo : 31;
                const uint32_t bit = 1u << sh;
                GS_COUNT(10);
                if (!((s_any >> sh) & 1u)) continue;
                ObjTablePtr ob = as_const(h_objects) + o;
                const GqFar fp = gq_load_far(ob);
                if (fp.disabled > 0) continue;  // .cu:115-116
                GS_COUNT(3);
                const bool queued = oo < h_tbl;  // objects beyond the LDS records (rare) are evaluated on the spot
#ifdef OMGX_GS_COUNT  // what tiles of 2 waypoints x 2 links would do here (verdict round 3, item 2 i): far tests and enqueue calls per half wave
                if (LB == 2) {
                    bool lv = false;
#pragma unroll
                    for (int k = 0; k < LB; ++k) {
                        const float ux = __builtin_fmaf(fp.T[2], pz[k], __builtin_fmaf(fp.T[1], py[k], __builtin_fmaf(fp.T[0], px[k], fp.T[3])));
                        const float uy = __builtin_fmaf(fp.T[6], pz[k], __builtin_fmaf(fp.T[5], py[k], __builtin_fmaf(fp.T[4], px[k], fp.T[7])));
                        const float uz = __builtin_fmaf(fp.T[10], pz[k], __builtin_fmaf(fp.T[9], py[k], __builtin_fmaf(fp.T[8], px[k], fp.T[11])));
                        const bool inside = rbox_inside(ux - fp.lo[0], uy - fp.lo[1], uz - fp.lo[2], fp.rc, fp.rh, fp.rr2);
                        lv = lv || ((msk[k] & bit) && (inside || !fp.cullable));
                    }
                    const unsigned long long hb = __ballot(((msk[0] | msk[1]) & bit) != 0), lb_ = __ballot(lv);
                    GS_COUNT_N(12, ((hb & 0xffffffffull) ? 1 : 0) + ((hb >> 32) ? 1 : 0));
                    GS_COUNT_N(13, ((lb_ & 0xffffffffull) ? 1 : 0) + ((lb_ >> 32) ? 1 : 0));
                }
#endif
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    if (!((sm[k] >> sh) & 1u)) continue;  // none of this link's four rows reaches the object
                    // SE3(pose) * point (.cu:125-133) and the far test of pair_prepare
                    GS_MARGINAL_COST_PROBE(px[k], py[k], pz[k], w[k], oo, k);  // (measurement builds only)
                    float ux, uy, uz;
                    se3_apply_s(fp.T, px[k], py[k], pz[k], ux, uy, uz);
                    const float tx = ux - fp.lo[0], ty = uy - fp.lo[1], tz = uz - fp.lo[2];
                    const bool reach = (msk[k] & bit) != 0, inside = rbox_inside(tx, ty, tz, fp.rc, fp.rh, fp.rr2);
                    const bool live = reach && (inside || !fp.cullable);
                    // the same as a lane mask, from the two compares' own masks (a ballot of the combined bool goes through a select and a
                    // second compare per lane)
                    const unsigned long long live_mask = wave_ballot(reach) & (wave_ballot(inside) | (fp.cullable ? 0ull : ~0ull));
                    GS_COUNT(4);
                    if (live_mask == 0) continue;
                    GS_COUNT(5);
                    GS_COUNT_N(9, __popcll(live_mask));
                    if (!wdone[k]) {
                        GS_COUNT(6);  // ||(x_i - x_{i-1}) / dt|| in float32 (config.py:162-187, cost.py:260-275), once per (row, link)
                        const int l = l0 + k;
                        float qx, qy, qz;
                        // the previous configuration's pose: cfg ci - 1 of the same link, i.e. one 72-byte record back (ci = 0: the start, record 0)
                        pose9_apply((lds_pose + l * h_ps * 9) + cic9, (pts + 3 * l * h_P) + pc3, qx, qy, qz);
                        const float vx = (px[k] - qx) * h_idt, vy = (py[k] - qy) * h_idt, vz = (pz[k] - qz) * h_idt;
                        w[k] = sqrtf(vx * vx + vy * vy + vz * vz);
                        wdone[k] = true;
                    }
                    const uint32_t soft = (h_soft && l0 + k >= 8) ? 1u : 0u;
                    if (queued) {
                        GS_COUNT(7);
                        enqueue(live, live_mask, tx, ty, tz, w[k], (uint32_t)oo | (soft << 16));
                    } else if (live) {
                        ObjParams op = load_object(ob);
                        // (an object beyond the LDS records is rare: keep what only this branch needs — 0.5 * (double)eps of the hinge —
                        // from being hoisted into every (tile, object) iteration: two half-rate instructions each)
                        asm volatile("" : "+s"(op.eps), "+s"(op.clr));
                        Accum one{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        pair_exact<false>(op, h_pool + ob->grid_offset, tx, ty, tz, one);
                        if (soft) { one.pot *= 0.1f; one.col = 0.0f; }
                        add_term(one.pot != 0.0f ? one.pot * w[k] : 0.0f);
                        tcol += one.col;
                    }
                }
            }
        }
    }
    if (inflight) consume();
    if (pending > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        issue(pending);
        consume();
    }
#if defined(OMGX_GS_CLOCK)
    if (lane == 0 && blockIdx.x < (1u << 16)) {  // when the first / the last wave of the workgroup left its main loop
        const unsigned long long now = wall_clock64();
        atomicMin(&g_gs_wg[GS_WG_IDX][5], now);
        atomicMax(&g_gs_wg[GS_WG_IDX][6], now);
    }
#endif
    {
        const double ws_ = wave_sum(tsum);  // exact (see tsum)
        const float wc_ = wave_sum(tcol);
        double* red = reinterpret_cast<double*>(lds_bytes + L.tbl_off);  // [W] sums, then as many float counts: the records are dead once every wave has flushed its queue
        float* redc = reinterpret_cast<float*>(red + W);
        __syncthreads();
        if (lane == 0) { red[wave] = ws_; redc[wave] = wc_; }
        __syncthreads();
        if (tid == 0) {
            const int64_t k = (int64_t)s * a.NCH + chunk;
            double rs = ((red[0] + red[1]) + red[2]) + red[3];   // exact: the order does not matter (tsum)
            float rc = ((redc[0] + redc[1]) + redc[2]) + redc[3];  // integers below 2^24: exact too
#pragma unroll
            for (int w = 4; w < W; ++w) { rs += red[w]; rc += redc[w]; }
            if (a.chunk_cost) gq_st_out<PERSIST>(a.chunk_cost + k, (float)rs);
            if (a.chunk_col) gq_st_out<PERSIST>(a.chunk_col + k, rc);
            if (STAMP) { const unsigned long long dt = wall_clock64() - work_t0; a.work[k] = dt < 1 ? 1u : (dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt); }
        }
    }
    GS_FREQ_END();
    GS_WG_STAMP(4);
}

// Workgroups per CU the WIDE instantiations are compiled for: six waves x 3 = 18 waves (96 VGPRs like the 4 x 5 of the plain kernel), eight
// waves x 2 = 16 (128 VGPRs).
template <int W> struct GqWide { static constexpr int wg_per_cu = W == GQ_WAVES ? GQ_WG_PER_CU : (W <= 6 ? 3 : 2); };
// What a workgroup of the launch does: find its item (scene, trajectory-layer piece | goal [, part]) and run it.  Two kernels share it:
// k_goalset_queue and k_goalset_range (RANGE).
template <int LB, bool STAMP, bool LAT, bool SPLIT, bool PRE, int W, bool RANGE>
__device__ __forceinline__ void goalset_queue_body(const ChunkArgs& a, double* const lds_pose) {
    GS_WG_STAMP(0);
#ifdef OMGX_GS_PRIO
    __builtin_amdgcn_s_setprio(OMGX_GS_PRIO);
#endif
    const int xcd = blockIdx.x & 7;
    // with a trajectory layer, a.layer_parts workgroups per scene compute it (a.layer_lg link groups x a.layer_nb blocks of a.layer_cb
    // configurations: 5 x 1 in a batch, finer in latency mode); those lead the grid
    // (at the end of the grid, as fillers of the launch's tail, they cost 3 %: measured)
    const int LPARTS = a.layer_parts;
    const bool spread = LAT;
    static_assert(!(LAT && SPLIT), "latency mode has its parts already");
    const int NP = (LAT || SPLIT) ? a.NP : 1;
    bool is_layer;
    int s, chunk, layer_part;
    bool scheduled = false;
    if (spread) {
        // latency mode (a handful of scenes): workgroup b -> (scene, item) in plain order, the scene's workgroups spread over all XCDs
        const int nl = a.wp_traj ? a.S * LPARTS : 0;
        is_layer = (int)blockIdx.x < nl;
        const int j = (int)blockIdx.x - nl;
        s = is_layer ? (int)blockIdx.x / LPARTS : j / a.NCH;
        layer_part = (int)blockIdx.x - s * LPARTS;
        chunk = is_layer ? 0 : j - s * a.NCH;
    } else {
        const int nlayer = a.wp_traj ? ((a.S + 7) >> 3) * LPARTS : 0;
        is_layer = (int)(blockIdx.x >> 3) < nlayer;
        const int j = (int)(blockIdx.x >> 3) - nlayer;
        const int sgrp = is_layer ? (int)(blockIdx.x >> 3) / LPARTS : j / a.NCH;
        layer_part = (int)(blockIdx.x >> 3) - sgrp * LPARTS;
        s = sgrp * 8 + xcd;
        chunk = is_layer ? 0 : j - sgrp * a.NCH;
        scheduled = a.schedule && !is_layer;
        if (scheduled) {  // goal workgroup b of the launch works on item schedule[b] (ChunkArgs)
            const int item = as_const(a.schedule)[(int)blockIdx.x - nlayer * 8];
            if (item < 0 || item >= a.S * a.NCH) return;
            s = item / a.NCH;
            chunk = item - s * a.NCH;
            if (a.active && as_const(a.active)[s] == 0) { if (STAMP && threadIdx.x == 0) a.work[item] = 0u; return; }
        }
    }
    if (s >= a.S) return;
    if (a.active && !scheduled && !spread) {
        // Scenes the planner has left (planner.py:626) get no workgroups, and the remaining ones are dealt out again so that
        // every XCD keeps an equal share: slot k = sgrp * 8 + xcd works on the k-th ACTIVE scene (ascending).  Every wave
        // finds it by itself with ballots over the mask (S / 64 steps, wave-uniform): no barrier, no extra launch, same
        // result in all waves.  With all scenes active this is the identity.
        const int k = s, ln = threadIdx.x & 63;
        int seen = 0;
        s = -1;
        for (int base = 0; base < a.S; base += 64) {
            const int i = base + ln;
            const unsigned long long bal = __ballot(i < a.S && a.active[i] != 0);
            const int cnt = __popcll(bal);
            if (k < seen + cnt) {
                unsigned long long m = bal;
                for (int q = k - seen; q > 0; --q) m &= m - 1;  // drop the k - seen lowest set bits
                s = base + __builtin_ctzll(m);
                break;
            }
            seen += cnt;
        }
    }
    if (s < 0) return;  // fewer active scenes than slots
    if (spread && a.active && as_const(a.active)[s] == 0) return;
    if constexpr (W != GQ_WAVES) {
        // a trajectory-layer piece is written for four waves: the others leave (a finished wave no longer counts at the workgroup's barriers)
        if (is_layer && threadIdx.x >= 64 * GQ_WAVES) return;
    }
    gq_item<LB, STAMP, LAT, SPLIT, PRE, false, 0, W, RANGE>(a, lds_pose, s, is_layer, layer_part, chunk, NP);
}

template <int LB, bool STAMP = false, bool LAT = false, bool SPLIT = false, bool PRE = false, int W = GQ_WAVES>
__global__ __launch_bounds__(LAT ? 256 : 64 * W, LAT ? 2 : GqWide<W>::wg_per_cu) void k_goalset_queue(ChunkArgs a) {  // LAT: a workgroup per CU or two, registers are free
    extern __shared__ __attribute__((aligned(16))) double lds_pose[];  // no static LDS: 31744 B is the most a workgroup may use at 5 per CU
    goalset_queue_body<LB, STAMP, LAT, SPLIT, PRE, W, false>(a, lds_pose);
}

// The batch kernel with a goal's window cut into two waypoint RANGES (gq_item: RANGE), a kernel name of its own.
template <int LB, bool STAMP = false>
__global__ __launch_bounds__(GQ_NT, GQ_WG_PER_CU) void k_goalset_range(ChunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds_pose[];
    goalset_queue_body<LB, STAMP, false, true, false, GQ_WAVES, true>(a, lds_pose);
}
