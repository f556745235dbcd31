"""ChompEngine — device-resident batched CHOMP planning over S independent scenes.

What the reference does per scene on the host (``Planner.plan``, omg/planner.py:600-653)::

    for t in range(optim_steps + extra_smooth_steps):
        if goal_set_proj and t < optim_steps: learner.update_goal()      # cost_vector -> batch_obstacle_cost
        info = optim.optimize(traj, force_update=True)                   # compute_total_loss + update

is run here for all scenes of this rank at once, with every tensor resident in HBM and no host round
trip inside an iteration: per iteration three C-ABI calls on one stream

    omgx_goalset_cost   (S x G goal candidates, arc-length weighted obstacle cost)   [t < optim_steps]
    omgx_fk_sdf         (S x n waypoint configurations -> potentials / gradients)
    omgx_chomp_optimize (S trajectories: loss, gradient, projected A^-1 step, joint limits)

plus one omgx_goal_update launch (Learner.update_goal, omg/online_learner.py:104-249: cost-vector tail, FTL / FTC /
Exp / MD mirror descent / Proj, argmax, goal gather) — SURVEY.md §8f-1.

Multi-GPU: scenes are independent (omg/core.py:869-885 loops them), so rank r of R owns a contiguous block
of the scene list; nothing is exchanged during planning and one all-gather of the final per-scene costs
closes the job (``gather_costs``).
"""
from __future__ import annotations

import operator
import os

import numpy as np
import torch

from . import _lib, ops
from .config import Config
from .robot import PandaModel
from .scenes import SceneBatch


def shard_range(num_items: int, rank: int, world: int) -> range:
    """Contiguous block partition of independent units over ranks (sizes differ by at most one)."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


_SIDE_STREAMS: dict = {}


def _side_stream(device: torch.device, i: int) -> "torch.cuda.Stream":
    """Side stream i of the device, shared by all engines: the first use of a new HIP stream costs milliseconds (its
    hardware queue is created), which a per-engine stream would pay in every plan."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), i)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)  # same priority as the caller's: a high-priority side stream costs 3 %
    return st


_ALL_DONE_CHECKS = frozenset((1, 2, 3, 5, 8, 13, 21, 34, 55))  # iterations after which plan() looks at `active` (small batches only)


class ChompEngine:
    # learner and step of a scene in different workgroups of omgx_goal_update_optimize (None: whenever both sets are resident
    # at once, 2 S <= CUs — beyond that the single-workgroup kernel is a little faster; True / False force it)
    split_update = None
    HOT_FIXED_GOAL = True  # fixed-goal iterations through the prepared calls (_iterate_hot_fixed); False: the general path (A/B, tests)
    # True: iterate() goes through iterate_separate(), the five separate entry points (cross-checks)
    separate_launches = False
    # Under early stop: iterations between rebuilds of the schedule without the terminated scenes (omgx_goalset_schedule, one
    # ~10 us launch); 0: scene-major order once scenes drop out (the kernel then deals the remaining scenes to the XCDs itself).
    # Measured equal within 1 % for 100 and 13 scenes x 64 goals (tools/experiments/ab_plan.py: 12.5 / 6.3 ms per early-stop plan either way), so
    # the simpler policy is the default.
    reschedule_every = 0
    schedule_slack = 2  # goal workgroup slots per XCD in units of the even share (see build_schedule)
    # Software pipeline over scene sub-ranges (see _iterate_pipelined): an integer k runs every iterate() as k parts on k HIP
    # streams (1: never); None: plan() pipelines two parts from PIPELINE_MIN_ITEMS (scene, goal) items on — below, the host's
    # four launches per iteration (~95 us) take longer than the GPU needs (tools/experiments/ab_pipeline.py: 8 x 64: 90 -> 96 us,
    # 16 x 64: 109 -> 95, 13 x 128: 139 -> 112, 50 x 64: 191 -> 172, 100 x 64: 312 -> 285) — a bare iterate() does not: its
    # caller owns the synchronisation (join()).
    pipeline = None
    PIPELINE_MIN_ITEMS = 768
    # Between PIPELINE_MIN_ITEMS and this many items three parts beat two: while one part's update launch (two latency-bound
    # workgroups per scene) runs, the other TWO parts' goal-set launches keep the chip busy instead of one.  Round 5, after the
    # goal-set kernel got 8 % shorter and the update's share of a step grew (bench.py --pipeline 2 / 3, ms per step): 35 x 64
    # 0.1004 / 0.0981, 50 x 64 0.1291 / 0.1228, 100 x 64 0.1945 / 0.1898, 200 x 64 0.3987 / 0.3873, 100 x 128 0.3309 / 0.3324,
    # 400 x 64 0.8274 / 0.8412 (two parts win again: three launches of 133 scenes lose more to their tails than the idle update
    # phases cost); four parts at 100 x 64: 0.214.  (Rounds 2-4: three parts only below 2048 items.)
    PIPELINE_THREE_BELOW = 16384
    MEASURE_MIN_ITEMS = 256   # (scene, goal, part) items from which the second launch's measured durations order the dispatch schedule
    # Up to this many items the measured schedule runs an XCD's items LONGEST FIRST across its scenes (omgx_goalset_schedule_ordered):
    # a launch of a round or two of the chip's 1280 workgroup slots ends with what starts last (13 x 128 in three pipeline parts:
    # 0.0864 -> 0.080 ms per step with the measured scene-major schedule, -> 0.0785 longest first; 100 x 64 longest first: +16 %,
    # a scene's volumes leave the L2 — tools/experiments/ab_schedule_order.py).  The order changes no result: the goal sums are exact.
    LONGEST_FIRST_MAX_ITEMS = 2048

    @classmethod
    def auto_parts(cls, num_scenes: int, num_goals: int) -> int:
        items = num_scenes * num_goals
        if items < cls.PIPELINE_MIN_ITEMS:
            return 1
        return min(num_scenes, 3 if items < cls.PIPELINE_THREE_BELOW else 2)

    # ---------------------------------------------------------------------------------------------
    # The size-aware layout: ONE documented rule, a pure function of the shape, so that every rank of a job picks the same one
    # (a goal's cost is a float32 sum whose order depends on goal_parts / latency mode: shards computed under different layouts
    # would not be bit-comparable).  Measured on MI355X (tools/experiments/ab_parts_graph.py, profiles/r04a_layout_sweep.json), ms per step
    # of bench.py's step, best (goal_parts, pipeline parts) against the plain batch layout:
    #   1 x 64   latency mode 0.050          plain 0.065      |   6 x 64   (2, 3) 0.067   plain 0.073
    #   2 x 64   (4, 1) 0.059                plain 0.068      |  10 x 64   (2, 2) 0.070   plain 0.075
    #   4 x 64   (4, 2) 0.060                plain 0.068      |   6 x 128  (2, 2) 0.071   plain 0.078
    #   2 x 128  (4, 2) 0.060                plain 0.070      |  16 x 64   (2, 2) 0.078   plain 0.079
    #   5 x 64   (4, 2) 0.066                plain 0.072      |  20 x 64 and beyond: plain (split goals cost 6-30 % there)
    # A goal workgroup's life is a latency-bound prologue (kinematics of the window) + its tiles; splitting a goal shortens the
    # launch only while the chip has idle workgroup slots — every part repeats the prologue — i.e. below about 0.7 rounds of the
    # 1280 slots.  Beyond that the batch is bound by the chip's capacity and the plain layout wins.
    # Fixed-goal iterations (the plan's smoothing phase: a layer-only launch + a step launch) of a batch below this many scenes run as ONE
    # pipeline part: their kernels are short chains on an almost empty chip, and three parts cost the host three times the launches — with the
    # layer in 40 pieces per scene (ops.IterationCalls._layer_only_tiling) a plan of 13 x 128 goes 4.72 -> 4.26 ms, 16 x 64 4.84 -> 4.24, 25 x 64
    # 4.97 -> 4.64 (tools/experiments/ab_smooth_small.sh).  Larger batches keep their parts (100 scenes: 8.80-8.86 as one part against 8.72-8.74).
    SMOOTH_SINGLE_PART_BELOW = int(os.environ.get("OMGX_SMOOTH_SINGLE_PART_BELOW", 32))
    WIDE_WINDOW_FROM = 57  # cfg.timesteps from which a plan's goal-set launches (their LDS follows the trajectory layer: all n waypoints) need more than 53 248 B per four-wave goal workgroup (15-16 points per link): eight waves

    @classmethod
    def layout(cls, num_scenes: int, num_goals: int, n_waypoints: int = 30, for_plan: bool = False) -> dict:
        """-> {"latency_mode", "goal_parts", "pipeline"} for a rank that plans num_scenes x num_goals goals over n_waypoints.
        The work of a goal workgroup grows with the window, so the thresholds count (scene, goal) items scaled by n / 30.
        for_plan: the engine will run WHOLE plans (shrinking windows), not the pinned first-iteration step the rule was fitted to — two to four
        scenes then plan in latency mode like one (round 6, tools/experiments/plan_once_n.py with OMGX_PLAN_LATENCY: plan of 2 / 3 / 4 x 64 x 30
        waypoints 3.63 / 3.66 / 3.71 -> 2.89 / 2.99 / 3.30 ms, 2 / 3 x 64 x 50 x 13 objects 5.20 / 5.19 -> 4.69 / 4.91; their pinned step is SLOWER in
        latency mode — 0.0455 / 0.0444 / 0.0456 -> 0.0497 / 0.0547 / 0.0664 ms — so the default stays what it was)."""
        load = num_scenes * num_goals * max(n_waypoints, 1) / 30.0
        if num_scenes == 1 or (for_plan and num_scenes <= (4 if n_waypoints <= 32 else 3)):
            return {"latency_mode": True, "goal_parts": 1, "pipeline": 1}
        if n_waypoints >= cls.WIDE_WINDOW_FROM and num_scenes * num_goals <= 4096:
            # Plans of 57-64 waypoints (round 6): the library runs their whole goals on EIGHT-wave workgroups (two per CU by LDS either way:
            # omg_kernels.hip gs_wide_waves) and those beat split goals wherever the rule below would split — bench.py --waypoints 64, ms per
            # step, rule / whole goals on (1, 2, 3) pipeline parts: 4 x 64 0.0815 / 0.0795, 0.0807, 0.0842; 8 x 64 0.106 / 0.096, 0.104, 0.110;
            # 16 x 64 0.159 / 0.123, 0.115, 0.116; 32 x 64 0.179 / 0.184, 0.173, 0.179; 16 x 64 x 60 waypoints x 13 objects 0.231 / 0.209,
            # 0.211, 0.211.  (A whole PLAN does not care — 16 x 64 x 64: 9.40 against 9.38 ms — its late windows are short either way.)
            return {"latency_mode": False, "goal_parts": 1, "pipeline": 1 if num_scenes * num_goals <= 512 else max(1, min(2, num_scenes))}
        if n_waypoints > 40 and 320 < load < 2560:
            # Long windows (round 6; BASELINE config 5 plans with 50 waypoints): a goal workgroup's poses alone are 37 KB, three
            # workgroups per CU instead of five, and its 65 tiles make a long life — two workgroups per goal pay up to twice the size the
            # 30-waypoint sweep found (bench.py --waypoints 50 --objects 12, ms per step, (goal_parts, pipeline)): 4 x 64 (2, 2) 0.091
            # against (1, 3) 0.126; 8 x 64 (2, 2) 0.125, (2, 3) 0.120 against 0.170; 16 x 64 (2, 2) 0.165 against 0.189 — (1, 1) 0.183,
            # (2, 3) 0.193; 32 x 64 and 16 x 128: whole goals again ((1, 3) 0.275 / 0.256 against (2, 2) 0.297 / 0.261)
            return {"latency_mode": False, "goal_parts": 2, "pipeline": max(1, min(2, num_scenes))}
        if for_plan and n_waypoints <= 40 and load <= 896:
            # whole plans of small batches: split goals repeat the prologue in every part, and late in a plan the prologue is most of a goal —
            # plan of 5 / 6 / 8 / 12 x 64 and 6 x 128: 3.62 / 3.50 / 3.49 / 3.74 / 3.68 ms by the rule below, 3.44 / 3.43 / 3.46 / 3.58 / 3.51 with whole goals
            # on two pipeline parts (profiles/r06i_ab_plan_layout.log; three parts: 4.0-4.7)
            return {"latency_mode": False, "goal_parts": 1, "pipeline": max(1, min(2, num_scenes))}
        if load <= 320:
            gp, pipe = 4, 2
        elif load <= 896:
            gp, pipe = 2, 2
        elif load < cls.PIPELINE_THREE_BELOW:
            gp, pipe = 1, 3
        else:
            gp, pipe = 1, 2
        return {"latency_mode": False, "goal_parts": gp, "pipeline": max(1, min(pipe, num_scenes))}

    @classmethod
    def auto(cls, model, batch, cfg, start, goal_set, layout_scenes: "int | None" = None, for_plan: bool = False, **kw) -> "ChompEngine":
        """An engine laid out by ChompEngine.layout for its shape.  `layout_scenes`: the scene count the RULE is evaluated for —
        in a multi-rank job every rank passes the same number (the largest shard, ceil(total / world)), so that shards of 13 and
        12 scenes run the same layout and stay bit-comparable with each other and with a single-process run given that number."""
        S, G = goal_set.shape[0], goal_set.shape[1]
        lay = cls.layout(S if layout_scenes is None else int(layout_scenes), G, cfg.timesteps, for_plan=for_plan)
        if lay["latency_mode"] and S > 4:  # the latency-mode kernel runs two workgroups per CU: a rule evaluated for another size must not force it on a batch
            lay = dict(lay, latency_mode=False, goal_parts=4)
        eng = cls(model, batch, cfg, start, goal_set, latency_mode=lay["latency_mode"], goal_parts=lay["goal_parts"], **kw)
        if not lay["latency_mode"] and eng.stream is None:
            eng.pipeline = max(1, min(lay["pipeline"], S))
        eng.layout_used = lay
        return eng

    # per-scene tensors: a part of the pipeline works on the rows [lo, hi) of each
    _PART_TENSORS = ("start", "goal_set", "reach", "cv_goals", "goal_idx", "goal_count", "eta_s", "traj", "end", "goal_rows", "goal_point",
                     "pot", "pgrad", "col", "grad", "cost_traj", "info", "goal_cost", "goal_col", "learner_state", "cost_vec", "_active",
                     "_scene_flags", "wp_pose", "start_pose", "end_pose", "goal_pose_tab")

    # Latency mode (ChompEngine(latency_mode=True); omgx_goalset_cost_layer_tiled): a goal's tiles dealt over up to LAT_GOAL_PARTS
    # workgroups, the trajectory layer in LAT_LAYER_LINK_GROUPS x ceil(n / LAT_LAYER_BLOCK) workgroups, all spread over the XCDs.
    LAT_GOAL_PARTS = 4
    LAT_LAYER_LINK_GROUPS = 10
    LAT_LAYER_BLOCK = 4   # one link x 4 waypoints per layer workgroup: its four waves take the objects side by side
    LAT_HAND_OVER_POSES = True  # inside plan(), both layouts: link poses handed between the launches (False: every kernel runs its own kinematics; same bits)

    PREPASS_DEFAULT = False

    def __init__(self, model: PandaModel, batch: SceneBatch, cfg: Config, start: np.ndarray, goal_set: np.ndarray,
                 reach_grasps: np.ndarray | None = None, traj_init: np.ndarray | None = None, device="cuda:0",
                 ol_alg: str = "FTL", stream: "torch.cuda.Stream | None" = None, goal_counts=None, latency_mode: bool = False,
                 goal_parts: int = 1, prepass: "bool | None" = None):
        """start [S,9]; goal_set [S,G,9]; reach_grasps [S,G,c,9] (needed when cfg.use_standoff).
        `stream`: run every launch of this engine on that HIP stream (several engines holding disjoint scene
        subsets on different streams overlap each other's latency-bound kernels).
        `goal_counts` [S]: ragged goal sets — scene s uses goal_set[s, :goal_counts[s]] (and the matching reach_grasps); the
        rest of its rows is padding that no launch reads.  Each scene then computes exactly what it would compute alone.
        `latency_mode`: for ONE or a few scenes (BASELINE configs 1-2) — the goal-set batch and the trajectory layer are cut
        into many small workgroups spread over the whole chip instead of one workgroup per goal on the scene's XCD (a third
        of the launch's latency).  A goal's cost is then the float32 sum of its parts' sums — another summation order
        than the batch layout's (~1e-7 relative), everything else is bit-identical; `goal_cost` / `goal_col` hold the partial
        sums (goal_cost_total() adds them).  No pipeline, no dispatch schedule in this mode.
        `goal_parts` (1, 2, 4, 8; batch layout only): MID-SIZE batches — a goal's tiles dealt over up to that many workgroups of
        the batch kernel (omgx_goalset_cost_layer_parts), scene per XCD, dispatch schedule and pipeline as usual; goal costs are
        partial sums as in latency mode.  ChompEngine.layout() names the rule that picks all of this from the shape.
        `prepass`: the goals' kinematics and row culling run as a launch of their own ahead of every goal-set launch
        (k_goalset_kin, one lane per (goal, configuration), through a workspace in HBM) instead of in every goal workgroup's
        prologue — the same bits from three launches per iteration instead of two."""
        self.cfg = cfg
        if prepass is None:  # the class default, or OMGX_PREPASS=0/1 (A/B runs of one command line)
            import os
            env = os.environ.get("OMGX_PREPASS", "")
            prepass = self.PREPASS_DEFAULT if env == "" else env not in ("0", "false", "no")
        self.prepass = bool(prepass)
        self.latency = bool(latency_mode)
        self.goal_parts = 1 if self.latency else int(goal_parts)
        if self.goal_parts not in (1, 2, 4, 8):
            raise ValueError("goal_parts must be 1, 2, 4 or 8")
        self.stream = stream
        self.model = model
        self.device = torch.device(device)
        self.S, self.G = goal_set.shape[0], goal_set.shape[1]
        self.n = cfg.timesteps
        self.P = model.points_per_link
        self.ol_alg = ol_alg
        if ol_alg not in ("FTL", "FTC", "Exp", "MD", "Proj", "Baseline"):
            raise ValueError(f"unknown ol_alg {ol_alg!r}")
        assert batch.num_scenes == self.S
        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.robot = ops.robot_blob(model, dev)
        self.scenes = batch if isinstance(batch, ops.DeviceScenes) else ops.DeviceScenes(batch, dev)  # a DeviceScenes: shared, and changeable between plans (set_object_pose, replace_grid)
        if torch.device(self.scenes.device) != dev and not (self.scenes.device.type == dev.type == "cuda" and (self.scenes.device.index or 0) == (dev.index or 0)):
            raise ValueError(f"the DeviceScenes live on {self.scenes.device}, the engine was asked for {dev}")
        self.start = torch.as_tensor(start, **f64).contiguous()
        self.goal_set = torch.as_tensor(goal_set, **f64).contiguous()
        self.use_standoff = bool(cfg.use_standoff)
        self.c = cfg.reach_tail_length if self.use_standoff else 1
        if self.use_standoff:
            if reach_grasps is None:
                raise ValueError("cfg.use_standoff needs reach_grasps [S,G,c,9]")
            self.reach = torch.as_tensor(reach_grasps, **f64).contiguous()
            self.cv_goals = self.reach[:, :, -1, :].contiguous()  # online_learner.py:121-125
        else:
            self.reach = None
            self.cv_goals = self.goal_set
        # (Set-up builds its tensors on the HOST and uploads them: the first use of a torch kernel — fill, strided copy,
        # index_select, where — loads another piece of torch's code objects, 20-120 ms each on a cold process, where a copy is DMA.)
        def dzeros(shape, dtype):
            return torch.from_numpy(np.zeros(shape, dtype)).to(dev)
        self.goal_idx = dzeros(self.S, np.int32)
        self.goal_count = self.eta_s = None
        if goal_counts is not None:
            gc = np.asarray(goal_counts, np.int64).reshape(-1)
            if gc.shape[0] != self.S or gc.min() < 1 or gc.max() > self.G:
                raise ValueError("goal_counts must hold one count in [1, G] per scene")
            self.goal_count = torch.as_tensor(gc, dtype=torch.int32, device=dev)
            self.eta_s = torch.as_tensor(np.sqrt(np.log(gc + 1) / cfg.optim_steps), dtype=torch.float64, device=dev)  # online_learner.py:80 per scene
            self._goal_counts_host = gc
        S, n, P, G = self.S, self.n, self.P, self.G
        if traj_init is None:
            from .scenes import cubic_init
            traj_init = np.stack([cubic_init(start[s], goal_set[s, 0], n) for s in range(S)])
        self.traj = torch.from_numpy(np.ascontiguousarray(traj_init, np.float64).copy()).to(dev)
        gs_h = np.ascontiguousarray(goal_set, np.float64)
        self.end = torch.from_numpy(gs_h[:, 0].copy()).to(dev)
        # the goal the plan starts from is goal 0 (goal_idx = 0): traj.end / chosen rows / goal point (online_learner.py:243-245, optimizer.py:93-99)
        self.goal_point = torch.from_numpy(gs_h[:, 0].copy()).to(dev)
        rows_h = np.ascontiguousarray(reach_grasps, np.float64)[:, 0] if self.use_standoff else gs_h[:, 0][:, None, :]
        self.goal_rows = torch.from_numpy(np.ascontiguousarray(rows_h).copy()).to(dev)
        # preallocated outputs: nothing is allocated inside an iteration
        f32 = dict(dtype=torch.float32, device=dev)
        self.pot = torch.empty((S, n, 10, P), **f32)
        self.pgrad = torch.empty((S, n, 10, P, 3), **f32)
        self.col = torch.empty((S, n, 10, P), **f32)
        self.grad = torch.empty((S, n, 9), **f64)
        self.cost_traj = torch.empty((S, n), **f64)
        self.info = dzeros((S, _lib.INFO_STRIDE), np.float64)
        # latency mode: [S][G][parts] partial sums, parts = ceil(window / LAT_GOAL_PARTS) shrinking with the window (flat buffer)
        self._parts_max = ops.goalset_parts(n, self.LAT_GOAL_PARTS) if self.latency else (ops.goalset_parts(n, self.goal_parts) if self.goal_parts > 1 else 1)
        self._parts_last = 1
        # latency mode, inside plan(): link poses handed between the launches instead of being recomputed (omgx_pose_table, ABI 7)
        # — the waypoints' poses from the layer workgroups to the step, the start's and the goals' poses tabulated per plan
        self._poses_on = False
        self.wp_pose = torch.empty((S, n, 10, 12), **f64)
        self.start_pose = torch.empty((S, 10, 12), **f64)
        self.end_pose = dzeros((S, 10, 12), np.float64)
        self.goal_pose_tab = torch.empty((S, G, 10, 12), **f64)
        self.goal_cost = dzeros((S, G * self._parts_max), np.float32)
        self.goal_col = dzeros((S, G * self._parts_max), np.float32)
        self.learner_state = ops.learner_state(S, G, dev, goal_counts)  # sum_costs | p | experts_p | q | experts_costs
        self.cost_vec = dzeros((S, G), np.float64)
        self.eta = float(np.sqrt(np.log(G + 1) / cfg.optim_steps))  # online_learner.py:80
        self._active = torch.from_numpy(np.ones(S, np.int32)).to(dev)
        self._masked = False  # becomes True with the first early_stop iteration or when `active` is assigned: launches then take the mask
        self.step_count = 0  # Optimizer.step
        self.t = 0           # Learner.t
        self._scene_flags = dzeros(S, np.int32)  # omgx_goal_update_optimize's rendezvous
        # Dispatch order of the goal-set launch.  The first launch of a plan runs in scene-major order and records every goal
        # workgroup's duration in `work`; build_schedule() turns that — on the device, no host sync — into the order of all
        # later launches: scenes dealt to the 8 XCDs by weight (heaviest first, serpentine), each scene's goals longest first.
        # A scene keeps all its workgroups on one XCD (its SDF volumes stay in that XCD's L2: without this affinity the
        # launch takes 1.7x as long).  Results do not depend on the order.
        self.work = dzeros(S * G * (1 if self.latency else self._parts_max), np.int32)  # one counter per (scene, goal, part) workgroup
        self.auto_schedule = True
        self.schedule = None
        self._sched_np = 1  # parts per goal the current schedule's items count
        self._gs_launches = 0
        self._measured = False
        self._sched_buf, self._sched_flip, self._sched_age = None, 0, None
        self._uniform_cache = None
        # ONE ticket source for this engine and every pipeline part made from it (the parts' flags are slices of _scene_flags):
        # a ticket must differ from every value still stored in the flags it is compared with (include/omg_hip.h), whoever
        # launched last — the whole engine, a part, parts of an earlier split.  A one-element list, shared by reference.
        self._ticket_src = [0]
        with torch.cuda.device(dev):
            self._num_cus = int(_lib.lib().omgx_device_cu_count())  # (torch.cuda.get_device_properties costs 110 ms on its first call)
        if self._num_cus <= 0:
            self._num_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        self._parts, self._forked, self._in_plan = None, False, False
        self._hot = None
        self._capturing = False
        self._scratch_state = self._cubic_h = self._cubic_tmp = self._cubic_diff = None

    @property
    def active(self) -> torch.Tensor:
        """[S] int32: 0 = the scene has left the planner loop (planner.py:626) and every launch skips it."""
        return self._active

    @active.setter
    def active(self, value: torch.Tensor):
        self.join()
        self._active = value
        self._masked = True
        self._refresh_parts()

    def _next_ticket(self) -> int:
        """The next rendezvous ticket of omgx_goal_update_optimize: monotonically increasing over the engine AND its pipeline
        parts, never 0 (the flags start at 0), wrapping below 2^24 (the scene's word is (ticket << 8) | chosen goal) long before any flag could
        still hold the value."""
        src = self._ticket_src
        src[0] = src[0] + 1 if src[0] < 0xfffff0 else 1  # the flag word is (ticket << 8) | goal index: tickets stay below 2^24
        return src[0]

    def _mask(self):
        """The mask for a launch — None while no scene can be inactive (the goal-set kernel then skips its slot look-up)."""
        return self._active if self._masked else None

    # ---------------------------------------------------------------------------------------------
    def _params(self, do_update: bool) -> _lib.ChompParams:
        cfg = self.cfg
        p = _lib.ChompParams()
        p.n_waypoints, p.n_points = self.n, self.P
        p.top_k = int(cfg.top_k_collision)
        p.consider_finger = int(cfg.consider_finger)
        p.goal_set_proj = int(cfg.goal_set_proj)
        p.constraint_num = self.c
        p.use_standoff = int(self.use_standoff)
        p.uncheck_finger_collision = int(cfg.uncheck_finger_collision)
        p.joint_limit_max_steps = int(cfg.joint_limit_max_steps)
        p.allow_collision_point = int(cfg.allow_collision_point)
        p.pre_terminate = int(cfg.pre_terminate)
        p.do_update = int(do_update)
        p.time_interval = float(cfg.time_interval)
        p.obstacle_weight = float(cfg.obstacle_weight)
        p.smoothness_weight = float(cfg.smoothness_weight)
        p.step_size = float(cfg.step_size)
        p.clip_grad_scale = float(cfg.clip_grad_scale)
        p.terminate_smooth_loss = float(cfg.terminate_smooth_loss)
        for d in range(9):
            p.link_smooth_weight[d] = float(cfg.link_smooth_weight[d])
        if self._poses_on:
            p.waypoint_poses, p.start_poses, p.end_poses = self.wp_pose.data_ptr(), self.start_pose.data_ptr(), self.end_pose.data_ptr()
        return p

    def _schedule(self):
        """Optimizer.update (omg/optimizer.py:59-80)."""
        cfg = self.cfg
        self.step_count += 1
        k = self.step_count
        cfg.obstacle_weight = cfg.base_obstacle_weight * cfg.cost_schedule_decay ** k
        cfg.smoothness_weight = cfg.smoothness_base_weight * cfg.cost_schedule_boost ** k
        cfg.grasp_weight = cfg.base_grasp_weight * cfg.cost_schedule_decay ** k
        cfg.step_size = cfg.step_decay_rate ** k * cfg.base_step_size

    def _gather_goal(self):
        """traj.end / chosen goal rows for the current goal_idx (online_learner.py:243-245, optimizer.py:93-99)."""
        idx = self.goal_idx.long()
        ar = torch.arange(self.S, device=self.device)
        torch.index_select(self.goal_set.view(self.S * self.G, 9), 0, ar * self.G + idx, out=self.goal_point)
        self.end.copy_(self.goal_point)
        if self.use_standoff:
            self.goal_rows.copy_(self.reach[ar, idx])
        else:
            self.goal_rows.copy_(self.goal_point[:, None, :])

    # ---------------------------------------------------------------------------------------------
    def _learner_params(self) -> _lib.LearnerParams:
        cfg = self.cfg
        p = _lib.LearnerParams()
        p.alg = _lib.ALG[self.ol_alg]
        p.num_goals, p.n_waypoints = self.G, self.n
        p.start_idx = min(int((self.t / cfg.optim_steps) * cfg.timesteps), cfg.timesteps - 1)  # online_learner.py:109-110
        p.constraint_num, p.use_standoff = self.c, int(self.use_standoff)
        p.normalize_cost = int(cfg.normalize_cost)
        p.base_obstacle_weight = float(cfg.base_obstacle_weight)
        p.smooth_weight = float(cfg.smoothness_base_weight * cfg.dist_eps)
        p.eta = self.eta
        p.cost_parts = ops.goalset_parts(cfg.timesteps - p.start_idx, self.LAT_GOAL_PARTS) if self.latency else (self._np(cfg.timesteps - p.start_idx) if self.goal_parts > 1 else 0)
        if self._poses_on:
            p.goal_pose_table, p.end_poses_out = self.goal_pose_tab.data_ptr(), self.end_pose.data_ptr()
        return p

    def _refresh_pose_tables(self):
        """Latency mode, at the start of a plan (after the initial goal pick): the start configurations' and all goal
        configurations' link poses (two small launches) and the current goal's poses — from then on the fused launches keep
        `end_pose` current themselves.  Valid while start, goal_set and goal_idx are only changed by the plan's own launches."""
        ops.pose_table(self.robot, self.P, self.start, out=self.start_pose)
        ops.pose_table(self.robot, self.P, self.goal_set, out=self.goal_pose_tab)
        ar = torch.arange(self.S, device=self.device)
        torch.index_select(self.goal_pose_tab.view(self.S * self.G, 120), 0, ar * self.G + self.goal_idx.long(), out=self.end_pose.view(self.S, 120))

    def pose_hand_over(self, on: bool = True):
        """Outside plan() (bench.py's step, callers that drive iterate() themselves): switch the hand-over of link poses between
        the launches on or off.  On: the start's and all goals' poses are tabulated now (two small launches) — valid while start,
        goal_set and goal_idx are only changed by the engine's own launches or restore(); every result keeps its bits."""
        self.join()
        if on and not self.separate_launches:
            self._refresh_pose_tables()
            self._poses_on = True
        else:
            self._poses_on = False

    def _tiling(self):
        return (self.LAT_GOAL_PARTS, self.LAT_LAYER_LINK_GROUPS, self.LAT_LAYER_BLOCK, 1)

    def _np(self, n_remaining: int) -> int:
        """Workgroups per goal of a batch-layout goal-set launch over a window of n_remaining configurations."""
        return ops.goalset_parts(n_remaining, self.goal_parts) if self.goal_parts > 1 else 1

    def goal_cost_total(self) -> torch.Tensor:
        """[S,G] float32 goal costs of the last goal-set launch (latency mode: the parts' sums added in part order, as the
        learner adds them)."""
        self.join()
        if not self.latency and self.goal_parts == 1:
            return self.goal_cost
        k = self._parts_last
        parts = self.goal_cost.reshape(-1)[: self.S * self.G * k].reshape(self.S, self.G, k)
        tot = parts[:, :, 0].clone()
        for j in range(1, k):
            tot += parts[:, :, j]
        return tot

    # ---------------------------------------------------------------------------------------------
    # Software pipeline.  One iteration is a goal-set launch that fills the GPU (~275 us for 100 scenes x 64 goals) followed by
    # the update launch (2 workgroups per scene, latency-bound, ~35 us): the tail of the first, the second and the ramp-up of
    # the next goal-set launch leave most of the GPU idle for ~80 us of every ~320.  Scenes are independent, so the batch is cut
    # into parts (contiguous scene ranges) whose iterations are enqueued alternately on different streams with no dependency
    # between them.  The parts mostly run in step — their goal-set launches at once, ramp-ups and tails overlapping, then their
    # update launches at once (DESIGN.md section 4 item 8 has the kernel trace) — which packs the same work into less time (measured:
    # 312 -> 285-290 us per iteration of 100 scenes; tools/experiments/ab_pipeline.py).  A part is a ChompEngine whose per-scene tensors are
    # row views of this engine's — same code, same results bit for bit; its dispatch schedule, work counters and flags are its own.
    def _make_part(self, lo: int, hi: int, stream):
        import copy
        part = object.__new__(ChompEngine)
        part.__dict__.update(self.__dict__)
        part.cfg = copy.copy(self.cfg)  # the weight schedule's fields are set per iteration (_iterate_pipelined)
        part.S, part.stream, part._lo = hi - lo, stream, lo
        part._batch_scenes = self.S  # the scenes of the whole batch: what the chip holds while this part's launches run
        part._parts, part._forked, part.pipeline = None, False, 1
        part._hot = None
        sc = object.__new__(ops.DeviceScenes)
        sc.__dict__.update(self.scenes.__dict__)
        sc.num_scenes, sc.scene_begin = hi - lo, self.scenes.scene_begin[lo:hi + 1]  # object offsets stay absolute
        part.scenes = sc
        part.work = torch.zeros(part.S * self.G * (1 if self.latency else self._parts_max), dtype=torch.int32, device=self.device)
        part.schedule, part._gs_launches, part._measured, part._sched_np = None, 0, False, 1
        part._sched_buf, part._sched_flip, part._sched_age = None, 0, None
        part._uniform_cache = None
        if self.goal_count is not None:
            part._goal_counts_host = self._goal_counts_host[lo:hi]
        self._bind_part(part)
        return part

    def _bind_part(self, part):
        lo, hi = part._lo, part._lo + part.S
        for k in self._PART_TENSORS:
            v = getattr(self, k)
            setattr(part, k, None if v is None else v[lo:hi])

    def _refresh_parts(self):
        """After one of this engine's per-scene tensors has been REBOUND (not written in place): the parts' views follow."""
        for part in self._parts or ():
            self._bind_part(part)

    def _get_parts(self, k: int):
        if self._parts is None or len(self._parts) != k:
            self.join()
            cuts = [self.S * i // k for i in range(k + 1)]
            self._parts = [self._make_part(cuts[i], cuts[i + 1], None if i == 0 else _side_stream(self.device, i)) for i in range(k)]
        return self._parts

    def join(self):
        """Make the current stream wait for everything the pipeline's side streams have been given.  Whole-batch operations
        of the engine call it themselves; a caller that reads the engine's tensors after bare pipelined iterate() calls
        needs it too (or a device-wide synchronize)."""
        if self._forked:
            cur = torch.cuda.current_stream(self.device)
            for part in self._parts[1:]:
                cur.wait_stream(part.stream)
            self._forked = False
        if self.__dict__.get("_persist_unchecked"):
            # a persistent launch reports what went wrong inside it (a bounded wait that ran out) through its control block only: look at it
            # before anything else is done with the engine's tensors (one small download per persistent launch)
            self._persist_unchecked = False
            st = self.persistent_status()
            if st["failure"]:
                raise _lib.OmgHipError(f"omgx_plan_persistent: failure code {st['failure']} inside the launch ({st}); the engine's tensors are not valid")

    _CFG_SCHEDULE = ("obstacle_weight", "smoothness_weight", "grasp_weight", "step_size")

    def _iterate_pipelined(self, t: int, early_stop: bool, k: int):
        parts = self._get_parts(k)
        if early_stop:
            self._masked = True
        if not self._forked:  # the side streams start behind whatever this stream has done to the engine's tensors
            cur = torch.cuda.current_stream(self.device)
            for part in parts[1:]:
                part.stream.wait_stream(cur)
            self._forked = True
        for part in parts:
            part.t, part.step_count, part._masked, part._poses_on = self.t, self.step_count, self._masked, self._poses_on
            for f in self._CFG_SCHEDULE:
                setattr(part.cfg, f, getattr(self.cfg, f))
            part.iterate(t, early_stop)
        p0 = parts[0]
        self.t, self.step_count = p0.t, p0.step_count
        for f in self._CFG_SCHEDULE:
            setattr(self.cfg, f, getattr(p0.cfg, f))

    def _pipeline_parts(self) -> int:
        # The pipeline forks from and joins into torch's CURRENT stream.  An engine bound to a stream of its own keeps every
        # launch on that stream — switching between the two modes would leave the streams unordered — so it never pipelines
        # by itself and refuses an explicit request.
        if self.latency:
            return 1
        if self.pipeline is not None:
            k = max(1, min(int(self.pipeline), self.S))
            if k > 1 and self.stream is not None:
                raise ValueError("ChompEngine(stream=...) cannot be pipelined: the pipeline's parts run on the current stream and the shared side streams")
            return k
        if self.stream is not None:
            return 1
        return self.auto_parts(self.S, self.G) if (self._in_plan and not self.separate_launches) else 1

    def update_goal(self, defer_update: bool = False, with_layer: bool = False):
        """Learner.update_goal (online_learner.py:237-249): omgx_goalset_cost + omgx_goal_update, no host sync.
        defer_update=True launches only the goal-set batch and returns the learner parameters: the caller then runs
        the goal update fused with the optimiser step (omgx_goal_update_optimize)."""
        self.join()
        self.t += 1
        if self.ol_alg == "Baseline":
            return None
        prm = self._learner_params()
        if self.ol_alg != "Proj":  # cost_vector's obstacle batch (online_learner.py:128-148)
            n_rem = self.cfg.timesteps - prm.start_idx
            traj_start = self.traj[:, prm.start_idx]  # strided view into the trajectory tensor: no copy kernel
            if self.latency:
                self._gs_launches += 1
                self._parts_last = ops.goalset_cost_layer_tiled(
                    self.robot, self.P, self.scenes, traj_start, self.cv_goals, n_rem, self.cfg.time_interval,
                    self.traj if with_layer else None, (self.pot, self.pgrad, self.col), (self.goal_cost, self.goal_col),
                    soften_fingers=False, layer_soften_fingers=self.cfg.uncheck_finger_collision == -1, active=self._mask(),
                    goal_count=self.goal_count, goal_parts=self.LAT_GOAL_PARTS, layer_link_groups=self.LAT_LAYER_LINK_GROUPS,
                    layer_config_block=self.LAT_LAYER_BLOCK, spread=True,
                    layer_poses=self.wp_pose if (self._poses_on and with_layer) else None, prepass=self.prepass)
            elif with_layer:  # the SDF layer of the current trajectories rides on the goal-set launch
                # the second launch is the measuring one (the first runs on cold caches and would distort the weights);
                # until then the items are split evenly by count.  Small batches keep the even split: measuring only pays
                # when the launch has several rounds of workgroups per CU.
                # With an active mask (early stop) the launch goes back to scene-major order, or — `reschedule_every` > 0 — the
                # schedule is rebuilt from the same measured durations without the scenes that have terminated.
                self._gs_launches += 1
                NP = self._np(n_rem)
                self._parts_last = NP
                use_sched = self.auto_schedule and (not self._masked or (self._measured and bool(self.reschedule_every)))
                measure = use_sched and not self._measured and self._gs_launches >= 2 and self.S * self.G * NP >= self.MEASURE_MIN_ITEMS and NP == self._parts_max
                if use_sched and self._masked:
                    if self._sched_age is None or self._sched_age >= self.reschedule_every or self._sched_np != NP:
                        self.schedule = self.build_schedule(active=self._mask(), parts=NP, uniform=(NP != self._parts_max or not self._measured))
                        self._sched_age = 0
                    self._sched_age += 1
                if use_sched and (self.schedule is None or self._sched_np != NP):
                    # nothing measured yet, or the window has shrunk to another number of parts per goal (the last few iterations
                    # of a plan): the items in scene-major order, equal counts per XCD
                    self.schedule = self.build_schedule(uniform=True, parts=NP)
                ops.goalset_cost_layer(self.robot, self.P, self.scenes, traj_start, self.cv_goals, n_rem, self.cfg.time_interval,
                                       self.traj, (self.pot, self.pgrad, self.col), soften_fingers=False,
                                       layer_soften_fingers=self.cfg.uncheck_finger_collision == -1,
                                       out=(self.goal_cost, self.goal_col), active=self._mask(), goal_count=self.goal_count,
                                       schedule=self.schedule if use_sched else None, work=self.work[: self.S * self.G * NP] if measure else None,
                                       goal_parts=self.goal_parts, layer_poses=self.wp_pose if self._poses_on else None,
                                       prepass=self.prepass)
                if measure:
                    self._measured = True
                    self.schedule = self.build_schedule(parts=NP)
            elif self.goal_parts > 1:  # the batch alone, split goals: the same partial sums as the fused launch writes
                self._parts_last = ops.goalset_cost_layer_tiled(
                    self.robot, self.P, self.scenes, traj_start, self.cv_goals, n_rem, self.cfg.time_interval, None, None,
                    (self.goal_cost, self.goal_col), soften_fingers=False, active=self._mask(), goal_count=self.goal_count,
                    goal_parts=self.goal_parts, layer_link_groups=5, layer_config_block=0, spread=False, prepass=self.prepass)
            else:
                ops.goalset_cost(self.robot, self.P, self.scenes, traj_start, self.cv_goals, n_rem, self.cfg.time_interval,
                                 soften_fingers=False, out=(self.goal_cost, self.goal_col), active=self._mask(),
                                 goal_count=self.goal_count, prepass=self.prepass)
        elif with_layer:
            self._layer()
        if defer_update:
            return prm
        ops.goal_update(prm, self.traj, self.goal_set, self.reach, self.goal_cost, self.learner_state, self.goal_idx,
                        self.end, self.goal_rows, self.goal_point, self.cost_vec, active=self._mask(),
                        goal_count=self.goal_count, eta=self.eta_s)
        return None

    def _uniform_schedule(self, parts: int = 1) -> torch.Tensor:
        """The schedule before anything has been measured: the (scene, goal[, part]) items in scene-major order, cut into 8 pieces
        of equal COUNT, one per XCD — built on the host (numpy) and uploaded once."""
        S, G = self.S, self.G * parts
        if self.goal_count is not None:
            items = np.concatenate([s * G + np.arange(int(c) * parts) for s, c in enumerate(self._goal_counts_host)])
        else:
            items = np.arange(S * G)
        n = len(items)
        pos = np.arange(n)
        x = np.minimum(8 * pos // max(n, 1), 7)
        first = np.searchsorted(x, np.arange(8))
        rank = pos - first[x]
        slots = int(rank.max()) + 1 if n else 1
        sched = np.full(slots * 8, -1, np.int32)
        sched[rank * 8 + x] = items
        return torch.as_tensor(sched, device=self.device)

    def build_schedule(self, active: "torch.Tensor | None" = None, uniform: bool = False, parts: int = 1) -> torch.Tensor:
        """Dispatch order for omgx_goalset_cost_layer (omgx_goalset_schedule: one small launch on the current stream, no host
        sync).

        The (scene, goal) items are laid out scene by scene — scenes by decreasing measured work (`work`, the durations of
        the measuring launch), each scene's goals longest first — and this list is cut into 8 contiguous pieces of equal
        total work, one per XCD (goal workgroup b of the launch runs on XCD b % 8).  Every XCD then holds whole scenes except
        for at most two that it shares with a neighbour, so a scene's SDF volumes stay in one or two L2s, and all XCDs finish
        together whatever the number of scenes (12 or 13 scenes per GPU would otherwise leave 3 of 8 XCDs with half the
        load).  `uniform`: all items weigh the same.  `active` [S] int32: scenes with 0 are left out; ragged goal sets leave
        out their padding.  An XCD has room for `schedule_slack` (2) times its share of the items; the weights are clamped to a
        band [L, schedule_slack * L] around their mean, so no piece of the list can need more.
        `parts`: the launch deals every goal over that many workgroups (goal_parts): the items are (scene, goal, part)."""
        self._sched_np = parts
        if self.S * self.G * parts > 65536 or self.S > _lib.SCHEDULE_MAX_SCENES:  # beyond the scheduler kernel's single workgroup: even split by count
            if self._uniform_cache is None:
                self._uniform_cache = {}
            if parts not in self._uniform_cache:  # built once: the upload is a host-to-device copy, which a graph capture could not record
                self._uniform_cache[parts] = self._uniform_schedule(parts)
            return self._uniform_cache[parts]
        if self._sched_buf is None:
            self._sched_buf = {}
        # two buffers (per item count) in turn: a launch that still reads the previous schedule (another stream's view of it) is never overwritten
        self._sched_flip ^= 1
        key = (parts, self._sched_flip)
        out = ops.goalset_schedule(None if uniform else self.work[: self.S * self.G * parts], self.S, self.G, active=active, goal_count=self.goal_count,
                                   slack=self.schedule_slack, out=self._sched_buf.get(key), device=self.device, parts=parts,
                                   longest_first=(not uniform) and self.S * self.G * parts <= self.LONGEST_FIRST_MAX_ITEMS)
        self._sched_buf[key] = out
        return out

    def _layer(self):
        """SDF layer outputs of the current waypoints (first half of Cost.compute_total_loss)."""
        if self.latency:
            ops.goalset_cost_layer_tiled(self.robot, self.P, self.scenes, None, None, 1, self.cfg.time_interval, self.traj,
                                         (self.pot, self.pgrad, self.col), None, layer_soften_fingers=self.cfg.uncheck_finger_collision == -1,
                                         layer_link_groups=self.LAT_LAYER_LINK_GROUPS,
                                         layer_config_block=self.LAT_LAYER_BLOCK, spread=True,
                                         layer_poses=self.wp_pose if self._poses_on else None)
            return
        if self._poses_on:  # the same five layer workgroups per scene as omgx_fk_sdf launches, leaving the waypoints' poses for the step
            ops.goalset_cost_layer_tiled(self.robot, self.P, self.scenes, None, None, 1, self.cfg.time_interval, self.traj,
                                         (self.pot, self.pgrad, self.col), None, layer_soften_fingers=self.cfg.uncheck_finger_collision == -1,
                                         goal_parts=1, layer_link_groups=5, layer_config_block=0, spread=False, layer_poses=self.wp_pose)
            return
        ops.fk_sdf(self.robot, self.P, self.scenes, self.traj, soften_fingers=self.cfg.uncheck_finger_collision == -1,
                   out=(self.pot, self.pgrad, self.col))

    def _step(self, do_update: bool, learner_prm=None, stop_on_terminate: bool = False):
        if learner_prm is not None:  # goal update + step in one launch
            # learner and step in different workgroups of the launch: pays while both sets are resident at once (one
            # 92 KB-LDS workgroup per CU); beyond that the single-workgroup kernel is a little faster (measured at 200 / 400 scenes)
            split = 2 * self.S <= self._num_cus if self.split_update is None else bool(self.split_update)
            ticket = self._next_ticket()
            ops.goal_update_optimize(learner_prm, self.goal_set, self.reach, self.goal_cost, self.learner_state, self.goal_idx,
                                     self.robot, self._params(do_update), self.traj, self.start, self.end, self.goal_rows,
                                     self.goal_point, self.pot, self.pgrad, self.col, active=self.active,
                                     out=(self.grad, self.cost_traj, self.info), cost_vector=self.cost_vec,
                                     scene_flags=self._scene_flags if split else None, ticket=ticket,
                                     stop_on_terminate=stop_on_terminate, goal_count=self.goal_count, eta=self.eta_s)
            return self.info
        ops.chomp_optimize(self.robot, self._params(do_update), self.traj, self.start, self.end, self.goal_rows,
                           self.goal_point, self.pot, self.pgrad, self.col, active=self.active,
                           out=(self.grad, self.cost_traj, self.info), stop_on_terminate=stop_on_terminate)
        return self.info

    def optimize(self, do_update: bool = True):
        """Optimizer.optimize(traj, force_update=True) (omg/optimizer.py:115-135) for all scenes."""
        self.join()
        self._schedule()
        self._layer()
        return self._step(do_update)

    def iterate(self, t: int, early_stop: bool = False):
        """One pass of the planner loop body (planner.py:612-621) over all scenes: two launches on one stream —
        omgx_goalset_cost_layer (goal-set batch + SDF layer of the current trajectories) and omgx_goal_update_optimize
        (Learner.update_goal + Optimizer.optimize); once the goal is fixed (t >= optim_steps, "Proj", "Baseline") the layer
        launch and the step."""
        k = self._pipeline_parts()
        cfg = self.cfg
        if k > 1 and self.SMOOTH_SINGLE_PART_BELOW > self.S and not (cfg.goal_set_proj and t < cfg.optim_steps and self.ol_alg not in ("Baseline", "Proj")) \
                and self.HOT_FIXED_GOAL and not self.separate_launches and self.stream is None:
            self.join()  # (the side streams are joined once, at the phase's first iteration; scenes are independent: the same bits)
            k = 1
        if k > 1:
            return self._iterate_pipelined(t, early_stop, k)
        # planner.py:609-618: the learner runs only for the online-learning rules and only for the first optim_steps iterations
        select = cfg.goal_set_proj and t < cfg.optim_steps and self.ol_alg not in ("Baseline", "Proj")
        if not self.separate_launches and select and not self._forked:
            if early_stop:
                self._masked = True
            if self._iterate_hot(bool(early_stop and t > 0)):
                return None
        elif self.HOT_FIXED_GOAL and not self.separate_launches and not self._forked and not select:
            # the goal is fixed (the plan's last cfg.extra_smooth_steps iterations; "Proj" / "Baseline" throughout): layer launch + step
            # through the prepared calls
            if early_stop:
                self._masked = True
            self._iterate_hot_fixed(bool(early_stop and t > 0))
            return None
        if self.stream is not None and torch.cuda.current_stream(self.device) != self.stream:
            with torch.cuda.stream(self.stream):
                return self._iterate_general(t, early_stop)
        return self._iterate_general(t, early_stop)

    def _iterate_general(self, t: int, early_stop: bool):
        self.join()
        if self.separate_launches:
            return self.iterate_separate(t, early_stop)
        cfg = self.cfg
        # planner.py:609-618: the learner runs only for the online-learning rules; "Proj" and "Baseline" keep the goal that
        # was fixed before planning (select_initial_goal)
        select = cfg.goal_set_proj and t < cfg.optim_steps and self.ol_alg not in ("Baseline", "Proj")
        # planner.py:626-627 (a scene that terminates at t > 0 leaves the loop): the step itself clears active[s] — no extra
        # kernels between iterations; the goal-set launch, the goal update and the step skip scenes with active[s] == 0
        stop = bool(early_stop and t > 0)
        if early_stop:
            self._masked = True
        if select:
            lprm = self.update_goal(defer_update=True, with_layer=True)
            self._schedule()
            self._step(True, lprm, stop_on_terminate=stop)
        else:
            self._layer()
            self._schedule()
            self._step(True, None, stop_on_terminate=stop)

    _HOT_TENSORS = ("robot", "cv_goals", "traj", "pot", "pgrad", "col", "goal_cost", "goal_col", "goal_set", "reach", "learner_state",
                    "goal_idx", "start", "end", "goal_rows", "goal_point", "grad", "cost_traj", "info", "cost_vec", "_active",
                    "goal_count", "eta_s", "_scene_flags")

    def _iterate_hot(self, stop: bool) -> bool:
        """A goal-selecting iteration in its steady state — schedule settled (or none), nothing to measure — through
        ops.IterationCalls: the same two launches with argument lists prepared once, on this engine's stream without
        switching torch's current stream.  Returns False when the general path has to run (first launches of a plan, schedule
        rebuilds, the rules without a goal-set batch)."""
        if self.ol_alg in ("Baseline", "Proj"):
            return False
        use_sched = self.auto_schedule and not self._masked and not self.latency
        if not self.latency and self._masked and self.auto_schedule and self._measured and self.reschedule_every:
            return False
        if use_sched:
            cfg = self.cfg
            NP = self._np(cfg.timesteps - min(int(((self.t + 1) / cfg.optim_steps) * cfg.timesteps), cfg.timesteps - 1))  # the window of the launch to come
            if self.schedule is None or self._sched_np != NP or (not self._measured and self._gs_launches >= 1 and self.S * self.G * self._parts_max >= self.MEASURE_MIN_ITEMS):
                return False
        calls = self._hot_calls()
        stream = (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).cuda_stream
        self.t += 1
        prm = self._learner_params()
        self._gs_launches += 1
        self._parts_last = max(1, int(prm.cost_parts))
        calls.use_layer_poses = self._poses_on
        calls.goalset_layer(prm.start_idx, self._masked, self.schedule if use_sched else None, None, stream)
        self._schedule()
        split = 2 * self.S <= self._num_cus if self.split_update is None else bool(self.split_update)
        calls.update(prm, self._params(True), split, self._next_ticket(), stop, stream)
        return True

    def _hot_calls(self) -> "ops.IterationCalls":
        # the prepared calls belong to these very tensor objects (held here, so none of them can be freed and its identity reused)
        key = [getattr(self, k) for k in self._HOT_TENSORS] + [self.scenes.scene_begin]
        baked = (float(self.cfg.time_interval), self.cfg.uncheck_finger_collision == -1)  # the scalars the calls carry
        hot = self._hot
        if hot is None or hot[2] != baked or not all(map(operator.is_, hot[0], key)):
            calls = ops.IterationCalls(self.robot, self.P, self.scenes, self.cv_goals, self.cfg.time_interval, self.traj,
                                       (self.pot, self.pgrad, self.col), (self.goal_cost, self.goal_col), self.goal_set, self.reach,
                                       self.learner_state, self.goal_idx, self.start, self.end, self.goal_rows, self.goal_point,
                                       (self.grad, self.cost_traj, self.info), self.cost_vec, self._active, self.goal_count, self.eta_s,
                                       self._scene_flags, layer_soften_fingers=self.cfg.uncheck_finger_collision == -1,
                                       tiling=self._tiling() if self.latency else None,
                                       layer_poses=self.wp_pose, goal_parts=self.goal_parts, prepass=self.prepass)
            calls.batch_scenes = int(self.__dict__.get("_batch_scenes", self.S))  # (a pipeline part: the whole engine's count)
            hot = self._hot = (key, calls, baked)
        return hot[1]

    def _iterate_hot_fixed(self, stop: bool):
        """An iteration with the goal fixed — the layer launch and the step (`_layer()` + `_step()` of the general path: the same two
        entry points on the same tensors) — through ops.IterationCalls, on this engine's stream without switching torch's current
        stream.  The general path costs ~50 us of host time per part and iteration (argument checks, stream context), more than
        the two launches need on the device: the last 20 iterations of a pipelined plan were bound by the host (kernel trace of a
        100-scene plan: a part's period 150 us against 58 us of kernels)."""
        calls = self._hot_calls()
        stream = (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).cuda_stream
        calls.use_layer_poses = self._poses_on
        calls.layer_only(stream)
        self._schedule()
        calls.step(self._params(True), stop, stream)

    def iterate_separate(self, t: int, early_stop: bool = False):
        """The same iteration composed from the separate entry points the drop-in classes use — omgx_goalset_cost,
        omgx_goal_update, omgx_fk_sdf, omgx_chomp_optimize (five launches).  Same results as iterate(); kept as the
        cross-check of the fused launches (tests) and as the reference composition for callers of the C ABI."""
        cfg = self.cfg
        select = cfg.goal_set_proj and t < cfg.optim_steps and self.ol_alg not in ("Baseline", "Proj")
        stop = bool(early_stop and t > 0)
        if early_stop:
            self._masked = True
        self._layer()
        if select:
            self.update_goal()
        self._schedule()
        self._step(True, None, stop_on_terminate=stop)

    # ---------------------------------------------------------------------------------------------
    # The persistent planner launch (omgx_plan_persistent, csrc/omg_persist.h): K iterations of iterate() for all scenes in ONE
    # launch, scheduled by the per-scene dependency instead of by launch boundaries.  Bit for bit what K calls of iterate() compute.
    PERSISTENT_ALGS = ("FTL", "FTC", "Exp", "MD")

    def persistent_ok(self) -> bool:
        """Can run_persistent() serve this engine?  Batch layout with whole goals, an online-learning rule with a goal-set batch
        (or fixed-goal iterations only), the pose hand-over available."""
        return (not self.latency and self.goal_parts == 1 and not self.separate_launches and self.stream is None and self.S <= 65535)

    def _iter_table(self, ts, early_stop: bool, pin_window: bool):
        """The omgx_plan_iter records of iterate(t) for t in ts — and, like the calls it replaces, the host-side counters advance:
        Learner.t, Optimizer.update's schedule (omg/optimizer.py:59-80)."""
        cfg = self.cfg
        recs = (_lib.PlanIter * len(ts))()
        for k, t in enumerate(ts):
            select = bool(cfg.goal_set_proj and t < cfg.optim_steps and self.ol_alg not in ("Baseline", "Proj"))
            r = recs[k]
            if select:
                if pin_window:
                    self.t = 0
                self.t += 1
                r.start_idx = min(int((self.t / cfg.optim_steps) * cfg.timesteps), cfg.timesteps - 1)  # online_learner.py:109-110
            r.mode = int(select)
            self._schedule()
            r.obstacle_weight, r.smoothness_weight, r.step_size = float(cfg.obstacle_weight), float(cfg.smoothness_weight), float(cfg.step_size)
            r.do_update = 1
            r.stop_on_terminate = int(bool(early_stop and t > 0))
        return recs

    def run_persistent(self, ts, early_stop: bool = False, pin_window: bool = False, max_workgroups: int = 0, update_cus: int = -1):
        """iterate(t, early_stop) for every t of `ts`, in order, as ONE launch on the current stream (no host sync).  pin_window: the
        learner's window stays at its first-iteration size (bench.py's step: Learner.t = 0 before every iteration)."""
        if not self.persistent_ok():
            raise ValueError("run_persistent: batch layout with whole goals on the current stream only")
        ts = [int(t) for t in ts]
        if not ts:
            return
        self.join()
        if any(self.cfg.goal_set_proj and t < self.cfg.optim_steps for t in ts) and self.ol_alg not in self.PERSISTENT_ALGS and self.ol_alg not in ("Baseline", "Proj"):
            raise ValueError(f"run_persistent: ol_alg {self.ol_alg!r}")
        if not self._poses_on:
            self._refresh_pose_tables()
            self._poses_on = True
        if early_stop:
            self._masked = True
        recs = self._iter_table(ts, early_stop, pin_window)
        raw = bytes(recs)
        cache = self.__dict__.setdefault("_persist_tables", {})
        d_iters = cache.get(raw)
        if d_iters is None:
            if len(cache) > 64:
                cache.clear()
            d_iters = cache[raw] = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(self.device)
        ws = self.__dict__.get("_persist_ws")
        if ws is None:
            ws = self._persist_ws = torch.zeros(int(_lib.lib().omgx_plan_persistent_workspace_bytes(self.S, self.n)), dtype=torch.uint8, device=self.device)
        lprm = self._learner_params()
        lprm.cost_parts = 0
        self._parts_last = 1
        ops.plan_persistent(self.robot, self.P, self.scenes, self.cv_goals, self.cfg.time_interval, self.traj, (self.pot, self.pgrad, self.col),
                            self.wp_pose, (self.goal_cost, self.goal_col), lprm, self.goal_set, self.reach, self.learner_state, self.goal_idx,
                            self.cost_vec, self._params(True), self.start, self.end, self.goal_rows, self.goal_point,
                            (self.grad, self.cost_traj, self.info), recs, d_iters, ws, active=self._mask(), goal_count=self.goal_count, eta=self.eta_s,
                            soften_fingers=False, layer_soften_fingers=self.cfg.uncheck_finger_collision == -1, max_workgroups=max_workgroups, update_cus=update_cus)
        self._persist_unchecked = True  # join() reads the launch's status

    def persistent_status(self) -> dict:
        return ops.plan_persistent_status(self._persist_ws, self.S)

    # ---------------------------------------------------------------------------------------------
    _STATE = ("traj", "end", "goal_rows", "goal_point", "goal_idx", "learner_state", "info", "active", "goal_cost", "goal_col",
              "cost_vec", "grad", "cost_traj", "pot", "pgrad", "col", "end_pose")

    def snapshot(self) -> dict:
        """Everything a plan mutates (device tensors cloned + the host-side counters): restore() brings the engine back to
        this point — e.g. to plan the same scenes again, or to keep a benchmark's workload stationary."""
        self.join()
        snap = {k: getattr(self, k).clone() for k in self._STATE}
        snap["_host"] = (self.step_count, self.t, self.cfg.obstacle_weight, self.cfg.smoothness_weight, self.cfg.grasp_weight, self.cfg.step_size)
        snap["_masked"] = self._masked
        return snap

    def restore(self, snap: dict):
        """Device-to-device copies on the current stream, no host sync."""
        self.join()
        dst, src = [], []
        for k in self._STATE:
            cur = getattr(self, k)
            if cur.shape != snap[k].shape:  # early_stop used to rebind self.active: keep one tensor
                setattr(self, k, snap[k].clone())
                self._refresh_parts()
            else:
                dst.append(cur)
                src.append(snap[k])
        if dst:  # one fused copy per dtype instead of a launch per tensor (17 of them: ~0.15 ms of launches every time a benchmark rewinds its workload)
            if hasattr(torch, "_foreach_copy_"):
                torch._foreach_copy_(dst, src)
            else:  # older torch builds: one copy per tensor
                for d_, s_ in zip(dst, src):
                    d_.copy_(s_)
        (self.step_count, self.t, self.cfg.obstacle_weight, self.cfg.smoothness_weight, self.cfg.grasp_weight, self.cfg.step_size) = snap["_host"]
        # the mask goes back with the flags it guards: a snapshot taken before any early stop has every scene active, and the
        # launches after the restore run unmasked again (dispatch schedule in use, no mask look-up in the goal-set kernel)
        self._masked = bool(snap.get("_masked", self._masked))

    def select_initial_goal(self):
        """Learner.__init__ (online_learner.py:96-102): before planning, pick the cheapest goal by one cost_vector
        evaluation at t = 0 and re-interpolate the trajectory towards it (Trajectory.interpolate_waypoints, cubic)."""
        self.join()
        if self.ol_alg in ("Proj", "Baseline"):
            # planner.py:200-222 (goal setup before planning): "Proj" takes the goal closest to the START in the
            # link_smooth_weight metric, "Baseline" cfg.goal_idx (>= 0: that goal; the default -2: goal 0; -1 would need the
            # grasp potentials of the scene file, which the engine does not carry: goal 0); then traj.end and the cubic init
            if self.ol_alg == "Proj":
                w = torch.as_tensor(np.broadcast_to(np.asarray(self.cfg.link_smooth_weight, np.float64).ravel(), (9,)).copy(),
                                    dtype=torch.float64, device=self.device)
                d = torch.linalg.norm((self.start[:, None, :] - self.goal_set) * w, dim=-1)
                if self.goal_count is not None:  # padding of a ragged goal set never wins
                    d = torch.where(torch.arange(self.G, device=self.device)[None, :] < self.goal_count[:, None], d, torch.full_like(d, float("inf")))
                self.goal_idx.copy_(torch.argmin(d, dim=1).to(torch.int32))
            else:
                gi = int(getattr(self.cfg, "goal_idx", -2))
                self.goal_idx.fill_(gi if 0 <= gi < self.G else 0)
                if self.goal_count is not None:  # a scene with fewer goals than cfg.goal_idx falls back to goal 0, never to padding
                    self.goal_idx.copy_(torch.where(self.goal_idx < self.goal_count, self.goal_idx, torch.zeros_like(self.goal_idx)))
            self._gather_goal()
            self._cubic_to_end()
            return
        saved_t, saved_alg = self.t, self.ol_alg
        if self._scratch_state is None:
            self._scratch_state = torch.empty_like(self.learner_state)
        scratch = self._scratch_state  # (no allocation per plan)
        scratch.copy_(self.learner_state)
        keep = self.learner_state
        try:
            self.t, self.ol_alg, self.learner_state = -1, "FTC", scratch  # update_goal increments t to 0: start_idx 0, argmin(costs)
            self.update_goal()
        finally:
            self.t, self.ol_alg, self.learner_state = saved_t, saved_alg, keep
        self._cubic_to_end()

    def _cubic_to_end(self):
        """Trajectory.interpolate_waypoints towards the chosen end: the clamped cubic through (0, start), (1, end).  The weights are
        computed once per engine; the same three elementwise operations (difference, product, sum) as before, into preallocated
        tensors."""
        if self._cubic_h is None:
            tt = (torch.arange(1, self.n + 1, device=self.device, dtype=torch.float64) / (self.n + 1.0))[None, :, None]
            self._cubic_h = 3.0 * tt * tt - 2.0 * tt * tt * tt
            self._cubic_tmp = torch.empty_like(self.traj)
            self._cubic_diff = torch.empty_like(self.start)
        torch.sub(self.end, self.start, out=self._cubic_diff)
        torch.mul(self._cubic_h, self._cubic_diff[:, None, :], out=self._cubic_tmp)
        torch.add(self.start[:, None, :], self._cubic_tmp, out=self.traj)

    def plan(self, early_stop: bool = True, initial_goal: bool = True) -> torch.Tensor:
        """Planner.plan (planner.py:600-653): up to optim_steps + extra_smooth_steps iterations, then one
        info-only evaluation; returns the final info [S,16] (device).  cfg.timeout (3 s; -1: none) is the reference's
        wall-clock budget (planner.py:629): once it is spent after an iteration t > 0, the loop ends for every scene.  The
        engine's loop is asynchronous, so with a budget set the host keeps at most PLAN_LOOKAHEAD iterations ahead of the
        device (an event every fourth iteration; the host waits for the one PLAN_LOOKAHEAD iterations back — the launch queue stays full, nothing
        is lost) and the clock it reads is the device's progress to within those few iterations.  A captured plan
        (capture_plan) has no host in its loop and no budget.
        A plan started with early_stop=False on an engine whose mask is already in use (an earlier early-stop plan that was
        not restore()d, or `active` assigned) keeps skipping the scenes that are switched off."""
        import time
        cfg = self.cfg
        if initial_goal and cfg.goal_set_proj:
            self.select_initial_goal()
        if self.LAT_HAND_OVER_POSES and not self.separate_launches:
            self._refresh_pose_tables()
            self._poses_on = True
        self.iterations_run = 0
        self.timed_out = False
        t_start = time.time()
        budget = cfg.timeout != -1 and not self._capturing
        marks = []
        self._in_plan = True  # plan() joins the pipeline's streams itself (the final optimize below), so it may use them
        try:
            for t in range(cfg.optim_steps + cfg.extra_smooth_steps):
                self.iterate(t, early_stop)
                self.iterations_run = t + 1
                if budget:
                    if t % 4 == 3:  # a mark every 4 iterations, PLAN_LOOKAHEAD / 4 of them outstanding
                        marks.append(self._record_progress())
                        if len(marks) > self.PLAN_LOOKAHEAD // 4:
                            for ev in marks.pop(0):
                                ev.synchronize()
                    if t > 0 and time.time() - t_start > cfg.timeout:
                        self.timed_out = True
                        break
                if self._plan_all_done(early_stop, t):
                    break
        except BaseException:
            self._poses_on = False
            raise
        finally:
            self._in_plan = False
        try:
            return self.optimize(False)
        finally:
            self._poses_on = False

    PLAN_LOOKAHEAD = 8  # iterations the host may be ahead of the device while plan() watches cfg.timeout

    def _record_progress(self) -> list:
        """Events behind everything enqueued so far: one on this engine's stream and one on every side stream the pipeline is
        using (no cross-stream waits: the parts stay independent)."""
        main = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        streams = [main] + ([part.stream for part in self._parts[1:]] if self._forked else [])
        evs = []
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        return evs

    def _plan_all_done(self, early_stop: bool, t: int) -> bool:
        """Every scene may have left the loop (planner.py:626 breaks at once; a lone scene often terminates after two
        iterations): look at the mask at a thinning set of iterations and stop launching no-ops.  Each look is a host
        sync that drains the launch queue (measured: ~0.4 ms each with 100 scenes in flight, where it never pays; 13 scenes:
        6.5 ms per plan with the looks, 6.3 without), so only the smallest batches do it — where all scenes often ARE done early."""
        if self._capturing:  # a graph has no host in its loop
            return False
        if not (early_stop and self.S <= 4 and t in _ALL_DONE_CHECKS):
            return False
        self.join()  # the mask rows of a pipelined engine's side stream
        return not self._active.cpu().numpy().any()  # a plain copy: a torch reduction would load its kernel (~10 ms) on first use

    def capture_plan(self, early_stop: bool = True, initial_goal: bool = True) -> "PlanGraph":
        """plan() as ONE HIP graph: the initial goal pick, all optim_steps + extra_smooth_steps iterations (on both streams of the
        pipeline where it applies) and the final evaluation are recorded once and replayed with a single launch — the planner
        loop of planner.py:600-653 without the host in it.  Everything the host decides per iteration is frozen at capture: the
        weight schedule (Optimizer.update), the goal-set window, the iteration count; what depends on the data stays on the
        device — a scene that terminates (`early_stop`) is skipped by every later launch through its `active` flag, but the
        launches themselves remain, and cfg.timeout has no meaning.  Replays compute exactly what plan() computes from the
        same state (tests/test_gpu_pipeline.py).  The engine's state after this call is its state before it.

            fresh = eng.snapshot(); graph = eng.capture_plan()
            for problem in problems:                       # same shapes, same scenes
                eng.restore(fresh); eng.start.copy_(...); eng.goal_set.copy_(...); eng.traj.copy_(...)
                info = graph.replay()
        """
        state = self.snapshot()
        self.plan(early_stop, initial_goal)  # warm-up: workspaces, dispatch schedules, pipeline parts and their streams exist afterwards
        self.restore(state)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with torch.cuda.graph(graph):
                info = self.plan(early_stop, initial_goal)
                self.join()
        finally:
            self._capturing = False
        self.restore(state)  # nothing has run, but the host-side counters went through a plan
        torch.cuda.synchronize(self.device)
        return PlanGraph(self, graph, info)

    def final_costs(self) -> torch.Tensor:
        self.join()
        return self.info[:, 0].contiguous()


class PlanGraph:
    """A captured ChompEngine.plan (ChompEngine.capture_plan)."""

    def __init__(self, engine: ChompEngine, graph: "torch.cuda.CUDAGraph", info: torch.Tensor):
        self.engine, self.graph, self.info = engine, graph, info

    def replay(self) -> torch.Tensor:
        """Run the plan from the engine's current device state on the current stream: one launch, no host sync.  Returns the
        engine's info tensor [S,16] (valid once the stream has caught up)."""
        self.engine.join()
        self.graph.replay()
        return self.info


def _collective_on(world: int) -> bool:
    """A process group exists (launched by torch.distributed.run): the collective runs even for a single rank, so that a
    one-GPU box exercises the same RCCL path as a node."""
    if world > 1:
        return True
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def gather_costs_equal(local_costs: torch.Tensor, world: int) -> torch.Tensor:
    """All-gather for equal-sized shards (bench.py's weak-scaling layout): one collective, no host sync."""
    if not _collective_on(world):
        return local_costs
    import torch.distributed as dist
    out = torch.empty(world * local_costs.numel(), dtype=local_costs.dtype, device=local_costs.device)
    dist.all_gather_into_tensor(out, local_costs.contiguous())
    return out


def gather_costs(local_costs: torch.Tensor, world: int, total: int | None = None) -> torch.Tensor:
    """The one collective of the job: all-gather of per-scene final costs over RCCL (backend "nccl") or gloo.
    `total`: number of scenes in all — rank r then holds the block shard_range(total, r, world), whose size every rank knows:
    the (at most one scene) shorter shards are padded to the longest with NaN, ONE all_gather_into_tensor moves them and the
    padding is cut away by index, without any host synchronisation.  Without `total` the shards are taken to be equal."""
    if not _collective_on(world):
        return local_costs
    import torch.distributed as dist
    if total is None:
        return gather_costs_equal(local_costs, world)
    sizes = [len(shard_range(total, r, world)) for r in range(world)]
    mx = max(sizes)
    if local_costs.numel() != sizes[dist.get_rank()]:
        raise ValueError(f"rank {dist.get_rank()} holds {local_costs.numel()} costs, shard_range says {sizes[dist.get_rank()]}")
    pad = torch.full((mx,), float("nan"), dtype=local_costs.dtype, device=local_costs.device)
    pad[: local_costs.numel()] = local_costs
    out = torch.empty(world * mx, dtype=local_costs.dtype, device=local_costs.device)
    dist.all_gather_into_tensor(out, pad)
    if all(n == mx for n in sizes):
        return out
    keep = torch.as_tensor([r * mx + k for r in range(world) for k in range(sizes[r])], device=out.device)  # built from host-known sizes
    return out.index_select(0, keep)
