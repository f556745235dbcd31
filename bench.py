#!/usr/bin/env python3
"""bench.py — CHOMP iterations/sec over batched scenes on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its N ranks itself, as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "100 generated table-top scenes batched, 30 waypoints, 1 MI355X with A^-1 covariant
update on-device", with configs[1]'s 64-goal goal-set batch evaluated every iteration as the planner does,
omg/planner.py:612-621): per GPU 100 synthetic table-top scenes (4 YCB-like 64^3 SDFs + a 128x96x32 table slab each,
private copies per scene), Panda 9-dof, 30 waypoints, 15 collision points per link, 64 goal candidates per scene.

One "step" = one planner-loop iteration for every scene of the rank, two launches on one stream:
    omgx_goalset_cost_layer    Learner.cost_vector's obstacle batch (S x 64 goals x 30 interpolated waypoints; window pinned at
                               the full 30 waypoints = the most expensive iteration) + the SDF layer of the current trajectories
    omgx_goal_update_optimize  Learner.update_goal (mirror descent `MD`, the reference default) + Optimizer.optimize
value = (scenes on all ranks) x K / (max over ranks of the timed region) in scene-iterations per second.

Scaling: scenes are independent, no data-path collective; one all-gather of the final per-scene costs is inside the timed
region.  Default (weak): every GPU holds `--scenes` scenes.  `--total-scenes T` (strong, BASELINE configs[3] with
`--goals 128`): T scenes in all, rank r plans the contiguous block shard_range(T, r, N); whole scenes per GPU, the
(scene, goal) items of a GPU are dealt to its 8 XCDs by work (ChompEngine.build_schedule).

Prints ONE JSON line (rank 0) with the driver's contract keys + "roofline", "cpu_baseline" and "parity_sample".
"""
from __future__ import annotations

import argparse
import copy
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# the roofline's constants and its one formula live in tools/roofline.py (bound: VALU instruction issue, priced by tools/valu_peak.hip)


def build_workload(num_scenes, num_goals, n, grid, seed0, share_grids, num_objects=4, timing=None, device=None):
    """timing: optional dict that receives the time of the scene table's build (object records + SDF pool + the fitted influence
    regions: the set-up a batch pays once, outside every timed region).  device: build the table ON that device
    (ops.DeviceScenes.from_scenes: one copy per volume, every region fitted by omgx_fit_influence_regions) and return the
    DeviceScenes in place of the host SceneBatch; None: scenes.pack_table on the host (the specification)."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    cfg = Config(timesteps=n, use_standoff=False)  # omg.core -exp sets use_standoff=False (core.py:873)
    model = rb.PandaModel(seed=0)
    scenes = [sc.make_tabletop_scene(seed0 + s, num_objects=num_objects, grid=grid) for s in range(num_scenes)]
    if not share_grids:  # every scene owns private SDF volumes, like the reference's per-scene sdf_torch
        for scn in scenes:
            for ob in scn.objects:
                ob.sdf = sc.SdfGrid(ob.sdf.data.copy(), ob.sdf.origin, ob.sdf.delta)
    t0 = time.perf_counter()
    if device is not None:
        import torch
        from omg_planner_amd import ops
        parts = {}
        batch = ops.DeviceScenes.from_scenes(scenes, cfg.layer_kwargs(), device, share_grids=share_grids, timing=parts)
        torch.cuda.synchronize(device)
        if timing is not None:
            timing["scene_table"] = dict(parts, built="on the device: DeviceScenes.from_scenes (records on the host, one copy per volume, omgx_fit_influence_regions)")
    else:
        batch = sc.pack_table(scenes, cfg.layer_kwargs(), ragged=True, share_grids=share_grids)
    if timing is not None:
        timing["pack_table_ms"] = (time.perf_counter() - t0) * 1e3
    start = np.tile(rb.HOME_CONFIG, (num_scenes, 1))
    # grasp-like goal sets: the hand ends 10-16 cm from the target, approach axis towards it
    goals = np.stack([sc.make_reach_goals(scenes[s], model, num_goals, seed0 + s) for s in range(num_scenes)])
    return cfg, model, batch, start, goals


class _stdout_to_stderr:
    """While RCCL comes up: its version banner goes to the C library's stdout (and is flushed whenever that buffer is) — the driver reads
    ONE JSON line from this process's stdout.  File descriptor 1 points at stderr for the duration; the C buffers are flushed before it
    comes back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            self._libc.fflush(None)
        finally:
            sys.stdout.flush()
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        f = Path("/sys/fs/cgroup/cpu.max")
        if f.exists():
            q, per = f.read_text().split()[:2]
            return None if q == "max" else float(q) / float(per)
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(cfg, model, batch, start, goals, n, budget_s=10.0):
    """The oracle (CPU port) timed on a bounded sample of the same step with 1, 16, 64 and all hardware threads (and the
    container's CPU quota, if it has one): `value` is the BEST of them, with the thread count that gave it and its efficiency
    against the single-thread figure.  The step's work is the 64-goal goal-set cost, an OpenMP loop over (scene, goal) items
    with schedule(dynamic, 1) — embarrassingly parallel (tools/cpu_scaling_probe.py: 72x on 128 threads in a 2 ms burst); what
    limits a sustained run on the GPU boxes of this pool is the cgroup quota (cpu.max = 16 CPUs), which throttles any larger
    team.  A reported baseline, not the target (see `roofline`)."""
    from oracle import oracle as orc
    from omg_planner_amd import scenes as sc
    cores = os.cpu_count() or 1
    quota = cpu_quota()
    P = model.points_per_link
    blob = model.blob()

    def one_step(k):
        e = int(batch.scene_begin[k])
        b = sc.SceneBatch(batch.objects[:e], batch.scene_begin[: k + 1], batch.pool)
        traj = np.stack([sc.cubic_init(start[s], goals[s, 0], n) for s in range(k)])
        t0 = time.perf_counter()
        cost, _ = orc.goalset_cost(blob, P, b, traj[:, 0], goals[:k], n, cfg.time_interval)
        gi = cost.argmin(-1)
        end = goals[np.arange(k), gi]
        pot, pg, col = orc.fk_sdf(blob, P, b, traj)
        prm = orc.ChompParams()
        prm.n_waypoints, prm.n_points, prm.top_k, prm.goal_set_proj, prm.constraint_num = n, P, cfg.top_k_collision, 1, 1
        prm.joint_limit_max_steps, prm.allow_collision_point, prm.pre_terminate, prm.do_update = 10, 5, 1, 1
        prm.time_interval, prm.obstacle_weight, prm.smoothness_weight, prm.step_size = cfg.time_interval, 1.0, 0.102, 0.1
        prm.clip_grad_scale, prm.terminate_smooth_loss = 10.0, 35.0
        for d in range(9):
            prm.link_smooth_weight[d] = 1.0
        orc.chomp_optimize(blob, prm, traj, start[:k], end, end[:, None], end, pot, pg, col)
        return time.perf_counter() - t0

    def timed(threads, budget):
        orc.set_threads(threads)
        k = min(2 if threads == 1 else 4, len(start))
        per_scene = one_step(k) / k  # pilot
        k2 = int(max(1, min(len(start), budget / max(per_scene, 1e-6))))
        total, reps = 0.0, 0
        while reps < 1 or (total < 0.8 * budget and reps < 8):  # sustained: a burst would not see the quota's throttling
            total += one_step(k2)
            reps += 1
        return k2 / (total / reps), k2, reps

    teams = sorted({1, min(16, cores), min(64, cores), cores} | ({max(1, min(cores, int(quota)))} if quota else set()))
    share = budget_s / (len(teams) + 1)
    v_one, k_one, r_one = timed(1, 2.0 * share)
    tried = {1: v_one}
    best = (v_one, 1, k_one, r_one)
    for th in teams[1:]:
        v, k, r = timed(th, share)
        tried[th] = v
        if v > best[0]:
            best = (v, th, k, r)
    orc.set_threads(cores)
    v_best, th_best, k_best, r_best = best
    return {"value": v_best, "unit": "iterations/s", "cores": th_best, "kind": "port",
            "sample": f"{k_best} scenes x 1 planner iteration (64-goal goal-set cost + optimize step), mean of {r_best} run(s), "
                      f"oracle/omg_oracle.c with OpenMP on {th_best} threads (the best of the teams tried)",
            "threads_tried": {str(k): v for k, v in tried.items()},
            "efficiency_vs_single_thread": v_best / (v_one * th_best),
            "host": {"hardware_threads": cores, "cgroup_cpu_quota": quota,
                     "note": "the container's cgroup quota caps sustained CPU time; teams larger than the quota are throttled" if quota else "no cgroup quota"},
            "single_thread": {"value": v_one, "unit": "iterations/s", "cores": 1,
                              "sample": f"{k_one} scene(s) x 1 planner iteration, mean of {r_one} run(s), same code on 1 thread"},
            "reference_python": "the reference's own numpy path (omg/cost.py, omg/optimizer.py, omg/online_learner.py) cannot travel to the GPU "
                                "box; timed in the build container by tools/time_reference_cpu.py: BASELINE.md section 3.1"}


def shape_step_timing(dev, ol_alg, scenes, goals_n, n=30, objects=4, layout_scenes=None, steps=100, regions=3):
    """ms per step of ANOTHER shape on this GPU, laid out by ChompEngine.layout like a rank that owns it would be, with the roofline
    block of its own goal-set launches (HIP events on the dispatch, per-launch counts from the profiled workload of that shape:
    tools/roofline.py).  -> (ms_per_step, layout, roofline block)."""
    import ctypes as C
    import torch
    from omg_planner_amd import _lib
    from omg_planner_amd.engine import ChompEngine
    from tools.roofline import roofline_block
    cfg, model, batch, start, goals = build_workload(scenes, goals_n, n, 64, seed0=0, share_grids=False, num_objects=objects, device=dev)
    eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=scenes if layout_scenes is None else layout_scenes,
                           device=dev, ol_alg=ol_alg)
    eng.pose_hand_over(True)
    snap = eng.snapshot()
    count = [0]

    def step():
        if count[0] and count[0] % cfg.optim_steps == 0:
            eng.restore(snap)
        count[0] += 1
        eng.t = 0
        eng.iterate(0)

    for _ in range(20):
        step()
    lib = _lib.lib()
    parts = 1 if eng.latency else max(1, int(eng.pipeline or 1))
    lib.omgx_timing_enable(5 if steps * parts >= 25 else 1)
    out = []
    for _ in range(regions):
        eng.join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.join()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    buf, kinds = (C.c_float * 4096)(), (C.c_int32 * 4096)()
    nrec = lib.omgx_timing_collect(buf, kinds, 4096)
    lib.omgx_timing_enable(0)
    goal_ms = np.array([buf[i] for i in range(nrec) if kinds[i] == 0], dtype=np.float64)
    ms = sorted(out)[len(out) // 2]
    P = model.points_per_link
    alg_bytes = (scenes * goals_n * n * 10 * P + scenes * n * 10 * P) / parts * (32 + 128 * (objects + 1))
    roof = roofline_block(ROOT / "profiles" / "roofline_inputs.json", float(goal_ms.mean()) if len(goal_ms) else ms, int(len(goal_ms)),
                          5 if steps * parts >= 25 else 1, alg_bytes,
                          {"scenes": scenes, "goals": goals_n, "waypoints": n, "points_per_link": P, "grid": 64, "pipeline": parts, "objects": objects + 1},
                          launches_per_step=parts, ms_per_step=ms)
    roof.pop("note", None)
    return ms, dict(eng.layout_used, pipeline=parts), roof


def shape_plan_timing(dev, ol_alg, scenes, goals_n, n=30, objects=4, reps=5):
    """ms per whole plan (cfg.optim_steps goal-selecting + cfg.extra_smooth_steps fixed-goal iterations, no early stop) of ANOTHER shape on
    this GPU, laid out by ChompEngine.layout; best of `reps` from the same fresh state (device-to-device restore outside the timing)."""
    import torch
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = build_workload(scenes, goals_n, n, 64, seed0=0, share_grids=False, num_objects=objects, device=dev)
    eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=scenes, device=dev, ol_alg=ol_alg)
    snap = eng.snapshot()
    best = float("inf")
    for _ in range(reps + 1):  # (the first pass warms the schedule measurement and the prepared calls)
        eng.restore(snap)
        eng.join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.plan(early_stop=False)
        eng.join()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def rank_share_config4(dev, ol_alg, steps=100, regions=3):
    """ms per step of ONE GPU's share of BASELINE config 4 on 8 GPUs (100 scenes x 128 goals -> 13 scenes x 128 goals), laid out
    by ChompEngine.layout like any rank of that job would be: what the 8-GPU strong-scaling ceiling hangs on."""
    return shape_step_timing(dev, ol_alg, 13, 128, layout_scenes=13, steps=steps, regions=regions)


def config5_shape(dev, ol_alg, steps=100, regions=3):
    """BASELINE config 5's shape on one GPU: 16 cluttered scenes (12 obstacles + the table), 50 waypoints, 64 goals."""
    return shape_step_timing(dev, ol_alg, 16, 64, n=50, objects=12, steps=steps, regions=regions)


def strong_scaling_estimate(dev, ol_alg, share8=None, steps=60, regions=3):
    """BASELINE config 4 (100 scenes x 128 goals) on 1 / 2 / 4 / 8 GPUs, ESTIMATED from this one GPU: a rank of an N-GPU job owns
    ceil(100 / N) whole scenes and exchanges nothing per iteration (engine.shard_range), so the job's step is the
    step of its largest shard — timed here shard by shard (100, 50, 25, 13 scenes x 128 goals), each laid out by ChompEngine.layout like
    the rank that would own it, each with its roofline block.  speedup = ms(100 scenes) / ms(shard): what N GPUs could give at best
    (the collective and the ranks' skew come on top: `collective`).  An estimate from 1-GPU runs, NOT a measured scaling curve."""
    out = {}
    for n_gpus, scenes in ((1, 100), (2, 50), (4, 25), (8, 13)):
        ms, lay, roof = share8 if (n_gpus == 8 and share8 is not None) else shape_step_timing(dev, ol_alg, scenes, 128, layout_scenes=scenes, steps=steps, regions=regions)
        out[str(n_gpus)] = {"scenes_per_gpu": scenes, "ms_per_step": ms, "layout": lay, "frac": roof.get("frac"), "avg_launch_ms": roof.get("avg_launch_ms")}
    base = out["1"]["ms_per_step"]
    for k, v in out.items():
        v["speedup"] = base / v["ms_per_step"]
        v["efficiency"] = v["speedup"] / int(k)
    out["what"] = "estimate: step of the largest shard of 100 scenes x 128 goals, timed on ONE GPU; no N-GPU node has run this"
    return out


def collective_cost(dev, num_scenes, reps=50):
    """What the job's ONE collective costs per call: the all-gather of the final per-scene costs (engine.gather_costs) over RCCL with
    one rank — the floor of its latency on this box (kernel launch + RCCL's own bookkeeping; over xGMI with N ranks the ring adds
    N - 1 hops of a few microseconds for these <= 51 KB).  Weak scaling exchanges nothing else: this cost is paid once per timed
    region / plan, not per step."""
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return None
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    try:
        from omg_planner_amd.engine import gather_costs_equal
        x = torch.zeros(num_scenes, dtype=torch.float64, device=dev)
        with _stdout_to_stderr():
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
            for _ in range(5):
                gather_costs_equal(x, 1)
            torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            gather_costs_equal(x, 1)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        return {"all_gather_us_one_rank": ts[len(ts) // 2], "min_us": ts[0], "bytes": num_scenes * 8, "backend": "nccl (RCCL), world_size 1",
                "per": "timed region / plan (one call closes the job), not per step"}
    except Exception as e:  # a box without RCCL support for a one-rank group: say so instead of failing the bench
        return {"error": repr(e)[:200]}
    finally:
        with _stdout_to_stderr():
            if dist.is_initialized():
                dist.destroy_process_group()


def persistent_launch_timing(dev, cfg, model, batch, start, goals, ol_alg, steps=50, regions=4):
    """The same step as ONE persistent launch per block of iterations (ChompEngine.run_persistent, omgx_plan_persistent: workgroups
    claim items, a scene's last item runs its learner and step and activates the scene's next iteration — DESIGN.md section 4.7)
    against the launches per iteration of the timed region, same engine layout rule, same pinned window: ms per step of both and whether
    every bit agrees.  Measured here so that the comparison is the bench line's, not a side script's."""
    import torch
    from omg_planner_amd.engine import ChompEngine
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=ol_alg)
    if not eng.persistent_ok() or ol_alg not in ChompEngine.PERSISTENT_ALGS:
        return None
    eng.pose_hand_over(True)
    snap = eng.snapshot()
    ref = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=ol_alg)
    ref.pose_hand_over(True)
    rsnap = ref.snapshot()

    def run(persistent):
        e, sn = (eng, snap) if persistent else (ref, rsnap)
        e.restore(sn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if persistent:
            e.run_persistent([0] * steps, pin_window=True)
        else:
            for _ in range(steps):
                e.t = 0
                e.iterate(0)
            e.join()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    run(True), run(False)
    a = sorted(run(True) for _ in range(regions))
    b = sorted(run(False) for _ in range(regions))
    same = all(bool(torch.equal(getattr(eng, k), getattr(ref, k))) for k in ("traj", "info", "goal_idx", "learner_state", "goal_cost", "cost_traj"))
    return {"ms_per_step_persistent": a[len(a) // 2], "ms_per_step_launches": b[len(b) // 2], "steps_per_launch": steps, "bits_equal": same,
            "status": eng.persistent_status(), "note": "one launch for `steps_per_launch` iterations of all scenes; not the path the timed region uses"}


def scene_update_timing(dev, cfg, model, batch, start, goals, ol_alg):
    """One perception frame's scene change on the device, for ONE scene of the workload (include/omg_hip.h section 8): two objects
    moved (48-byte writes), one obstacle's volume replaced by a fresh point-cloud SDF (4096 points -> a ~45^3 grid written
    straight into the pool, omg/core.py:426-457) with its influence region fitted on the device — wall time incl. the final
    synchronisation — and, beside it, what the same change costs through the host (volume back to the host + scenes.pack_table's
    fit).  Then the SAME engine plans the changed scene; its result against an engine packed afresh is part of the line."""
    import torch
    from omg_planner_amd import ops, scenes as sc
    from omg_planner_amd.engine import ChompEngine
    one = batch.subset(0, 1)
    ds = ops.DeviceScenes(one, dev, reserve_voxels=400_000)
    eng = ChompEngine.auto(model, ds, copy.deepcopy(cfg), start[:1], goals[:1], device=dev, ol_alg=ol_alg)
    fresh = eng.snapshot()
    eng.plan(early_stop=False)
    rng = np.random.RandomState(7)
    cloud = rng.uniform([0.35, -0.2, 0.05], [0.65, 0.2, 0.35], size=(4096, 3))
    pts = torch.as_tensor(cloud, dtype=torch.float64, device=dev)
    lo, hi = cloud.min(0) - 0.24, cloud.max(0) + 0.24
    shape = tuple(len(np.arange(lo[a], hi[a], 0.02)) for a in range(3))
    best = float("inf")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.set_object_pose(0, 0, sc._yaw_pose(0.5, 0.1 + 0.01 * rep, 0.15, 0.3))
        ds.set_object_pose(0, 1, sc._yaw_pose(0.6, -0.15, 0.16, -0.8))
        slot = ds.grid_slot(0, 2, shape)
        grid, origin, res = ops.point_cloud_sdf(pts, 0.02, 0.24, out=slot)
        ds.replace_grid(0, 2, grid, origin, res, fit="device")
        ds.set_object_pose(0, 2, np.eye(4))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    g_host = grid.cpu().numpy()
    obj = sc.SceneObject("cloud", np.eye(4), sc.SdfGrid(g_host, origin, res))
    sc.pack_table([sc.Scene([obj], 0)], cfg.layer_kwargs())
    host_ms = (time.perf_counter() - t0) * 1e3
    eng.restore(fresh)
    info = eng.plan(early_stop=False)
    torch.cuda.synchronize()
    hb = ds.host_batch()
    eng2 = ChompEngine.auto(model, sc.SceneBatch(hb.objects, hb.scene_begin, hb.pool), copy.deepcopy(cfg), start[:1], goals[:1], device=dev, ol_alg=ol_alg)
    info2 = eng2.plan(early_stop=False)
    torch.cuda.synchronize()
    same = bool(torch.equal(eng.traj, eng2.traj) and torch.equal(info, info2))
    return {"scene_update_ms": best, "what": f"2 poses + a {shape[0]}x{shape[1]}x{shape[2]} point-cloud SDF of 4096 points built, placed and region-fitted on the device (best of 3, incl. synchronize)",
            "same_change_through_the_host_ms": host_ms, "replan_equals_fresh_engine": same}


def drop_in_plan_timing(dev, ol_alg="MD", reps=8):
    """ms per plan through the DROP-IN classes (Trajectory / Cost / Learner / Optimizer) on bench scene 0 — the loop of
    Planner.plan (omg/planner.py:612-653) as the reference writes it: update_goal, a look at traj.goal_idx, optimize(force_update),
    a copy of traj.data, a look at info["terminate"]; 50 + 20 iterations (no early exit, like the other plan timings) and the
    closing info-only call.  Best of `reps`, each on a fresh Trajectory / Learner / Optimizer over a warm Cost."""
    import types
    import torch
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.online_learner import Learner
    from omg_planner_amd.optimizer import Optimizer
    from omg_planner_amd.trajectory import Trajectory
    n, G = 30, 64
    model = rb.PandaModel(seed=0)
    scene = sc.make_tabletop_scene(0, grid=64)
    sdf, lim = sc.pack_padded(scene.objects)  # Env.combine_sdfs layout (omg/core.py:366-411)
    goals = sc.make_reach_goals(scene, model, G, 0)
    robot = types.SimpleNamespace(collision_points=model.collision_points, joint_lower_limit=model.joint_lower_limit,
                                  joint_upper_limit=model.joint_upper_limit)
    best, iters, goal_changes = float("inf"), 0, 0
    cost = None
    for rep in range(reps + 1):  # the first pass warms everything up (object table, influence regions, buffers)
        cfg = Config(timesteps=n, use_standoff=False, ol_alg=ol_alg)
        objs = [types.SimpleNamespace(name=o.name, pose_mat=o.pose_mat, attached=False, reach_grasps=goals[:, None, :]) for o in scene.objects]
        if cost is None:
            env = types.SimpleNamespace(robot=robot, objects=objs, target_idx=scene.target_idx, config=cfg,
                                        sdf_torch=torch.as_tensor(sdf, device=dev), sdf_limits=torch.as_tensor(lim, device=dev))
            cost = Cost(env)
        else:
            env.config, env.objects, cost.cfg = cfg, objs, cfg
            cost.target_obj = objs[scene.target_idx]
        traj = Trajectory(cfg=cfg)
        traj.start, traj.goal_set, traj.end = rb.HOME_CONFIG.copy(), goals, goals[0].copy()
        traj.interpolate_waypoints()
        learner = Learner(env, traj, cost)
        optim = Optimizer(types.SimpleNamespace(config=cfg, robot=robot), cost)
        import gc
        gc.collect()  # (the previous pass's garbage is not this pass's time)
        gc_was_on = gc.isenabled()
        gc.disable()  # ... nor is a collection pass in the middle of 71 iterations of 80 us (one run in five read 6.3 instead of 5.8 ms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infos, history, selected = [], [np.copy(traj.data)], []
        for t in range(cfg.optim_steps + cfg.extra_smooth_steps):
            if cfg.goal_set_proj and ol_alg not in ("Baseline", "Proj") and t < cfg.optim_steps:
                learner.update_goal()
                selected.append(traj.goal_idx)
            infos.append(optim.optimize(traj, force_update=True))
            history.append(np.copy(traj.data))
            _ = infos[-1]["terminate"] and t > 0
        infos.append(optim.optimize(traj, info_only=True))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if gc_was_on:
            gc.enable()
        if rep > 0:
            best = min(best, ms)
        iters = len(history) - 1
        sel = [int(g) for g in selected]
        goal_changes = int(sum(1 for a, b in zip(sel[:-1], sel[1:]) if a != b))
    return {"ms_per_plan_drop_in_classes": best, "iterations": iters, "goal_changes": goal_changes, "final_cost": float(infos[-1]["cost"]),
            "what": "Planner.plan's loop (omg/planner.py:612-653) written with the drop-in Trajectory / Cost / Learner / Optimizer on bench scene 0 "
                    f"(64 goals, {ol_alg}), {iters} iterations + the closing info-only call, best of {reps} on a warm Cost"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=100, help="scenes per GPU (weak scaling)")
    ap.add_argument("--total-scenes", type=int, default=0, help="scenes in all, sharded over the GPUs (strong scaling)")
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--objects", type=int, default=4, help="obstacles per scene besides the table (BASELINE config 5's clutter: 12 with --waypoints 50)")
    ap.add_argument("--share-grids", action="store_true", help="store identical SDF volumes once (model library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-pack", action="store_true", help="build the scene table with scenes.pack_table on the host (round 1-4) instead of on the device")
    ap.add_argument("--pipeline", type=int, default=0, help="parts of the engine's software pipeline (0: ChompEngine.layout's choice)")
    ap.add_argument("--layout-scenes", type=int, default=0, help="scene count ChompEngine.layout is evaluated for (0: the largest shard, ceil(total / world)); "
                    "a single-process run given a multi-rank job's number computes the same bits as that job")
    ap.add_argument("--goal-parts", type=int, default=0, help="workgroups per goal in the batch layout (0: the layout rule's choice)")
    ap.add_argument("--regions", type=int, default=0, help="timed regions of --steps steps each (0: 1, or 5 when a region is shorter than 50 ms); ms_per_step is their median")
    ap.add_argument("--no-plan", action="store_true", help="skip timing a full 70-iteration plan (ms_per_plan)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of three scenes after the timed region")
    ap.add_argument("--ol-alg", default="MD", help="goal-selection rule (reference default: MD, omg/config.py:67)")
    ap.add_argument("--dump-costs", default=None, help="rank 0 writes the gathered final per-scene costs to this .npy file (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed plainly: start the N ranks as a CHILD process group (torch.distributed.run, one rank per
        # GPU) before this process has imported torch or touched the GPU — never os.exec*: replacing a process that has initialised
        # HIP takes the machine down on this pool — relay its output (rank 0 prints the JSON line) and leave with its exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE = {world}: one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # one rank per GPU; OMGX_BENCH_BACKEND=gloo lets several ranks share one GPU for a functional test of this path
    backend = os.environ.get("OMGX_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # Launched by torch.distributed.run (RANK in the environment) the process group is created even for ONE rank, so that a
    # one-GPU box runs the same RCCL initialisation and the same all-gather on device tensors as a node (tests/test_gpu_round3.py)
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()  # the communicator (and RCCL's banner) comes up HERE, not inside the first timed collective
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    from omg_planner_amd import _lib
    from omg_planner_amd.engine import ChompEngine, gather_costs, gather_costs_equal, shard_range

    G, n = args.goals, args.waypoints
    strong = args.total_scenes > 0
    if strong:  # rank r plans scenes shard_range(T, r, N); scene s is the same scene whatever N is (seed = s)
        mine = shard_range(args.total_scenes, rank, world)
        S, seed0, total_scenes = len(mine), mine.start, args.total_scenes
    else:
        S, seed0, total_scenes = args.scenes, rank * args.scenes, world * args.scenes
    setup = {}
    # the HIP runtime's own first-use costs (context, the staging buffers of the first pageable copy, loading the library's code
    # objects) are not the scene table's: paid here, reported beside it
    t_w = time.perf_counter()
    torch.from_numpy(np.zeros(1 << 18, np.float32)).to(dev)
    ops_warm = torch.zeros((2, 3), dtype=torch.float64, device=dev)
    from omg_planner_amd import ops as _ops
    _ops.point_cloud_sdf(ops_warm, 0.5, 0.1)
    torch.cuda.synchronize()
    setup["runtime_warmup_ms"] = (time.perf_counter() - t_w) * 1e3
    cfg, model, batch, start, goals = build_workload(S, G, n, args.grid, seed0=seed0, share_grids=args.share_grids, num_objects=args.objects, timing=setup,
                                                     device=None if args.host_pack else dev)
    scenes_dev = None if args.host_pack else batch
    _host = [batch if args.host_pack else None]

    def host_batch():  # the oracle's legs (cpu_baseline, parity_sample) and the one-scene engines read the table from the host
        if _host[0] is None:
            from omg_planner_amd import scenes as sc_
            hb = scenes_dev.host_batch()
            _host[0] = sc_.SceneBatch(hb.objects, hb.scene_begin, hb.pool)
        return _host[0]
    # The layout — latency mode, split goals, pipeline parts — follows ONE rule of the shape (ChompEngine.layout), evaluated on every
    # rank for the same scene count (the largest shard), so that all shards of a job compute comparable bits
    S_max = len(shard_range(args.total_scenes, 0, world)) if strong else S
    layout_scenes = args.layout_scenes if args.layout_scenes > 0 else S_max
    t_init = time.perf_counter()
    eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=layout_scenes, device=dev, ol_alg=args.ol_alg)
    if args.goal_parts > 0 and not eng.latency and args.goal_parts != eng.goal_parts:
        eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=args.ol_alg, goal_parts=args.goal_parts)
        eng.pipeline = ChompEngine.layout(layout_scenes, G, n)["pipeline"]
        eng.layout_used = {"latency_mode": False, "goal_parts": args.goal_parts, "pipeline": eng.pipeline}
    torch.cuda.synchronize()
    setup["engine_init_ms"] = (time.perf_counter() - t_init) * 1e3  # uploads (records, SDF pool, goals) + allocations
    # the engine's software pipeline: the rank's scenes as independent parts on their own streams — their launches run concurrently
    # and pack ramp-ups and tails into less time than one launch after the other; same results bit for bit
    if args.pipeline > 0 and not eng.latency:
        eng.pipeline = min(args.pipeline, S)
    parts = 1 if eng.latency else max(1, int(eng.pipeline or 1))
    layout = dict(eng.layout_used, pipeline=parts, evaluated_for_scenes=layout_scenes)
    # the step is an iteration INSIDE a plan: like plan() itself, the launches hand link poses to each other (the start's and the
    # goals' poses tabulated once, the waypoints' poses left by the layer workgroups) instead of running the same kinematics again
    eng.pose_hand_over(True)
    lib = _lib.lib()

    # The workload must not drift with the number of steps: a trajectory that has been optimised for hundreds of iterations
    # has left the obstacles' influence regions (the kernel's culling then retires nearly every pair) and Optimizer.update's
    # schedule grows without bound (1.02^k).  A plan runs cfg.optim_steps = 50 goal-selecting iterations, so every 50 steps the
    # engine goes back to the fresh plan (device-to-device copies inside the timed region, no host sync): each block of 50
    # steps is the first 50 iterations of a plan, whatever --steps is.
    snap = eng.snapshot()
    count = [0]

    def step():
        if count[0] and count[0] % cfg.optim_steps == 0:
            eng.restore(snap)
        count[0] += 1
        eng.t = 0  # pin the goal-set window at the full n waypoints (first-iteration workload)
        eng.iterate(0)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def gather(c):  # the job's one collective: sizes follow from shard_range on every rank, nothing else is exchanged
        c = c if backend == "nccl" else c.cpu()  # RCCL all-gather over xGMI on device tensors; host tensors under the gloo test backend
        return gather_costs(c, world, total_scenes) if strong else gather_costs_equal(c, world)

    for _ in range(args.warmup):
        step()
    _ = gather(eng.final_costs())  # the epilogue's kernels / collective are loaded too
    barrier()
    # every 5th launch of the dominant kernel is bracketed by HIP events attached to the dispatch (all of them would cost
    # 1.7 % of the step time, a fifth 0.3 %); an odd stride, so that the samples alternate between the pipeline's parts
    stride = 5 if args.steps * parts >= 25 else 1  # a very short run still gets its launches timed
    lib.omgx_timing_enable(stride)

    def region():
        """EXACTLY --steps steps + the job's one collective, bracketed by barrier + synchronize on both sides; -> (this rank's
        seconds, MAX over ranks)."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        costs = eng.final_costs()  # enqueued behind the last step: no host sync before the collective
        allc = gather(costs)
        assert allc.numel() == total_scenes
        barrier()
        el = time.perf_counter() - t0
        mx = el
        if dist_on:
            import torch.distributed as dist
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mx = float(tmax.item())
        return el, mx, allc

    # A region of the driver's size (20 steps of ~0.2 ms) is a few milliseconds: one such interval says little — and the first ones
    # run while the GPU's clocks are still coming up (the goal-set launches of five successive 20-step regions: 179, 174, 173, 168,
    # 166 us in the median; tools/trace_regions.py).  When the first region is shorter than 50 ms, more regions of the same --steps
    # are timed — as many as it takes to time ~60 ms, between 5 and 15 — and ms_per_step is the MEDIAN region (the spread is
    # reported beside it); `steps` stays what was asked for.
    first = region()
    n_regions = args.regions if args.regions > 0 else (max(5, min(15, int(np.ceil(0.060 / max(first[1], 1e-6))))) if first[1] < 0.050 else 1)
    # (first[1] is the MAX over the ranks: every rank computes the same number)
    regs = [first] + [region() for _ in range(n_regions - 1)]
    order = sorted(range(len(regs)), key=lambda i: regs[i][1])
    mid = order[len(order) // 2]
    elapsed_local, elapsed, allc = regs[mid]
    region_ms = [r[1] / args.steps * 1e3 for r in regs]
    if args.dump_costs and rank == 0:
        np.save(args.dump_costs, regs[0][2].cpu().numpy())  # the FIRST region's costs: which region is the median depends on the clock
    buf = (C.c_float * 4096)()
    kinds = (C.c_int32 * 4096)()
    nrec = lib.omgx_timing_collect(buf, kinds, 4096)
    lib.omgx_timing_enable(0)

    # every rank's own roofline block: its launch durations (HIP events), its scenes, its wall time — gathered so that an N-GPU
    # line carries the achieved rates per GPU (north_star: "achieved ... vs roofline reported at 1/2/4/8")
    durs = np.array([buf[i] for i in range(nrec)], dtype=np.float64)
    kind = np.array([kinds[i] for i in range(nrec)])
    goal_ms = durs[kind == 0]  # the goal-set launch (goal-set batch + trajectory layer) = the dominant kernel
    avg_ms = float(goal_ms.mean()) if len(goal_ms) else elapsed_local / args.steps * 1e3  # no goal-set launch recorded (rules without a goal-set batch)
    P = model.points_per_link
    O_active = args.objects + 1
    # N (32 + 128 O_active) algorithmic bytes for the N points of one launch — the goal-set batch
    # plus the S x n x 150 points of the trajectory layer.  NOT a measure of what the kernel moves: 85 % of the (point,
    # object) pairs retire in registers before any load and the rest hit L2.
    pts_per_launch = (S * G * n * 10 * P + S * n * 10 * P) / parts  # a launch handles one part of the pipeline
    alg_bytes = pts_per_launch * (32 + 128 * O_active)
    from tools.roofline import roofline_block
    roof = roofline_block(ROOT / "profiles" / "roofline_inputs.json", avg_ms, int(len(goal_ms)), stride, alg_bytes,
                          {"scenes": S, "goals": G, "waypoints": n, "points_per_link": P, "grid": args.grid, "pipeline": parts, "objects": O_active},
                          launches_per_step=parts, ms_per_step=elapsed_local / args.steps * 1e3)
    per_rank = None
    if dist_on:
        import torch.distributed as dist
        mine = {"rank": rank, "scenes": S, "ms_per_step": elapsed_local / args.steps * 1e3, "avg_launch_ms": avg_ms, "launches": int(len(goal_ms)),
                "achieved": roof["achieved"], "frac": roof["frac"], "hbm_GBs": None if roof["hbm_real"] is None else roof["hbm_real"]["GBs"],
                "algorithmic_equiv_GBs": roof["algorithmic_equiv_GBs"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    parity = None
    if not args.no_parity and rank == 0:
        # the timed workload against the oracle: three scenes of this rank, the first three iterations of the plan the timed
        # region repeats (same pinned window), trajectories / final costs / chosen goals
        from oracle.check import engine_vs_oracle
        eng.restore(snap)
        parity = engine_vs_oracle(eng, host_batch(), sorted({0, S // 2, S - 1}), steps=3, pin_window=True)
        # the same sample against the oracle in SOPHUS MODE (the reference kernel's quaternion round trip, which the product path does
        # not restate): the one knowing deviation as a number on the timed workload; the bar is north_star's 1e-4
        eng.restore(snap)
        sm = engine_vs_oracle(eng, host_batch(), sorted({0, S // 2, S - 1}), steps=3, pin_window=True, sophus=True)
        parity["sophus_mode"] = {k: sm[k] for k in ("max_traj_err", "max_cost_rel_err", "goal_idx_equal", "ok")}
        parity["ok"] = bool(parity["ok"] and sm["ok"])

    ms_per_plan = ms_plan_early = ms_single = ms_single_batch_layout = terminated = ms_graph_early = ms_graph_single = None
    share4 = scene_upd = drop_in = cfg5 = scaling = coll = persist = plans_other = None
    if not args.no_plan and rank == 0 and world == 1:
        share4 = rank_share_config4(dev, args.ol_alg)
        cfg5 = config5_shape(dev, args.ol_alg)
        plans_other = (shape_plan_timing(dev, args.ol_alg, 13, 128), shape_plan_timing(dev, args.ol_alg, 16, 64, n=50, objects=12))
        scaling = strong_scaling_estimate(dev, args.ol_alg, share8=share4)
        persist = persistent_launch_timing(dev, cfg, model, batch, start, goals, args.ol_alg)
        scene_upd = scene_update_timing(dev, cfg, model, host_batch(), start, goals, args.ol_alg)
        drop_in = drop_in_plan_timing(dev, args.ol_alg)
    if not args.no_plan and rank == 0:
        ms_per_plan = float("inf")
        for _ in range(2):  # best of 2: the first plan pays one-off costs (code-object load of the 30 window sizes)
            eng2 = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=layout_scenes, device=dev, ol_alg=args.ol_alg)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            eng2.plan(early_stop=False)
            torch.cuda.synchronize()
            ms_per_plan = min(ms_per_plan, (time.perf_counter() - tp) * 1e3)
            del eng2
        # the same plan as the reference runs it: a scene that terminates leaves the loop (planner.py:626) — its goal-set
        # batch, goal update and step are skipped from then on (active mask)
        eng3 = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=layout_scenes, device=dev, ol_alg=args.ol_alg)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        eng3.plan(early_stop=True)
        torch.cuda.synchronize()
        ms_plan_early = (time.perf_counter() - tp) * 1e3
        terminated = int((eng3.active == 0).sum().item())
        del eng3
        one = host_batch().subset(0, 1)  # BASELINE configs[0]/[1] shape: ONE scene, 64 goals — latency of a whole plan
        # latency mode (ChompEngine(latency_mode=True): the launches cut into many small workgroups over the whole chip) and, for
        # comparison, the batch layout a 100-scene run uses (one workgroup per goal on the scene's XCD)
        ms_single, ms_single_batch_layout = float("inf"), float("inf")
        for lat in (True, False):
            # one engine per mode, planned again and again from the same fresh state (restore(): device-to-device copies, untimed) —
            # the first plan pays the engine's one-off host work (argument lists checked and prepared once) and is not timed
            e1 = ChompEngine(model, one, copy.deepcopy(cfg), start[:1], goals[:1], device=dev, ol_alg=args.ol_alg, latency_mode=lat)
            fresh1 = e1.snapshot()
            e1.plan(early_stop=False)
            for _ in range(3):
                e1.restore(fresh1)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                e1.plan(early_stop=False)
                torch.cuda.synchronize()
                dt_ = (time.perf_counter() - tp) * 1e3
                if lat:
                    ms_single = min(ms_single, dt_)
                else:
                    ms_single_batch_layout = min(ms_single_batch_layout, dt_)
        # the same plans as ONE HIP graph each (ChompEngine.capture_plan: no host in the loop), replayed from the fresh state
        def graph_ms(e):
            fresh = e.snapshot()
            pg = e.capture_plan(early_stop=True)
            best = float("inf")
            for _ in range(3):
                e.restore(fresh)
                torch.cuda.synchronize()
                tp_ = time.perf_counter()
                pg.replay()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - tp_) * 1e3)
            return best
        ms_graph_early = graph_ms(ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=layout_scenes, device=dev, ol_alg=args.ol_alg))
        ms_graph_single = graph_ms(ChompEngine(model, one, copy.deepcopy(cfg), start[:1], goals[:1], device=dev, ol_alg=args.ol_alg, latency_mode=True))

    if rank == 0:
        if per_rank is not None:
            roof["per_rank"] = per_rank
            roof["all_ranks"] = {"achieved": None if any(r["achieved"] is None for r in per_rank) else sum(r["achieved"] for r in per_rank),
                                 "hbm_GBs": None if any(r["hbm_GBs"] is None for r in per_rank) else sum(r["hbm_GBs"] for r in per_rank),
                                 "note": "sums over the ranks' own figures (each GPU against its own peak: per_rank[i].frac)"}
        out = {
            "metric": "CHOMP iterations/sec (batched scenes) + ms/plan, Panda 7-DoF 30-wp",  # BASELINE.json's metric: `value` is the first half, ms_per_plan* the second
            "value": total_scenes * args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_spread": [min(region_ms), max(region_ms)],  # over the timed regions of --steps steps each; ms_per_step / value = the median region
            "regions": len(region_ms),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32 SDF / f64 kinematics+update",
            "data": "synthetic",
            "config": {"workload": (f"{total_scenes} table-top scenes sharded over {world} GPU(s)" if strong else f"{S} table-top scenes/GPU") +
                                   f" x ({G}-goal goal-set cost + CHOMP step), Panda 9-dof, {n} waypoints",
                       "scenes_per_gpu": S, "total_scenes": total_scenes, "goals": G, "waypoints": n, "objects_per_scene": O_active,
                       "sdf_grid": f"{args.objects}x{args.grid}^3 + 128x96x32 per scene, {'shared' if args.share_grids else 'private'}",
                       "goal_selection": f"{args.ol_alg} on device (omgx_goal_update_optimize)", "launches_per_iteration": 2 * parts,
                       "pipeline_parts": parts, "layout": layout,
                       "top_k_collision": cfg.top_k_collision, "plan_restart_every_steps": cfg.optim_steps},
            "roofline": roof,
        }
        if parity is not None:
            out["parity_sample"] = parity
        out["setup_ms"] = dict(setup, what=("pack_table_ms: the scene table (records, SDF pool, fitted influence regions) — " + ("scenes.pack_table on the host" if args.host_pack else "built on the device, see scene_table") +
                                            "; engine_init_ms: uploads + allocations; once per batch, outside the timed regions"))
        if share4 is not None:
            out["ms_per_step_rank_share_config4"] = share4[0]  # 13 scenes x 128 goals on this GPU: one rank's share of BASELINE config 4 on 8 GPUs
            out["rank_share_config4_layout"] = share4[1]
            out["roofline_rank_share_config4"] = share4[2]  # the same block as `roofline`, for that shape's own goal-set launches
        if scaling is not None:
            out["strong_scaling_estimate"] = scaling  # BASELINE config 4 on 1 / 2 / 4 / 8 GPUs from 1-GPU runs of the shards: ms_per_step, speed-up, frac per N
        if persist is not None:
            out["persistent_launch"] = persist
        if plans_other is not None:
            out["ms_per_plan_rank_share_config4"], out["ms_per_plan_config5_shape"] = plans_other  # whole plans of the two other BASELINE shapes (13 x 128; 16 x 64 x 50 waypoints x 13 objects), best of 5
        if cfg5 is not None:
            out["ms_per_step_config5_shape"] = cfg5[0]  # 16 scenes x 64 goals, 50 waypoints, 12 obstacles + table (BASELINE config 5's shape)
            out["config5_shape_layout"] = cfg5[1]
            out["roofline_config5_shape"] = cfg5[2]
        if scene_upd is not None:
            out["scene_update_ms"] = scene_upd["scene_update_ms"]
            out["scene_update"] = scene_upd
        if drop_in is not None:
            out["ms_per_plan_drop_in_classes"] = drop_in["ms_per_plan_drop_in_classes"]  # the reference's own call surface: Learner.update_goal + Optimizer.optimize per iteration
            out["drop_in_plan"] = drop_in
        if ms_per_plan is not None:
            out["plan_timing_version"] = 2  # since round 3: single-scene plans on a WARM engine restored from a snapshot, best of 3 (round 1-2: first plan of a fresh batch-layout engine); round 4: the batch's engines laid out by ChompEngine.layout
            out["ms_per_plan"] = ms_per_plan  # Planner.plan for all scenes of rank 0: initial goal pick + 50 + 20 iterations + final info
            out["ms_per_plan_per_scene"] = ms_per_plan / S
            out["ms_per_plan_early_stop"] = ms_plan_early  # with the reference's break on `terminate` (informational)
            out["scenes_terminated_early"] = terminated
            out["ms_per_plan_single_scene"] = ms_single  # one scene alone in latency mode (latency-bound): a warm engine, best of 3 plans
            out["ms_per_plan_single_scene_batch_layout"] = ms_single_batch_layout  # the same plan with the launches of the batched path
            out["ms_per_plan_early_stop_graph"] = ms_graph_early  # the early-stop plan replayed as one HIP graph (capture_plan)
            out["ms_per_plan_single_scene_graph"] = ms_graph_single
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, host_batch(), start, goals, n)
        if world == 1 and not dist_on and not args.no_plan:
            coll = collective_cost(dev, S)
            if coll is not None:
                out["collective"] = coll  # the job's one collective (final costs): what weak scaling pays on top of the ranks' own steps
        print(json.dumps(out))
    if dist_on:
        import torch.distributed as dist
        sys.stdout.flush()
        with _stdout_to_stderr():
            dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"parity_sample failed: {parity}")


if __name__ == "__main__":
    main()
