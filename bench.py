#!/usr/bin/env python3
"""bench.py — CHOMP iterations/sec over batched scenes on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "100 generated table-top scenes batched, 30 waypoints, 1 MI355X with
A^-1 covariant update on-device", with configs[1]'s 64-goal goal-set batch evaluated every iteration as
the planner does, omg/planner.py:612-621): per GPU 100 synthetic table-top scenes (4 YCB-like 64^3 SDFs
+ a 128x96x32 table slab each, private copies per scene), Panda 9-dof, 30 waypoints, 15 collision
points per link, 64 goal candidates per scene.

One "step" = one planner-loop iteration for every scene of the rank:
    Learner.cost_vector  -> omgx_goalset_cost  (S x 64 goals x 30 interpolated waypoints; window pinned
                                               at the full 30 waypoints = the most expensive iteration)
    Learner.update_goal  -> omgx_goal_update   (cost-vector tail + mirror descent `MD`, the reference default)
    Optimizer.optimize   -> omgx_fk_sdf + omgx_chomp_optimize (loss, gradient, projected A^-1 step, limits)
value = (scenes on all ranks) x K / (max over ranks of the timed region) in scene-iterations per second.
Scenes are independent, so N GPUs hold N x 100 scenes (weak scaling); one all-gather of the final
per-scene costs is inside the timed region.

Prints ONE JSON line (rank 0) with the driver's contract keys + "roofline" and "cpu_baseline".
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def build_workload(num_scenes, num_goals, n, grid, seed0, share_grids):
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    cfg = Config(timesteps=n, use_standoff=False)  # omg.core -exp sets use_standoff=False (core.py:873)
    model = rb.PandaModel(seed=0)
    scenes = [sc.make_tabletop_scene(seed0 + s, grid=grid) for s in range(num_scenes)]
    if not share_grids:  # every scene owns private SDF volumes, like the reference's per-scene sdf_torch
        for scn in scenes:
            for ob in scn.objects:
                ob.sdf = sc.SdfGrid(ob.sdf.data.copy(), ob.sdf.origin, ob.sdf.delta)
    batch = sc.pack_table(scenes, cfg.layer_kwargs(), ragged=True, share_grids=share_grids)
    start = np.tile(rb.HOME_CONFIG, (num_scenes, 1))
    # grasp-like goal sets: the hand ends 10-16 cm from the target, approach axis towards it
    goals = np.stack([sc.make_reach_goals(scenes[s], model, num_goals, seed0 + s) for s in range(num_scenes)])
    return cfg, model, batch, start, goals


def cpu_baseline(cfg, model, batch, start, goals, n, budget_s=15.0):
    """The oracle (CPU port, OpenMP over scenes/goals) timed on a bounded sample of the same step."""
    from oracle import oracle as orc
    from omg_planner_amd import scenes as sc
    from omg_planner_amd._lib import ChompParams  # struct layout only
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    P = model.points_per_link
    blob = model.blob()

    def sub_batch(k):
        e = int(batch.scene_begin[k])
        return sc.SceneBatch(batch.objects[:e], batch.scene_begin[: k + 1], batch.pool)

    def one_step(k):
        b = sub_batch(k)
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

    k = min(4, len(start))
    per_scene = one_step(k) / k  # pilot
    k2 = int(max(k, min(len(start), budget_s / max(per_scene, 1e-6))))
    total, reps = 0.0, 0
    while reps < 1 or (total < 10.0 and reps < 6):  # about 10-30 s of CPU work
        total += one_step(k2)
        reps += 1
    t = total / reps
    return {"value": k2 / t, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{k2} scenes x 1 planner iteration (64-goal goal-set cost + optimize step), mean of {reps} run(s), "
                      f"oracle/omg_oracle.c with OpenMP on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=100, help="scenes per GPU")
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--share-grids", action="store_true", help="store identical SDF volumes once (model library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plan", action="store_true", help="skip timing a full 70-iteration plan (ms_per_plan)")
    ap.add_argument("--streams", type=int, default=1, help="split the rank's scenes over this many HIP streams")
    ap.add_argument("--ol-alg", default="MD", help="goal-selection rule (reference default: MD, omg/config.py:67)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if args.gpus > 1 and world == 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # one rank per GPU; OMGX_BENCH_BACKEND=gloo lets several ranks share one GPU for a functional test of this path
    backend = os.environ.get("OMGX_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from omg_planner_amd import _lib
    from omg_planner_amd.engine import ChompEngine

    S, G, n = args.scenes, args.goals, args.waypoints
    cfg, model, batch, start, goals = build_workload(S, G, n, args.grid, seed0=rank * S, share_grids=args.share_grids)
    # Scenes are independent: the rank's scenes are split over `--streams` engines on separate HIP streams so the
    # latency-bound kernels of one subset (FK, waypoint SDF, k_chomp_optimize: ~100-200 workgroups) overlap the
    # throughput-bound goal-set kernel of another.
    import copy
    ns = max(1, min(args.streams, S))
    engines = []
    for k in range(ns):
        idx = list(range(k * S // ns, (k + 1) * S // ns))
        sub = batch.subset(idx[0], idx[-1] + 1) if ns > 1 else batch
        engines.append(ChompEngine(model, sub, copy.deepcopy(cfg), start[idx], goals[idx], device=dev, ol_alg=args.ol_alg,
                                   stream=torch.cuda.Stream(device=dev) if ns > 1 else None))
    eng = engines[0]
    lib = _lib.lib()

    # The workload must not drift with the number of steps: a trajectory that has been optimised for hundreds of iterations
    # has left the obstacles' influence regions (the kernel's culling then retires nearly every pair: 4000 consecutive steps
    # measured 0.15 ms/step) and Optimizer.update's schedule grows without bound (1.02^k).  A plan runs cfg.optim_steps = 50
    # goal-selecting iterations, so every 50 steps the engines go back to the fresh plan (device-to-device copies inside the
    # timed region, no host sync): each block of 50 steps is the first 50 iterations of a plan, whatever --steps is.
    snaps = [e.snapshot() for e in engines]
    count = [0]

    def step():
        if count[0] and count[0] % cfg.optim_steps == 0:
            for e, sn in zip(engines, snaps):
                e.restore(sn)
        count[0] += 1
        for e in engines:
            e.t = 0  # pin the goal-set window at the full n waypoints (first-iteration workload)
            e.iterate(0)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # every 4th launch of the dominant kernel is bracketed by HIP events (attached to the dispatch): bracketing all of them
    # costs 1.7 % of the step time, a quarter of them 0.4 %; OMGX_TIMING_STRIDE=1 times every launch
    lib.omgx_timing_enable(0 if os.environ.get("OMGX_NO_TIMING") else int(os.environ.get("OMGX_TIMING_STRIDE", "4")))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    costs = torch.cat([e.final_costs() for e in engines])  # enqueued behind the last step: no host sync before the collective
    from omg_planner_amd.engine import gather_costs_equal
    # the job's one collective (RCCL all-gather over xGMI; host tensors under the gloo test backend)
    allc = gather_costs_equal(costs if backend == "nccl" else costs.cpu(), world)
    assert allc.numel() == world * S
    barrier()
    elapsed = time.perf_counter() - t0
    buf = (C.c_float * 4096)()
    kinds = (C.c_int32 * 4096)()
    nrec = lib.omgx_timing_collect(buf, kinds, 4096)
    lib.omgx_timing_enable(0)
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms_per_plan = None
    if not args.no_plan and rank == 0:
        ms_per_plan = float("inf")
        for _ in range(2):  # best of 2: the first plan pays one-off costs (code-object load of the 30 window sizes)
            eng2 = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=args.ol_alg)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            eng2.plan(early_stop=False)
            torch.cuda.synchronize()
            ms_per_plan = min(ms_per_plan, (time.perf_counter() - tp) * 1e3)
            del eng2
        # the same plan as the reference runs it: a scene that terminates leaves the loop (planner.py:626) — its goal-set
        # batch, goal update and step are skipped from then on (active mask)
        eng3 = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=args.ol_alg)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        eng3.plan(early_stop=True)
        torch.cuda.synchronize()
        ms_plan_early = (time.perf_counter() - tp) * 1e3
        terminated = int((eng3.active == 0).sum().item())
        del eng3

    ms_single = None
    if not args.no_plan and rank == 0:  # BASELINE configs[0]/[1] shape: ONE scene, 64 goals — latency of a whole plan
        one = batch.subset(0, 1)
        best = float("inf")
        for _ in range(3):
            e1 = ChompEngine(model, one, copy.deepcopy(cfg), start[:1], goals[:1], device=dev, ol_alg=args.ol_alg)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            e1.plan(early_stop=False)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - tp) * 1e3)
        ms_single = best

    if rank == 0:
        durs = np.array([buf[i] for i in range(nrec)], dtype=np.float64)
        kind = np.array([kinds[i] for i in range(nrec)])
        goal_ms = durs[kind == 0]  # potentials-only variant = the goal-set batch (dominant kernel)
        wp_ms = durs[kind == 1]    # gradient variant = the waypoint batch of the optimiser step
        O_active = 5
        # points of one launch of the dominant kernel: the goal-set batch plus (in the default two-launch iteration) the
        # S x n x 150 points of the trajectory layer it also computes
        pts_per_launch = engines[0].S * G * n * 10 * model.points_per_link
        if os.environ.get("OMGX_ITERATION", "fused") == "fused" and not os.environ.get("OMGX_NO_OVERLAP"):
            pts_per_launch += engines[0].S * n * 10 * model.points_per_link
        alg_bytes = pts_per_launch * (32 + 128 * O_active)  # N (32 + 128 O_active)
        avg_ms = float(goal_ms.mean()) if len(goal_ms) else float("nan")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, valu_busy, l2_hit = None, None, None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():  # PMC numbers of the same command, collected by tools/collect_profiles.sh
            tj = json.loads(tfile.read_text())
            traffic, valu_busy, l2_hit = tj.get("goalset_kernel_bytes_per_launch", tj.get("k_sdf_chunks_goalset_bytes_per_launch")), tj.get("valu_busy_frac"), tj.get("l2_hit_rate")
        out = {
            "metric": "CHOMP iterations/sec (batched scenes)",
            "value": world * S * args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 SDF / f64 kinematics+update",
            "data": "synthetic",
            "config": {"workload": "100 table-top scenes/GPU x (64-goal goal-set cost + CHOMP step), Panda 9-dof, 30 waypoints",
                       "scenes_per_gpu": S, "goals": G, "waypoints": n, "objects_per_scene": O_active,
                       "sdf_grid": f"4x{args.grid}^3 + 128x96x32 per scene, {'shared' if args.share_grids else 'private'}",
                       "goal_selection": f"{args.ol_alg} on device (omgx_goal_update_optimize)", "launches_per_iteration": 2, "top_k_collision": cfg.top_k_collision, "streams": ns,
                       "plan_restart_every_steps": cfg.optim_steps},
            "roofline": {"bound": "hbm", "kernel": "k_goalset_compact<2> (goal-set batch + trajectory layer: FK + SDF + arc-length cost)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": int(len(goal_ms)), "timing_stride": int(os.environ.get("OMGX_TIMING_STRIDE", "4")), "algorithmic_bytes_per_launch": alg_bytes,
                         "waypoint_launch_avg_ms": float(wp_ms.mean()) if len(wp_ms) else None,
                         "pairs_per_s": pts_per_launch * O_active / (avg_ms * 1e-3), "valu_busy_frac_pmc": valu_busy,
                         "l2_hit_rate_pmc": l2_hit,
                         "note": "frac > 1: the algorithmic figure charges the reference's 56 loads to every (point, object) pair; "
                                 "the kernel retires 83% of pairs before any load and the rest hit L2, so it is VALU/latency-bound "
                                 "(see DESIGN.md section 5)"},
        }
        if ms_per_plan is not None:
            out["ms_per_plan"] = ms_per_plan  # Planner.plan for all scenes of rank 0: initial goal pick + 50 + 20 iterations + final info
            out["ms_per_plan_per_scene"] = ms_per_plan / S
            out["ms_per_plan_early_stop"] = ms_plan_early  # with the reference's break on `terminate` (informational)
            out["scenes_terminated_early"] = terminated
        if ms_single is not None:
            out["ms_per_plan_single_scene"] = ms_single  # one scene alone (launch-latency bound), best of 3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, batch, start, goals, n)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
