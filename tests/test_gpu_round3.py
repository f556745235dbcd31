"""Round-3 GPU tests: the rendezvous tickets of the pipeline's parts, an engine bound to a stream, the mask after restore(), a
scene LOADED FROM THE REFERENCE'S FILE FORMATS through the HIP path, the Sophus gap of the SDF op bounded on the
HIP op itself (layers/sdf_matching_loss_kernel.cu:125-133,176), and bench.py's collective path over RCCL with one rank.
"""
from __future__ import annotations

import copy
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parents[1]

STATE = ("traj", "info", "goal_idx", "learner_state", "goal_cost", "end", "goal_rows", "cost_traj", "grad", "pot", "col")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _make(dev, S, G, grid=32, **kw):
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, grid, 0, False)
    return lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", **kw)


def _assert_same(a, b):
    torch.cuda.synchronize()
    for k in STATE:
        x, y = getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy()
        assert np.array_equal(x, y, equal_nan=True), k


def test_bare_iterate_then_pipelined_iterate_draws_fresh_tickets(dev):
    """A bare split-update iterate() leaves ticket 1 in every scene's flag; a pipelined iterate() afterwards must not use 1 again
    (its step workgroups would skip the wait for the learner's): the engine and its parts draw from ONE counter.  Also after
    the number of parts changes and after the engine runs unpipelined again."""
    make = _make(dev, 70, 64)
    one, two = make(), make()
    one.pipeline = 1
    for e in (one, two):
        e.iterate(0)  # bare: one split-update launch over all 70 scenes
    torch.cuda.synchronize()
    assert int((two._scene_flags >> 8).max().item()) == 1 and two._ticket_src[0] == 1  # the word: (ticket << 8) | chosen goal
    assert torch.equal(two._scene_flags & 0xff, two.goal_idx)  # ... and its low byte IS the goal the learner chose (ABI 9)
    two.pipeline = 2
    for t in range(1, 4):
        one.iterate(t); two.iterate(t)
    two.join()
    assert all(p._ticket_src is two._ticket_src for p in two._parts)
    flags = two._scene_flags.cpu().numpy() >> 8
    lo, hi = two._parts[0].S, two.S
    assert flags[:lo].max() != flags[lo:hi].max() and two._ticket_src[0] == 1 + 3 * 2  # every launch drew its own ticket
    _assert_same(one, two)
    two.pipeline = 3  # new parts: they continue the same counter
    for t in range(4, 6):
        one.iterate(t); two.iterate(t)
    two.join()
    two.pipeline = 1  # and the whole engine after its parts
    one.iterate(6); two.iterate(6)
    assert two._ticket_src[0] == 7 + 2 * 3 + 1
    _assert_same(one, two)


def test_a_goal_that_never_arrives_ends_in_nan_not_in_a_hang(dev):
    """k_update_optimize_split: a step workgroup waits for its scene's learner workgroup through a ticket.  Forward progress rests
    on the learner workgroups leading the grid; the wait is bounded all the same (2 s), and a goal that never arrives must show
    as a NaN cost — never as a hung device and never as a plausible number.  omgx_debug_drop_ticket makes the learner workgroups
    publish a wrong ticket (test hook, not part of the ABI)."""
    import time
    from omg_planner_amd import _lib
    eng = _make(dev, 2, 8)()
    eng.split_update = True
    snap = eng.snapshot()
    eng.iterate(0)
    torch.cuda.synchronize()
    assert np.isfinite(eng.info.cpu().numpy()[:, 0]).all()
    lib = _lib.lib()
    lib.omgx_debug_drop_ticket(1)
    try:
        t0 = time.perf_counter()
        eng.iterate(1)
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
    finally:
        lib.omgx_debug_drop_ticket(0)
    assert 1.5 < waited < 10.0, waited                      # the 2 s bound of the wait, not forever
    assert np.isnan(eng.info.cpu().numpy()[:, 0]).all()      # fail loudly
    eng.restore(snap)                                        # and the device is as usable as before
    eng.iterate(0)
    torch.cuda.synchronize()
    assert np.isfinite(eng.info.cpu().numpy()[:, 0]).all()


def test_engine_bound_to_a_stream_does_not_pipeline(dev):
    side = torch.cuda.Stream(device=dev)
    make = _make(dev, 16, 64, stream=side)
    eng = make()
    assert eng.auto_parts(16, 64) > 1
    eng._in_plan = True
    assert eng._pipeline_parts() == 1  # plan() on such an engine stays on its stream
    eng._in_plan = False
    eng.pipeline = 2
    with pytest.raises(ValueError):
        eng.iterate(0)
    eng.pipeline = None
    ref = _make(dev, 16, 64)()
    ref.pipeline = 1
    for t in range(3):
        eng.iterate(t); ref.iterate(t)
    side.synchronize()
    _assert_same(eng, ref)


def test_restore_brings_back_the_unmasked_launches(dev):
    """An early-stop plan switches the launches to the `active` mask; restore() to a snapshot from before switches them back
    (dispatch schedule in use again), and a second plan from there repeats the first bit for bit."""
    eng = _make(dev, 40, 64)()
    snap = eng.snapshot()
    assert not eng._masked
    info1 = eng.plan(early_stop=True).clone()
    assert eng._masked and int((eng.active == 0).sum().item()) > 0
    eng.restore(snap)
    assert not eng._masked and int((eng.active == 0).sum().item()) == 0
    eng.iterate(0)
    assert eng._mask() is None
    eng.restore(snap)
    info2 = eng.plan(early_stop=True)
    torch.cuda.synchronize()
    assert np.array_equal(info1.cpu().numpy(), info2.cpu().numpy(), equal_nan=True)


def test_plan_respects_the_time_budget_on_the_devices_clock(dev):
    """cfg.timeout (planner.py:629): the host may not run more than PLAN_LOOKAHEAD iterations ahead of the device, so a budget
    shorter than the plan stops it after about budget / iteration time + PLAN_LOOKAHEAD iterations, not after all 70."""
    import time
    eng = _make(dev, 100, 64, grid=64)()
    eng.plan(early_stop=False)  # warm: code objects, schedules, side streams
    torch.cuda.synchronize()
    eng2 = _make(dev, 100, 64, grid=64)()
    eng2.cfg.timeout = 0.004  # the whole plan takes ~13 ms
    t0 = time.time()
    eng2.plan(early_stop=False)
    torch.cuda.synchronize()
    assert eng2.timed_out and eng2.iterations_run < 70, eng2.iterations_run
    assert eng2.iterations_run <= 0.004 / 0.00020 + eng2.PLAN_LOOKAHEAD + 4, eng2.iterations_run  # an iteration takes >= 0.2 ms on the device
    assert time.time() - t0 < 1.0
    eng3 = _make(dev, 100, 64, grid=64)()
    eng3.cfg.timeout = -1
    eng3.plan(early_stop=False)
    assert not eng3.timed_out and eng3.iterations_run == 70


def test_scene_loaded_from_the_reference_file_formats_plans_like_the_oracle(dev):
    """SURVEY §8f-4 on the GPU: tests/golden/scene_mat/scene_0.mat + its SDF .pth volumes (written key by key like
    bullet/gen_data.py:21-34, volumes read back identically by the reference's own SignedDensityField.from_pth) -> scene_io
    (omg/core.py:258-278, omg/planner.py:155-174, omg/sdf_tools.py:186-193) -> pack_table -> ChompEngine with the file's
    goal set and standoff tails -> 6 planner iterations, against the oracle-driven loop on the same loaded scene."""
    from omg_planner_amd import robot as rb, scene_io, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    root = ROOT / "tests" / "golden" / "scene_mat"
    got = scene_io.load_scene_mat(str(root / "scene_0.mat"), str(root))
    assert got.goals.shape == (6, 9) and got.reach_grasps.shape == (6, 5, 9) and len(got.scene.objects) == 3
    for alg, standoff in (("MD", True), ("FTL", False)):
        cfg = Config(timesteps=30, use_standoff=standoff)
        model = rb.PandaModel(seed=0)
        batch = sc.pack_table([got.scene], cfg.layer_kwargs())
        # the table of the loaded scene: target = the object `target_name` names (the LAST one here), with the target's epsilon
        assert batch.num_scenes == 1 and len(batch.objects) == 3
        assert float(batch.objects["epsilon"][got.scene.target_idx]) == pytest.approx(cfg.target_epsilon)
        # the fixture's volumes are small (20-30 cm) and its goal configurations random: the arm STARTS next to the target (a
        # reach configuration found by sampling) so that the loaded volumes shape both the goal choice and the step
        start = sc.make_reach_goals(got.scene, model, 1, 3)
        eng = ChompEngine(model, batch, cfg, start, got.goals[None], reach_grasps=got.reach_grasps[None] if standoff else None,
                          device=dev, ol_alg=alg)
        eng.select_initial_goal()
        res = engine_vs_oracle(eng, batch, [0], steps=6, pin_window=False)
        assert res["ok"] and res["goal_idx_equal"], res
        assert res["max_traj_err"] < 1e-9 and res["max_cost_rel_err"] < 1e-9, res
        pot, gc = eng.pot.cpu().numpy(), eng.goal_cost.cpu().numpy()
        assert np.isfinite(pot).all() and (pot > 0).sum() > 20 and (gc > 0).all() and gc.max() > 0.5, "the loaded volumes must matter to the plan"


def test_hip_sdf_op_under_sophus_style_transform_is_within_the_parity_bar(dev):
    """a1's kernel body cannot be compiled here (Sophus / Eigen / nvcc absent).  The one place where the build's arithmetic
    knowingly differs from it is the pose: the reference converts the float32 matrix to a unit quaternion, rotates points
    with it and rotates gradients back with the matrix regenerated from it (.cu:125-133,176); the HIP op multiplies by the
    float32 matrix.  Here the HIP op itself (omgx_sdf_loss_forward through the drop-in omg_cuda module) is fed both forms —
    the matrix as the product path does, and Sophus' float32 restatement applied on the host with an identity pose on the
    device — on fixture-like scenes with se3_inverse-style float32 poses: potentials, gradients and collides agree far
    inside north_star's 1e-4."""
    from omg_planner_amd import omg_cuda, scenes as sc
    import importlib.util
    spec = importlib.util.spec_from_file_location("loose_ends", ROOT / "tests" / "test_reference_loose_ends.py")  # the float32 restatement of Sophus / Eigen
    le = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(le)
    F, _eigen_quaternion_from_matrix, _quaternion_matrix, _sophus_rotate = le.F, le._eigen_quaternion_from_matrix, le._quaternion_matrix, le._sophus_rotate
    rng = np.random.RandomState(23)
    worst_pot = worst_grad = 0.0
    flips = total = 0
    for trial in range(8):
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        A *= np.sign(np.linalg.det(A))
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = A, rng.uniform(-0.8, 0.8, 3)
        inv = sc.se3_inverse(pose).astype(F)
        grid = sc.sphere_sdf(0.08, (24, 24, 24), 0.5 / 24) if trial % 2 else sc.box_sdf((0.06, 0.09, 0.05), (24, 20, 28), 0.02)
        sdf, lim = sc.pack_padded([sc.SceneObject("o", pose, grid)])
        local = rng.uniform(grid.min_coords - 0.03, grid.min_coords + np.array(grid.data.shape) * grid.delta + 0.03, (20000, 3))
        pts = (local @ A.T + pose[:3, 3]).astype(F)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        eps, pad, clr, dis = (t(np.array([v], F)) for v in (0.2, 1.0, 0.01, 0.0))
        pot_a, grad_a, col_a = (x.cpu().numpy() for x in omg_cuda.sdf_loss_forward(t(inv[None]), t(sdf), t(lim), t(pts), eps, pad, clr, dis))
        q = _eigen_quaternion_from_matrix(inv[:3, :3])
        u_b = np.stack([_sophus_rotate(q, p) + inv[:3, 3] for p in pts]).astype(F)
        pot_b, grad_o, col_b = (x.cpu().numpy() for x in omg_cuda.sdf_loss_forward(t(np.eye(4, dtype=F)[None]), t(sdf), t(lim), t(u_b), eps, pad, clr, dis))
        grad_b = (grad_o.astype(F) @ _quaternion_matrix(q)).astype(F)  # rotationMatrix().transpose() * vgrad (.cu:176-179)
        worst_pot = max(worst_pot, float(np.abs(pot_a - pot_b).max()))
        worst_grad = max(worst_grad, float(np.abs(grad_a - grad_b).max()))
        flips += int((col_a != col_b).sum())
        total += len(pts)
        assert (pot_a != 0).mean() > 0.05
    assert worst_pot < 1e-6, worst_pot          # north_star: 1e-4 on cost values
    assert worst_grad < 1e-4, worst_grad        # central differences over one voxel amplify a 3e-7 m coordinate change
    assert flips <= 2e-4 * total, (flips, total)


def test_fitted_influence_regions_change_no_result(dev):
    """The influence region only decides which (point, object) pairs are skipped.  Against the same table with the default
    region (the whole grid): every per-point output of the layer — a sum over the objects in index order, to which a skipped
    pair would have added 0 — is bit-identical, with and without per-point potentials of the goal-set batch; a goal's cost is a
    float32 sum in the order its pairs were queued, so it moves by that sum's rounding only (1e-6) and stays within the tests'
    1e-5 of the oracle, which culls nothing."""
    import bench
    from omg_planner_amd import ops, scenes as sc
    from oracle import oracle as orc
    S, G, n = 6, 16, 30
    cfg, model, _, start, goals = bench.build_workload(S, G, n, 32, 0, True)
    scenes = [sc.make_tabletop_scene(s, grid=32) for s in range(S)]
    tight = sc.pack_table(scenes, cfg.layer_kwargs(), tight=True)
    loose = sc.pack_table(scenes, cfg.layer_kwargs(), tight=False)
    assert (loose.objects["rb_r2"] == 0).all() and (tight.objects["rb_r2"] > 0).any()
    P = model.points_per_link
    robot = ops.robot_blob(model, dev)
    dt, dl = ops.DeviceScenes(tight, dev), ops.DeviceScenes(loose, dev)
    traj = torch.as_tensor(np.stack([sc.cubic_init(start[s], goals[s, 0], n) for s in range(S)]), device=dev)
    a, b = ops.fk_sdf(robot, P, dt, traj), ops.fk_sdf(robot, P, dl, traj)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert float((a[0] > 0).float().mean()) > 0.02
    g = torch.as_tensor(goals, device=dev)
    ts = traj[:, 0]
    ct, colt, pt = ops.goalset_cost(robot, P, dt, ts, g, n, cfg.time_interval, want_potentials=True)
    cl, coll, pl = ops.goalset_cost(robot, P, dl, ts, g, n, cfg.time_interval, want_potentials=True)
    assert torch.equal(pt, pl) and torch.equal(ct, cl) and torch.equal(colt, coll)  # per-point path: sums in a fixed per-point order
    qt, qcolt, _ = ops.goalset_cost(robot, P, dt, ts, g, n, cfg.time_interval)
    ql, qcoll, _ = ops.goalset_cost(robot, P, dl, ts, g, n, cfg.time_interval)
    torch.cuda.synchronize()
    assert torch.equal(qcolt, qcoll)  # collision counts are sums of 0 / 1: exact in any order
    np.testing.assert_allclose(qt.cpu().numpy(), ql.cpu().numpy(), rtol=2e-6, atol=1e-7)
    ref, _ = orc.goalset_cost(model.blob(), P, tight, traj[:, 0].cpu().numpy(), goals, n, cfg.time_interval)
    np.testing.assert_allclose(qt.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_bench_collective_path_runs_over_rccl_with_one_rank(tmp_path):
    """bench.py under torch.distributed.run with ONE rank and the default backend: RCCL initialisation, the barrier, the
    all_gather_into_tensor of the final costs on DEVICE tensors and the MAX all-reduce of the timing all run for real (the
    2-rank test shares the one GPU over gloo and never touches RCCL).  Costs equal the plain single-process run bit for bit;
    the line carries the per-rank roofline list of an N-GPU line."""
    common = ["--total-scenes", "6", "--goals", "8", "--grid", "24", "--steps", "4", "--warmup", "1", "--no-plan", "--no-cpu-baseline", "--no-parity"]
    env = {k: v for k, v in os.environ.items() if k not in ("OMGX_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    one, two = tmp_path / "one.npy", tmp_path / "two.npy"
    r1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *common, "--dump-costs", str(one)], env=env, capture_output=True,
                        text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    port = 29500 + os.getpid() % 2000
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", *common, "--dump-costs", str(two)], env=env,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout[-1000:], r2.stderr[-3000:])
    a, b = np.load(one), np.load(two)
    assert a.shape == (6,) and np.array_equal(a, b), (a, b)
    j = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["scaling"] == "strong"
    pr = j["roofline"]["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["scenes"] == 6 and pr[0]["avg_launch_ms"] > 0
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert "per_rank" not in j1["roofline"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_when_typed_plainly(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (what a driver types): bench.py starts torch.distributed.run as a child
    before it touches the GPU, rank 0's JSON line comes back on stdout, exit code 0.  Both ranks share the one GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMGX_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--total-scenes", "5", "--goals", "8", "--grid", "24", "--steps", "3",
                        "--warmup", "1", "--no-plan", "--no-cpu-baseline", "--no-parity"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and len(j["roofline"]["per_rank"]) == 2
