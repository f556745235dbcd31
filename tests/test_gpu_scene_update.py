"""Scenes that change while they are resident in HBM (include/omg_hip.h section 8, ABI 8): DeviceScenes.set_object_pose,
grid_slot / replace_grid, omgx_fit_influence_region.  The reference rebuilds its object parameters on every call
(omg/cost.py:296-335) and replaces an obstacle's volume per perception frame (omg/core.py:426-457); here nothing leaves the device.
What has to hold: the region fitted on the device is the host fit (scenes.influence_rbox, the specification) field for field; a
ChompEngine plans again after poses and a volume have changed — without being rebuilt — exactly like an engine packed afresh from
the changed scene, and like the oracle."""
from __future__ import annotations

import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _volumes():
    from omg_planner_amd import scenes as sc
    rng = np.random.RandomState(11)
    out = {
        "sphere": sc.sphere_sdf(0.07, (40, 40, 40), 0.6 / 40),
        "box": sc.box_sdf((0.025, 0.05, 0.10), (48, 40, 36), 0.6 / 48),
        "slab": sc.box_sdf((0.6, 0.4, 0.02), (64, 48, 16), 1.5 / 64),
        "cloud": sc.point_cloud_sdf(rng.uniform([0.3, -0.3, 0.0], [0.7, 0.3, 0.4], size=(600, 3)), 0.02, 0.24),
    }
    noise = rng.uniform(-0.05, 0.6, size=(24, 20, 28)).astype(np.float32)
    out["noise"] = sc.SdfGrid(noise, np.array([-0.2, 0.1, 0.0]), 0.02)
    far = np.full((12, 12, 12), 0.9, np.float32)  # nothing within reach of any threshold: the empty region
    out["nothing"] = sc.SdfGrid(far, np.zeros(3), 0.05)
    weird = rng.uniform(0.0, 0.5, size=(16, 16, 16)).astype(np.float32)
    weird[3, 4, 5], weird[9, 9, 9], weird[0, 0, 0] = np.inf, np.nan, -np.inf
    out["nonfinite"] = sc.SdfGrid(weird, np.array([0.1, 0.1, 0.1]), 0.03)
    return out


@pytest.mark.parametrize("name", ["sphere", "box", "slab", "cloud", "noise", "nothing", "nonfinite"])
@pytest.mark.parametrize("eps,clr", [(0.2, 0.01), (0.1, 0.0), (0.05, 0.0)])
def test_device_fit_is_the_host_fit(dev, name, eps, clr):
    """omgx_object_set_grid + omgx_fit_influence_region against scenes.pack_table(tight=True) for the same volume: every field
    of the record — limits, derived constants, the fitted rounded box — bit for bit."""
    from omg_planner_amd import ops, scenes as sc
    vol = _volumes()[name]
    kw = dict(epsilon=eps, target_epsilon=eps, clearance=clr, target_clearance=clr)
    scene = sc.Scene([sc.SceneObject("obj", np.eye(4), vol)], 0)
    want = sc.pack_table([scene], kw, tight=True).objects[0]
    # a device scene that starts with ANOTHER volume in the slot
    other = sc.Scene([sc.SceneObject("obj", np.eye(4), sc.sphere_sdf(0.05, (8, 8, 8), 0.05))], 0)
    ds = ops.DeviceScenes(sc.pack_table([other], kw, tight=True), dev, reserve_voxels=int(vol.data.size))
    slot = ds.grid_slot(0, 0, vol.data.shape)
    slot.copy_(torch.from_numpy(np.ascontiguousarray(vol.data, np.float32)))
    ds.replace_grid(0, 0, slot, vol.origin, vol.delta, fit="device")
    torch.cuda.synchronize()
    got = ds.sync_host()[0]
    for f in ("lo", "hi", "dim", "delta", "inv_extent", "inv_delta", "epsilon", "clearance", "inv_2eps", "inv_eps", "pose_inv"):
        assert np.array_equal(got[f], want[f]), f
    for f in ("rb_c", "rb_h", "rb_r", "rb_r2"):
        assert np.array_equal(got[f], want[f]), (f, got[f], want[f])
    # the loose region is the default of finish_records
    ds.replace_grid(0, 0, slot, vol.origin, vol.delta, fit="loose")
    torch.cuda.synchronize()
    loose = sc.pack_table([scene], kw, tight=False).objects[0]
    got = ds.sync_host()[0]
    for f in ("rb_c", "rb_h", "rb_r", "rb_r2", "lo", "hi", "dim"):
        assert np.array_equal(got[f], loose[f]), f


def _workload(S, G, seed=0, grid=32):
    import bench
    return bench.build_workload(S, G, 30, grid, seed, False)


@pytest.mark.parametrize("alg,goal_parts", [("MD", 1), ("FTL", 2)])
def test_plan_again_after_a_device_side_scene_change(dev, alg, goal_parts):
    """Plan; move two objects and swap in a fresh point-cloud SDF (built on the device, straight into the pool); plan again with
    the SAME engine: trajectories, goals and info equal to an engine packed afresh from the changed scenes (bit for bit: the same
    records, the same volumes) and to the oracle-driven loop."""
    import bench
    from omg_planner_amd import ops, robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    S, G, n = 3, 16, 30
    cfg = Config(timesteps=n, use_standoff=False)
    model = rb.PandaModel(seed=0)
    scenes = [sc.make_tabletop_scene(s, grid=32, table_grid=(48, 32, 16)) for s in range(S)]
    for scn in scenes:  # private volumes (the device pool holds one copy per object)
        for ob in scn.objects:
            ob.sdf = sc.SdfGrid(ob.sdf.data.copy(), ob.sdf.origin, ob.sdf.delta)
    kw = cfg.layer_kwargs()
    batch = sc.pack_table(scenes, kw, ragged=True, share_grids=False)
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    goals = np.stack([sc.make_reach_goals(scenes[s], model, G, s) for s in range(S)])
    ds = ops.DeviceScenes(batch, dev, reserve_voxels=200_000)
    eng = ChompEngine(model, ds, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg, goal_parts=goal_parts)
    fresh = eng.snapshot()
    eng.plan(early_stop=False)
    torch.cuda.synchronize()
    before = eng.traj.cpu().numpy().copy()
    # ---- the change, all on the device's stream
    eng.restore(fresh)
    new_pose1 = sc._yaw_pose(0.45, 0.12, 0.16, 0.7)
    new_pose2 = torch.as_tensor(sc._yaw_pose(0.55, -0.2, 0.15, -1.1), dtype=torch.float64, device=dev)  # a device tensor works too
    ds.set_object_pose(0, 1, new_pose1)
    ds.set_object_pose(2, 0, new_pose2)
    rng = np.random.RandomState(5)
    cloud = rng.uniform([0.35, -0.1, 0.05], [0.6, 0.15, 0.3], size=(2048, 3))
    pts = torch.as_tensor(cloud, dtype=torch.float64, device=dev)
    lo, hi = cloud.min(0) - 0.24, cloud.max(0) + 0.24
    shape = tuple(len(np.arange(lo[a], hi[a], 0.02)) for a in range(3))
    slot = ds.grid_slot(1, 2, shape)  # outgrows the object's 32^3 slot: space from the reserve
    grid, origin, res = ops.point_cloud_sdf(pts, 0.02, 0.24, out=slot)
    ds.replace_grid(1, 2, grid, origin, res, fit="device")
    ds.set_object_pose(1, 2, np.eye(4))  # a perceived cloud lives in the robot's base frame
    eng.plan(early_stop=False)
    torch.cuda.synchronize()
    after = {k: getattr(eng, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "goal_cost", "pot", "col")}
    assert np.abs(after["traj"] - before).max() > 1e-4  # the change matters
    # ---- a fresh engine from the changed scenes
    scenes[0].objects[1].pose_mat = new_pose1
    scenes[2].objects[0].pose_mat = new_pose2.cpu().numpy()
    scenes[1].objects[2] = sc.SceneObject("obj_2", np.eye(4), sc.SdfGrid(grid.cpu().numpy().copy(), origin.copy(), res))
    batch2 = sc.pack_table(scenes, kw, ragged=True, share_grids=False)
    got = ds.sync_host()
    for f in ("pose_inv", "lo", "hi", "dim", "delta", "inv_extent", "rb_c", "rb_h", "rb_r", "rb_r2", "epsilon", "clearance"):
        assert np.array_equal(got[f], batch2.objects[f]), f
    eng2 = ChompEngine(model, batch2, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg, goal_parts=goal_parts)
    eng2.plan(early_stop=False)
    torch.cuda.synchronize()
    for k, v in after.items():
        assert np.array_equal(v, getattr(eng2, k).cpu().numpy(), equal_nan=True), k
    # ---- and the oracle on the scene as the device holds it now
    eng.restore(fresh)
    eng.select_initial_goal()
    r = engine_vs_oracle(eng, ds.host_batch(), [0, 1, 2], steps=8, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-9 and r["max_cost_rel_err"] <= 1e-5, r


def test_pool_reserve_and_argument_errors(dev):
    from omg_planner_amd import _lib, ops, scenes as sc
    scene = sc.Scene([sc.SceneObject("obj", np.eye(4), sc.sphere_sdf(0.05, (8, 8, 8), 0.05))], 0)
    ds = ops.DeviceScenes(sc.pack_table([scene]), dev)  # no reserve
    with pytest.raises(_lib.OmgHipError):
        ds.grid_slot(0, 0, (9, 9, 9))
    assert ds.grid_slot(0, 0, (8, 8, 4)).shape == (8, 8, 4)  # a smaller volume reuses the slot
    with pytest.raises(IndexError):
        ds.set_object_pose(0, 3, np.eye(4))
    l = _lib.lib()
    assert l.omgx_region_scratch_bytes(0, 4, 4) == 0 and l.omgx_region_scratch_bytes(4, 4, 4) >= 64
    assert l.omgx_fit_influence_region(None, None, None, None, None, 0.1, 0.0, None, None) == _lib.OMGX_ERR_INVALID
    assert l.omgx_object_set_grid(None, None, None, None, 0.1, 0, None) == _lib.OMGX_ERR_INVALID


def test_replacing_a_shared_volume_leaves_the_other_scenes_alone(dev):
    """scenes.pack_table(share_grids=True) stores identical volumes once: several records point at one grid_offset.  Replacing ONE
    object's volume must not write into that slot (copy-on-write: the object gets space of its own), must leave every other
    record and every other scene's costs untouched, and hand a slot nobody uses any more to the next volume that fits it."""
    import bench
    from omg_planner_amd import ops, scenes as sc
    from omg_planner_amd.engine import ChompEngine
    S, G = 4, 8
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 32, 0, True)  # share_grids=True
    offs = batch.objects["grid_offset"]
    shared = [int(o) for o in np.unique(offs) if (offs == o).sum() > 1]
    assert shared, "the workload must share at least one volume between scenes"
    ds = ops.DeviceScenes(batch, dev, reserve_voxels=3 * 40 ** 3)
    eng = ChompEngine(model, ds, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    fresh = eng.snapshot()

    def goal_costs():
        eng.restore(fresh)
        eng.t = 0
        eng.iterate(0)
        torch.cuda.synchronize()
        return eng.goal_cost_total().cpu().numpy().copy(), eng.pot.cpu().numpy().copy()
    before, pot_before = goal_costs()
    # the object of scene 1 that shares its volume with other scenes gets a new (different) volume of the same shape
    lo1, hi1 = int(batch.scene_begin[1]), int(batch.scene_begin[2])
    o = next(k for k in range(lo1, hi1) if int(offs[k]) in shared)
    users = [k for k in range(len(offs)) if int(offs[k]) == int(offs[o]) and k != o]
    pool_before = ds.pool[: ds.pool_used].clone()
    recs_before = ds.sync_host().copy()
    shape = tuple(int(d) for d in batch.objects["dim"][o])
    vol = sc.sphere_sdf(0.04, shape, float(batch.objects["delta"][o]))
    slot = ds.grid_slot(1, o - lo1, shape)
    assert slot.data_ptr() != ds.pool.data_ptr() + 4 * int(offs[o])   # not the shared slot
    assert int(ds.host_objects[o]["grid_offset"]) == int(offs[o])        # nothing changes before replace_grid
    slot.copy_(torch.from_numpy(np.ascontiguousarray(vol.data, np.float32)))
    ds.replace_grid(1, o - lo1, slot, batch.objects["lo"][o].astype(np.float64), float(batch.objects["delta"][o]))
    torch.cuda.synchronize()
    assert torch.equal(ds.pool[: pool_before.numel()], pool_before)      # no byte of the old pool was written
    recs = ds.sync_host()
    for k in range(len(recs)):
        if k != o:
            assert recs[k].tobytes() == recs_before[k].tobytes(), k      # every other record as it was
    assert int(recs[o]["grid_offset"]) >= pool_before.numel() and all(int(recs[k]["grid_offset"]) == int(offs[o]) for k in users)
    after, pot_after = goal_costs()
    for s in (0, 2, 3):
        assert np.array_equal(after[s], before[s]) and np.array_equal(pot_after[s], pot_before[s]), s
    assert not np.array_equal(after[1], before[1])
    # ... and like an engine packed afresh from what the device holds now
    hb = ds.host_batch()
    eng2 = ChompEngine(model, sc.SceneBatch(hb.objects, hb.scene_begin, hb.pool), copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    eng2.t = 0
    eng2.iterate(0)
    torch.cuda.synchronize()
    assert np.array_equal(eng2.goal_cost_total().cpu().numpy(), after)
    # the object's own slot is reused in place from now on; a slot that is outgrown goes to the free list and is taken by the next fit
    used = ds.pool_used
    assert ds.grid_slot(1, o - lo1, shape).data_ptr() == slot.data_ptr() and ds.pool_used == used
    big = tuple(d + 4 for d in shape)
    s2 = ds.grid_slot(1, o - lo1, big)
    assert ds.pool_used == used + int(np.prod(big))
    s2.fill_(0.5)
    ds.replace_grid(1, o - lo1, s2, np.zeros(3), 0.02, fit="loose")
    other = next(k for k in range(int(batch.scene_begin[2]), int(batch.scene_begin[3])) if int(offs[k]) in shared)
    s3 = ds.grid_slot(2, other - int(batch.scene_begin[2]), tuple(int(d) for d in batch.objects["dim"][other]))
    if int(np.prod(batch.objects["dim"][other])) <= int(np.prod(shape)):
        assert s3.data_ptr() == slot.data_ptr() and ds.pool_used == used + int(np.prod(big))  # the freed slot, not the reserve
    # a slot handed out and never committed returns to the free list
    s4 = ds.grid_slot(2, other - int(batch.scene_begin[2]), tuple(int(d) for d in batch.objects["dim"][other]))
    assert s4.data_ptr() == s3.data_ptr()


@pytest.mark.parametrize("S,share", [(100, False), (12, True)])
def test_first_build_on_the_device_is_the_host_pack_table(dev, S, share):
    """DeviceScenes.from_scenes (records on the host, one copy per volume, ALL influence regions fitted by
    omgx_fit_influence_regions in seven launches) against scenes.pack_table — the specification — at the bench's size: every
    record field for field, the pool byte for byte; an engine built that way plans like one built from the host table and like
    the oracle."""
    from omg_planner_amd import ops, robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    cfg = Config(timesteps=30, use_standoff=False)
    scenes = [sc.make_tabletop_scene(s, num_objects=4, grid=64 if S > 50 else 32) for s in range(S)]
    if not share:
        for scn in scenes:
            for ob in scn.objects:
                ob.sdf = sc.SdfGrid(ob.sdf.data.copy(), ob.sdf.origin, ob.sdf.delta)
    want = sc.pack_table(scenes, cfg.layer_kwargs(), ragged=True, share_grids=share)
    tm = {}
    ds = ops.DeviceScenes.from_scenes(scenes, cfg.layer_kwargs(), dev, share_grids=share, timing=tm)
    got = ds.sync_host()
    assert got.shape == want.objects.shape and np.array_equal(ds.host_scene_begin, want.scene_begin)
    for name in want.objects.dtype.names:
        assert np.array_equal(got[name], want.objects[name], equal_nan=True), name
    assert np.array_equal(ds.pool[: ds.pool_used].cpu().numpy(), want.pool)
    assert tm["total_ms"] > 0 and set(tm) >= {"records_ms", "upload_ms", "fit_ms"}
    if S > 50:
        return
    model = rb.PandaModel(seed=0)
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    goals = np.stack([sc.make_reach_goals(scenes[s], model, 8, s) for s in range(S)])
    a = ChompEngine(model, ds, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    b = ChompEngine(model, want, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    for e in (a, b):
        e.plan(early_stop=True)
    torch.cuda.synchronize()
    for k in ("traj", "info", "goal_idx", "learner_state"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    c = ChompEngine(model, ds, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    c.select_initial_goal()
    r = engine_vs_oracle(c, want, [0, S // 2, S - 1], steps=6, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-9 and r["max_cost_rel_err"] <= 1e-5, r
