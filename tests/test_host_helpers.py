"""CPU tests of the host-side mirrors (util / config / trajectory / Cost helper methods) against the reference
fixtures."""
import numpy as np

from tests import helpers as H


def test_wrap_helpers():
    from omg_planner_amd import util
    assert util.wrap_index(3) == [0, 1, 2] and util.wrap_index(8) == list(range(7)) and util.wrap_index(9) == list(range(8))
    assert util.wrap_index(10) == list(range(7)) + [8]
    assert util.wrap_joint(8) == list(range(7)) and util.wrap_joint(9) == list(range(7)) + [8] and util.wrap_joint(10) == list(range(7)) + [9]
    q = np.arange(9) * 0.1
    w = util.wrap_value(q)
    assert w.shape == (10,) and w[7] == 0 and np.allclose(w[8:], np.rad2deg(q[7:]))
    assert np.allclose(util.wrap_values(q[None])[0], w)
    assert util.safe_div(1.0, 1.0) == 1.0 / (1.0 + 1e-8)


def test_config_matrices_and_derivatives_match_reference():
    from omg_planner_amd.config import Config
    fx = H.load("matrices.npz")
    for n, gsp in ((30, True), (30, False), (50, True)):
        cfg = Config(timesteps=n, goal_set_proj=gsp)
        tag = f"n{n}_g{int(gsp)}_dt0.10"
        np.testing.assert_allclose(cfg.diff_matrices[0], fx[tag + "_D1"], atol=1e-12)
        np.testing.assert_allclose(cfg.diff_matrices[1], fx[tag + "_D2"], atol=1e-12)
        np.testing.assert_allclose(cfg.A, fx[tag + "_A"], atol=1e-9)
        np.testing.assert_allclose(cfg.Ainv, fx[tag + "_Ainv"], rtol=1e-9, atol=1e-12)
    cfg = Config()
    cfg.get_global_param(50)
    assert abs(cfg.time_interval - 0.06) < 1e-15  # dt from the PREVIOUS timesteps (config.py:201)
    np.testing.assert_allclose(cfg.Ainv, fx["n50_g1_dt0.06_Ainv"], rtol=1e-9, atol=1e-12)
    # finite differences reproduce the reference's v and a of forward_kinematics_obstacle
    full = H.load("cost_topk1000.npz")
    cfg = Config(timesteps=30)
    x = full["x"]  # [n,10,p,3]
    m = H.model_from(full)
    from oracle import oracle as orc
    xs = orc.config_points(m.blob(), m.points_per_link, full["start"]).transpose(1, 0, 2)  # [p,10,3]
    xe = orc.config_points(m.blob(), m.points_per_link, full["end"]).transpose(1, 0, 2)
    ws = x.transpose(2, 1, 0, 3)  # [p,10,n,3]
    np.testing.assert_allclose(cfg.get_derivative(ws, xs, xe, 1).transpose(2, 1, 0, 3), full["v"], atol=1e-9)
    np.testing.assert_allclose(cfg.get_derivative(ws, xs, xe, 2).transpose(2, 1, 0, 3), full["a"], atol=1e-7)


def test_trajectory_container():
    from scipy import interpolate
    from omg_planner_amd.config import Config
    from omg_planner_amd.trajectory import Trajectory
    cfg = Config(timesteps=30)
    t = Trajectory(cfg=cfg)
    assert t.data.shape == (30, 9)
    # the closed-form blend equals scipy's clamped cubic spline through the two knots (omg/util.py:238-258)
    f = interpolate.CubicSpline(np.linspace(0, 1, 2), np.stack([t.start, t.end]), bc_type="clamped")
    np.testing.assert_allclose(t.data, f(np.linspace(0, 1, 32)[1:-1]), atol=1e-12)
    g = np.ones((30, 9))
    before = t.data.copy()
    t.update(g)
    np.testing.assert_allclose(t.data[:, :7], before[:, :7] + 1)
    assert (t.data[:, 7:] <= 0.04).all() and (t.data[:, 7:] >= 0).all()


def test_linear_interpolation_is_bit_identical_to_the_references_interp1d():
    """omg/util.py:238-290 ("linear"): interp1d over x = [0, 1] at t = linspace(0, 1, n + 2)[1:-1].  numpy's linspace is
    i * fl(1 / (n + 1)), not i / (n + 1) — the device kernels and the oracle use the same expression (omg_kernels.hip,
    omg_oracle.c), scenes.linear_init is checked here for every trajectory length."""
    from scipy import interpolate
    from omg_planner_amd import scenes as sc
    rng = np.random.RandomState(0)
    for n in range(1, 65):
        a, b = rng.uniform(-3, 3, 9), rng.uniform(-3, 3, 9)
        f = interpolate.interp1d(np.linspace(0, 1, 2), np.stack([a, b]), "linear", axis=0)
        want = f(np.linspace(0, 1, n + 2)[1:-1])
        assert np.array_equal(sc.linear_init(a, b, n), want), n
        i = np.arange(1, n + 1)[:, None]
        assert np.array_equal(a + (i * (1.0 / (n + 1.0))) * (b - a), want), n  # the expression of the kernels / the oracle


def test_trajectory_dynamic_timestep():
    """cfg.dynamic_timestep (omg/core.py:64-76): n = clip(int(|start - end| / traj_delta), traj_min_step, traj_max_step);
    cfg.timesteps is set FIRST, so get_global_param(n) keeps dt = 0.1 * n / n (config.py:201) and only resizes the matrices."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.trajectory import Trajectory
    cfg = Config(timesteps=30, dynamic_timestep=True)
    t = Trajectory(cfg=cfg)
    want = min(max(int(np.linalg.norm(t.start - t.end) / 0.05), 2), 50)
    assert want == 50 and t.data.shape == (50, 9) and cfg.timesteps == 50 and t.timesteps == 50
    assert abs(cfg.time_interval - 0.1) < 1e-15 and cfg.A.shape == (50, 50)
    t.end = t.start + np.array([0.3, 0, 0, 0.4, 0, 0, 0, 0, 0])  # distance 0.5 -> 10 waypoints (0.5 / 0.05 = 10.000000000000002)
    t.interpolate_waypoints(mode="linear")
    assert t.data.shape == (int(np.linalg.norm(t.start - t.end) / 0.05), 9) and cfg.timesteps == t.data.shape[0]
    np.testing.assert_allclose(t.data[0], t.start + (t.end - t.start) / (t.data.shape[0] + 1), atol=1e-15)
    t.end = t.start + 1e-3
    t.interpolate_waypoints()
    assert t.data.shape == (2, 9)  # traj_min_step


def test_cost_numpy_helpers_match_reference_intermediates():
    """functional_grad / compute_point_jacobian reproduce obs_grad of the clean branch from the fixture's x, v, a, J."""
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.util import wrap_index
    fx = H.load("cost_topk1000.npz")
    c = Cost.__new__(Cost)
    n = fx["xi"].shape[0]
    total = np.zeros((n, 9))
    for j in range(10):
        k = len(wrap_index(j + 1))
        cost_j, grad_j = c.functional_grad(fx["v"][:, j], fx["a"][:, j], fx["J"][:, j, :, :k], fx["potentials"][:, j].astype(np.float64),
                                           fx["potential_grads"][:, j].astype(np.float64))
        total[:, wrap_index(j + 1)] += grad_j.sum(1)
    # cross-check against the oracle's clean branch on the same inputs
    from oracle import oracle as orc
    m = H.model_from(fx)
    prm = H.params_from(dict(fx, cfg_top_k=0), orc.ChompParams, n, m.points_per_link, 0, 1.0, 0.0)
    prm.clip_grad_scale = 1e30
    col = np.zeros_like(fx["potentials"])[None]
    _, grad, _, _ = orc.chomp_optimize(m.blob(), prm, fx["xi"][None], fx["start"][None], fx["end"][None], fx["end"][None, None],
                                       fx["end"][None], fx["potentials"][None], fx["potential_grads"][None], col)
    np.testing.assert_allclose(total, grad[0], rtol=1e-9, atol=1e-9)


def test_select_goals_mirrors_setup_goal_set_quirks():
    """planner.py:526-575: collision threshold, greedy diversity with the j / j+1 index quirk, sampling from np.random."""
    from omg_planner_amd.goalset import select_goals
    goals = [np.full(9, v, float) for v in (0.0, 0.05, 1.0, 1.02, 2.0, 3.0)]
    reach = [g[None] + 0.0 for g in goals]
    collide = np.array([0, 0, 9, 0, 5, 6])           # goals 2 and 5 collide too much (allow 5)
    pots = np.arange(6, dtype=np.float32)
    # survivors: goals 0, 1, 3, 4 -> filtered list [0.0, 0.05, 1.02, 2.0]; diversity keeps 0.0 (seed), drops 0.05 (too close),
    # keeps 1.02 and 2.0 but records j = 1 and j = 2 (indices of 0.05 and 1.02 in the filtered list): the reference's quirk
    rng = np.random.RandomState(0)
    grasps, r, p, chosen = select_goals(goals, reach, collide, pots, rng=rng)
    assert sorted(int(c) for c in chosen) == [1, 2]
    assert sorted(float(g[0]) for g in grasps) == [0.05, 1.02]
    np.testing.assert_array_equal(np.sort(p), [1.0, 3.0])
    # same numpy stream -> same order as np.random.choice on [1, 2]
    np.testing.assert_array_equal(chosen, np.random.RandomState(0).choice([1, 2], 2, replace=False))
    # without the diversity filter every collision-free goal is a candidate; the cap applies
    g2, _, _, c2 = select_goals(goals, reach, collide, pots, goal_set_max_num=3, filter_diversity=False, rng=np.random.RandomState(1))
    assert len(g2) == 3 and set(int(c) for c in c2) <= {0, 1, 2, 3}
    # nothing survives
    assert select_goals(goals, reach, np.full(6, 99), pots) == ([], [], [], [])


def test_optimizer_and_cost_numpy_methods_match_reference():
    """The small numpy methods that keep the reference's signatures (Optimizer.update / goal_set_projection /
    compute_traj_v / handle_joint_limit / check_joint_limit, Cost.forward_points / color_point) against outputs of the
    reference's own methods (tests/golden/make_golden.py:_fixed_host_helpers; tests/fuzz/fuzz_host_mirror.py: random cases)."""
    import types

    import torch
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.optimizer import Optimizer
    fx = H.load("host_helpers.npz")
    robot = types.SimpleNamespace(joint_lower_limit=fx["joint_lower_limit"], joint_upper_limit=fx["joint_upper_limit"])
    k = 0
    while f"c{k}_n" in fx:
        g = lambda name: fx[f"c{k}_{name}"]  # noqa: E731
        n = int(g("n"))
        cfg = Config()
        cfg.use_standoff, cfg.goal_set_proj, cfg.joint_limit_max_steps = bool(g("standoff")), True, int(g("joint_limit_max_steps"))
        cfg.timesteps = int(g("timesteps_before"))
        cfg.get_global_param(n)
        assert abs(cfg.time_interval - float(g("time_interval"))) < 1e-15
        opt = Optimizer(types.SimpleNamespace(config=cfg, robot=robot),
                        types.SimpleNamespace(target_obj=types.SimpleNamespace(reach_grasps=g("reach"))))
        for _ in range(int(g("updates"))):
            opt.update()
        np.testing.assert_allclose([cfg.obstacle_weight, cfg.smoothness_weight, cfg.grasp_weight, cfg.step_size], g("schedules"), rtol=1e-14)
        data = g("data")
        traj = types.SimpleNamespace(data=data.copy(), end=data[-1].copy(), goal_set=g("goal_set"), goal_idx=int(g("goal_idx")))
        np.testing.assert_allclose(opt.goal_set_projection(traj, g("grad")), g("projection"), rtol=1e-8, atol=1e-8)
        assert np.array_equal(opt.compute_traj_v(data), g("traj_v"))
        np.testing.assert_allclose(opt.handle_joint_limit(data.copy()), g("limited"), rtol=1e-9, atol=1e-9)
        rs = np.random.RandomState(k)
        lo, hi = robot.joint_lower_limit[0], robot.joint_upper_limit[0]
        for kind, violate, terminate in g("limit_flags"):
            probe = rs.uniform(lo + 0.1, hi - 0.1, (n, 9))
            if int(kind) & 1:
                probe[1, 2] = -10.0
            if int(kind) & 2:
                probe[2, 3] = 10.0
            info = {"terminate": True}
            opt.check_joint_limit(probe, info)
            assert (info["violate_limit"], info["terminate"]) == (bool(violate), bool(terminate)), (k, kind)
        k += 1
    assert k == 5
    c = Cost.__new__(Cost)
    assert np.array_equal(c.forward_points(fx["fp_pose"], fx["fp_pts"]), fx["fp_out"])
    assert np.array_equal(c.forward_points(fx["fp_pose"], fx["fp_pts"], fx["fp_normals"]), fx["fp_out_normals"])
    vis = fx["cp_vis"].copy()
    c.color_point(vis, torch.as_tensor(fx["cp_collide"]))
    assert np.array_equal(vis, fx["cp_out"])


def test_lazy_info_builds_collision_pts_on_first_access():
    """cost.LazyInfo: info["collision_pts"] is built by the first subscript access (the reference's consumers index it,
    omg/core.py:661) and is an ordinary key afterwards; other missing keys still raise KeyError."""
    import pytest
    from omg_planner_amd.cost import LazyInfo
    calls = []
    info = LazyInfo({"cost": 1.0}, collision_pts=lambda: calls.append(1) or np.ones((2, 10, 3, 12)))
    assert "collision_pts" not in info and len(info) == 1 and not calls
    assert info["collision_pts"].shape == (2, 10, 3, 12) and calls == [1]
    assert info["collision_pts"].shape == (2, 10, 3, 12) and calls == [1] and "collision_pts" in info and len(info) == 2
    with pytest.raises(KeyError):
        info["nope"]
    with pytest.raises(KeyError):
        LazyInfo({"cost": 1.0})["collision_pts"]


def test_scene_from_env_takes_the_gpu_copy_of_the_volumes():
    """scenes.scene_from_env: a reference-shaped Env -> Scene; the volume is sdf.data_torch (un-penalised), and packing the
    result reproduces Env.combine_sdfs' padded tensor and limits (omg/core.py:366-411) for the same objects."""
    import types

    import torch
    from omg_planner_amd import scenes as sc
    rng = np.random.RandomState(0)
    objs = []
    for k, dims in enumerate([(6, 5, 4), (3, 8, 7)]):
        g = rng.normal(0, 0.1, dims).astype(np.float32)
        pen = g.copy()
        pen[pen < 0] *= 5.0  # Model's penalize_constant on the numpy copy (core.py:110)
        origin, delta = rng.uniform(-0.2, 0, 3), 0.02 * (k + 1)
        sdf = types.SimpleNamespace(data=pen, data_torch=torch.from_numpy(g), min_coords=origin, max_coords=origin + delta * np.array(dims), delta=delta)
        pose = np.eye(4)
        pose[:3, 3] = rng.normal(size=3)
        objs.append(types.SimpleNamespace(name=f"obj{k}", pose_mat=pose, attached=bool(k), sdf=sdf))
    env = types.SimpleNamespace(objects=objs, target_idx=1)
    scene = sc.scene_from_env(env)
    assert scene.target_idx == 1 and [o.name for o in scene.objects] == ["obj0", "obj1"] and scene.objects[1].attached
    for o, ref in zip(scene.objects, objs):
        assert np.array_equal(o.sdf.data, ref.sdf.data_torch.numpy()) and not np.array_equal(o.sdf.data, ref.sdf.data)
        assert np.array_equal(o.pose_mat, ref.pose_mat) and o.sdf.delta == ref.sdf.delta
    # Env.combine_sdfs restated on the same objects (core.py:366-411)
    mx = np.array([o.sdf.data.shape for o in objs]).max(axis=0)
    want = np.ones((2, *mx), np.float32)
    lim = np.zeros((2, 10), np.float32)
    for i, o in enumerate(objs):
        size = o.sdf.data.shape
        want[i, :size[0], :size[1], :size[2]] = o.sdf.data_torch.numpy()
        lo, hi = o.sdf.min_coords, o.sdf.max_coords
        for a in range(3):
            lim[i, a] = lo[a]
            lim[i, 3 + a] = lo[a] + (hi[a] - lo[a]) * mx[a] / size[a]
            lim[i, 6 + a] = mx[a]
        lim[i, 9] = o.sdf.delta
    sdf, limits = sc.pack_padded(scene.objects)
    assert np.array_equal(sdf, want) and np.array_equal(limits, lim)


def test_lazy_array_materialises_on_first_use():
    """cost.LazyArray (vis_pts of batch_obstacle_cost): shape / len without building; indexing, np.asarray, ndarray methods
    and item assignment build once and then behave like the array."""
    from omg_planner_amd.cost import LazyArray
    calls = []
    la = LazyArray((4, 10, 3, 12), lambda: calls.append(1) or np.arange(4 * 10 * 3 * 12, dtype=np.float64).reshape(4, 10, 3, 12))
    assert la.shape == (4, 10, 3, 12) and len(la) == 4 and la.ndim == 4 and not calls
    assert la[1, 2, 0, 5] == 1 * 360 + 2 * 36 + 5 and calls == [1]
    assert np.asarray(la).sum() == la.sum() and np.asarray(la, dtype=np.float32).dtype == np.float32 and calls == [1]
    la[0, 0, 0, 0] = -1.0
    assert la.reshape(-1)[0] == -1.0 and (la[..., :3] + 1).shape == (4, 10, 3, 3) and calls == [1]
