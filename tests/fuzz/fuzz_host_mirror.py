"""Randomised check of the numpy utilities of the drop-in classes against THE REFERENCE's own methods (build container
only: needs /root/reference, no GPU).  `Optimizer.optimize` / `Cost.compute_total_loss` run on the device and are pinned
by the fixtures and the GPU fuzzers; this tool covers the small host-side methods that keep the reference's signatures for
callers which use them on their own:

    Optimizer.goal_set_projection   (closed-form projector vs the reference's explicit C / inverse, optimizer.py:88-113)
    Optimizer.compute_traj_v, handle_joint_limit, check_joint_limit   (optimizer.py:137-174)
    Optimizer.update   (schedules written into cfg, optimizer.py:59-80)
    Cost.forward_points, color_point, functional_grad, compute_point_jacobian   (cost.py:24-110)
    scene_io.save_sdf_pth / load_sdf_pth vs SignedDensityField.from_pth (+ .resize)   (sdf_tools.py:37-45,186-193)
    config.get_global_param matrices (A, Ainv, diff) for random trajectory lengths / link weights / time steps

    python tests/fuzz/fuzz_host_mirror.py [trials] [seed]
"""
import importlib.util
import sys
import time
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402


def load_generators():
    spec = importlib.util.spec_from_file_location("make_golden", ROOT / "tests" / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    if not Path("/root/reference").exists():
        raise SystemExit("needs /root/reference (build container only)")
    mg = load_generators()
    config, cost_mod, opt_mod, util, rk = mg.load_reference()
    rcfg = config.cfg
    from omg_planner_amd import robot as rb
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.optimizer import Optimizer

    model = rb.PandaModel(seed=0)
    lo, hi = model.joint_lower_limit, model.joint_upper_limit
    stats, fails = {}, []
    t0 = time.time()
    import importlib
    import tempfile
    sdf_tools = importlib.import_module("omg.sdf_tools")
    from omg_planner_amd import scene_io
    from omg_planner_amd.scenes import SdfGrid
    tmpdir = tempfile.mkdtemp(prefix="omg_pth_")

    def check(name, a, b, rtol=1e-9, atol=1e-10):
        stats[name] = stats.get(name, 0) + 1
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or not np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True):
            d = float(np.nanmax(np.abs(a - b))) if a.shape == b.shape else float("nan")
            fails.append(f"{name}: shapes {a.shape} {b.shape}, max abs {d:.3e}")

    for k in range(trials):
        n = int(rng.choice([5, 8, 12, 30, 30, 50, 64]))
        m_prev = n if rng.rand() < 0.4 else int(rng.randint(max(1, n // 2), 2 * n))  # get_global_param(n) called from an m_prev-step state
        dt = 0.1 * m_prev / n
        standoff = bool(rng.rand() < 0.5)
        over = dict(timesteps=n, time_interval=dt, use_standoff=standoff, goal_set_proj=True,
                    joint_limit_max_steps=int(rng.choice([10, 2, 0, 25])), base_step_size=float(rng.choice([0.1, 0.02])),
                    cost_schedule_boost=float(rng.choice([1.02, 1.0])), cost_schedule_decay=float(rng.choice([1.0, 0.98])),
                    step_decay_rate=float(rng.choice([1.0, 0.97])))
        lsw = np.where(rng.rand(9) < 0.3, 0.5, 1.0)
        mg.reset_cfg(rcfg, **over)
        rcfg.link_smooth_weight = lsw.copy()
        cfg = Config()
        for key, v in over.items():
            if key not in ("timesteps", "time_interval"):
                setattr(cfg, key, v)
        cfg.link_smooth_weight = lsw.copy()
        cfg.timesteps = m_prev
        cfg.get_global_param(n)
        check("time_interval", cfg.time_interval, rcfg.time_interval, rtol=1e-15, atol=0)
        check("A", cfg.A, rcfg.A); check("Ainv", cfg.Ainv, rcfg.Ainv, rtol=1e-8, atol=1e-9 * np.abs(rcfg.Ainv).max())
        check("diff_matrices", cfg.diff_matrices[0], rcfg.diff_matrices[0])

        c = cfg.reach_tail_length if standoff else 1
        robot = types.SimpleNamespace(joint_lower_limit=lo, joint_upper_limit=hi)
        G = int(rng.randint(1, 6))
        goal_set = rng.uniform(lo[0], hi[0], (G, 9))
        reach = rng.uniform(lo[0], hi[0], (G, c, 9))
        tgt = types.SimpleNamespace(reach_grasps=reach)
        ref_cost = types.SimpleNamespace(target_obj=tgt)
        r_opt = opt_mod.Optimizer(types.SimpleNamespace(config=rcfg, robot=robot), ref_cost)
        m_opt = Optimizer(types.SimpleNamespace(config=cfg, robot=robot), types.SimpleNamespace(target_obj=tgt))

        # schedules
        for _ in range(int(rng.randint(1, 6))):
            r_opt.update(); m_opt.update()
        for key in ("obstacle_weight", "smoothness_weight", "grasp_weight", "step_size"):
            check("update." + key, getattr(cfg, key), getattr(rcfg, key))

        wide = float(rng.choice([0.0, 0.05, 0.5]))
        data = rng.uniform(lo[0] - wide, hi[0] + wide, (n, 9))
        grad = rng.normal(0, float(rng.choice([1.0, 30.0])), (n, 9))
        traj = types.SimpleNamespace(data=data.copy(), end=data[-1].copy(), goal_set=goal_set, goal_idx=int(rng.randint(0, G)))
        check("goal_set_projection", m_opt.goal_set_projection(traj, grad), r_opt.goal_set_projection(traj, grad),
              rtol=1e-8, atol=1e-9 * max(1.0, np.abs(grad).max()))
        check("compute_traj_v", m_opt.compute_traj_v(data), r_opt.compute_traj_v(data), rtol=0, atol=0)
        check("handle_joint_limit", m_opt.handle_joint_limit(data.copy()), r_opt.handle_joint_limit(data.copy()), rtol=1e-9, atol=1e-9)
        # check_joint_limit needs BOTH kinds of violation in the reference: build each combination
        probe = rng.uniform(lo[0] + 0.1, hi[0] - 0.1, (n, 9))
        kind = int(rng.randint(0, 4))
        if kind & 1:
            probe[rng.randint(0, n), rng.randint(0, 9)] = -10.0
        if kind & 2:
            probe[rng.randint(0, n), rng.randint(0, 9)] = 10.0
        for term in (True, False):
            ia, ib = {"terminate": term}, {"terminate": term}
            m_opt.check_joint_limit(probe, ia); r_opt.check_joint_limit(probe, ib)
            check("check_joint_limit", [ia["violate_limit"], ia["terminate"]], [bool(ib["violate_limit"]), bool(ib["terminate"])], rtol=0, atol=0)

        # Cost helpers (no device needed: bypass __init__)
        r_c, m_c = object.__new__(cost_mod.Cost), object.__new__(Cost)
        p, links, npts = int(rng.randint(1, 9)), 10, int(rng.randint(1, 17))
        pose = rng.normal(size=(p, links, 4, 4))
        pts = rng.normal(size=(links, 3, npts))
        nrm = rng.normal(size=(links, 3, npts))
        check("forward_points", m_c.forward_points(pose, pts), r_c.forward_points(pose, pts), rtol=0, atol=0)
        check("forward_points[normals]", m_c.forward_points(pose, pts, nrm), r_c.forward_points(pose, pts, nrm), rtol=0, atol=0)
        vis = rng.uniform(0, 1, (p, 11, npts, 12))
        if rng.rand() < 0.2:
            vis[..., 6] = 0.25  # flat potentials: the 1e-8 guards decide
        col = rng.rand(p, 11, npts) < 0.1
        va, vb = vis.copy(), vis.copy()
        import torch
        m_c.color_point(va, torch.as_tensor(col.astype(np.float32))); r_c.color_point(vb, torch.as_tensor(col.astype(np.float32)))
        check("color_point", va, vb, rtol=0, atol=0)
        v, a = rng.normal(size=(p, npts, 3)), rng.normal(size=(p, npts, 3))
        if rng.rand() < 0.2:
            v[0] = 0.0  # zero speed: the 1e-8 guards decide
        JT = rng.normal(size=(p, npts, 9, 3))
        wc, wg = rng.uniform(0, 1, (p, npts)), rng.normal(size=(p, npts, 3))
        (ca, ga), (cb, gb) = m_c.functional_grad(v, a, JT, wc, wg), r_c.functional_grad(v, a, JT, wc, wg)
        check("functional_grad.cost", ca, cb); check("functional_grad.grad", ga, gb, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(gb).max()))
        nj = int(rng.randint(1, 10))
        org, ax, x = rng.normal(size=(p, nj, 3)), rng.normal(size=(p, nj, 3)), rng.normal(size=(npts, p, 3))
        for ty in ("revolute", "prsimatic"):
            check(f"compute_point_jacobian[{ty}]", m_c.compute_point_jacobian(org, x, ax, None, ty),
                  r_c.compute_point_jacobian(org, x, ax, None, ty), rtol=0, atol=0)
        # finite differences with the boundary terms (omg/config.py:134-187), numpy and torch float32
        import torch
        pshape = (int(rng.randint(1, 4)), int(rng.randint(1, 4)))
        dat, st_, en_ = rng.normal(size=pshape + (n, 3)), rng.normal(size=pshape + (3,)), rng.normal(size=pshape + (3,))
        for order in (1, 2):
            check(f"get_derivative[{order}]", cfg.get_derivative(dat.copy(), st_, en_, order), rcfg.get_derivative(dat.copy(), st_, en_, order),
                  rtol=1e-12, atol=1e-9)
        td, ts_, te_ = (torch.as_tensor(x, dtype=torch.float32) for x in (dat, st_, en_))
        check("get_derivative_torch", cfg.get_derivative_torch(td.clone(), ts_, te_).numpy(), rcfg.get_derivative_torch(td.clone(), ts_, te_).numpy(),
              rtol=1e-5, atol=1e-4)
        # util helpers (omg/util.py:129-135,181-220)
        from omg_planner_amd import util as mutil
        qv = rng.uniform(-3, 3, int(rng.choice([7, 9])))
        check("util.wrap_value", mutil.wrap_value(qv), util.wrap_value(qv), rtol=0, atol=0)
        qs = rng.uniform(-3, 3, (int(rng.randint(1, 5)), 9))
        check("util.wrap_values", mutil.wrap_values(qs), util.wrap_values(qs), rtol=0, atol=0)
        for j in range(1, 11):
            check("util.wrap_index", mutil.wrap_index(j), util.wrap_index(j), rtol=0, atol=0)
            check("util.wrap_joint", mutil.wrap_joint(j), util.wrap_joint(j), rtol=0, atol=0)
        RT = np.eye(4)
        RT[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        RT[:3, 3] = rng.normal(size=3)
        inv_a, inv_b = mutil.se3_inverse(RT), util.se3_inverse(RT)
        check("util.se3_inverse", inv_a, inv_b, rtol=0, atol=0)
        check("util.se3_inverse.dtype", np.array([inv_a.dtype == inv_b.dtype]), np.array([True]), rtol=0, atol=0)
        check("util.rad2deg", mutil.rad2deg(qv), util.rad2deg(qv), rtol=0, atol=0)
        check("util.deg2rad", mutil.deg2rad(qv * 50), util.deg2rad(qv * 50), rtol=0, atol=0)
        check("util.safe_div", mutil.safe_div(qv, qv[::-1]), util.safe_div(qv, qv[::-1]), rtol=0, atol=0)
        # Trajectory initialisation (omg/util.py:238-258 through scipy): linear bit-identical, cubic to round-off
        from omg_planner_amd import scenes as msc
        a9, b9 = rng.uniform(-3, 3, 9), rng.uniform(-3, 3, 9)
        check("interpolate[linear]", msc.linear_init(a9, b9, n), util.interpolate_waypoints(np.stack([a9, b9]), n, 9, "linear"), rtol=0, atol=0)
        check("interpolate[cubic]", msc.cubic_init(a9, b9, n), util.interpolate_waypoints(np.stack([a9, b9]), n, 9, "cubic"), rtol=0, atol=1e-14)
        if k % 10 == 0:  # the SDF volume file format: our writer -> the reference's reader, and our reader on the same file
            dims = tuple(int(d) for d in rng.randint(2, 12, 3))
            grid = SdfGrid(rng.normal(0, 0.1, dims).astype(np.float32), rng.uniform(-0.3, 0.0, 3), float(rng.choice([0.01, 0.02, 0.005])))
            path = f"{tmpdir}/m{k}.pth"
            scene_io.save_sdf_pth(path, grid)
            ref = sdf_tools.SignedDensityField.from_pth(path)
            ratio = float(rng.choice([1.0, 1.0, 0.6, 1.3]))
            if ratio != 1.0:
                ref.data = ref.data.copy()
                ref.origin = ref.origin.copy()
                ref.resize(ratio)
            ours = scene_io.load_sdf_pth(path, resize=ratio)
            check("pth.data", ours.data, ref.data, rtol=0, atol=0)
            check("pth.origin", ours.origin, ref.origin, rtol=1e-15, atol=0)
            check("pth.delta", ours.delta, ref.delta, rtol=1e-15, atol=0)
            check("pth.shape", np.array(ours.data.shape), np.array([ref.nx, ref.ny, ref.nz]), rtol=0, atol=0)
        if fails:
            print(f"trial {k} [n={n} dt={dt} standoff={standoff} c={c}]: FAIL " + "; ".join(fails[:4]), flush=True)
            break
    import shutil
    shutil.rmtree(tmpdir, ignore_errors=True)
    total = sum(stats.values())
    print(f"{'FAILED' if fails else 'all agree'}: {total} comparisons over {k + 1} trials ({', '.join(sorted(stats))}); {time.time() - t0:.0f} s")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
