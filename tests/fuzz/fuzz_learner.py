"""Randomised differential test of omgx_goal_update on the GPU box against the CPU oracle: all five rules, 1..256 goals,
sequences of updates that accumulate state (sum of costs, expert distributions, mixture weights), goal costs that are
sparse / equal (ties) / tiny / huge / all zero (NaN cost vector), with and without standoff tails and normalisation.

    python tests/fuzz/fuzz_learner.py [trials] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from omg_planner_amd import _lib, ops
from oracle import oracle as orc

STATS = {"updates": 0, "nan_vectors": 0, "tied_choices": 0}


def trial(rng, dev):
    S, G = int(rng.randint(1, 5)), int(rng.choice([1, 2, 3, 7, 16, 63, 64, 65, 128, 200, 256]))
    n = int(rng.choice([5, 30, 50]))
    alg = str(rng.choice(["FTL", "FTC", "Exp", "MD", "MD", "Proj"]))
    standoff = bool(rng.rand() < 0.3)
    c = int(rng.randint(1, 6)) if standoff else 1
    steps = int(rng.randint(1, 13))
    traj = rng.uniform(-2, 2, (S, n, 9))
    goals = rng.uniform(-2, 2, (S, G, 9))
    if rng.rand() < 0.2:
        goals[:, 1:] = goals[:, :1] + rng.normal(0, 1e-3, (S, max(G - 1, 0), 9))  # nearly identical goals
    reach = rng.uniform(-2, 2, (S, G, c, 9)) if standoff else None
    pd, po = _lib.LearnerParams(), orc.LearnerParams()
    base = dict(alg=_lib.ALG[alg], num_goals=G, n_waypoints=n, constraint_num=c, use_standoff=int(standoff),
                normalize_cost=int(rng.rand() < 0.85), base_obstacle_weight=float(rng.choice([1.0, 0.1, 10.0])),
                smooth_weight=float(rng.choice([0.01, 0.0, 1.0])), eta=float(np.sqrt(np.log(G + 1) / rng.choice([5, 50]))))
    st_ref = orc.learner_state_init(S, G)
    st = ops.learner_state(S, G, dev)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)  # noqa: E731
    tj, gs, rc_ = t(traj), t(goals), (t(reach) if standoff else None)
    idx = torch.zeros(S, dtype=torch.int32, device=dev)
    end, rows, gp = (torch.zeros((S, 9), dtype=torch.float64, device=dev), torch.zeros((S, c, 9), dtype=torch.float64, device=dev),
                     torch.zeros((S, 9), dtype=torch.float64, device=dev))
    cv = torch.zeros((S, G), dtype=torch.float64, device=dev)
    errs = []
    for k in range(steps):
        kind = rng.randint(0, 6)
        if kind == 0:
            gc = rng.uniform(0, 5, (S, G))
        elif kind == 1:
            gc = rng.uniform(0, 5, (S, G)) * (rng.rand(S, G) < 0.2)
        elif kind == 2:
            gc = np.full((S, G), float(rng.choice([0.0, 1.0, 37.5])))
        elif kind == 3:
            gc = rng.randint(0, 3, (S, G)).astype(np.float64)  # many ties
        elif kind == 4:
            gc = rng.uniform(0, 1e-6, (S, G))
        else:
            gc = rng.uniform(0, 1e4, (S, G))
        gc = gc.astype(np.float32)
        base["start_idx"] = int(rng.randint(0, n))
        for key, v in base.items():
            setattr(pd, key, v); setattr(po, key, v)
        if rng.rand() < 0.1 and base["smooth_weight"] == 0.0:
            gc[:] = 0  # zero cost vector -> NaN after normalisation
        r_idx, r_end, r_rows, r_gp, r_cv = orc.goal_update(po, traj, goals, reach, gc, st_ref)
        ops.goal_update(pd, tj, gs, rc_, t(gc, torch.float32), st, idx, end, rows, gp, cv)
        torch.cuda.synchronize()
        STATS["updates"] += S
        STATS["nan_vectors"] += int(np.isnan(r_cv).all(axis=1).sum()) if alg != "Proj" else 0
        tag = f"step {k} kind {kind}"
        if alg != "Proj" and not np.allclose(cv.cpu().numpy(), r_cv, rtol=1e-12, atol=0, equal_nan=True):
            errs.append(f"{tag}: cost vector")
        gi = idx.cpu().numpy()
        if not np.array_equal(gi, r_idx):
            # a different index is legitimate only if the two candidates tie to round-off in the oracle's own score
            p_ref = st_ref[:, G:2 * G]
            ok = all(abs(p_ref[s, gi[s]] - p_ref[s, r_idx[s]]) <= 1e-9 * max(1e-300, abs(p_ref[s, r_idx[s]]))
                     for s in range(S) if gi[s] != r_idx[s]) and alg in ("Exp", "MD")
            if ok:
                STATS["tied_choices"] += 1
                return errs, f"S={S} G={G} alg={alg} steps={steps}"  # states diverge from here on: stop this trial
            pg = st.cpu().numpy()[:, G:2 * G]
            errs.append(f"{tag}: goal index {gi} vs {r_idx}; p_gpu at (gpu, ref) choice " +
                        str([(float(pg[s, gi[s]]), float(pg[s, r_idx[s]])) for s in range(S) if gi[s] != r_idx[s]]) + " p_ref " +
                        str([(float(p_ref[s, gi[s]]), float(p_ref[s, r_idx[s]])) for s in range(S) if gi[s] != r_idx[s]]) +
                        f" nan gpu {int(np.isnan(pg).sum())} ref {int(np.isnan(p_ref).sum())}")
            break
        if not np.array_equal(end.cpu().numpy(), r_end) or not np.array_equal(rows.cpu().numpy(), r_rows):
            errs.append(f"{tag}: goal rows")
        if not np.allclose(st.cpu().numpy(), st_ref, rtol=1e-5, atol=1e-8, equal_nan=True):
            errs.append(f"{tag}: state differs by {np.nanmax(np.abs(st.cpu().numpy() - st_ref)):.2e}")
            break
    return errs, f"S={S} G={G} n={n} alg={alg} standoff={standoff} c={c} steps={steps} norm={base['normalize_cost']}"


def main(trials=None, seed=None):
    trials = int(trials if trials is not None else (sys.argv[1] if len(sys.argv) > 1 else 50))
    seed = int(seed if seed is not None else (sys.argv[2] if len(sys.argv) > 2 else 0))
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda:0")
    bad, t0 = 0, time.time()
    for k in range(trials):
        try:
            errs, tag = trial(rng, dev)
        except Exception as e:  # noqa: BLE001
            errs, tag = [f"exception {type(e).__name__}: {e}"], "?"
        if errs:
            bad += 1
            print(f"trial {k} [{tag}]: FAIL " + "; ".join(errs[:3]), flush=True)
    print(f"{trials - bad}/{trials} trials agree; {STATS['updates']} updates, {STATS['nan_vectors']} with a NaN cost vector, "
          f"{STATS['tied_choices']} round-off ties in the arg-max; {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
