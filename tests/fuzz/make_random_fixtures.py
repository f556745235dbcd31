"""Build-container half of the device-vs-REFERENCE check on random cases: drives the generators of
tests/golden/make_golden.py (which RUN THE REFERENCE, /root/reference) with random parameters and writes the fixtures
to tests/golden_random/ (git-ignored; travels to the GPU box with the gpurun snapshot).  The GPU half is
tests/fuzz/check_random_fixtures.py.

    python tests/fuzz/make_random_fixtures.py [count] [seed]
"""
import importlib.util
import os
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

OUT = ROOT / "tests" / "golden_random"


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    spec = importlib.util.spec_from_file_location("make_golden", ROOT / "tests" / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    shutil.rmtree(OUT, ignore_errors=True)
    OUT.mkdir(parents=True)
    made = {"cost": 0, "opt": 0, "batch": 0, "learner": 0}

    def script(ns):
        devnull = open(os.devnull, "w")
        old = sys.stdout
        for k in range(count):
            kind = rng.choice(["cost", "opt", "batch", "learner"], p=[0.3, 0.3, 0.2, 0.2])
            seed = int(rng.randint(0, 10_000))
            tag = f"r{k:04d}"
            over = dict(allow_collision_point=int(rng.choice([5, 0, 100])), pre_terminate=bool(rng.rand() < 0.8),
                        terminate_smooth_loss=float(rng.choice([35.0, 1e9])), clip_grad_scale=float(rng.choice([10.0, 0.3])))
            sys.stdout = devnull
            try:
                if kind == "cost":
                    n = int(rng.choice([5, 12, 30, 30, 50]))
                    ns.run_cost_case(tag, seed, n, int(rng.choice([0, 60, 300, 1000])), goal_set_proj=bool(rng.rand() < 0.7),
                                     uncheck=int(rng.choice([0, -1])), consider_finger=bool(rng.rand() < 0.3),
                                     dt=(0.06 if (n == 50 and rng.rand() < 0.5) else None), attached=bool(rng.rand() < 0.2),
                                     floor=bool(rng.rand() < 0.3), wiggle=float(rng.choice([0.0, 0.01, 0.03])),
                                     use_standoff=bool(rng.rand() < 0.5), cfg_over=over)
                elif kind == "opt":
                    n = int(rng.choice([8, 12, 30, 30, 50]))
                    at_goal = bool(rng.rand() < 0.3)
                    ns.run_opt_case(tag, seed, n, int(rng.randint(1, 7)), bool(rng.rand() < 0.5), goal_set_proj=bool(rng.rand() < 0.75),
                                    top_k=int(rng.choice([0, 300, 1000])), bad_limits=(not at_goal) and bool(rng.rand() < 0.35),
                                    dt=(0.06 if (n == 50 and rng.rand() < 0.5) else None), force_update=bool(rng.rand() < 0.6),
                                    at_goal=at_goal, cfg_over=dict(over, joint_limit_max_steps=int(rng.choice([10, 2, 0]))))
                elif kind == "batch":
                    arc = bool(rng.rand() < 0.7)
                    ns.run_batch_case(tag, seed, int(rng.randint(1, 9)), int(rng.choice([1, 3, 7, 12, 30])) if arc else 1, arc,
                                      int(rng.choice([0, -1])) if arc else -1, attached=bool(rng.rand() < 0.25), floor=bool(rng.rand() < 0.3))
                else:
                    alg = str(rng.choice(["FTL", "FTC", "Exp", "MD"]))
                    ns.run_learner_case(alg, seed, int(rng.randint(2, 17)), int(rng.randint(2, 9)), bool(rng.rand() < 0.4),
                                        spread=float(rng.choice([0.12, 0.015, 0.3])), tag="_" + tag,
                                        cfg_over=dict(normalize_cost=bool(rng.rand() < 0.75), base_obstacle_weight=float(rng.choice([1.0, 5.0, 0.2])),
                                                      smoothness_base_weight=float(rng.choice([0.1, 1.0])), dist_eps=float(rng.choice([0.1, 0.5])),
                                                      optim_steps=int(rng.choice([50, 10]))))
                made[kind] += 1
            except AssertionError:
                pass  # tied potentials in the top-k set: no canonical answer (numpy's unstable argsort)
            finally:
                sys.stdout = old
        devnull.close()

    mg.main(out_dir=OUT, script=script)
    size = sum(f.stat().st_size for f in OUT.glob("*.npz"))
    print(f"{made} -> {len(list(OUT.glob('*.npz')))} fixtures, {size / 1e6:.1f} MB in {OUT}")


if __name__ == "__main__":
    main()
