"""GPU-box half of the device-vs-REFERENCE check on random cases: runs the fixture-based GPU tests of
tests/test_gpu_parity.py (C ABI entry points and the drop-in Cost / Optimizer classes against the reference's outputs) on
every fixture that tests/fuzz/make_random_fixtures.py wrote to tests/golden_random/.

    python tests/fuzz/check_random_fixtures.py
"""
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from tests import helpers as H  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402


def main():
    src = ROOT / "tests" / "golden_random"
    files = sorted(src.glob("*.npz"))
    if not files:
        raise SystemExit(f"no fixtures in {src}: run tests/fuzz/make_random_fixtures.py in the build container first")
    H.GOLDEN = src
    dev = torch.device("cuda:0")
    stats, bad = {}, 0

    def run(kind, name, fn):
        nonlocal bad
        s = stats.setdefault(kind, [0, 0])
        s[0] += 1
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            s[1] += 1
            bad += 1
            msg = str(e).strip().splitlines()
            print(f"FAIL {kind} {name}: {type(e).__name__} {msg[0] if msg else ''} {msg[1] if len(msg) > 1 else ''}", flush=True)
            if "-v" in sys.argv:
                traceback.print_exc()

    for f in files:
        name = f.name
        if name.startswith("cost_"):
            case = name[5:-4]
            fx = H.load(name)
            run("sdf op (bit-exact)", name, lambda: T.test_sdf_loss_forward_bit_exact_vs_oracle(dev, name))
            run("fk_sdf vs reference layer", name, lambda: T.test_fk_sdf_matches_reference_fixture(dev, name))
            run("chomp_optimize totals", name, lambda: T.test_total_loss_matches_reference_fixture(dev, case))
            if int(fx["collision_points"].shape[1]) == 15:
                run("Cost class", name, lambda: T.test_cost_class_matches_reference_fixture(dev, case))
        elif name.startswith("opt_"):
            case = name[4:-4]
            run("chomp_optimize sequence", name, lambda: T.test_optimizer_sequence_matches_reference_fixture(dev, case))
            run("Optimizer class", name, lambda: T.test_optimizer_class_matches_reference_fixture(dev, case))
        elif name.startswith("batch_"):
            case = name[6:-4]
            fx = H.load(name)
            if int(fx["arc_length"]):
                run("goalset_cost", name, lambda: T.test_goalset_cost_matches_reference_fixture(dev, case))
            run("Cost.batch_obstacle_cost", name, lambda: T.test_cost_batch_obstacle_cost_matches_reference_fixture(dev, case))
        elif name.startswith("learner_"):
            case = name[8:-4]
            run("goal_update sequence", name, lambda: T.test_goal_update_matches_reference_learner_fixture(dev, case))
    for k, (n, b) in stats.items():
        print(f"{k:28s} {n - b}/{n}")
    print(f"{len(files)} fixtures, {sum(v[0] for v in stats.values()) - bad}/{sum(v[0] for v in stats.values())} checks agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
