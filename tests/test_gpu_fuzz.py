"""Short runs of the randomised differential tests in tests/fuzz/fuzz_*.py (GPU vs CPU oracle); the long campaigns are quoted in
DESIGN.md section 2.  Seeds differ from the ones used there."""
import importlib.util
from pathlib import Path

import pytest

TOOLS = Path(__file__).resolve().parent / "fuzz"
pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, TOOLS / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("tool,trials", [("fuzz_parity", 25), ("fuzz_sdf", 40), ("fuzz_chomp", 80), ("fuzz_learner", 120), ("fuzz_misc", 12)])
def test_fuzz_smoke(tool, trials, capsys):
    rc = _load(tool).main(trials=trials, seed=12345)
    out = capsys.readouterr().out
    assert rc == 0, out[-2000:]
