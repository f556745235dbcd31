#!/usr/bin/env python3
"""bench.py on a VARIANT build of the library (measurement builds only: counters, ablations — never the shipped one).

    make -C omg-planner_amd/csrc BUILD=build_noexact OUT=libomg_hip_noexact.so EXTRA=-DOMGX_GS_NO_EXACT=1
    python tools/bench_variant.py libomg_hip_noexact.so --steps 5 --warmup 1 --no-cpu-baseline --no-plan --no-parity
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from omg_planner_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / sys.argv[1]
sys.argv = [str(ROOT / "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
