"""Register / spill budget of the dominant kernel.  k_goalset_compact<2> runs at 6 workgroups per CU only with <= 80 VGPRs, and
its main loop is sensitive to SGPR allocation: an innocent-looking second early exit near the top once grew the spill area from
144 to 172 bytes and cost 25 % of the kernel's speed.  This compiles the file to assembly (no GPU
needed) and checks the figures the measured numbers were obtained with."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_goalset_kernel_register_and_spill_budget(tmp_path):
    out = tmp_path / "omg_kernels.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT / 'include'}",
             f"-I{ROOT / 'omg-planner_amd' / 'csrc'}", "--cuda-device-only", "-S"]
    subprocess.run([HIPCC, *flags, str(ROOT / "omg-planner_amd" / "csrc" / "omg_kernels.hip"), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    start = text.index("_Z17k_goalset_compactILi2ELb0EEv9ChunkArgs:")
    block = text[start: text.index("; Occupancy:", start) + 40]
    vgprs = int(re.search(r"; NumVgprs: (\d+)", block).group(1))
    scratch = int(re.search(r"; ScratchSize: (\d+)", block).group(1))
    occupancy = int(re.search(r"; Occupancy: (\d+)", block).group(1))
    assert vgprs <= 80 and occupancy >= 6, (vgprs, occupancy)
    assert scratch <= 32, f"k_goalset_compact<2> spills {scratch} bytes per lane (budget 32): check the main loop's speed on the GPU"
