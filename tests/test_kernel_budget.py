"""Register / LDS budget of the dominant kernel.  k_goalset_queue<2> runs at 5 workgroups per CU with <= 96 VGPRs and at most
31 744 B of LDS per workgroup (tools/lds_occupancy_probe.hip: the CU admits 5 workgroups up to that size, 4 at 32 768 B
although the occupancy API still answers 5), and it must not spill: scratch traffic in its main loop once cost 25 % of the
kernel's speed.  This compiles the file to assembly (no GPU needed) and checks the figures the
measured numbers were obtained with."""
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
    variants = [(st, lat, sp, pre, 4) for pre in (0, 1) for (st, lat, sp) in ((0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0))]
    variants += [(st, 0, 0, 0, w) for w in (6, 8) for st in (0, 1)]  # round 6: the WIDE instantiations (six / eight waves: 3 / 2 workgroups per CU)
    for st, lat, sp, pre, w in variants:  # batch, measuring, batch with split goals (x2), latency mode — each with its own kinematics and behind the pre-pass
        name = f"_Z15k_goalset_queueILi2ELb{st}ELb{lat}ELb{sp}ELb{pre}ELi{w}EEv9ChunkArgs"  # <LB, STAMP, LAT, SPLIT, PRE, W>
        start = text.index(name + ":")
        block = text[start: text.index("; Occupancy:", start) + 40]
        vgprs = int(re.search(r"; NumVgprs: (\d+)", block).group(1))
        scratch = int(re.search(r"; ScratchSize: (\d+)", block).group(1))
        occupancy = int(re.search(r"; Occupancy: (\d+)", block).group(1))
        if not lat:  # the latency-mode variant (LAT = true) runs one or two workgroups per CU
            assert vgprs <= 96 and occupancy >= 5, (name, vgprs, occupancy)
        assert scratch == 0, f"{name} spills {scratch} bytes per lane: the goal path must stay in registers"


def test_goalset_kernel_lds_fits_five_workgroups_per_cu():
    """The launcher's LDS layout for the bench shape (30 waypoints, 15 points per link) — restated from GqLayout /
    gq_choose_tbl_n (omg_goalset_queue.h) — stays within the 31 744 B that still admit 5 workgroups per CU, with 8
    exact-path records staged; the largest shape the ABI accepts (64 waypoints x 16 points) stays below the 64 KB a launch may
    ask for without opting in."""
    src = (ROOT / "omg-planner_amd" / "csrc" / "omg_goalset_queue.h").read_text()
    steps = [int(x) for x in re.search(r"static const int step\[\] = \{([^}]*)\}", src).group(1).split(",")]
    tbl_max = int(re.search(r"#define GQ_TBL_MAX (\d+)", src).group(1))

    def total(PS, MR, P, n):
        mask_off = PS * 90 * 8
        tbl_off = mask_off + ((10 * MR * 4 + 16 + 15) & ~15)  # row masks + the tile bits
        pts_off = tbl_off + n * 64
        stage_off = pts_off + ((10 * P * 3 * 8 + 15) & ~15)
        return max(stage_off + 4 * 64 * 16, PS * 90 * 8 + PS * 14 * 8)

    def choose(PS, MR, P):
        base = total(PS, MR, P, 0)
        for st in steps:
            if base + 4 * 64 <= st:
                return min(tbl_max, (st - base) // 64)
        return 0

    n = choose(31, 30, 15)
    assert n == 8 and total(31, 30, 15, n) <= 31744
    assert total(65, 64, 16, choose(65, 64, 16)) <= 64 * 1024
    assert "__shared__ float" not in src  # no static LDS on top of the dynamic allocation


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_update_kernels_have_no_static_lds(tmp_path):
    """k_update_optimize_split / k_update_optimize / k_chomp_optimize ask for up to all of a CU's LDS as DYNAMIC memory
    (hipFuncSetAttribute(..., 160 KB)): any static allocation on top — a `__shared__` variable, or a library helper that brings one,
    like __syncthreads_or — makes that call fail on the device.  Checked here on the compiled metadata, without a GPU."""
    out = tmp_path / "omg_chomp.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT / 'include'}",
             f"-I{ROOT / 'omg-planner_amd' / 'csrc'}", "--cuda-device-only", "-S"]
    subprocess.run([HIPCC, *flags, str(ROOT / "omg-planner_amd" / "csrc" / "omg_chomp.hip"), "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    seen = 0
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+)[\s\S]*?\.name:\s+(\S+)", text):
        size, name = int(m.group(1)), m.group(2)
        if "k_update_optimize" in name or "k_chomp_optimize" in name:
            seen += 1
            assert size == 0, f"{name} has {size} B of static LDS"
    assert seen >= 4, seen  # split, fused and plain kernels in both item-count variants
