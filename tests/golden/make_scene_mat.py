#!/usr/bin/env python3
"""Write the scene-file fixture tests/golden/scene_mat/ the way the REFERENCE writes and reads such files.

Writer followed key by key: bullet/gen_data.py:21-34 (`record_traj`): scene_mat["path"] = the list of object directories
(python strings of unequal length — scipy stores a space-padded char matrix, which is why omg/core.py:264 strips them),
["pose"] = the list of 4x4 object poses, ["goals"], ["target_name"] = a python string; plus ["reach_grasps"] /
["grasp_qualities"] / ["grasp_potentials"] as the shipped demo scenes carry them (read by omg/planner.py:155-174).
SDF volumes: the dict layout SignedDensityField.from_pth expects (omg/sdf_tools.py:186-193).

    python tests/golden/make_scene_mat.py     # rewrites the fixture; in the build container it also reads the volumes back
                                              # with the reference's own SignedDensityField.from_pth
The fixture is data only.  tests/test_scene_io.py checks scene_io.load_scene_mat against expected.npz.
"""
import sys
from pathlib import Path

import numpy as np
import scipy.io as sio
import torch

HERE = Path(__file__).resolve().parent
OUT = HERE / "scene_mat"
sys.path.insert(0, str(HERE.parents[1]))


def main():
    rng = np.random.RandomState(5)
    names = ["data/objects/003_cracker_box", "data/objects/table", "data/objects/025_mug"]
    poses, grids = [], []
    for k, rel in enumerate(names):
        th = rng.uniform(-3, 3)
        T = np.eye(4)
        T[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
        T[:3, 3] = rng.uniform([0.3, -0.3, 0.0], [0.7, 0.3, 0.4])
        poses.append(T)
        shape = [(10, 12, 14), (16, 12, 6), (9, 9, 11)][k]
        data = rng.uniform(-0.05, 0.3, shape).astype(np.float32)
        mn = rng.uniform(-0.2, -0.1, 3)
        delta = float(rng.choice([0.02, 0.03]))
        grids.append((data, mn, delta))
        d = OUT / rel
        d.mkdir(parents=True, exist_ok=True)
        # stored tensor = the grid with its first two axes swapped, [1,1,B,A,C] (sdf_tools.py:189 permutes them back)
        torch.save({"min_coords": torch.from_numpy(mn), "max_coords": torch.from_numpy(mn + delta * np.array(shape)), "delta": delta,
                    "sdf_torch": torch.from_numpy(data).permute(1, 0, 2)[None, None].contiguous()}, str(d / "model_normalized_chomp.pth"))
    goals = rng.uniform(-1, 1, (6, 9))
    reach = rng.uniform(-1, 1, (6, 5, 9))
    scene_mat = {}
    scene_mat["path"] = names                      # gen_data.py:26
    scene_mat["pose"] = poses                      # gen_data.py:27
    scene_mat["goals"] = goals                     # gen_data.py:32
    scene_mat["target_name"] = "025_mug"           # gen_data.py:33
    scene_mat["reach_grasps"] = reach              # demo scenes (planner.py:167)
    scene_mat["grasp_qualities"] = rng.uniform(0, 1, 6)
    scene_mat["grasp_potentials"] = rng.uniform(0, 1, 6)
    sio.savemat(str(OUT / "scene_0.mat"), scene_mat)
    np.savez(OUT / "expected.npz", names=np.array([n.split("/")[-1] for n in names]), pose=np.stack(poses), goals=goals, reach_grasps=reach,
             target_idx=np.int64(2), grasp_qualities=scene_mat["grasp_qualities"], grasp_potentials=scene_mat["grasp_potentials"],
             **{f"grid{k}": g[0] for k, g in enumerate(grids)}, **{f"origin{k}": g[1] for k, g in enumerate(grids)},
             deltas=np.array([g[2] for g in grids]))
    # the reference's reading conventions on this file (core.py:261-278, planner.py:163-171)
    scene = sio.loadmat(str(OUT / "scene_0.mat"))
    assert [p.strip() + "/" for p in scene["path"]] == [n + "/" for n in names]
    assert scene["target_name"][0] == "025_mug"
    assert scene["pose"].shape == (3, 4, 4) and scene["grasp_qualities"][0].shape == (6,)
    ref = Path("/root/reference/omg/sdf_tools.py")
    if ref.exists():  # build container: the reference's own reader on the fixture's volumes
        import importlib.util
        import types
        for m in ("IPython",):
            sys.modules.setdefault(m, types.ModuleType(m))
        if not hasattr(np, "int"):
            np.int = int
        torch.Tensor.cuda = lambda self, *a, **k: self  # the reference constructor uploads the volume (sdf_tools.py:31)
        spec = importlib.util.spec_from_file_location("ref_sdf_tools", ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for k, rel in enumerate(names):
            s = mod.SignedDensityField.from_pth(str(OUT / rel / "model_normalized_chomp.pth"))
            assert np.array_equal(np.asarray(s.data), grids[k][0]) and np.allclose(np.asarray(s.origin), grids[k][1]) and float(s.delta) == grids[k][2]
        print("reference SignedDensityField.from_pth reads the fixture's volumes identically")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
