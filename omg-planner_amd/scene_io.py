"""On-disk formats of the reference's `data/` directory, read straight into the engine's scene types.

* SDF volume ``model_normalized_chomp.pth`` (omg/sdf_tools.py:186-193): a ``torch.save``d dict
  ``{min_coords, max_coords, delta, sdf_torch[1,1,A,B,C]}``; the grid the planner uses is ``sdf_torch[0,0]`` with its
  first two axes swapped (``permute(1,0,2)``), origin ``min_coords``, voxel size ``delta``.
* scene ``scene_N.mat`` (omg/core.py:258-278, omg/planner.py:155-174): ``pose [O,4,4]`` object poses, ``path [O]`` object
  directories relative to the repo root (first entry = grasp target unless ``target_name`` says otherwise),
  ``goals [G,9]``, ``reach_grasps [G,5,9]``, optional ``grasp_qualities`` / ``grasp_potentials`` / ``target_name``.

The 600 MB data set itself is not redistributable with this repository; these loaders let a user who has it plan
the reference's own scenes (`ChompEngine` or the `Cost` / `Optimizer` drop-ins).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .scenes import Scene, SceneObject, SdfGrid


def load_sdf_pth(path: str, resize: float = 1.0) -> SdfGrid:
    """SignedDensityField.from_pth (omg/sdf_tools.py:186-193) (+ .resize, :37-45)."""
    import torch
    d = torch.load(path, map_location="cpu", weights_only=True)  # tensors and a float: no pickled code is ever executed
    data = d["sdf_torch"][0, 0].permute(1, 0, 2).contiguous().numpy().astype(np.float32)
    origin = np.asarray(d["min_coords"], dtype=np.float64).copy()
    delta = float(d["delta"])
    if resize != 1.0:
        data, origin, delta = data * resize, origin * resize, delta * resize
    return SdfGrid(data, origin, delta)


def save_sdf_pth(path: str, grid: SdfGrid) -> None:
    """Inverse of load_sdf_pth (for tests and for exporting synthetic scenes in the reference's format)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(grid.data, np.float32)).permute(1, 0, 2)[None, None].contiguous()
    torch.save({"min_coords": torch.from_numpy(np.asarray(grid.min_coords)), "max_coords": torch.from_numpy(np.asarray(grid.max_coords)),
                "delta": float(grid.delta), "sdf_torch": t}, path)


@dataclass
class LoadedScene:
    scene: Scene
    goals: np.ndarray           # [G,9]   traj.goal_set
    reach_grasps: np.ndarray    # [G,c,9] target_obj.reach_grasps
    grasp_qualities: np.ndarray | None = None
    grasp_potentials: np.ndarray | None = None


def load_scene_mat(mat_path: str, root_dir: str, sdf_name: str = "model_normalized_chomp.pth", target_size: float = 1.0,
                   sdf_cache: dict | None = None) -> LoadedScene:
    """Env.__init__'s scene branch + Planner.load_goal_from_scene.  Objects share SdfGrid instances through
    `sdf_cache` (keyed by directory), so `pack_table(share_grids=True)` stores each model once."""
    import scipy.io as sio
    m = sio.loadmat(mat_path)
    poses = np.asarray(m["pose"], dtype=np.float64)
    paths = [str(p).strip() + "/" for p in m["path"]]
    cache = sdf_cache if sdf_cache is not None else {}
    objs = []
    for i, rel in enumerate(paths):
        full = os.path.join(root_dir, rel)
        if full not in cache:
            cache[full] = load_sdf_pth(os.path.join(full, sdf_name), target_size)
        name = rel.rstrip("/").split("/")[-1]  # Model.model_name: the object's directory name (omg/core.py:99)
        objs.append(SceneObject(name, poses[i], cache[full]))
    target_idx = 0
    if "target_name" in m:
        tn = str(np.asarray(m["target_name"]).ravel()[0]).strip()
        names = [o.name for o in objs]
        if tn in names:
            target_idx = names.index(tn)
    goals = np.asarray(m["goals"], dtype=np.float64) if "goals" in m else np.zeros((0, 9))
    reach = np.asarray(m["reach_grasps"], dtype=np.float64) if "reach_grasps" in m else np.zeros((0, 5, 9))
    gq = np.asarray(m["grasp_qualities"])[0] if "grasp_qualities" in m else None
    gp = np.asarray(m["grasp_potentials"])[0] if "grasp_potentials" in m else None
    return LoadedScene(Scene(objs, target_idx), goals, reach, gq, gp)


def save_scene_mat(mat_path: str, root_dir: str, scene: Scene, goals: np.ndarray, reach_grasps: np.ndarray,
                   rel_dirs: list | None = None, sdf_name: str = "model_normalized_chomp.pth") -> None:
    """Write a scene (+ its SDF volumes) in the reference's layout."""
    import scipy.io as sio
    rel_dirs = rel_dirs or [f"data/objects/{o.name}" for o in scene.objects]
    for o, rel in zip(scene.objects, rel_dirs):
        d = os.path.join(root_dir, rel)
        os.makedirs(d, exist_ok=True)
        f = os.path.join(d, sdf_name)
        if not os.path.exists(f):
            save_sdf_pth(f, o.sdf)
    sio.savemat(mat_path, {"pose": np.stack([o.pose_mat for o in scene.objects]), "path": np.array(rel_dirs),
                           "goals": goals, "reach_grasps": reach_grasps, "target_name": np.array([scene.objects[scene.target_idx].name])})
