#!/usr/bin/env python3
"""Re-encode the Panda kinematic constants of the reference's robot_p3.pkl as a plain .npz.

Run once in the build container (the reference tree does not exist on the GPU box):

    python tools/convert_robot_pkl.py /root/reference/ycb_render/robotPose/robot_p3.pkl

Only numeric tables are kept (data, not code): _pose_0, _tip2joint, center_offset [10,4,4],
_joint_axis [10,3] and the URDF joint limits in the 9-dof order used by omg/core.py:152-164
(the dummy hand joint removed).  `_joint_origin` is deliberately NOT exported: the reference never
reads it (robot_pykdl.py:104 loads `_joint_axis` under that name).
"""
import pickle
import sys
from pathlib import Path

import numpy as np


class _NumericOnly(pickle.Unpickler):
    """The file comes from an untrusted tree: only numpy's array reconstruction and builtin containers may be unpickled."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
                ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}")
        return super().find_class(module, name)


def main(src: str) -> None:
    with open(src, "rb") as fid:
        info = _NumericOnly(fid, encoding="latin1").load()
    names = list(info["_joint_name"])
    del names[-3]  # remove the dummy hand joint, omg/core.py:154
    limits = np.array([info["_joint_limits"][n] for n in names], dtype=np.float64)  # [9,2]
    out = Path(__file__).resolve().parent.parent / "omg-planner_amd" / "data" / "panda_fk.npz"
    np.savez(
        out,
        pose_0=np.asarray(info["_pose_0"], dtype=np.float64),
        tip2joint=np.asarray(info["_tip2joint"], dtype=np.float64),
        center_offset=np.asarray(info["center_offset"], dtype=np.float64),
        joint_axis=np.asarray(info["_joint_axis"], dtype=np.float64),
        joint_limits=limits,
        joint_names=np.array(names),
        link_names=np.array(info["_link_names"]),
    )
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ycb_render/robotPose/robot_p3.pkl")
