"""CPU test: libomg_hip.so loads without a GPU and exports every function include/omg_hip.h declares
(no compute calls), and the POD structs have the sizes the header documents."""
import ctypes as C
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def _declared_functions():
    text = (ROOT / "include" / "omg_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omgx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from omg_planner_amd import _lib
    names = _declared_functions()
    assert len(names) >= 12 and "omgx_sdf_loss_forward" in names
    lib = _lib.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names  # the loader's list is the header's list
    assert lib.omgx_abi_version() >= 1


def test_struct_layouts_match_header():
    from omg_planner_amd import _lib, scenes
    assert scenes.OBJECT_DTYPE.itemsize == 184
    assert scenes.OBJECT_DTYPE.fields["grid_offset"][1] == 104 and scenes.OBJECT_DTYPE.fields["inv_extent"][1] == 112
    assert C.sizeof(_lib.ChompParams) == 12 * 4 + 6 * 8 + 9 * 8 + 3 * 8   # ABI 7: three optional pose pointers at the end
    assert C.sizeof(_lib.LearnerParams) == 8 * 4 + 3 * 8 + 2 * 8          # ABI 7: goal pose table, end poses out
    from oracle import oracle as orc
    assert orc.lib().orc_sizeof_object() == 184 and orc.lib().orc_sizeof_params() == C.sizeof(_lib.ChompParams)


def test_workspace_and_aux_sizes_without_gpu():
    from omg_planner_amd import _lib
    lib = _lib.lib()
    assert lib.omgx_chomp_aux_doubles(30) == 30 * 9 + 30 * 10 + 30 * 9 + 31
    # ABI 10: the kinematics pre-pass's scratch — per goal 10 links x 9 doubles x (n + 1) configurations + 10 x n mask words
    assert lib.omgx_goalset_workspace_bytes(100, 64, 30, 15) == 100 * 64 * (90 * 31 * 8 + 300 * 4)
    assert lib.omgx_goalset_workspace_bytes(1, 1, 1, 15) == 1488 and lib.omgx_goalset_workspace_bytes(0, 64, 30, 15) == 0
    assert lib.omgx_fk_sdf_workspace_bytes(100, 30, 15) >= 100 * 10 * 32 * 12 * 8
    assert lib.omgx_fk_sdf_workspace_bytes(0, 30, 15) == 0


def test_robot_blob_layout():
    from omg_planner_amd import robot as rb
    m = rb.PandaModel(seed=0)
    b = m.blob()
    P = m.points_per_link
    assert b.size == 528 + 30 * P + 356 + 30 * P
    np.testing.assert_array_equal(b[:160].reshape(10, 4, 4), m.pose_0)
    np.testing.assert_array_equal(b[528:528 + 30 * P].reshape(10, P, 3), m.collision_points)
    # derived joint-0 matrices reproduce pose_0[0] @ Rz(q) @ Rx(0)
    D = 528 + 30 * P
    U, V, W = b[D:D + 9].reshape(3, 3), b[D + 9:D + 18].reshape(3, 3), b[D + 18:D + 27].reshape(3, 3)
    q = 0.37
    Rz = np.array([[np.cos(q), -np.sin(q), 0], [np.sin(q), np.cos(q), 0], [0, 0, 1]])
    np.testing.assert_allclose(np.cos(q) * U + np.sin(q) * V + W, m.pose_0[0][:3, :3] @ Rz, atol=1e-15)


def test_goalset_parts_and_tiled_argument_checks_without_gpu():
    """Host logic of the latency-mode launch (no GPU needed: argument errors are reported before anything is launched).
    omgx_goalset_parts: the largest power of two <= goal_parts that leaves every workgroup at least 4 of the window's
    ceil(n / 4) x 5 tiles."""
    from omg_planner_amd import _lib
    lib = _lib.lib()
    want = {(30, 4): 4, (30, 8): 8, (30, 1): 1, (30, 2): 2, (12, 8): 2, (13, 4): 4, (8, 4): 2, (5, 8): 2, (4, 8): 1, (1, 8): 1, (64, 8): 8}
    for (n, p), np_ in want.items():
        assert lib.omgx_goalset_parts(n, p) == np_, (n, p)
        tiles = ((n + 3) // 4) * 5
        assert np_ == 1 or tiles // np_ >= 4
    assert lib.omgx_goalset_parts(0, 4) == 0 and lib.omgx_goalset_parts(30, 0) == 0 and lib.omgx_goalset_parts(30, 9) == 0
    d = C.c_void_p(4096)  # never dereferenced: every call below fails its checks first

    def call(goal_parts=4, lg=10, cb=4, spread=1, goals=d, traj=d, G=64, n_rem=30):
        return lib.omgx_goalset_cost_layer_tiled(d, 15, d, d, d, d, 270, goals, 1, G, n_rem, 0.1, 0, d, d, traj, 30, 0, d, d, d, None, None,
                                                 goal_parts, lg, cb, spread, None, None, None)
    assert call(goal_parts=0) == _lib.OMGX_ERR_INVALID and call(goal_parts=9) == _lib.OMGX_ERR_INVALID
    assert call(lg=3) == _lib.OMGX_ERR_INVALID and call(lg=11) == _lib.OMGX_ERR_INVALID and call(cb=-1) == _lib.OMGX_ERR_INVALID
    # (goal_parts > 1 without `spread` is the batch kernel with split goals since ABI 8: omgx_goalset_cost_layer_parts)

    def parts(goal_parts=2, traj=d, sched=None, slen=0):
        return lib.omgx_goalset_cost_layer_parts(d, 15, d, d, d, d, 270, d, 1, 64, 30, 0.1, 0, d, d, traj, 30, 0, d, d, d, None, None,
                                                 sched, slen, None, goal_parts, None, None, None)
    assert parts(traj=None) == _lib.OMGX_ERR_INVALID and parts(goal_parts=0) == _lib.OMGX_ERR_INVALID and parts(goal_parts=16) == _lib.OMGX_ERR_INVALID
    assert parts(sched=d, slen=12) == _lib.OMGX_ERR_INVALID  # a schedule's length is a multiple of 8
    assert lib.omgx_goalset_schedule_parts(None, None, None, 4, 64, 3, 2, d, None) == _lib.OMGX_ERR_INVALID
    assert lib.omgx_goalset_schedule_parts(None, None, None, 4, 64, 2, 2, None, None) == _lib.OMGX_ERR_INVALID
    assert lib.omgx_goalset_schedule_ordered(None, None, None, 4, 64, 1, 2, 2, d, None) == _lib.OMGX_ERR_INVALID  # no such order
    assert lib.omgx_goalset_schedule_ordered(None, None, None, 4, 64, 3, 2, 1, d, None) == _lib.OMGX_ERR_INVALID
    assert call(goals=None, traj=None, G=0) == _lib.OMGX_ERR_INVALID
    assert call(goals=None) == _lib.OMGX_ERR_INVALID
    assert call(n_rem=0) == _lib.OMGX_ERR_UNSUPPORTED and call(n_rem=65) == _lib.OMGX_ERR_UNSUPPORTED
    assert lib.omgx_pose_table(None, 15, d, 4, d, None) == _lib.OMGX_ERR_INVALID and lib.omgx_pose_table(d, 15, d, 0, d, None) == _lib.OMGX_OK
    assert lib.omgx_pose_table(d, 99, d, 4, d, None) == _lib.OMGX_ERR_UNSUPPORTED and lib.omgx_pose_table(d, 15, d, -1, d, None) == _lib.OMGX_ERR_INVALID
