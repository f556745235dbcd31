"""ctypes loader of libomg_hip.so (the hand-written gfx950 kernels + C ABI of include/omg_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.  Nothing in
this package computes the hot path on the CPU.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = _CSRC / "libomg_hip.so"

OMGX_OK, OMGX_ERR_INVALID, OMGX_ERR_LAUNCH, OMGX_ERR_UNSUPPORTED = 0, -1, -2, -3
NUM_DOF, INFO_STRIDE = 9, 16
SCHEDULE_MAX_SCENES = 1792  # OMGX_SCHEDULE_MAX_SCENES
SCHEDULE_SCENE_MAJOR, SCHEDULE_LONGEST_FIRST = 0, 1  # OMGX_SCHEDULE_*: the order inside an XCD (omgx_goalset_schedule_ordered)
SCHEDULE_LONGEST_FIRST_MAX_ITEMS = 8192
ABI_VERSION = 11  # omgx_abi_version() of the library these argtypes describe

# every symbol include/omg_hip.h declares
EXPORTS = ["omgx_sdf_loss_forward", "omgx_fk_sdf_workspace_bytes", "omgx_fk_sdf", "omgx_forward_kinematics", "omgx_pose_table",
           "omgx_goalset_workspace_bytes", "omgx_goalset_cost", "omgx_goalset_cost_layer", "omgx_goalset_parts", "omgx_goalset_cost_layer_tiled", "omgx_goalset_cost_layer_parts", "omgx_goalset_schedule_len", "omgx_goalset_schedule", "omgx_goalset_schedule_parts", "omgx_goalset_schedule_ordered", "omgx_region_scratch_bytes", "omgx_object_set_grid", "omgx_fit_influence_region", "omgx_regions_scratch_bytes", "omgx_fit_influence_regions", "omgx_volume_hashes", "omgx_chomp_aux_doubles", "omgx_chomp_optimize",
           "omgx_learner_state_doubles", "omgx_goal_update", "omgx_goal_update_optimize", "omgx_point_cloud_sdf", "omgx_last_error", "omgx_abi_version", "omgx_device_arch", "omgx_device_cu_count", "omgx_download_sync",
           "omgx_timing_enable", "omgx_timing_collect", "omgx_plan_persistent_workspace_bytes", "omgx_plan_persistent", "omgx_plan_persistent_status"]


class OmgHipError(RuntimeError):
    pass


class ChompParams(C.Structure):
    """Mirror of `omgx_chomp_params` (include/omg_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_waypoints", "n_points", "top_k", "consider_finger", "goal_set_proj", "constraint_num",
        "use_standoff", "uncheck_finger_collision", "joint_limit_max_steps", "allow_collision_point",
        "pre_terminate", "do_update")] + [(n, C.c_double) for n in (
        "time_interval", "obstacle_weight", "smoothness_weight", "step_size", "clip_grad_scale",
        "terminate_smooth_loss")] + [("link_smooth_weight", C.c_double * NUM_DOF)] + [(n, C.c_void_p) for n in (
        "waypoint_poses", "start_poses", "end_poses")]


class LearnerParams(C.Structure):
    """Mirror of `omgx_learner_params` (include/omg_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("alg", "num_goals", "n_waypoints", "start_idx", "constraint_num", "use_standoff",
                                          "normalize_cost", "cost_parts")] + [(n, C.c_double) for n in (
        "base_obstacle_weight", "smooth_weight", "eta")] + [(n, C.c_void_p) for n in ("goal_pose_table", "end_poses_out")]


class PlanIter(C.Structure):
    """Mirror of `omgx_plan_iter` (include/omg_hip.h): one iteration of omgx_plan_persistent."""
    _fields_ = [(n, C.c_int32) for n in ("mode", "start_idx", "stop_on_terminate", "do_update")] + [(n, C.c_double) for n in (
        "obstacle_weight", "smoothness_weight", "step_size")]


ALG = {"FTL": 0, "FTC": 1, "Exp": 2, "MD": 3, "Proj": 4}


def build(force: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", str(_CSRC)] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise OmgHipError(f"{LIB_PATH} is missing: build it with `make -C {_CSRC}` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            # torch first: its HIP runtime must be the one in the process before this library binds to it (loaded the other way
            # round — build() followed by smoke() in one process — the two runtimes disagree and no device is found)
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(str(LIB_PATH))
        vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        l.omgx_sdf_loss_forward.argtypes = [vp] * 8 + [i64, i32] + [vp] * 3 + [vp]
        l.omgx_fk_sdf_workspace_bytes.argtypes = [i32, i32, i32]
        l.omgx_fk_sdf_workspace_bytes.restype = i64
        l.omgx_fk_sdf.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, f64, vp, vp, vp, vp, vp]
        l.omgx_forward_kinematics.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp]
        l.omgx_chomp_aux_doubles.argtypes = [i32]
        l.omgx_chomp_aux_doubles.restype = i64
        l.omgx_goalset_workspace_bytes.argtypes = [i32, i32, i32, i32]
        l.omgx_goalset_workspace_bytes.restype = i64
        l.omgx_goalset_cost.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, i32, i32, i32, f64, i32, vp, vp, vp, vp, vp, vp, vp]
        l.omgx_goalset_cost_layer.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, i32, i32, i32, f64, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        l.omgx_goalset_cost_layer.restype = C.c_int
        l.omgx_goalset_parts.argtypes = [i32, i32]
        l.omgx_goalset_parts.restype = i32
        l.omgx_goalset_cost_layer_tiled.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, i32, i32, i32, f64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        l.omgx_goalset_cost_layer_parts.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, i32, i32, i32, f64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp]
        l.omgx_goalset_cost_layer_parts.restype = C.c_int
        l.omgx_goalset_schedule_parts.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
        l.omgx_goalset_schedule_parts.restype = C.c_int
        l.omgx_goalset_schedule_ordered.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        l.omgx_goalset_schedule_ordered.restype = C.c_int
        l.omgx_region_scratch_bytes.argtypes = [i32, i32, i32]
        l.omgx_region_scratch_bytes.restype = i64
        l.omgx_object_set_grid.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32), C.c_float, i64, vp]
        l.omgx_object_set_grid.restype = C.c_int
        l.omgx_fit_influence_region.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, vp, vp]
        l.omgx_fit_influence_region.restype = C.c_int
        l.omgx_regions_scratch_bytes.argtypes = [i32, i64]
        l.omgx_regions_scratch_bytes.restype = i64
        l.omgx_fit_influence_regions.argtypes = [vp, i32, vp, vp, vp, i32, i64, vp, vp, vp]
        l.omgx_fit_influence_regions.restype = C.c_int
        l.omgx_volume_hashes.argtypes = [vp, i32, vp, vp, vp]
        l.omgx_volume_hashes.restype = C.c_int
        l.omgx_pose_table.argtypes = [vp, i32, vp, i64, vp, vp]
        l.omgx_pose_table.restype = C.c_int
        l.omgx_goalset_cost_layer_tiled.restype = C.c_int
        l.omgx_goalset_schedule_len.argtypes = [i32, i32, i32]
        l.omgx_goalset_schedule_len.restype = i32
        l.omgx_goalset_schedule.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
        l.omgx_goalset_schedule.restype = C.c_int
        l.omgx_chomp_optimize.argtypes = [vp, C.POINTER(ChompParams)] + [vp] * 9 + [i32] + [vp] * 4 + [i32, vp]
        l.omgx_learner_state_doubles.argtypes = [i32]
        l.omgx_learner_state_doubles.restype = i64
        l.omgx_goal_update.argtypes = [C.POINTER(LearnerParams)] + [vp] * 5 + [i32] + [vp] * 5 + [vp, vp, vp, vp]
        l.omgx_goal_update.restype = C.c_int
        l.omgx_goal_update_optimize.argtypes = ([C.POINTER(LearnerParams)] + [vp] * 6 + [vp, C.POINTER(ChompParams)] + [vp] * 9 +
                                                [i32] + [vp] * 4 + [vp, i32, i32, vp, vp] + [vp])
        l.omgx_goal_update_optimize.restype = C.c_int
        l.omgx_point_cloud_sdf.argtypes = [vp, i32, C.POINTER(C.c_double), f64, C.POINTER(i32), vp, vp]
        l.omgx_point_cloud_sdf.restype = C.c_int
        l.omgx_last_error.restype = C.c_char_p
        l.omgx_device_arch.argtypes = [C.c_char_p, i32]
        l.omgx_device_cu_count.restype = i32
        l.omgx_download_sync.argtypes = [vp, vp, i64, vp]
        l.omgx_download_sync.restype = C.c_int
        l.omgx_timing_enable.argtypes = [i32]
        l.omgx_timing_collect.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), i32]
        l.omgx_plan_persistent_workspace_bytes.argtypes = [i32, i32]
        l.omgx_plan_persistent_workspace_bytes.restype = i64
        l.omgx_plan_persistent.argtypes = ([vp, i32, vp, vp, vp, vp, i32, i32, f64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp] +       # goal-set batch + layer
                                           [C.POINTER(LearnerParams), vp, vp, vp, vp, vp, vp] +                                       # learner
                                           [C.POINTER(ChompParams), vp, vp, vp, vp, vp, vp, vp] +                                     # step
                                           [C.POINTER(PlanIter), vp, i32, vp, i64, i32, i32, vp])                                          # the plan
        l.omgx_plan_persistent.restype = C.c_int
        l.omgx_plan_persistent_status.argtypes = [vp, i32, C.POINTER(i32), vp]
        l.omgx_plan_persistent_status.restype = C.c_int
        for name in ("omgx_sdf_loss_forward", "omgx_fk_sdf", "omgx_forward_kinematics", "omgx_pose_table", "omgx_goalset_cost", "omgx_chomp_optimize",
                     "omgx_abi_version", "omgx_device_arch", "omgx_timing_enable", "omgx_timing_collect"):
            getattr(l, name).restype = C.c_int
        if l.omgx_abi_version() != ABI_VERSION:
            raise OmgHipError(f"{LIB_PATH} has ABI {l.omgx_abi_version()}, this package needs {ABI_VERSION}: rebuild it (make -C {_CSRC})")
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc == OMGX_OK:
        return
    if rc == OMGX_ERR_LAUNCH:
        raise OmgHipError(f"{what}: {lib().omgx_last_error().decode()}")
    names = {OMGX_ERR_INVALID: "invalid argument", OMGX_ERR_UNSUPPORTED: "size not supported by this build"}
    raise OmgHipError(f"{what}: {names.get(rc, rc)}")


def device_arch() -> str:
    buf = C.create_string_buffer(64)
    check(lib().omgx_device_arch(buf, 64), "omgx_device_arch")
    return buf.value.decode()
