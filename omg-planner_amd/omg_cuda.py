"""Drop-in for the reference's compiled extension module ``omg_cuda`` (layers/setup.py:8,
layers/omg_layers.cpp:24-49): put this directory on ``sys.path`` (or copy this file next to the
reference's ``layers/``) and the unmodified ``layers/sdf_matching_loss.py`` runs on MI355X.

    outputs = omg_cuda.sdf_loss_forward(pose_init, sdf_grids, sdf_limits, points,
                                        epsilons, padding_scales, clearances, disables)
    potentials, potential_grads, collides = outputs      # [N], [N,3], [N]  float32 device tensors

Differences from the CUDA original, all deliberate: one fused launch instead of four and no device-wide
synchronisation (the call is asynchronous on torch's current stream); the sum over objects is in index
order (deterministic) instead of atomicAdd order; bad inputs raise instead of asserting, and a launch
failure raises instead of calling exit(-1) (layers/sdf_matching_loss_kernel.cu:241-246).
"""
try:  # imported as a top-level module named `omg_cuda` (directory on sys.path) or as a package member
    from omg_planner_amd.ops import sdf_loss_forward  # noqa: F401
except ImportError:  # pragma: no cover
    from .ops import sdf_loss_forward  # noqa: F401

__all__ = ["sdf_loss_forward"]
