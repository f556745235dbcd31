"""MI355X-native CHOMP trajectory-update engine for OMG-Planner's hot path.

Layout (only what the path needs):
  csrc/        hand-written HIP kernels for gfx950 + the C ABI of include/omg_hip.h  -> libomg_hip.so
  _lib.py      ctypes loader of libomg_hip.so (fails loudly when the library is missing)
  ops.py       thin tensor-level wrappers over the C ABI (torch tensors own the device memory)
  cost.py      `Cost`      — host-side mirror of omg/cost.py's class surface
  optimizer.py `Optimizer` — host-side mirror of omg/optimizer.py's class surface
  online_learner.py `Learner` — host-side mirror of omg/online_learner.py's class surface (goal selection on the device)
  config.py    `cfg`       — the hyper-parameters the path reads (omg/config.py)
  trajectory.py / util.py   — `Trajectory` container and the index/angle helpers of the path
  engine.py    `ChompEngine` — batched, device-resident planner loop over S scenes (+ sharding over ranks)
  robot.py / scenes.py      — robot constants, SDF volume layouts, synthetic scenes
  scene_io.py               — the reference's scene .mat / SDF .pth file formats
  omg_cuda.py  drop-in for the reference's `omg_cuda` extension module (sdf_loss_forward)

Import as ``omg_planner_amd`` (shim at the repo root).
"""
__version__ = "0.1.0"
