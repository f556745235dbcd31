"""CPU test: the oracle's trilinear value / central-difference gradient against THE REFERENCE'S OWN helpers —
layers/sdf_matching_loss_kernel.cu:15-86 (lerp, getValue, getValueInterpolated, getGradientInterpolated), compiled for the
host from the reference tree by `make -C oracle ref` into oracle/_ref/ (two builds: without and with FMA contraction).

This pins the numerically delicate part of the SDF op (the -0.5 voxel-centre shift in double, truncation toward zero,
out-of-range 1.0, one-voxel central differences divided by delta in double).  The kernel body (.cu:96-181: pose transform,
hinge, rotate-back, reduction) needs ATen + Eigen + Sophus + nvcc and is covered by the known-answer tests instead."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as orc

REF_DIR = Path(__file__).resolve().parents[1] / "oracle" / "_ref"


def _load(name):
    p = REF_DIR / name
    if not p.exists():
        pytest.skip(f"{p} not built (needs /root/reference: `make -C oracle ref`)")
    lib = C.CDLL(str(p))
    lib.ref_value_interpolated.restype = C.c_float
    lib.ref_value_interpolated.argtypes = [C.c_float] * 3 + [C.c_int] * 3 + [C.c_void_p]
    lib.ref_gradient_interpolated.restype = None
    lib.ref_gradient_interpolated.argtypes = [C.c_float] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_float, C.c_void_p]
    return lib


def _cases(rng, dims, n):
    d = np.array(dims, np.float32)
    g = rng.uniform(-2.5, 1.0, size=(n, 3)).astype(np.float32) + rng.uniform(0, 1, size=(n, 3)).astype(np.float32) * (d + 3.0)
    edge = np.array([[0.2, 3, 3], [-0.2, 3, 3], [-0.6, 3, 3], [0.5, 0.5, 0.5], [1.5, 1.5, 1.5], [d[0] - 0.5, 3, 3], [d[0] - 0.51, 3, 3],
                     [d[0] - 1.5, 3, 3], [3, d[1] - 0.49, 3], [3, 3, d[2] + 5], [1e9, 0, 0], [-1e9, 0, 0], [2.5, 2.5, 2.5]], np.float32)
    return np.concatenate([edge, g])


@pytest.mark.parametrize("build,tol_ulps", [("libsdf_ref_helpers_fma.so", 0), ("libsdf_ref_helpers.so", 4)])
def test_value_and_gradient_match_the_reference_helpers(build, tol_ulps):
    ref = _load(build)
    o = orc.lib()
    o.orc_value_interpolated.restype = C.c_float
    o.orc_value_interpolated.argtypes = [C.c_float] * 3 + [C.c_int] * 3 + [C.c_void_p]
    o.orc_gradient_interpolated.restype = None
    o.orc_gradient_interpolated.argtypes = [C.c_float] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_float, C.c_void_p]
    rng = np.random.RandomState(0)
    worst_v = worst_g = 0
    for dims, delta in (((9, 7, 11), 0.05), ((16, 16, 16), 0.03125), ((5, 24, 6), 0.0117)):
        grid = rng.normal(0.05, 0.1, size=dims).astype(np.float32)
        gp = grid.ctypes.data_as(C.c_void_p)
        a3, b3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        for gx, gy, gz in _cases(rng, dims, 4000):
            va = ref.ref_value_interpolated(gx, gy, gz, *dims, gp)
            vb = o.orc_value_interpolated(gx, gy, gz, *dims, gp)
            ref.ref_gradient_interpolated(gx, gy, gz, *dims, gp, delta, a3.ctypes.data_as(C.c_void_p))
            o.orc_gradient_interpolated(gx, gy, gz, *dims, gp, delta, b3.ctypes.data_as(C.c_void_p))
            if tol_ulps == 0:
                assert np.float32(va).view(np.int32) == np.float32(vb).view(np.int32), (gx, gy, gz, va, vb)
                assert np.array_equal(a3.view(np.int32), b3.view(np.int32)), (gx, gy, gz, a3, b3)
            else:  # without contraction lerp rounds twice: a few ulps of the value scale
                worst_v = max(worst_v, abs(va - vb))
                worst_g = max(worst_g, np.abs(a3 - b3).max())
    if tol_ulps:
        assert worst_v < 1e-7 and worst_g < 1e-4  # 1e-7 on values ~0.1; gradients are differences / delta (~1e-2)
