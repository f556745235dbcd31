"""Index / angle helpers of the path: `safe_div`, `rad2deg`, `deg2rad`, `wrap_value(s)`, `wrap_index`, `wrap_joint`, `se3_inverse`
(SURVEY.md §8a row 23).

These are the reference's helpers of the same names (omg/util.py:65-76, 129-135, 181-220) — a few lines each, called by `Cost`,
`Optimizer` and user code, so names, argument meaning and results ARE the interface; the results are reproduced exactly
(`tests/test_host_helpers.py`; `tests/fuzz/fuzz_host_mirror.py` against the reference's own functions in the build container);
the bodies are written from the tables below rather than as the reference's if-chains.

The Panda is addressed two ways: the trajectory has 9 columns (7 arm joints + 2 fingers) while the kinematics tables have 10
entries (a dummy hand joint at index 7).  Link numbers are 1-based: 1..7 arm links, 8 hand, 9 left finger, 10 right finger.
"""
from __future__ import annotations

import numpy as np

_ARM = list(range(7))
# link number -> trajectory columns that move it (fingers: the arm + their own column 7 / 8)
_TRAJECTORY_COLUMNS = {8: _ARM, 9: _ARM + [7], 10: _ARM + [8]}
# link number -> entries of the 10-joint tables that move it (index 7 is the fixed hand joint)
_TABLE_ENTRIES = {8: _ARM, 9: _ARM + [8], 10: _ARM + [9]}


def safe_div(dividend, divisor, eps=1e-8):
    """a / (b + eps)  (omg/util.py:181-182)."""
    return dividend / (divisor + eps)


def _elementwise(f, x):
    return [f(v) for v in x] if type(x) is list else f(x)


def rad2deg(rad):
    """rad / pi * 180 in this order (omg/util.py:73-76) — NOT np.rad2deg, which multiplies by the constant 180 / pi and differs in
    the last bit; the kinematics round-trip angles through this expression."""
    return _elementwise(lambda v: v / np.pi * 180, rad)


def deg2rad(deg):
    """deg / 180 * pi in this order (omg/util.py:67-70)."""
    return _elementwise(lambda v: v / 180.0 * np.pi, deg)


def _with_hand_joint(value):
    """Degrees, with a zero inserted for the hand joint when the fingers are present (last axis 9 -> 10)."""
    value = np.asarray(value, dtype=np.float64)
    deg = rad2deg(value)
    return deg if value.shape[-1] <= 7 else np.insert(deg, 7, 0.0, axis=-1)


def wrap_value(value):
    """One configuration: radians -> degrees, 9 -> 10 entries (util.py:185-191)."""
    return _with_hand_joint(value)


def wrap_values(value):
    """A batch [B, dof] of configurations (util.py:194-202)."""
    return _with_hand_joint(value)


def wrap_index(value):
    """Trajectory columns driven by link number `value` (util.py:205-210; numbers beyond the table fall back to its arithmetic)."""
    if value in _TRAJECTORY_COLUMNS:
        return list(_TRAJECTORY_COLUMNS[value])
    return list(range(value - 1 if value > 7 else value))


def wrap_joint(value):
    """Entries of the 10-joint tables that move link number `value` (util.py:213-220)."""
    return list(_TABLE_ENTRIES.get(value, range(value)))


def se3_inverse(RT):
    """[R t]^-1 as float32 (util.py:129-135)."""
    from .scenes import se3_inverse as _inv
    return _inv(np.asarray(RT))
