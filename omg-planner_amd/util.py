"""Small index/angle helpers of the path (the hot-path subset of omg/util.py).

The Panda is addressed two ways: the trajectory has 9 columns (7 arm joints + 2 fingers) while the kinematics
tables have 10 entries (a dummy hand joint at index 7).  These helpers translate between the two.
"""
from __future__ import annotations

import numpy as np


def safe_div(dividend, divisor, eps=1e-8):
    """a / (b + eps)  (omg/util.py:181-182)."""
    return dividend / (divisor + eps)


def rad2deg(rad):
    """rad / pi * 180 (omg/util.py:73-76) — NOT np.rad2deg, which multiplies by the constant 180 / pi and differs in the last bit."""
    if type(rad) is list:
        return [x / np.pi * 180 for x in rad]
    return rad / np.pi * 180


def deg2rad(deg):
    """deg / 180 * pi (omg/util.py:67-70)."""
    if type(deg) is list:
        return [x / 180.0 * np.pi for x in deg]
    return deg / 180.0 * np.pi


def wrap_value(value):
    """One configuration: radians -> degrees, 9 -> 10 entries with a zero for the hand joint (util.py:185-191)."""
    value = np.asarray(value, dtype=np.float64)
    if value.shape[0] <= 7:
        return rad2deg(value)
    out = np.zeros(value.shape[0] + 1)
    out[:7] = rad2deg(value[:7])
    out[8:] = rad2deg(value[7:])
    return out


def wrap_values(value):
    """A batch [B, dof] of configurations (util.py:194-202)."""
    value = np.asarray(value, dtype=np.float64)
    if value.shape[1] <= 7:
        return rad2deg(value)
    out = np.zeros((value.shape[0], value.shape[1] + 1))
    out[:, :7] = rad2deg(value[:, :7])
    out[:, 8:] = rad2deg(value[:, 7:])
    return out


def wrap_index(value):
    """Trajectory columns driven by link number `value` (1-based): util.py:205-210."""
    if value == 10:  # right finger
        return list(range(7)) + [8]
    if value > 7:
        return list(range(value - 1))
    return list(range(value))


def wrap_joint(value):
    """Entries of the 10-joint tables that move link number `value` (1-based): util.py:213-220."""
    if value == 8:
        return list(range(7))
    if value == 9:
        return list(range(7)) + [8]
    if value == 10:
        return list(range(7)) + [9]
    return list(range(value))


def se3_inverse(RT):
    """[R t]^-1 as float32 (util.py:129-135)."""
    from .scenes import se3_inverse as _inv
    return _inv(np.asarray(RT))
