"""CPU tests of the two exactness claims the HIP SDF lookup relies on (csrc/omg_device.h):

1. axis_of: the reference's double-precision split  s = (double)g - 0.5; i0 = (int)s; f = (float)(s - i0)
   (layers/sdf_matching_loss_kernel.cu:39-41) equals a float32 computation with one patched edge;
2. the IEEE float32 quotient t / w equals (float)((double)t * fl64(1 / w)).
Both are checked in numpy (IEEE float32/float64 arithmetic) on adversarial neighbourhoods + random data."""
import numpy as np


def _neigh(center, k=4096):
    """All float32 values within k ulps of center."""
    c = np.float32(center)
    bits = c.view(np.int32) if c.ndim else np.array([c]).view(np.int32)[0]
    b = np.arange(-k, k + 1, dtype=np.int64) + int(bits)
    return b.astype(np.int32).view(np.float32)


def _split_double(g):
    s = g.astype(np.float64) - 0.5
    i0 = np.trunc(s)
    return i0.astype(np.int64), (s - i0).astype(np.float32)


def _split_float(g):
    s = (g - np.float32(0.5)).astype(np.float32)
    i0 = np.trunc(s)
    f = (s - i0.astype(np.float32)).astype(np.float32)
    edge = (s == np.float32(-1.0)) & (g > np.float32(-0.5))
    return np.where(edge, 0, i0).astype(np.int64), np.where(edge, np.float32(-1.0), f).astype(np.float32)


def test_float_axis_split_equals_double_split():
    rng = np.random.RandomState(0)
    vals = [rng.uniform(-4, 4, 2_000_000).astype(np.float32), rng.uniform(0, 2 ** 22, 2_000_000).astype(np.float32),
            (rng.uniform(0, 1100, 2_000_000).astype(np.float32))]
    for c in [-1.5, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.5, 63.5, 64.0, 127.5, 1023.5, 4194303.5]:
        vals.append(_neigh(c))
    g = np.concatenate(vals)
    g = g[np.isfinite(g)]
    i_d, f_d = _split_double(g)
    i_f, f_f = _split_float(g)
    inrange_d, inrange_f = i_d >= 0, i_f >= 0
    # whenever either says "index >= 0" (the only case the lookup can be in range) they must agree exactly
    m = inrange_d | inrange_f
    assert np.array_equal(i_d[m], i_f[m])
    assert np.array_equal(f_d[m].view(np.int32), f_f[m].view(np.int32))
    # and below the grid both are negative (out of range -> 1.0)
    assert np.array_equal(inrange_d, inrange_f)
    # the patched edge really occurs and really differs without the patch
    e = np.nextafter(np.float32(-0.5), np.float32(0.0))  # -0.5 + 2^-25
    assert _split_double(np.array([e], np.float32))[0][0] == 0
    assert np.trunc(np.float32(e) - np.float32(0.5)) == -1
    assert _split_float(np.array([e], np.float32))[0][0] == 0


def test_double_reciprocal_gives_ieee_float_quotient():
    rng = np.random.RandomState(1)
    n = 4_000_000
    w = np.concatenate([rng.uniform(1e-3, 4.0, n).astype(np.float32), _neigh(0.6, 2000), _neigh(1.0, 2000), _neigh(1.5, 2000)])
    t = rng.uniform(-2.0, 6.0, w.size).astype(np.float32)
    # adversarial numerators: integer multiples and near-midpoint quotients
    t[: n // 4] = (np.round(t[: n // 4] / w[: n // 4] * 64) / 64 * w[: n // 4]).astype(np.float32)
    q_ref = (t / w).astype(np.float32)
    q_new = (t.astype(np.float64) * (1.0 / w.astype(np.float64))).astype(np.float32)
    assert np.array_equal(q_ref.view(np.int32), q_new.view(np.int32))


def test_double_reciprocal_gives_reference_gradient_quotient():
    """getGradientInterpolated (.cu:82-84): (float)(0.5 * (double)(f_p - f_m) / (double)delta) equals
    (float)(0.5 * (double)(f_p - f_m) * fl64(1 / delta))."""
    rng = np.random.RandomState(2)
    n = 4_000_000
    d = (rng.uniform(-0.05, 0.05, n).astype(np.float32) - rng.uniform(-0.05, 0.05, n).astype(np.float32)).astype(np.float32)
    delta = rng.choice(np.array([0.6 / 64, 0.01, 0.0125, 0.02, 0.03, 0.03125, 0.05, 1.5 / 128], np.float32), n)
    ref = (0.5 * d.astype(np.float64) / delta.astype(np.float64)).astype(np.float32)
    new = (0.5 * d.astype(np.float64) * (1.0 / delta.astype(np.float64))).astype(np.float32)
    assert np.array_equal(ref.view(np.int32), new.view(np.int32))
