"""Scene data for the CHOMP engine: SDF volumes, object tables, synthetic table-top scenes.

Two HBM layouts of the same information:

* the **reference boundary layout** of ``Env.combine_sdfs`` (omg/core.py:366-411): every grid padded
  with 1.0 to the per-scene max shape, ``sdf_torch[O,X,Y,Z]`` float32 + ``sdf_limits[O,10]`` with the
  max coordinate *stretched* to the padded shape -> :func:`pack_padded`;
* the **engine layout** of ``include/omg_hip.h``: one ragged float32 pool holding every grid of every
  scene back to back + a 128-byte ``omgx_object`` record per object + ``scene_begin[S+1]``
  -> :func:`pack_table` / :class:`SceneBatch`.

The reference's ``data/`` (YCB SDFs, scene .mat files) is a 600 MB download that is not available
offline, so benchmark and test scenes are synthetic: analytic spheres / boxes sampled at
voxel centres, a table slab, table-top poses.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# numpy mirror of `omgx_object` (include/omg_hip.h), 184 bytes
OBJECT_DTYPE = np.dtype([
    ("pose_inv", np.float32, (12,)),
    ("lo", np.float32, (3,)),
    ("hi", np.float32, (3,)),
    ("dim", np.int32, (3,)),
    ("delta", np.float32),
    ("epsilon", np.float32),
    ("padding_scale", np.float32),
    ("clearance", np.float32),
    ("disabled", np.int32),
    ("grid_offset", np.int64),
    ("inv_extent", np.float64, (3,)),
    ("rb_c", np.float32, (3,)),
    ("rb_h", np.float32, (3,)),
    ("rb_r", np.float32),
    ("rb_r2", np.float32),
    ("inv_delta", np.float64),
    ("inv_2eps", np.float32),
    ("inv_eps", np.float32),
], align=True)
assert OBJECT_DTYPE.itemsize == 184


def finish_records(rec: np.ndarray) -> np.ndarray:
    """Fill the derived fields (include/omg_hip.h): inv_extent = 1 / float64(float32(hi) - float32(lo)) and the default
    influence region — a rounded box  sum_k max(|t_k - c_k| - h_k, 0)^2 <= R^2  in offset-from-lo coordinates — as the plain
    box (R = 0) [-1.5 voxel, extent + 1.5 voxel].  A lookup is in range only for grid coordinates in (-0.5, dim - 0.5), so
    1.5 voxels of slack absorb every rounding; degenerate extents get an infinite box (never reject)."""
    w = (rec["hi"].astype(np.float32) - rec["lo"].astype(np.float32)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        rec["inv_extent"] = 1.0 / w.astype(np.float64)
        vox = w / np.maximum(rec["dim"], 1).astype(np.float32)
    ok = (w > 0) & (rec["dim"] > 0)
    with np.errstate(invalid="ignore"):
        rec["rb_c"] = np.where(ok, 0.5 * w, 0.0).astype(np.float32)
        rec["rb_h"] = np.where(ok, 0.5 * w + 1.5 * vox, np.inf).astype(np.float32)
    rec["rb_r"] = 0.0
    rec["rb_r2"] = 0.0
    with np.errstate(divide="ignore"):
        rec["inv_delta"] = 1.0 / rec["delta"].astype(np.float64)
        eps = rec["epsilon"].astype(np.float32)
        rec["inv_2eps"] = np.float32(1.0) / (np.float32(2.0) * eps)   # float32 arithmetic like the kernel's literals
        rec["inv_eps"] = np.float32(1.0) / eps
    return rec


def _needed_windows(grid: np.ndarray, epsilon: float, clearance: float):
    """(need, ge, margin): `need[w]` (shape = grid dims) says whether a lookup through WINDOW w can return a value <= epsilon
    or < clearance (anything else adds neither potential, gradient nor collision, .cu:150-171).

    Window w = (wx, wy, wz) is the trilinear polynomial of voxels w-1 .. w per axis, evaluated at fractions in [0, 1]; voxel -1 is
    the linear extension 2 v[0] - v[1].  A lookup with base index b and fractions f in [0,1)^3 uses window b + 1; on an axis
    where b == 0 the reference's trunc-toward-zero also lets f run through (-1, 0) (.cu:39-48), which is window 0 of that
    axis.  In grid coordinates g (the kernels' `gx`), window w covers g_k in [w_k - 0.5, w_k + 0.5).  A multilinear function
    takes its extrema at the vertices of a box, so the minimum over a window is the minimum of its 8 corner voxels.  A relative
    margin absorbs the float32 rounding of the lerp chain; non-finite voxels count as reachable.  `ge` = the grid in float64
    with the extension layer prepended on every axis (ge index = voxel index + 1), non-finite entries as -inf."""
    g = np.asarray(grid, np.float32).astype(np.float64)
    finite = np.isfinite(g)
    margin = 1e-5 * max(1.0, float(np.abs(g[finite]).max(initial=1.0)))
    for ax in range(3):  # prepend the extended layer on every axis
        with np.errstate(invalid="ignore"):  # inf - inf of non-finite voxels: NaN, counted as reachable below
            first = np.take(g, [0], axis=ax) * 2.0 - np.take(g, [1], axis=ax)
        g = np.concatenate([first, g], axis=ax)
    ge = np.where(np.isfinite(g), g, -np.inf)
    m = ge
    for ax in range(3):
        n = m.shape[ax]
        m = np.minimum(np.take(m, range(0, n - 1), axis=ax), np.take(m, range(1, n), axis=ax))
    need = (m <= float(np.float32(epsilon)) + margin) | (m < float(np.float32(clearance)) + margin)
    return need, ge, margin


def influence_range(grid: np.ndarray, epsilon: float, clearance: float):
    """Per-axis range [bmin, bmax] of the lookup BASE indices whose trilinear value can be <= epsilon or < clearance, or None
    when no base index can (the bounding box of the needed windows, see _needed_windows)."""
    if min(np.asarray(grid).shape) < 2:
        return None
    need, _, _ = _needed_windows(grid, epsilon, clearance)
    if not need.any():
        return None
    lo, hi = [], []
    for ax in range(3):
        idx = np.flatnonzero(need.any(axis=tuple(a for a in range(3) if a != ax)))
        lo.append(max(int(idx[0]) - 1, 0)); hi.append(max(int(idx[-1]) - 1, 0))  # window w <-> base w - 1 (window 0 belongs to base 0)
    return np.array(lo), np.array(hi)


RBOX_REFINE = 4  # sub-cells per axis the boundary windows are split into when the region is fitted


def influence_rbox(grid: np.ndarray, epsilon: float, clearance: float, vox: np.ndarray):
    """A rounded box (c, h, R) — in the kernels' offset coordinates t = R_obj p + t_obj - lo, metres — that contains every
    point whose lookup can add anything:  sum_k max(|t_k - c_k| - h_k, 0)^2 <= R^2.  None when nothing can.

    The set to cover is the union of the needed windows (_needed_windows), boxes g in [w - 0.5, w + 0.5] in grid coordinates;
    windows on its boundary are split RBOX_REFINE times per axis and only the sub-boxes that can still reach the thresholds are
    kept — then merged into one bounding box per window — (the interpolant restricted to a sub-box is multilinear too: its minimum sits at a sub-box vertex).  For a given inner
    box the smallest admissible R is the largest distance from the inner box to the far corner of any kept box — the distance
    to a box is convex, so its maximum over the union is attained at such a corner.  The inner box is chosen among a small
    family (the bounding box of the needed windows eroded by 0 .. its smallest half-width; the bounding box of the non-positive
    voxels shrunk towards its centre) for the smallest volume of the resulting rounded box: a sphere-like object gets a ball, a
    box-like object its own box grown by epsilon with rounded edges, a volume that fills its grid the plain grid box (R = 0).
    Conservative for ANY volume — a field that is no distance field just gets a large R.  `vox` = voxel extents per axis."""
    dim = np.array(np.asarray(grid).shape)
    if dim.min() < 2:
        return None
    vox = np.asarray(vox, np.float64)
    need, ge, margin = _needed_windows(grid, epsilon, clearance)
    if not need.any():
        return None
    eps_t, clr_t = float(np.float32(epsilon)) + margin, float(np.float32(clearance)) + margin
    # boundary windows: needed, with a face neighbour that is not (windows on the grid's faces: the outside is not needed)
    pad = np.pad(need, 1, constant_values=False)
    inner = pad[1:-1, 1:-1, 1:-1].copy()
    for ax in range(3):
        for sh in (0, 2):
            sl = [slice(1, -1)] * 3
            sl[ax] = slice(sh, sh + need.shape[ax])
            inner &= pad[tuple(sl)]
    bidx = np.argwhere(need & ~inner)
    # kept boxes of the boundary windows in grid coordinates
    r = RBOX_REFINE
    fr = np.arange(r + 1) / r
    lo_list, hi_list = [], []
    for k0 in range(0, len(bidx), 20000):  # chunks keep the lattice arrays small
        w = bidx[k0: k0 + 20000]
        c = [[[ge[w[:, 0] + a, w[:, 1] + b, w[:, 2] + cc] for cc in (0, 1)] for b in (0, 1)] for a in (0, 1)]
        fx, fy, fz = fr[None, :, None, None], fr[None, None, :, None], fr[None, None, None, :]
        e = lambda a: a[:, None, None, None]
        lerp = lambda a, b, t: a + (b - a) * t
        with np.errstate(invalid="ignore"):
            v = lerp(lerp(lerp(e(c[0][0][0]), e(c[1][0][0]), fx), lerp(e(c[0][1][0]), e(c[1][1][0]), fx), fy),
                     lerp(lerp(e(c[0][0][1]), e(c[1][0][1]), fx), lerp(e(c[0][1][1]), e(c[1][1][1]), fx), fy), fz)
        v = np.where(np.isfinite(v), v, -np.inf)  # -inf corners: inf - inf along the way
        for ax in (1, 2, 3):
            nn = v.shape[ax]
            v = np.minimum(np.take(v, range(0, nn - 1), axis=ax), np.take(v, range(1, nn), axis=ax))
        kept = (v <= eps_t) | (v < clr_t)  # [windows, r, r, r]
        # one box per window: the bounding box of its kept sub-boxes (keeps the windows' count small for the fit below)
        lo_w, hi_w = np.empty((len(w), 3)), np.empty((len(w), 3))
        for ax in range(3):
            along = kept.any(axis=tuple(a for a in (1, 2, 3) if a != ax + 1))  # [windows, r]
            first = along.argmax(1)
            last = r - 1 - along[:, ::-1].argmax(1)
            lo_w[:, ax] = w[:, ax] - 0.5 + first / r
            hi_w[:, ax] = w[:, ax] - 0.5 + (last + 1) / r
        some = kept.reshape(len(w), -1).any(1)  # (every needed window keeps at least the sub-box of its minimal corner)
        lo_list.append(lo_w[some]); hi_list.append(hi_w[some])
    # ... and, whole, the needed windows NEXT to the boundary (inner windows with a boundary window as a face neighbour).  Once the
    # boundary windows have been shrunk to their kept sub-boxes they no longer enclose what lies behind them, and a far corner of
    # such an inner window can be the farthest point of the union.  Deeper windows cannot: along every axis a deeper window lies
    # between two windows of this layer, so none of its corners is an extreme point of the union (omgx_fit_influence_region,
    # the device version of this fit, simply feeds every inner window: the same maximum).
    bnd = need & ~inner
    padb = np.pad(bnd, 1, constant_values=False)
    adj = np.zeros_like(need)
    for ax in range(3):
        for sh in (0, 2):
            sl = [slice(1, -1)] * 3
            sl[ax] = slice(sh, sh + need.shape[ax])
            adj |= padb[tuple(sl)]
    sidx = np.argwhere(inner & adj).astype(np.float64)
    if len(sidx):
        lo_list.append(sidx - 0.5); hi_list.append(sidx + 0.5)
    blo, bhi = np.concatenate(lo_list), np.concatenate(hi_list)
    if len(blo) == 0:  # cannot happen (a needed window keeps the sub-box of its minimal corner); stay safe
        widx = np.argwhere(need)
        blo, bhi = widx - 0.5, widx + 0.5
    # candidate inner boxes (centre, half-widths) in grid coordinates
    allw = np.argwhere(need)
    nlo, nhi = allw.min(0) - 0.5, allw.max(0) + 0.5
    bc, bh = (nlo + nhi) / 2, (nhi - nlo) / 2
    cands = []
    rmax = float((bh * vox).min())
    for k in range(9):
        cands.append((bc, np.maximum(bh - (rmax * k / 8.0) / vox, 0.0)))
    neg = np.argwhere(np.asarray(grid) <= 0)
    if len(neg):
        qlo, qhi = neg.min(0) + 0.5, neg.max(0) + 0.5  # voxel i sits at grid coordinate i + 0.5
        for sh in (0.0, 0.25, 0.5, 0.75, 1.0):
            cands.append(((qlo + qhi) / 2, (qhi - qlo) / 2 * (1.0 - sh)))
    best = None
    for ctr, h in cands:
        d = np.maximum(np.maximum(np.abs(blo - ctr), np.abs(bhi - ctr)) - h, 0.0) * vox
        R = float(np.sqrt((d * d).sum(1).max()))
        hm = h * vox
        vol = 8 * hm.prod() + 8 * R * (hm[0] * hm[1] + hm[1] * hm[2] + hm[0] * hm[2]) + 2 * np.pi * R * R * hm.sum() + 4.0 / 3.0 * np.pi * R ** 3
        if best is None or vol < best[0]:
            best = (vol, ctr * vox, hm, R)
    return best[1], best[2], best[3]


def tighten_far_boxes(rec: np.ndarray, pool: np.ndarray, cache: dict | None = None) -> np.ndarray:
    """Replace the default influence region of every record (the whole grid) by the fitted rounded box (influence_rbox) of
    the lookups that can matter, grown by 1 % of a voxel + 1e-6 m for the float32 rounding of the kernels' coordinates.
    Only meaningful where the kernels cull at all (epsilon < 1 and clearance <= 1: an out-of-range lookup returns 1.0).
    A record nothing can reach gets R^2 = -1 (every point is rejected).
    `cache` (optional dict) keeps the fits between calls (key: the volume's CONTENT, voxel sizes, epsilon, clearance — scenes
    with private copies of one model share the fit)."""
    import hashlib
    cache = {} if cache is None else cache
    for r in rec:
        d = r["dim"].astype(np.int64)
        w = (r["hi"].astype(np.float32) - r["lo"].astype(np.float32)).astype(np.float32)
        if not ((w > 0).all() and (d > 1).all() and r["epsilon"] < 1.0 and r["clearance"] <= 1.0):
            continue
        g = pool[int(r["grid_offset"]): int(r["grid_offset"]) + int(d.prod())]
        vox = (w / d.astype(np.float32)).astype(np.float64)
        key = (hashlib.blake2b(np.ascontiguousarray(g).tobytes(), digest_size=12).digest(), tuple(d), tuple(vox), float(r["epsilon"]), float(r["clearance"]))
        if key not in cache:
            cache[key] = influence_rbox(g.reshape(tuple(d)), key[3], key[4], vox)
        fit = cache[key]
        if fit is None:  # nothing reachable: an empty region rejects every point
            r["rb_c"] = 0.0
            r["rb_h"] = 0.0
            r["rb_r"] = 0.0
            r["rb_r2"] = -1.0
            continue
        c, h, R = fit
        R = R + 0.01 * float(vox.min()) + 1e-6
        r["rb_c"] = c.astype(np.float32)
        r["rb_h"] = np.nextafter(h.astype(np.float32), np.float32(np.inf))  # rounded up
        r["rb_r"] = np.nextafter(np.float32(R), np.float32(np.inf))
        r["rb_r2"] = np.nextafter(np.float32(float(r["rb_r"]) ** 2), np.float32(np.inf))
    return rec


@dataclass
class SdfGrid:
    """What the path reads of SignedDensityField (omg/sdf_tools.py:17-35): data[x,y,z], origin, delta."""
    data: np.ndarray  # [X,Y,Z] float32
    origin: np.ndarray  # min_coords [3]
    delta: float

    @property
    def min_coords(self) -> np.ndarray:
        return np.asarray(self.origin, dtype=np.float64)

    @property
    def max_coords(self) -> np.ndarray:  # sdf_tools.py:30
        return self.min_coords + self.delta * np.array(self.data.shape)


@dataclass
class SceneObject:
    """What the path reads of Model (omg/core.py:83-141): name, pose_mat, sdf, attached."""
    name: str
    pose_mat: np.ndarray  # [4,4] object -> world
    sdf: SdfGrid
    attached: bool = False


@dataclass
class Scene:
    objects: list = field(default_factory=list)
    target_idx: int = 0


def scene_from_env(env) -> Scene:
    """The `Scene` of a reference `Env` (omg/core.py:239-411), for `pack_table` / `ChompEngine`: every object's name, pose and
    attached flag and the volume the reference itself puts on the GPU — `obj.sdf.data_torch` (core.py:381), NOT `obj.sdf.data`,
    whose negative values `Model` has multiplied by cfg.penalize_constant on the numpy copy only (core.py:110) — with origin
    `sdf.min_coords` and voxel size `sdf.delta`."""
    objs = []
    for o in env.objects:
        vol = getattr(o.sdf, "data_torch", None)
        vol = np.asarray(o.sdf.data, np.float32) if vol is None else vol.detach().cpu().numpy().astype(np.float32)
        objs.append(SceneObject(str(o.name), np.array(o.pose_mat, np.float64), SdfGrid(np.ascontiguousarray(vol), np.array(o.sdf.min_coords, np.float64),
                                                                                   float(o.sdf.delta)), bool(getattr(o, "attached", False))))
    return Scene(objs, int(env.target_idx))


def se3_inverse(pose: np.ndarray) -> np.ndarray:
    """[R t]^-1 = [R^T, -R^T t] as float32 (omg/util.py:129-135)."""
    out = np.eye(4, dtype=np.float32)
    R = pose[:3, :3]
    out[:3, :3] = R.T
    out[:3, 3] = -(R.T @ pose[:3, 3])
    return out


# ------------------------------------------------------------------------------------------------
# analytic SDFs
# ------------------------------------------------------------------------------------------------
def _voxel_centres(shape, origin, delta):
    ax = [origin[i] + (np.arange(shape[i]) + 0.5) * delta for i in range(3)]
    return np.meshgrid(*ax, indexing="ij")


def sphere_sdf(radius: float, shape=(64, 64, 64), delta: float = 0.6 / 64) -> SdfGrid:
    origin = -0.5 * delta * np.array(shape, dtype=np.float64)
    x, y, z = _voxel_centres(shape, origin, delta)
    return SdfGrid((np.sqrt(x * x + y * y + z * z) - radius).astype(np.float32), origin, float(delta))


def box_sdf(half_extents, shape=(64, 64, 64), delta: float = 0.6 / 64) -> SdfGrid:
    origin = -0.5 * delta * np.array(shape, dtype=np.float64)
    x, y, z = _voxel_centres(shape, origin, delta)
    q = np.stack([np.abs(x) - half_extents[0], np.abs(y) - half_extents[1], np.abs(z) - half_extents[2]], -1)
    outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
    inside = np.minimum(q.max(-1), 0.0)
    return SdfGrid((outside + inside).astype(np.float32), origin, float(delta))


def point_cloud_sdf(points: np.ndarray, grid_resolution: float = 0.02, margin: float = 0.24) -> SdfGrid:
    """Unsigned nearest-point distance grid like PointEnv.compute_sdf_from_points (omg/core.py:426-457):
    grid nodes at arange(min - margin, max + margin, res) per axis, value = distance to nearest point."""
    from scipy.spatial import cKDTree

    lo, hi = points.min(0), points.max(0)
    ax = [np.arange(lo[i] - margin, hi[i] + margin, grid_resolution) for i in range(3)]
    g = np.stack(np.meshgrid(*ax, indexing="ij"), 0)
    d, _ = cKDTree(points).query(g.reshape(3, -1).T)
    return SdfGrid(d.reshape(g.shape[1:]).astype(np.float32), (lo - margin).astype(np.float64), float(grid_resolution))


# ------------------------------------------------------------------------------------------------
# per-object SDF-layer parameters (Cost.compute_obstacle_cost_layer, omg/cost.py:303-328)
# ------------------------------------------------------------------------------------------------
def layer_params(scene: Scene, epsilon=0.2, target_epsilon=0.1, clearance=0.01, target_clearance=0.0,
                 disable_collision_set=(), special_check_id=None):
    """-> poses_inv [O,4,4] f32, epsilons, padding_scales, clearances, disables [O] f32.

    ``special_check_id`` is accepted for signature parity; like the reference, the target is taken from
    ``scene.target_idx`` (cost.py:321 compares against ``self.env.target_idx``).
    """
    O = len(scene.objects)
    poses = np.zeros((O, 4, 4), np.float32)
    eps = np.full(O, epsilon, np.float32)
    pad = np.ones(O, np.float32)
    clr = np.full(O, clearance, np.float32)
    dis = np.zeros(O, np.float32)
    for i, ob in enumerate(scene.objects):
        if ob.name == "floor" or ob.name in disable_collision_set:
            dis[i] = 1
        poses[i] = se3_inverse(ob.pose_mat)
        if i == scene.target_idx:
            clr[i] = target_clearance
            eps[i] = target_epsilon
    if O > 0 and scene.objects[scene.target_idx].attached:  # cost.py:325-328 (table is last)
        clr[-1] = 0.0
        eps[-1] = 0.05
        pad[-1] = 0.5
    return poses, eps, pad, clr, dis


# ------------------------------------------------------------------------------------------------
# layouts
# ------------------------------------------------------------------------------------------------
def pack_padded(objects) -> tuple[np.ndarray, np.ndarray]:
    """Env.combine_sdfs (omg/core.py:366-411): -> sdf [O,X,Y,Z] f32 (pad value 1.0), limits [O,10] f32."""
    shapes = np.array([o.sdf.data.shape for o in objects])
    mx = shapes.max(0)
    sdf = np.ones((len(objects), mx[0], mx[1], mx[2]), np.float32)
    lim = np.zeros((len(objects), 10), np.float32)
    for i, o in enumerate(objects):
        sz = o.sdf.data.shape
        sdf[i, : sz[0], : sz[1], : sz[2]] = o.sdf.data.astype(np.float32)
        mn, mxc = o.sdf.min_coords, o.sdf.max_coords
        lim[i, 0:3] = mn
        for a in range(3):
            lim[i, 3 + a] = mn[a] + (mxc[a] - mn[a]) * mx[a] / sz[a]
        lim[i, 6:9] = mx
        lim[i, 9] = o.sdf.delta
    return sdf, lim


def table_from_padded(poses_inv, limits, eps, pad, clr, dis, grid_elems_per_object=None) -> np.ndarray:
    """omgx_object records that address the reference's padded [O,X,Y,Z] tensor in place."""
    O = limits.shape[0]
    rec = np.zeros(O, OBJECT_DTYPE)
    for o in range(O):
        rec[o]["pose_inv"] = np.asarray(poses_inv[o], np.float32)[:3, :4].ravel()
        rec[o]["lo"] = limits[o, 0:3]
        rec[o]["hi"] = limits[o, 3:6]
        rec[o]["dim"] = limits[o, 6:9].astype(np.int32)
        rec[o]["delta"] = limits[o, 9]
        rec[o]["epsilon"] = eps[o]
        rec[o]["padding_scale"] = pad[o]
        rec[o]["clearance"] = clr[o]
        rec[o]["disabled"] = 1 if dis[o] > 0 else 0
        d = rec[o]["dim"].astype(np.int64)
        rec[o]["grid_offset"] = o * int(d[0] * d[1] * d[2]) if grid_elems_per_object is None else o * grid_elems_per_object
    return finish_records(rec)


@dataclass
class SceneBatch:
    """Engine layout for S scenes: object records, scene_begin[S+1], one ragged float32 pool."""
    objects: np.ndarray  # OBJECT_DTYPE [sum O_s]
    scene_begin: np.ndarray  # int32 [S+1]
    pool: np.ndarray  # float32 [sum voxels]

    @property
    def num_scenes(self) -> int:
        return len(self.scene_begin) - 1

    def subset(self, first: int, last: int) -> "SceneBatch":
        """Scenes [first, last) as their own batch: records copied, scene_begin rebased, and the pool cut to
        the range those records address (grid offsets rebased)."""
        lo_o, hi_o = int(self.scene_begin[first]), int(self.scene_begin[last])
        rec = self.objects[lo_o:hi_o].copy()
        sizes = rec["dim"].astype(np.int64).prod(axis=1)
        p0 = int(rec["grid_offset"].min()) if len(rec) else 0
        p1 = int((rec["grid_offset"] + sizes).max()) if len(rec) else 0
        rec["grid_offset"] -= p0
        return SceneBatch(rec, (self.scene_begin[first:last + 1] - lo_o).astype(np.int32), self.pool[p0:p1])


def pack_table(scenes, cfg_kwargs=None, ragged: bool = True, share_grids: bool = True, tight: bool = True) -> SceneBatch:
    """Pack scenes into the engine layout.

    ragged=True keeps every grid at its own shape (true limits); ragged=False reproduces the
    reference's pad-to-max + stretched-limits layout per scene.  The two agree up to the float32
    rounding of the grid coordinate (u-lo)/(hi-lo)*dim; inside the stretched box the padded layout reads
    the 1.0 padding value where the ragged one returns the out-of-range 1.0 — the same number.
    share_grids=True stores a grid referenced by several scenes once (same ndarray object).
    tight=True shrinks each record's far box to the voxels that can contribute (tighten_far_boxes): same results,
    fewer exact lookups.
    """
    cfg_kwargs = cfg_kwargs or {}
    recs, begins, chunks, offset = [], [0], [], 0
    seen = {}
    for sc in scenes:
        poses, eps, pad, clr, dis = layer_params(sc, **cfg_kwargs)
        if ragged:
            for i, ob in enumerate(sc.objects):
                r = np.zeros((), OBJECT_DTYPE)
                r["pose_inv"] = poses[i][:3, :4].ravel()
                r["lo"] = ob.sdf.min_coords.astype(np.float32)
                # same float32 arithmetic as combine_sdfs with max_shape == size
                mn, mxc = ob.sdf.min_coords, ob.sdf.max_coords
                r["hi"] = np.array([mn[a] + (mxc[a] - mn[a]) * 1.0 for a in range(3)], np.float32)
                r["dim"] = ob.sdf.data.shape
                r["delta"] = ob.sdf.delta
                r["epsilon"], r["padding_scale"], r["clearance"] = eps[i], pad[i], clr[i]
                r["disabled"] = 1 if dis[i] > 0 else 0
                key = id(ob.sdf.data)
                if share_grids and key in seen:
                    r["grid_offset"] = seen[key]
                else:
                    r["grid_offset"] = offset
                    seen[key] = offset
                    chunks.append(np.ascontiguousarray(ob.sdf.data, np.float32).ravel())
                    offset += chunks[-1].size
                recs.append(r)
        else:
            sdf, lim = pack_padded(sc.objects)
            t = table_from_padded(poses, lim, eps, pad, clr, dis)
            t["grid_offset"] += offset
            chunks.append(sdf.ravel())
            offset += sdf.size
            recs.extend(list(t))
        begins.append(len(recs))
    rec = finish_records(np.array(recs, OBJECT_DTYPE))
    pool = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    if tight:
        tighten_far_boxes(rec, pool)
    return SceneBatch(rec, np.array(begins, np.int32), pool)


# ------------------------------------------------------------------------------------------------
# synthetic table-top scenes
# ------------------------------------------------------------------------------------------------
_SHAPE_CACHE: dict = {}


def _shape(kind: str, key, grid: int) -> SdfGrid:
    k = (kind, key, grid)
    if k not in _SHAPE_CACHE:
        delta = 0.6 / grid
        if kind == "sphere":
            _SHAPE_CACHE[k] = sphere_sdf(key, (grid,) * 3, delta)
        else:
            _SHAPE_CACHE[k] = box_sdf(key, (grid,) * 3, delta)
    return _SHAPE_CACHE[k]


def _yaw_pose(x, y, z, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    T[:3, 3] = [x, y, z]
    return T


def make_tabletop_scene(seed: int, num_objects: int = 4, grid: int = 64, table_grid=(128, 96, 32)) -> Scene:
    """`num_objects` YCB-like shapes (spheres r 6-9 cm, boxes 5-20 cm) on `grid`^3 volumes with
    delta = 0.6/grid m, plus a table slab 1.2 x 0.8 x 0.04 m as the LAST object (cost.py:325-328 treats
    objects[-1] as the table).  Object 0 is the grasp target.  A small palette of shapes is shared
    between scenes (like the 21 YCB models are shared between the reference's 100 scenes)."""
    rng = np.random.RandomState(seed)
    radii = (0.06, 0.07, 0.08, 0.09)
    boxes = ((0.025, 0.05, 0.10), (0.05, 0.05, 0.05), (0.04, 0.08, 0.03), (0.03, 0.03, 0.09))
    objs = []
    for i in range(num_objects):
        if rng.rand() < 0.5:
            sdf = _shape("sphere", radii[rng.randint(len(radii))], grid)
        else:
            sdf = _shape("box", boxes[rng.randint(len(boxes))], grid)
        pose = _yaw_pose(rng.uniform(0.3, 0.7), rng.uniform(-0.3, 0.3), 0.15 + rng.uniform(-0.02, 0.02),
                         rng.uniform(-np.pi, np.pi))
        objs.append(SceneObject(f"obj_{i}", pose, sdf))
    tk = ("table", tuple(table_grid))
    if tk not in _SHAPE_CACHE:
        tdelta = 1.5 / table_grid[0]
        _SHAPE_CACHE[tk] = box_sdf((0.6, 0.4, 0.02), table_grid, tdelta)
    objs.append(SceneObject("table", _yaw_pose(0.5, 0.0, 0.02, 0.0), _SHAPE_CACHE[tk]))
    return Scene(objs, target_idx=0)


def make_goal_set(seed: int, num_goals: int) -> np.ndarray:
    """Reach configurations near a table-top pre-grasp + N(0, 0.05^2) on the arm joints."""
    rng = np.random.RandomState(10_000 + seed)
    base = np.array([0.3, 0.2, 0.1, -1.6, 0.1, 1.9, 1.0, 0.04, 0.04])
    g = np.tile(base, (num_goals, 1))
    g[:, :7] += rng.normal(0.0, 0.05, size=(num_goals, 7))
    return g


_REACH_POOL: dict = {}


def _reach_pool(model, count: int = 400_000, seed: int = 1234):
    """Random arm configurations within the soft joint limits with their hand pose (numpy FK), cached."""
    key = (count, seed)
    if key not in _REACH_POOL:
        from scipy.spatial import cKDTree
        rng = np.random.RandomState(seed)
        lo, hi = model.joint_lower_limit[0], model.joint_upper_limit[0]
        q = rng.uniform(lo, hi, size=(count, 9))
        q[:, 7:] = 0.04
        pos, zax = hand_pose(model, q)
        keep = pos[:, 2] > 0.10
        q, pos, zax = q[keep], pos[keep], zax[keep]
        _REACH_POOL[key] = (q, pos, zax, cKDTree(pos))
    return _REACH_POOL[key]


def hand_pose(model, q: np.ndarray):
    """Position and approach (z) axis of the hand link for configurations q [B,9]: the first 8 frames of
    robot_pykdl.py:148-215 in plain numpy (host-side scene synthesis only, not on the hot path)."""
    offs = [0.0, -np.pi, np.pi, np.pi, -np.pi, np.pi, np.pi]
    B = q.shape[0]
    cur = np.tile(np.eye(4), (B, 1, 1))
    for i in range(7):
        c, s = np.cos(q[:, i]), np.sin(q[:, i])
        Rz = np.tile(np.eye(4), (B, 1, 1))
        Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1] = c, -s, s, c
        co, so = np.cos(offs[i]), np.sin(offs[i])
        Rx = np.array([[1, 0, 0, 0], [0, co, -so, 0], [0, so, co, 0], [0, 0, 0, 1.0]])
        b = model.pose_0[i] @ (Rz @ Rx)
        if i > 0:
            b[..., [1, 2]] *= -1
        cur = cur @ b
    hand = cur @ model.pose_0[7]
    return hand[:, :3, 3].copy(), hand[:, :3, 2].copy()


def make_reach_goals(scene: "Scene", model, num_goals: int, seed: int = 0) -> np.ndarray:
    """Grasp-like goal set for the scene's target: configurations whose hand sits 10-16 cm from the
    target's centre, above it, with the approach axis pointing at it — a stand-in for the IK'd grasp
    set of Planner.solve_goal_set_ik (omg/planner.py:296-455; IK itself is out of scope)."""
    q, pos, zax, tree = _reach_pool(model)
    ctr = scene.objects[scene.target_idx].pose_mat[:3, 3]
    rng = np.random.RandomState(20_000 + seed)
    radius = 0.16
    while True:
        idx = np.array(tree.query_ball_point(ctr, radius), dtype=np.int64)
        if idx.size:
            d = ctr[None] - pos[idx]
            dist = np.linalg.norm(d, axis=1)
            good = (dist > 0.10) & (np.einsum("ij,ij->i", d / dist[:, None], zax[idx]) > 0.8) & (pos[idx, 2] > ctr[2] + 0.02)
            idx = idx[good]
        if idx.size >= num_goals or radius > 0.4:
            break
        radius += 0.02
    if idx.size == 0:
        return make_goal_set(seed, num_goals)
    pick = rng.choice(idx, size=num_goals, replace=idx.size < num_goals)
    return q[pick].copy()


def linear_init(start: np.ndarray, end: np.ndarray, n: int) -> np.ndarray:
    """Interior waypoints linspace(0,1,n+2)[1:-1] between start and end (util.py:238-258, "linear"): bit-identical to
    the reference's interp1d result (linspace is i * step with step = fl(1 / (n + 1)))."""
    t = np.linspace(0, 1, n + 2)[1:-1, None]
    return start[None] + t * (end - start)[None]


def cubic_init(start: np.ndarray, end: np.ndarray, n: int) -> np.ndarray:
    """Clamped cubic spline through (0,start),(1,end) with zero end slopes = 3t^2-2t^3 blend
    (closed form of scipy CubicSpline(bc_type="clamped") on two knots, util.py:252)."""
    t = (np.arange(1, n + 1) / (n + 1.0))[:, None]
    h = 3.0 * t * t - 2.0 * t * t * t
    return start[None] + h * (end - start)[None]
