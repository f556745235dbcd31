"""Tensor-level wrappers over the C ABI.  torch owns device memory and streams; all compute is in
libomg_hip.so.  Every function enqueues on torch's current stream and returns without syncing."""
from __future__ import annotations

import os
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ChompParams, LearnerParams, check

_ws_cache: dict = {}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.OmgHipError(f"{name} must be a device tensor (the reference asserts CHECK_CUDA, omg_layers.cpp:5)")
    if not t.is_contiguous():
        raise _lib.OmgHipError(f"{name} must be contiguous (CHECK_CONTIGUOUS, omg_layers.cpp:6)")
    if t.dtype != dtype:
        raise _lib.OmgHipError(f"{name} must be {dtype}, got {t.dtype}")


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _kin_workspace(prepass, S: int, G: int, n_remaining: int, P: int, device):
    """Scratch of the kinematics pre-pass (omgx_goalset_workspace_bytes): None without it, the caller's own uint8 tensor, or the
    cached buffer of this (device, current stream) pair (`_workspace`: launches on different streams get different buffers)."""
    if prepass is None or prepass is False:
        return None
    need = _lib.lib().omgx_goalset_workspace_bytes(S, G, int(n_remaining), P)
    if isinstance(prepass, torch.Tensor):
        if not prepass.is_cuda or prepass.numel() * prepass.element_size() < need or not prepass.is_contiguous():
            raise _lib.OmgHipError(f"the pre-pass workspace must be a contiguous device tensor of at least {need} bytes")
        return prepass
    return _workspace(need, device)


def sdf_loss_forward(pose_init, sdf_grids, sdf_limits, points, epsilons, padding_scales, clearances, disables):
    """omg_cuda.sdf_loss_forward (layers/omg_layers.cpp:24-49): -> [potentials[N], potential_grads[N,3], collides[N]]."""
    for n, t in (("pose_init", pose_init), ("sdf_grids", sdf_grids), ("sdf_limits", sdf_limits), ("points", points),
                 ("epsilons", epsilons), ("padding_scales", padding_scales), ("clearances", clearances), ("disables", disables)):
        _need(t, torch.float32, n)
    N, O = points.shape[0], pose_init.shape[0]
    pot = torch.empty(N, dtype=torch.float32, device=points.device)
    grad = torch.empty((N, 3), dtype=torch.float32, device=points.device)
    col = torch.empty(N, dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        check(_lib.lib().omgx_sdf_loss_forward(_ptr(pose_init), _ptr(sdf_grids), _ptr(sdf_limits), _ptr(points),
                                               _ptr(epsilons), _ptr(padding_scales), _ptr(clearances), _ptr(disables),
                                               N, O, _ptr(pot), _ptr(grad), _ptr(col), _stream()), "omgx_sdf_loss_forward")
    return [pot, grad, col]


class DeviceScenes:
    """Scene table resident in HBM: object records, scene_begin, SDF pool (see scenes.SceneBatch) — and the ways to change it
    while it stays there (include/omg_hip.h section 8): `set_object_pose` (a 48-byte write), `replace_grid` (a volume that is
    already on the device, its influence region fitted on the device), `grid_slot` (where omgx_point_cloud_sdf can write a new
    volume directly).  Every change is ordered on torch's current stream; launches enqueued behind it see the new scene, and a
    ChompEngine built on these scenes plans again without being rebuilt (it holds pointers, not copies).
    `reserve_voxels`: spare float32 elements at the end of the pool for volumes that outgrow their slot (the pool is never
    reallocated: engines and prepared launches keep its address)."""

    def __init__(self, batch, device="cuda:0", reserve_voxels: int = 0):
        from . import scenes as _sc
        self.device = torch.device(device)
        self.num_scenes = batch.num_scenes
        self.host_objects = np.ascontiguousarray(batch.objects).copy()          # host mirror of the records (the region fields of a
        self.host_scene_begin = np.ascontiguousarray(batch.scene_begin, np.int32)  # device-fitted object are stale until sync_host())
        assert self.host_objects.dtype == _sc.OBJECT_DTYPE
        self.objects = torch.from_numpy(self.host_objects.view(np.uint8).copy()).to(self.device)
        self.scene_begin = torch.from_numpy(self.host_scene_begin).to(self.device)
        used = int(np.asarray(batch.pool).size)
        pool = np.ascontiguousarray(batch.pool, np.float32)
        if reserve_voxels > 0:
            self.pool = torch.empty(used + int(reserve_voxels), dtype=torch.float32, device=self.device)
            self.pool[:used].copy_(torch.from_numpy(pool))
        else:
            self.pool = torch.from_numpy(pool).to(self.device)
        self.pool_used = used
        sizes = self.host_objects["dim"].astype(np.int64).prod(axis=1)
        self._slot_cap = {i: int(sizes[i]) for i in range(len(sizes))}  # elements object i may use at its grid_offset
        # scenes.pack_table(share_grids=True) and scene_io store identical volumes once: several records then point at one offset.
        # A slot that is shared is never written in place (copy-on-write: the object that changes gets space of its own).
        from collections import Counter
        self._refs = Counter(int(o) for o in self.host_objects["grid_offset"])
        self._free = []       # [(offset, capacity)] slots no record points at any more: reused before the reserve is touched
        self._pending = {}    # object -> (offset, capacity) handed out by grid_slot, not yet committed by replace_grid
        self._scratch = None

    @classmethod
    def from_scenes(cls, scenes, cfg_kwargs=None, device="cuda:0", share_grids: bool = True, reserve_voxels: int = 0,
                    timing: "dict | None" = None) -> "DeviceScenes":
        """First build of a batch WITHOUT a host pass over the voxels (Env.combine_sdfs, omg/core.py:366-411; scenes.pack_table is
        the host-side specification): the records are written on the host from the scenes' poses / limits / thresholds (a few
        hundred bytes per object), every distinct volume goes to the pool with one copy, and ALL influence regions are fitted on
        the device in seven launches (omgx_fit_influence_regions) — volumes and thresholds that occur several times are fitted
        once.  share_grids: a volume referenced by several objects (the same ndarray) is stored once.  The device records equal
        scenes.pack_table(scenes, cfg_kwargs, ragged=True, share_grids=share_grids) field for field; the host mirror's region
        fields are the loose ones until sync_host()."""
        import time as _time
        from . import scenes as _sc
        t0 = _time.perf_counter()
        cfg_kwargs = cfg_kwargs or {}
        dev = torch.device(device)
        n_obj = sum(len(s.objects) for s in scenes)
        rec = np.zeros(n_obj, _sc.OBJECT_DTYPE)
        begins, chunks, seen, offset, k = [0], [], {}, 0, 0
        for s in scenes:
            poses, eps, pad, clr, dis = _sc.layer_params(s, **cfg_kwargs)
            m = len(s.objects)
            rec["pose_inv"][k: k + m] = poses[:, :3, :4].reshape(m, 12)
            rec["epsilon"][k: k + m], rec["padding_scale"][k: k + m], rec["clearance"][k: k + m] = eps, pad, clr
            rec["disabled"][k: k + m] = dis > 0
            for i, ob in enumerate(s.objects):
                mn, mxc = ob.sdf.min_coords, ob.sdf.max_coords
                rec["lo"][k + i] = mn.astype(np.float32)
                rec["hi"][k + i] = np.array([mn[a] + (mxc[a] - mn[a]) * 1.0 for a in range(3)], np.float32)  # scenes.pack_table (ragged)
                rec["dim"][k + i] = ob.sdf.data.shape
                rec["delta"][k + i] = ob.sdf.delta
                key = id(ob.sdf.data)
                if share_grids and key in seen:
                    rec["grid_offset"][k + i] = seen[key]
                else:
                    rec["grid_offset"][k + i] = offset
                    seen[key] = offset
                    chunks.append((offset, ob.sdf.data))
                    offset += int(ob.sdf.data.size)
            k += m
            begins.append(k)
        _sc.finish_records(rec)  # derived constants + the loose region
        t1 = _time.perf_counter()
        self = cls.__new__(cls)
        self.device = dev
        self.num_scenes = len(scenes)
        self.host_objects = rec
        self.host_scene_begin = np.array(begins, np.int32)
        with torch.cuda.device(dev):
            self.pool = torch.empty(offset + int(reserve_voxels), dtype=torch.float32, device=dev)
            for off, data in chunks:
                src = torch.from_numpy(np.ascontiguousarray(data, np.float32).reshape(-1))
                self.pool[off: off + src.numel()].copy_(src, non_blocking=True)
            self.objects = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
            self.scene_begin = torch.from_numpy(self.host_scene_begin).to(dev)
        self.pool_used = offset
        sizes = rec["dim"].astype(np.int64).prod(axis=1)
        self._slot_cap = {i: int(sizes[i]) for i in range(n_obj)}
        from collections import Counter
        self._refs = Counter(int(o) for o in rec["grid_offset"])
        self._free, self._pending, self._scratch = [], {}, None
        t2 = _time.perf_counter()
        self.fit_all()
        if timing is not None:
            torch.cuda.synchronize(dev)
            t3 = _time.perf_counter()
            timing.update(records_ms=(t1 - t0) * 1e3, upload_ms=(t2 - t1) * 1e3, fit_ms=(t3 - t2) * 1e3, total_ms=(t3 - t0) * 1e3)
        return self

    def fit_all(self) -> int:
        """Fit the influence region of every record the kernels cull for, on the device, in seven launches
        (omgx_fit_influence_regions); records with the same volume CONTENT (omgx_volume_hashes: private copies of one model in many
        scenes count as one), dims, voxel size and thresholds are fitted once.  Returns the number of distinct fits."""
        rec = self.host_objects
        w = (rec["hi"].astype(np.float32) - rec["lo"].astype(np.float32)).astype(np.float32)
        ok = (w > 0).all(axis=1) & (rec["dim"] > 1).all(axis=1) & (rec["epsilon"] < 1.0) & (rec["clearance"] <= 1.0)  # scenes.tighten_far_boxes
        l = _lib.lib()
        with torch.cuda.device(self.device):  # 128-bit content hashes of all volumes: one launch, one small download
            d_hash = torch.empty((len(rec), 2), dtype=torch.int64, device=self.device)
            check(l.omgx_volume_hashes(_ptr(self.objects), len(rec), _ptr(self.pool), _ptr(d_hash), _stream()), "omgx_volume_hashes")
            hashes = d_hash.cpu().numpy()
        copy_src = np.full(len(rec), -1, np.int32)
        leaders, first = [], {}
        for o in np.nonzero(ok)[0]:
            r = rec[o]
            key = (hashes[o].tobytes(), r["dim"].tobytes(), w[o].tobytes(), float(r["epsilon"]), float(r["clearance"]))
            lead = first.setdefault(key, int(o))
            copy_src[o] = lead
            if lead == o:
                leaders.append(int(o))
        if not leaders:
            return 0
        fit_list = np.array(leaders, np.int32)
        nvox = rec["dim"][fit_list].astype(np.int64).prod(axis=1)
        need_off = np.concatenate([[0], np.cumsum(nvox)[:-1]]).astype(np.int64)
        nbytes = int(l.omgx_regions_scratch_bytes(len(fit_list), int(nvox.sum())))
        with torch.cuda.device(self.device):
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            d_fit = torch.from_numpy(fit_list).to(self.device)
            d_off = torch.from_numpy(need_off).to(self.device)
            d_src = torch.from_numpy(copy_src).to(self.device)
            check(l.omgx_fit_influence_regions(_ptr(self.objects), len(rec), _ptr(self.pool), _ptr(d_fit), _ptr(d_off), len(fit_list),
                                               int(nvox.max()), _ptr(d_src), _ptr(scratch), _stream()), "omgx_fit_influence_regions")
            scratch.record_stream(torch.cuda.current_stream(self.device))
        return len(fit_list)

    def _index(self, scene: int, obj: int) -> int:
        lo, hi = int(self.host_scene_begin[scene]), int(self.host_scene_begin[scene + 1])
        if not 0 <= obj < hi - lo:
            raise IndexError(f"scene {scene} has {hi - lo} objects")
        return lo + obj

    def _record_ptr(self, idx: int) -> C.c_void_p:
        return C.c_void_p(self.objects.data_ptr() + idx * self.host_objects.dtype.itemsize)

    def set_object_pose(self, scene: int, obj: int, pose_mat) -> None:
        """Move an object: pose_mat [4,4] (object -> world; numpy or a device tensor).  The record's pose rows become
        se3_inverse(pose_mat) in float32 (omg/util.py:129-135, what Cost.compute_obstacle_cost_layer rebuilds per call,
        omg/cost.py:303-316); its influence region lives in object coordinates and stays."""
        from . import scenes as _sc
        idx = self._index(scene, obj)
        if isinstance(pose_mat, torch.Tensor):
            P = pose_mat.to(device=self.device, dtype=torch.float64)
            R, t = P[:3, :3], P[:3, 3]
            inv = torch.cat([R.T, -(R.T @ t)[:, None]], dim=1).to(torch.float32).contiguous()  # [3,4]
            self.host_objects[idx]["pose_inv"] = inv.cpu().numpy().ravel()
        else:
            inv_h = _sc.se3_inverse(np.asarray(pose_mat, np.float64))[:3, :4]
            self.host_objects[idx]["pose_inv"] = inv_h.ravel()
            inv = torch.from_numpy(np.ascontiguousarray(inv_h, np.float32)).to(self.device, non_blocking=True)
        off = idx * self.host_objects.dtype.itemsize
        self.objects[off: off + 48].copy_(inv.reshape(-1).view(torch.uint8))

    def grid_slot(self, scene: int, obj: int, shape) -> torch.Tensor:
        """A float32 [X,Y,Z] view into the pool where the object's NEXT volume can be written in place (e.g. by
        point_cloud_sdf(out=...)): the object's own slot if the shape fits and no other object shares it, else a slot nobody
        uses any more, else fresh space from the reserve.  Nothing about the object changes until replace_grid is given the view:
        a slot handed out and never used goes back to the free list with the next grid_slot call for the same object.
        Sizing the reserve: one extra volume per object whose cloud extents grow from frame to frame (a slot that is outgrown
        is reused by the next volume that fits it), plus one per object that shares its volume and will be changed."""
        idx = self._index(scene, obj)
        n = int(np.prod(shape))
        old = self._pending.pop(idx, None)
        if old is not None and old[0] != int(self.host_objects[idx]["grid_offset"]):
            self._free.append(old)  # handed out earlier, never committed
        off, cap = int(self.host_objects[idx]["grid_offset"]), self._slot_cap[idx]
        if n > cap or self._refs[off] > 1:
            for k, (fo, fc) in enumerate(self._free):
                if fc >= n:
                    off, cap = self._free.pop(k)
                    break
            else:
                if self.pool_used + n > self.pool.numel():
                    raise _lib.OmgHipError(f"the SDF pool has no room for {n} more voxels: build DeviceScenes with reserve_voxels")
                off, cap = self.pool_used, n
                self.pool_used += n
        self._pending[idx] = (off, cap)
        return self.pool[off: off + n].view(tuple(int(d) for d in shape))

    def _commit_slot(self, idx: int, n: int):
        """replace_grid: the object moves into the slot grid_slot handed out; a slot it leaves behind is free once no record uses it."""
        off, cap = self._pending.pop(idx)
        old = int(self.host_objects[idx]["grid_offset"])
        if off != old:
            self._refs[old] -= 1
            if self._refs[old] <= 0:
                del self._refs[old]
                self._free.append((old, self._slot_cap[idx]))
            self._refs[off] += 1
            self._slot_cap[idx] = cap
            self.host_objects[idx]["grid_offset"] = off
        return off

    def replace_grid(self, scene: int, obj: int, grid: torch.Tensor, origin, delta: float, fit: str = "device") -> None:
        """Give an object a new volume that is already on the device: grid [X,Y,Z] float32 (a view from grid_slot: used in place;
        anything else is copied into the pool, device to device), origin = min corner [3], delta = voxel size.  The record's
        limits follow like scenes.pack_table writes them; the influence region is fitted ON THE DEVICE (fit="device":
        omgx_fit_influence_region, the algorithm of scenes.influence_rbox) or left loose (fit="loose": the grid with 1.5 voxels
        of slack — same results, more exact lookups)."""
        if fit not in ("device", "loose"):
            raise ValueError("fit must be 'device' or 'loose'")
        _need(grid, torch.float32, "grid")
        if grid.dim() != 3:
            raise _lib.OmgHipError("grid must be [X,Y,Z]")
        idx = self._index(scene, obj)
        shape = tuple(int(d) for d in grid.shape)
        n = int(np.prod(shape))
        pend = self._pending.get(idx)
        if pend is not None and n <= pend[1] and grid.data_ptr() == self.pool.data_ptr() + 4 * pend[0]:
            slot = grid  # the view grid_slot handed out, filled in place
        else:
            slot = self.grid_slot(scene, obj, shape)
            slot.copy_(grid)
        self._commit_slot(idx, n)
        rec = self.host_objects[idx]
        mn = np.asarray(origin, np.float64)
        mxc = mn + float(delta) * np.array(shape)
        lo = mn.astype(np.float32)
        hi = np.array([mn[a] + (mxc[a] - mn[a]) * 1.0 for a in range(3)], np.float32)  # scenes.pack_table (ragged)
        dims = np.array(shape, np.int32)
        rec["lo"], rec["hi"], rec["dim"], rec["delta"] = lo, hi, dims, np.float32(delta)
        one = self.host_objects[idx: idx + 1]
        from . import scenes as _sc
        _sc.finish_records(one)  # host mirror: derived constants + the loose region
        l = _lib.lib()
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        with torch.cuda.device(self.device):
            check(l.omgx_object_set_grid(self._record_ptr(idx), lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), dims.ctypes.data_as(ip),
                                         float(np.float32(delta)), int(rec["grid_offset"]), _stream()), "omgx_object_set_grid")
            if fit == "device":
                need = int(l.omgx_region_scratch_bytes(*shape))
                if self._scratch is None or self._scratch.numel() < need:
                    self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
                check(l.omgx_fit_influence_region(self._record_ptr(idx), C.c_void_p(slot.data_ptr()), dims.ctypes.data_as(ip),
                                                  lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), float(rec["epsilon"]), float(rec["clearance"]),
                                                  _ptr(self._scratch), _stream()), "omgx_fit_influence_region")

    def sync_host(self) -> np.ndarray:
        """Bring the host mirror of the records up to date with the device (one small copy; tests, oracle comparisons)."""
        self.host_objects = self.objects.cpu().numpy().view(self.host_objects.dtype).copy()
        return self.host_objects

    def host_batch(self):
        """The scenes as they are on the device now, as a host SceneBatch (records + pool copied back): for oracle checks."""
        from . import scenes as _sc
        return _sc.SceneBatch(self.sync_host(), self.host_scene_begin.copy(), self.pool[: self.pool_used].cpu().numpy())


def robot_blob(model, device="cuda:0") -> torch.Tensor:
    return torch.from_numpy(model.blob()).to(device)


def fk_sdf(robot: torch.Tensor, P: int, scenes: DeviceScenes, joints: torch.Tensor, soften_fingers=False,
           want_grad=True, want_col=True, out=None, arc_length=0, arc_start=None, dt=0.1):
    """joints [S,C,9] f64 -> potentials [S,C,10,P], grads [S,C,10,P,3] | None, collides [S,C,10,P] | None.
    arc_length > 0: groups of arc_length waypoints, potentials weighted by the float32 point speed."""
    _need(joints, torch.float64, "joints")
    S, Cn = joints.shape[0], joints.shape[1]
    dev = joints.device
    if out is None:
        pot = torch.empty((S, Cn, 10, P), dtype=torch.float32, device=dev)
        grad = torch.empty((S, Cn, 10, P, 3), dtype=torch.float32, device=dev) if want_grad else None
        col = torch.empty((S, Cn, 10, P), dtype=torch.float32, device=dev) if want_col else None
    else:
        pot, grad, col = out
    l = _lib.lib()
    with torch.cuda.device(dev):
        ws = _workspace(l.omgx_fk_sdf_workspace_bytes(S, Cn, P), dev)
        if arc_length > 0:
            _need(arc_start, torch.float64, "arc_start")
        check(l.omgx_fk_sdf(_ptr(robot), P, _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool),
                            _ptr(joints), S, Cn, int(bool(soften_fingers)), int(arc_length), _ptr(arc_start), float(dt),
                            _ptr(pot), _ptr(grad), _ptr(col), _ptr(ws), _stream()), "omgx_fk_sdf")
    return pot, grad, col


def goalset_cost(robot, P, scenes: DeviceScenes, traj_start, goals, n_remaining, dt, soften_fingers=False,
                 want_potentials=False, out=None, active=None, goal_count=None, prepass=False):
    """traj_start [S,9], goals [S,G,9] f64 -> goal_cost [S,G] f32, collides [S,G] f32, potentials [S,G,n,10,P] | None.
    traj_start may be a strided row view such as traj[:, k] of a contiguous [S,n,9] tensor (no copy is made).
    active / goal_count [S] int32 (optional, not with want_potentials): scenes with 0 and the padding goals of a ragged
    goal set are skipped; their outputs keep their previous contents.
    prepass: the goals' kinematics and row masks as a launch of their own (k_goalset_kin, ABI 10) through a scratch workspace
    — same bits, two launches; a torch tensor of omgx_goalset_workspace_bytes is taken as that workspace."""
    if not (traj_start.is_cuda and traj_start.dtype == torch.float64 and traj_start.dim() == 2 and traj_start.shape[1] == 9
            and traj_start.stride(1) == 1 and (traj_start.shape[0] == 1 or traj_start.stride(0) >= 9)):
        raise _lib.OmgHipError("traj_start must be a float64 device tensor [S,9] with unit inner stride")
    ts_stride = traj_start.stride(0) if traj_start.shape[0] > 1 else 9
    _need(goals, torch.float64, "goals")
    S, G = goals.shape[0], goals.shape[1]
    dev = goals.device
    if out is None:
        cost = torch.empty((S, G), dtype=torch.float32, device=dev)
        col = torch.empty((S, G), dtype=torch.float32, device=dev)
    else:
        cost, col = out
    pots = torch.empty((S, G, n_remaining, 10, P), dtype=torch.float32, device=dev) if want_potentials else None
    l = _lib.lib()
    with torch.cuda.device(dev):
        ws = _kin_workspace(prepass, S, G, n_remaining, P, dev) if not want_potentials else None
        check(l.omgx_goalset_cost(_ptr(robot), P, _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool),
                                  _ptr(traj_start), ts_stride, _ptr(goals), S, G, n_remaining, float(dt), int(bool(soften_fingers)),
                                  _ptr(cost), _ptr(pots), _ptr(col), _ptr(ws), _ptr(_active(active, S)), _ptr(_active(goal_count, S)),
                                  _stream()), "omgx_goalset_cost")
    return cost, col, pots


def goalset_cost_layer(robot, P, scenes: DeviceScenes, traj_start, goals, n_remaining, dt, traj, layer_out, soften_fingers=False,
                       layer_soften_fingers=False, out=None, active=None, goal_count=None, schedule=None, work=None, goal_parts=1,
                       layer_poses=None, prepass=False):
    """goalset_cost (cost only) + fk_sdf(traj) in one launch (omgx_goalset_cost_layer).  traj [S,n,9] f64;
    layer_out = (potentials [S,n,10,P], grads [S,n,10,P,3], collides [S,n,10,P]) float32, written in place.
    active [S] int32 (optional): scenes with 0 are skipped, their outputs keep their previous contents.
    goal_parts > 1 (omgx_goalset_cost_layer_parts): a goal's tiles dealt over NP = goalset_parts(n_remaining, goal_parts) workgroups
    of the batch kernel; `out` must then hold S * G * NP elements each and receives [S][G][NP] PARTIAL sums, schedule / work count
    the S * G * NP (scene, goal, part) items.  prepass: see goalset_cost.  Returns (cost, collides) as given / allocated."""
    if not (traj_start.is_cuda and traj_start.dtype == torch.float64 and traj_start.dim() == 2 and traj_start.shape[1] == 9
            and traj_start.stride(1) == 1 and (traj_start.shape[0] == 1 or traj_start.stride(0) >= 9)):
        raise _lib.OmgHipError("traj_start must be a float64 device tensor [S,9] with unit inner stride")
    ts_stride = traj_start.stride(0) if traj_start.shape[0] > 1 else 9
    _need(goals, torch.float64, "goals")
    _need(traj, torch.float64, "traj")
    lp, lg, lc = layer_out
    for n_, t in (("layer potentials", lp), ("layer grads", lg), ("layer collides", lc)):
        _need(t, torch.float32, n_)
    S, G, n = goals.shape[0], goals.shape[1], traj.shape[1]
    if traj.shape[0] != S or lp.numel() != S * n * 10 * P or lg.numel() != 3 * lp.numel() or lc.numel() != lp.numel():
        raise _lib.OmgHipError("layer outputs must be [S,n,10,P], [S,n,10,P,3], [S,n,10,P]")
    dev = goals.device
    NP = goalset_parts(n_remaining, goal_parts) if int(goal_parts) != 1 else 1
    if NP < 1:
        raise _lib.OmgHipError("goal_parts must be 1, 2, 4 or 8")
    if out is None:
        cost = torch.empty((S, G) if NP == 1 else (S, G, NP), dtype=torch.float32, device=dev)
        col = torch.empty((S, G) if NP == 1 else (S, G, NP), dtype=torch.float32, device=dev)
    else:
        cost, col = out
    l = _lib.lib()
    if int(goal_parts) != 1 or layer_poses is not None:
        for n_, t in (("goal_cost", cost), ("goal collides", col)):
            _need(t, torch.float32, n_)
            if t.numel() < S * G * NP:
                raise _lib.OmgHipError(f"{n_} must hold S * G * parts = {S * G * NP} elements")
        if layer_poses is not None:
            _need(layer_poses, torch.float64, "layer_poses")
            if layer_poses.numel() != S * n * 120:
                raise _lib.OmgHipError("layer_poses must be [S,n,10,12]")
        with torch.cuda.device(dev):
            check(l.omgx_goalset_cost_layer_parts(_ptr(robot), P, _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool),
                                                  _ptr(traj_start), ts_stride, _ptr(goals), S, G, n_remaining, float(dt),
                                                  int(bool(soften_fingers)), _ptr(cost), _ptr(col), _ptr(traj), n,
                                                  int(bool(layer_soften_fingers)), _ptr(lp), _ptr(lg), _ptr(lc), _ptr(_active(active, S)),
                                                  _ptr(_active(goal_count, S)), _ptr(_i32n(schedule, None, "schedule")),
                                                  0 if schedule is None else schedule.numel(), _ptr(_i32n(work, S * G * NP, "work")),
                                                  int(goal_parts), _ptr(layer_poses), _ptr(_kin_workspace(prepass, S, G, n_remaining, P, dev)),
                                                  _stream()),
                  "omgx_goalset_cost_layer_parts")
        return cost, col
    with torch.cuda.device(dev):
        ws = _kin_workspace(prepass, S, G, n_remaining, P, dev)
        check(l.omgx_goalset_cost_layer(_ptr(robot), P, _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool),
                                        _ptr(traj_start), ts_stride, _ptr(goals), S, G, n_remaining, float(dt),
                                        int(bool(soften_fingers)), _ptr(cost), _ptr(col), _ptr(ws), _ptr(traj), n,
                                        int(bool(layer_soften_fingers)), _ptr(lp), _ptr(lg), _ptr(lc), _ptr(_active(active, S)),
                                        _ptr(_active(goal_count, S)), _ptr(_i32n(schedule, None, "schedule")), 0 if schedule is None else schedule.numel(), _ptr(_i32n(work, S * G, "work")),
                                        _stream()),
              "omgx_goalset_cost_layer")
    return cost, col


def goalset_parts(n_remaining: int, goal_parts: int) -> int:
    """Workgroups per goal of goalset_cost_layer_tiled for a window of n_remaining configurations (omgx_goalset_parts)."""
    return int(_lib.lib().omgx_goalset_parts(int(n_remaining), int(goal_parts)))


def goalset_cost_layer_tiled(robot, P, scenes: DeviceScenes, traj_start, goals, n_remaining, dt, traj, layer_out, out,
                             soften_fingers=False, layer_soften_fingers=False, active=None, goal_count=None, goal_parts=4,
                             layer_link_groups=10, layer_config_block=16, spread=True, layer_poses=None, prepass=False):
    """The goal-set batch and / or the trajectory layer cut into many small workgroups (omgx_goalset_cost_layer_tiled: latency
    mode for one or a few scenes).  goals None: only the layer; traj None: only the batch.  out = (cost, collides): float32
    device tensors with at least S * G * goalset_parts(n_remaining, goal_parts) elements, written as [S][G][parts] PARTIAL sums
    (the learner adds them: LearnerParams.cost_parts).  layer_poses: optional float64 [S,n,10,12] receiving the waypoints' link
    poses (ChompParams.waypoint_poses of the step that follows).  Returns the number of parts per goal."""
    dev = (goals if goals is not None else traj).device
    S = (goals if goals is not None else traj).shape[0]
    G, ts_stride, parts = 0, 9, 1
    cost = col = None
    if goals is not None:
        if not (traj_start.is_cuda and traj_start.dtype == torch.float64 and traj_start.dim() == 2 and traj_start.shape[1] == 9
                and traj_start.stride(1) == 1 and (traj_start.shape[0] == 1 or traj_start.stride(0) >= 9)):
            raise _lib.OmgHipError("traj_start must be a float64 device tensor [S,9] with unit inner stride")
        ts_stride = traj_start.stride(0) if traj_start.shape[0] > 1 else 9
        _need(goals, torch.float64, "goals")
        G = goals.shape[1]
        parts = goalset_parts(n_remaining, goal_parts)
        if parts < 1:
            raise _lib.OmgHipError("goal_parts must be 1, 2, 4 or 8")
        cost, col = out
        for n_, t in (("goal_cost", cost), ("goal collides", col)):
            _need(t, torch.float32, n_)
            if t.numel() < S * G * parts:
                raise _lib.OmgHipError(f"{n_} must hold S * G * parts = {S * G * parts} elements")
    lp = lg = lc = None
    n = 0
    if layer_poses is not None:
        _need(layer_poses, torch.float64, "layer_poses")
        if traj is None or layer_poses.numel() != S * traj.shape[1] * 120:
            raise _lib.OmgHipError("layer_poses must be [S,n,10,12] and needs the trajectory layer")
    if traj is not None:
        _need(traj, torch.float64, "traj")
        lp, lg, lc = layer_out
        for n_, t in (("layer potentials", lp), ("layer grads", lg), ("layer collides", lc)):
            _need(t, torch.float32, n_)
        n = traj.shape[1]
        if traj.shape[0] != S or lp.numel() != S * n * 10 * P or lg.numel() != 3 * lp.numel() or lc.numel() != lp.numel():
            raise _lib.OmgHipError("layer outputs must be [S,n,10,P], [S,n,10,P,3], [S,n,10,P]")
    with torch.cuda.device(dev):
        check(_lib.lib().omgx_goalset_cost_layer_tiled(
            _ptr(robot), P, _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool), _ptr(traj_start) if goals is not None else None,
            ts_stride, _ptr(goals), S, G, int(n_remaining) if goals is not None else 1, float(dt), int(bool(soften_fingers)), _ptr(cost), _ptr(col),
            _ptr(traj), n, int(bool(layer_soften_fingers)), _ptr(lp), _ptr(lg), _ptr(lc), _ptr(_active(active, S)),
            _ptr(_active(goal_count, S)), int(goal_parts), int(layer_link_groups), int(layer_config_block), int(bool(spread)),
            _ptr(layer_poses), _ptr(_kin_workspace(prepass, S, G, n_remaining, P, dev) if goals is not None else None), _stream()),
            "omgx_goalset_cost_layer_tiled")
    return parts


def goalset_schedule(work, num_scenes: int, num_goals: int, active=None, goal_count=None, slack: int = 2, out=None, device=None,
                     parts: int = 1, longest_first: bool = False):
    """Dispatch order for goalset_cost_layer (omgx_goalset_schedule): int32 device tensor [omgx_goalset_schedule_len].
    work: int32/uint32 device tensor [S*G] of durations (None: all items weigh the same).  Asynchronous, one small launch.
    The result lives on the device of `out`, else of the first tensor given, else `device`.
    parts > 1 (omgx_goalset_schedule_parts): the items are the S * G * parts (scene, goal, part) workgroups of a launch with
    split goals; work [S*G*parts]; goal_count still counts goals.
    longest_first (omgx_goalset_schedule_ordered): inside an XCD the items run by decreasing work across its scenes — for launches of
    a round or two of the chip's workgroup slots (falls back to the scene-major order above 8192 items)."""
    l = _lib.lib()
    parts = int(parts)
    n = int(l.omgx_goalset_schedule_len(num_scenes, num_goals * parts, slack))
    if out is None:
        for t in (work, active, goal_count):
            if t is not None:
                device = t.device
                break
        if device is None:
            raise _lib.OmgHipError("goalset_schedule needs a device: pass work, active, goal_count, out or device")
        out = torch.empty(n, dtype=torch.int32, device=device)
    _i32n(out, n, "schedule")
    _i32n(work, num_scenes * num_goals * parts, "work")
    with torch.cuda.device(out.device):
        if longest_first:
            check(l.omgx_goalset_schedule_ordered(_ptr(work), _ptr(_active(active, num_scenes)), _ptr(_active(goal_count, num_scenes)),
                                                  num_scenes, num_goals, parts, slack, _lib.SCHEDULE_LONGEST_FIRST, _ptr(out), _stream()),
                  "omgx_goalset_schedule_ordered")
        elif parts != 1:
            check(l.omgx_goalset_schedule_parts(_ptr(work), _ptr(_active(active, num_scenes)), _ptr(_active(goal_count, num_scenes)),
                                                num_scenes, num_goals, parts, slack, _ptr(out), _stream()), "omgx_goalset_schedule_parts")
        else:
            check(l.omgx_goalset_schedule(_ptr(work), _ptr(_active(active, num_scenes)), _ptr(_active(goal_count, num_scenes)), num_scenes,
                                          num_goals, slack, _ptr(out), _stream()), "omgx_goalset_schedule")
    return out


def forward_kinematics(robot, P, joints, want_joint_info=True):
    """joints [B,9] f64 -> link poses [B,10,4,4], joint origins [B,10,3] | None, joint axes [B,10,3] | None."""
    _need(joints, torch.float64, "joints")
    B, dev = joints.shape[0], joints.device
    poses = torch.empty((B, 10, 4, 4), dtype=torch.float64, device=dev)
    org = torch.empty((B, 10, 3), dtype=torch.float64, device=dev) if want_joint_info else None
    ax = torch.empty((B, 10, 3), dtype=torch.float64, device=dev) if want_joint_info else None
    with torch.cuda.device(dev):
        check(_lib.lib().omgx_forward_kinematics(_ptr(robot), P, _ptr(joints), B, _ptr(poses), _ptr(org), _ptr(ax), _stream()),
              "omgx_forward_kinematics")
    return poses, org, ax


def pose_table(robot, P, configs: torch.Tensor, out: "torch.Tensor | None" = None) -> torch.Tensor:
    """configs [..., 9] f64 -> link poses [..., 10, 12] f64 in the step's own layout (omgx_pose_table: rotation rows, translation;
    before center_offset): what ChompParams.start_poses / end_poses and LearnerParams.goal_pose_table point at."""
    _need(configs, torch.float64, "configs")
    N = configs.numel() // 9
    if out is None:
        out = torch.empty(tuple(configs.shape[:-1]) + (10, 12), dtype=torch.float64, device=configs.device)
    else:
        _need(out, torch.float64, "poses")
        if out.numel() != N * 120:
            raise _lib.OmgHipError("poses must hold 120 doubles per configuration")
    with torch.cuda.device(configs.device):
        check(_lib.lib().omgx_pose_table(_ptr(robot), P, _ptr(configs), N, _ptr(out), _stream()), "omgx_pose_table")
    return out


def chomp_optimize(robot, params: ChompParams, traj, start, end, goal, goal_point, pot, pgrad, col, active=None, out=None,
                   aux=None, stop_on_terminate=False):
    """In-place step on traj [S,n,9] f64 -> grad [S,n,9], cost_traj [S,n], info [S,16] (f64).
    aux: optional [S, omgx_chomp_aux_doubles(n)] f64 receiving obs_grad | obs_cost | smooth_grad | smooth_loss."""
    for n_, t in (("traj", traj), ("start", start), ("end", end), ("goal", goal), ("goal_point", goal_point)):
        _need(t, torch.float64, n_)
    for n_, t in (("potentials", pot), ("grads", pgrad), ("collides", col)):
        _need(t, torch.float32, n_)
    S, n = traj.shape[0], traj.shape[1]
    dev = traj.device
    if out is None:
        grad = torch.empty((S, n, 9), dtype=torch.float64, device=dev)
        cost_traj = torch.empty((S, n), dtype=torch.float64, device=dev)
        info = torch.zeros((S, _lib.INFO_STRIDE), dtype=torch.float64, device=dev)
    else:
        grad, cost_traj, info = out
    with torch.cuda.device(dev):
        check(_lib.lib().omgx_chomp_optimize(_ptr(robot), C.byref(params), _ptr(traj), _ptr(start), _ptr(end), _ptr(goal),
                                             _ptr(goal_point), _ptr(pot), _ptr(pgrad), _ptr(col), _ptr(active), S,
                                             _ptr(grad), _ptr(cost_traj), _ptr(info), _ptr(aux), int(bool(stop_on_terminate)),
                                             _stream()), "omgx_chomp_optimize")
    return grad, cost_traj, info


def learner_state(S: int, G: int, device, goal_count=None) -> torch.Tensor:
    """Initial Learner state [S, 7G+10] f64: sum_costs 0 | p 1/G | experts_p 1/G | q 1/5 | experts_costs 0
    (Learner.__init__, omg/online_learner.py:66-95).  goal_count [S] (ragged goal sets padded to G): scene s holds
    1 / goal_count[s] in its first goal_count[s] entries and 0 in the padding."""
    st = np.zeros((S, 7 * G + 10), np.float64)  # built on the host: one upload instead of four first-use torch kernels
    if goal_count is None:
        st[:, G:7 * G] = 1.0 / G
    else:
        cnt = np.asarray(goal_count, np.float64).reshape(S, 1)
        row = np.where(np.arange(G)[None, :] < cnt, 1.0 / cnt, 0.0)
        st[:, G:7 * G] = np.tile(row, (1, 6))
    st[:, 7 * G:7 * G + 5] = 0.2
    return torch.from_numpy(st).to(device)


def _eta(eta, S):
    if eta is not None and not (eta.is_cuda and eta.dtype == torch.float64 and eta.is_contiguous() and eta.numel() == S):
        raise _lib.OmgHipError("eta must be a contiguous float64 device tensor [S]")
    return eta


def _i32n(t, n, name):
    """Optional contiguous 4-byte integer device tensor of n elements (int32 schedules, uint32-as-int32 work counters)."""
    if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and (n is None or t.numel() == n)):
        raise _lib.OmgHipError(f"{name} must be a contiguous int32 device tensor" + (f" of {n} elements" if n is not None else ""))
    return t


def _active(active, S):
    if active is not None and not (active.is_cuda and active.dtype == torch.int32 and active.is_contiguous() and active.numel() == S):
        raise _lib.OmgHipError("active must be a contiguous int32 device tensor [S]")
    return active


def goal_update(params: LearnerParams, traj, goal_set, reach, goal_cost, state, goal_idx, end, goal_rows, goal_point,
                cost_vector=None, active=None, goal_count=None, eta=None):
    """Learner.update_goal for S scenes in one launch (omgx_goal_update); all outputs are written in place.
    active [S] int32 (optional): scenes with 0 keep goal, outputs and state."""
    _need(traj, torch.float64, "traj")
    _need(goal_set, torch.float64, "goal_set")
    _need(state, torch.float64, "state")
    if goal_idx.dtype != torch.int32:
        raise _lib.OmgHipError("goal_idx must be int32")
    with torch.cuda.device(traj.device):
        check(_lib.lib().omgx_goal_update(C.byref(params), _ptr(traj), _ptr(goal_set), _ptr(reach), _ptr(goal_cost), _ptr(state),
                                          traj.shape[0], _ptr(goal_idx), _ptr(end), _ptr(goal_rows), _ptr(goal_point),
                                          _ptr(cost_vector), _ptr(_active(active, traj.shape[0])),
                                          _ptr(_active(goal_count, traj.shape[0])), _ptr(_eta(eta, traj.shape[0])), _stream()), "omgx_goal_update")


def goal_update_optimize(lparams: LearnerParams, goal_set, reach, goal_cost, state, goal_idx, robot, params: ChompParams, traj,
                         start, end, goal, goal_point, pot, pgrad, col, active=None, out=None, aux=None, cost_vector=None,
                         scene_flags=None, ticket=0, stop_on_terminate=False, goal_count=None, eta=None):
    """goal_update followed by chomp_optimize in one launch (omgx_goal_update_optimize): same results as the two calls."""
    for n_, t in (("traj", traj), ("start", start), ("end", end), ("goal", goal), ("goal_point", goal_point),
                  ("goal_set", goal_set), ("state", state)):
        _need(t, torch.float64, n_)
    for n_, t in (("potentials", pot), ("grads", pgrad), ("collides", col)):
        _need(t, torch.float32, n_)
    if goal_idx.dtype != torch.int32:
        raise _lib.OmgHipError("goal_idx must be int32")
    if scene_flags is not None and (scene_flags.dtype != torch.int32 or scene_flags.numel() < traj.shape[0] or not scene_flags.is_cuda):
        raise _lib.OmgHipError("scene_flags must be an int32 device tensor [S]")
    S, n = traj.shape[0], traj.shape[1]
    dev = traj.device
    if out is None:
        grad = torch.empty((S, n, 9), dtype=torch.float64, device=dev)
        cost_traj = torch.empty((S, n), dtype=torch.float64, device=dev)
        info = torch.zeros((S, _lib.INFO_STRIDE), dtype=torch.float64, device=dev)
    else:
        grad, cost_traj, info = out
    with torch.cuda.device(dev):
        check(_lib.lib().omgx_goal_update_optimize(C.byref(lparams), _ptr(goal_set), _ptr(reach), _ptr(goal_cost), _ptr(state),
                                                   _ptr(goal_idx), _ptr(cost_vector), _ptr(robot), C.byref(params), _ptr(traj),
                                                   _ptr(start), _ptr(end), _ptr(goal), _ptr(goal_point), _ptr(pot), _ptr(pgrad),
                                                   _ptr(col), _ptr(active), S, _ptr(grad), _ptr(cost_traj), _ptr(info), _ptr(aux),
                                                   _ptr(scene_flags), int(ticket), int(bool(stop_on_terminate)),
                                                   _ptr(_active(goal_count, S)), _ptr(_eta(eta, S)), _stream()),
              "omgx_goal_update_optimize")
    return grad, cost_traj, info


class IterationCalls:
    """The two launches of a planner iteration (omgx_goalset_cost_layer with traj_start = traj[:, k], then
    omgx_goal_update_optimize) with their tensor arguments checked and converted ONCE: a steady-state iteration passes ~60
    pointers whose values never change, and re-checking / re-wrapping them costs the host more than the launches
    themselves (48 -> 27 us per iteration; it matters when the iteration is short: small batches, the late iterations of a
    plan, the engine's two-stream pipeline).  Same entry points, same argument meaning as goalset_cost_layer() and
    goal_update_optimize(); whoever rebinds one of the tensors builds a new object (ChompEngine keys it on their identities)."""

    def __init__(self, robot, P, scenes: DeviceScenes, goals, dt, traj, layer_out, goal_out, goal_set, reach, state, goal_idx,
                 start, end, goal_rows, goal_point, step_out, cost_vector, active, goal_count=None, eta=None, scene_flags=None,
                 layer_soften_fingers=False, tiling=None, layer_poses=None, goal_parts=1, prepass=False):
        lp, lg, lc = layer_out
        cost, col = goal_out
        grad, cost_traj, info = step_out
        for n_, t in (("goals", goals), ("traj", traj), ("goal_set", goal_set), ("state", state), ("start", start), ("end", end),
                      ("goal", goal_rows), ("goal_point", goal_point), ("grad", grad), ("cost_traj", cost_traj), ("info", info),
                      ("cost_vector", cost_vector)):
            _need(t, torch.float64, n_)
        if reach is not None:
            _need(reach, torch.float64, "reach")
        for n_, t in (("layer potentials", lp), ("layer grads", lg), ("layer collides", lc), ("goal_cost", cost), ("goal collides", col)):
            _need(t, torch.float32, n_)
        S, G, n = goals.shape[0], goals.shape[1], traj.shape[1]
        if traj.shape[0] != S or lp.numel() != S * n * 10 * P or lg.numel() != 3 * lp.numel() or lc.numel() != lp.numel():
            raise _lib.OmgHipError("layer outputs must be [S,n,10,P], [S,n,10,P,3], [S,n,10,P]")
        # tiling = (goal_parts, layer_link_groups, layer_config_block, spread): the launches go through omgx_goalset_cost_layer_tiled
        # (latency mode), goal_cost / collides then hold [S][G][parts] partial sums
        self._tiling = None if tiling is None else tuple(int(v) for v in tiling)
        # goal_parts > 1 without a tiling: the batch kernel with split goals (omgx_goalset_cost_layer_parts), schedules over (scene, goal, part)
        self._goal_parts = int(goal_parts) if self._tiling is None else 1
        if layer_poses is not None:
            _need(layer_poses, torch.float64, "layer_poses")
            if layer_poses.numel() != S * n * 120:
                raise _lib.OmgHipError("layer_poses must be [S,n,10,12]")
        self._layer_poses = _ptr(layer_poses)
        if self._tiling is not None or self._goal_parts > 1:
            need = S * G * goalset_parts(n, self._tiling[0] if self._tiling is not None else self._goal_parts)
            if cost.numel() < need or col.numel() < need:
                raise _lib.OmgHipError(f"goal_cost / collides must hold S * G * parts = {need} elements")
        if goal_idx.dtype != torch.int32 or not goal_idx.is_cuda or goal_idx.numel() != S:
            raise _lib.OmgHipError("goal_idx must be an int32 device tensor [S]")
        if scene_flags is not None and (scene_flags.dtype != torch.int32 or scene_flags.numel() < S or not scene_flags.is_cuda):
            raise _lib.OmgHipError("scene_flags must be an int32 device tensor [S]")
        _active(active, S); _active(goal_count, S); _eta(eta, S)
        self.S, self.G, self.n, self.P, self.dt = S, G, n, int(P), float(dt)
        self.device = traj.device
        self._dev_index = traj.device.index if traj.device.index is not None else torch.cuda.current_device()
        self._traj_addr = traj.data_ptr()
        l = _lib.lib()
        self._f_gs, self._f_up = l.omgx_goalset_cost_layer, l.omgx_goal_update_optimize
        self._f_gst = l.omgx_goalset_cost_layer_tiled
        self._f_gsp = l.omgx_goalset_cost_layer_parts
        self._layer_soft = int(bool(layer_soften_fingers))
        p = _ptr
        self._gs_head = (p(robot), self.P, p(scenes.objects), p(scenes.scene_begin), p(scenes.pool))
        # prepass: the goals' kinematics as a launch of their own (k_goalset_kin) through a workspace this object owns — launches of
        # different IterationCalls may run at once on different streams
        self.kin_workspace = (torch.empty(max(16, l.omgx_goalset_workspace_bytes(S, G, n, self.P)), dtype=torch.uint8, device=traj.device)
                              if prepass else None)
        self._kin_ws = p(self.kin_workspace)
        self._gs_mid = (p(cost), p(col), self._kin_ws, p(traj), n, self._layer_soft, p(lp), p(lg), p(lc))
        self._goals, self._active_p, self._goal_count = p(goals), p(active), p(goal_count)
        self._up_a = (p(goal_set), p(reach), p(cost), p(state), p(goal_idx), p(cost_vector), p(robot))
        self._up_b = (p(traj), p(start), p(end), p(goal_rows), p(goal_point), p(lp), p(lg), p(lc), p(active), S, p(grad), p(cost_traj),
                      p(info), None)
        self._flags, self._eta = p(scene_flags), p(eta)
        self.use_layer_poses = False  # set per launch by the owner: only while the step that follows takes the poses

    def _on_device(self):
        return torch.cuda.current_device() == self._dev_index

    def goalset_layer(self, start_idx: int, masked: bool, schedule, work, stream):
        """omgx_goalset_cost_layer for traj_start = traj[:, start_idx], n_remaining = n - start_idx.  schedule / work: checked
        int32 device tensors or None; stream: a HIP stream handle (int)."""
        if self._tiling is not None:
            if schedule is not None or work is not None:
                raise _lib.OmgHipError("a tiled goal-set launch takes no dispatch schedule")
            cost, col, _ws, traj, n, soft, lp, lg, lc = self._gs_mid
            args = (*self._gs_head, C.c_void_p(self._traj_addr + 72 * start_idx), self.n * 9, self._goals, self.S, self.G,
                    self.n - start_idx, self.dt, 0, cost, col, traj, n, soft, lp, lg, lc, self._active_p if masked else None,
                    self._goal_count, *self._tiling, self._layer_poses if self.use_layer_poses else None, self._kin_ws, C.c_void_p(stream))
            if self._on_device():
                check(self._f_gst(*args), "omgx_goalset_cost_layer_tiled")
            else:
                with torch.cuda.device(self.device):
                    check(self._f_gst(*args), "omgx_goalset_cost_layer_tiled")
            return
        if self._goal_parts > 1 or self.use_layer_poses:
            cost, col, _ws, traj, n, soft, lp, lg, lc = self._gs_mid
            NP = goalset_parts(self.n - start_idx, self._goal_parts) if self._goal_parts > 1 else 1
            args = (*self._gs_head, C.c_void_p(self._traj_addr + 72 * start_idx), self.n * 9, self._goals, self.S, self.G,
                    self.n - start_idx, self.dt, 0, cost, col, traj, n, soft, lp, lg, lc, self._active_p if masked else None,
                    self._goal_count, _ptr(_i32n(schedule, None, "schedule")), 0 if schedule is None else schedule.numel(),
                    _ptr(_i32n(work, self.S * self.G * NP, "work")), self._goal_parts, self._layer_poses if self.use_layer_poses else None,
                    self._kin_ws, C.c_void_p(stream))
            if self._on_device():
                check(self._f_gsp(*args), "omgx_goalset_cost_layer_parts")
            else:
                with torch.cuda.device(self.device):
                    check(self._f_gsp(*args), "omgx_goalset_cost_layer_parts")
            return
        args = (*self._gs_head, C.c_void_p(self._traj_addr + 72 * start_idx), self.n * 9, self._goals, self.S, self.G,
                self.n - start_idx, self.dt, 0, *self._gs_mid, self._active_p if masked else None, self._goal_count,
                _ptr(_i32n(schedule, None, "schedule")), 0 if schedule is None else schedule.numel(), _ptr(_i32n(work, self.S * self.G, "work")),
                C.c_void_p(stream))
        if self._on_device():
            check(self._f_gs(*args), "omgx_goalset_cost_layer")
        else:
            with torch.cuda.device(self.device):
                check(self._f_gs(*args), "omgx_goalset_cost_layer")

    def _call(self, fn, args, what):
        if self._on_device():
            check(fn(*args), what)
        else:
            with torch.cuda.device(self.device):
                check(fn(*args), what)

    # (goal_parts, layer_link_groups, layer_config_block, spread) of a layer-ONLY launch in the batch layout (the smoothing iterations of a
    # plan): any split gives the same bits (every element is computed on its own).  Five workgroups per scene (2 links x all waypoints)
    # take 38 us on a chip they fill to a third; TEN (one link each) shorten the launch: plan of 100 scenes 9.06 -> 8.73 ms, with early
    # stop 7.88 -> 7.62 (round 6, tools/experiments/ab_layer_tiling.sh; config blocks on top: nothing; 13 x 128 and 16 x 64: within the noise
    # either way).  OMGX_LAYER_ONLY_TILING="1,5,0,0" (experiments) overrides.
    LAYER_ONLY_TILING = tuple(int(x) for x in os.environ["OMGX_LAYER_ONLY_TILING"].split(",")) if os.environ.get("OMGX_LAYER_ONLY_TILING") else None

    def _layer_only_tiling(self):
        if self.LAYER_ONLY_TILING is not None:
            return self.LAYER_ONLY_TILING
        S = getattr(self, "batch_scenes", self.S)  # set by the engine (a pipeline part's calls know the batch)
        if self.n > 32:
            # Long plans (50 waypoints, a dozen objects): a piece of 2 links x all waypoints is 70 us of work and a batch of 16 scenes has 80 of
            # them on 1 280 slots.  About 1 100 pieces in all, ten link groups x blocks of waypoints: plan of 16 x 64 x 50 x 13 objects 9.94 ->
            # 8.85 ms, 8 x 64 x 50 7.92 -> 6.86, 32 x 64 x 50 13.41 -> 12.22, 100 x 64 x 50 13.23 -> 12.79 (tools/experiments/ab_layer_tiling_long.sh)
            blocks = max(1, min(7, round(110.0 / max(S, 1))))
            return (1, 10, 0 if blocks == 1 else -(-self.n // blocks), 0)
        return (1, 10, 0, 0) if S >= 32 else (1, 10, 8, 0)  # small batches: 40 pieces per scene (13 x 128: plan 4.72 -> 4.48 by the pieces, 4.26 with the phase as one part)

    def layer_only(self, stream):
        """The SDF layer of the current trajectories alone (omgx_goalset_cost_layer_tiled with num_goals = 0): what omgx_fk_sdf
        computes for the step, with this object's tiling (latency mode) or five workgroups per scene (batch layout)."""
        cost, col, _ws, traj, n, soft, lp, lg, lc = self._gs_mid
        tl = self._tiling if self._tiling is not None else self._layer_only_tiling()
        args = (*self._gs_head, None, 9, None, self.S, 0, 1, self.dt, 0, None, None, traj, n, soft, lp, lg, lc, None, None, *tl,
                self._layer_poses if self.use_layer_poses else None, None, C.c_void_p(stream))
        self._call(self._f_gst, args, "omgx_goalset_cost_layer_tiled")

    def step(self, params: ChompParams, stop_on_terminate: bool, stream):
        """omgx_chomp_optimize on the layer outputs this object's launches write."""
        robot = self._gs_head[0]
        traj, start, end, goal_rows, goal_point, lp, lg, lc, active, S, grad, cost_traj, info, _aux = self._up_b
        args = (robot, C.byref(params), traj, start, end, goal_rows, goal_point, lp, lg, lc, active, S, grad, cost_traj, info, None,
                int(bool(stop_on_terminate)), C.c_void_p(stream))
        self._call(_lib.lib().omgx_chomp_optimize, args, "omgx_chomp_optimize")

    def goal_update(self, lparams: LearnerParams, stream):
        """omgx_goal_update alone (the learner without the step) on the goal costs the last goalset_layer() left."""
        goal_set, reach, cost, state, goal_idx, cost_vector, _robot = self._up_a
        traj, _start, end, goal_rows, goal_point = self._up_b[:5]
        args = (C.byref(lparams), traj, goal_set, reach, cost, state, self.S, goal_idx, end, goal_rows, goal_point, cost_vector,
                None, self._goal_count, self._eta, C.c_void_p(stream))
        self._call(_lib.lib().omgx_goal_update, args, "omgx_goal_update")

    def update(self, lparams: LearnerParams, params: ChompParams, split: bool, ticket: int, stop_on_terminate: bool, stream):
        """omgx_goal_update_optimize."""
        args = (C.byref(lparams), *self._up_a, C.byref(params), *self._up_b, self._flags if split else None, int(ticket),
                int(bool(stop_on_terminate)), self._goal_count, self._eta, C.c_void_p(stream))
        if self._on_device():
            check(self._f_up(*args), "omgx_goal_update_optimize")
        else:
            with torch.cuda.device(self.device):
                check(self._f_up(*args), "omgx_goal_update_optimize")


def plan_persistent(robot, P, scenes: DeviceScenes, goals, dt, traj, layer_out, layer_poses, goal_out, lparams: LearnerParams, goal_set, reach,
                    state, goal_idx, cost_vector, params: ChompParams, start, end, goal_rows, goal_point, step_out, iters, d_iters, workspace,
                    active=None, goal_count=None, eta=None, soften_fingers=False, layer_soften_fingers=False, max_workgroups=0, update_cus=-1):
    """omgx_plan_persistent: len(iters) iterations of the planner loop (omg/planner.py:612-630) for every scene in ONE launch
    (csrc/omg_persist.h) on the current stream.  iters: a ctypes array of _lib.PlanIter (host); d_iters: a uint8 device tensor holding the
    same bytes; workspace: uint8 device tensor of omgx_plan_persistent_workspace_bytes(S, n).  The same tensors as
    goalset_cost_layer() + goal_update_optimize(); the pose hand-over (params.start_poses / end_poses, lparams.goal_pose_table /
    end_poses_out, layer_poses) is required."""
    lp, lg, lc = layer_out
    cost, col = goal_out
    grad, cost_traj, info = step_out
    for n_, t in (("goals", goals), ("traj", traj), ("goal_set", goal_set), ("state", state), ("start", start), ("end", end),
                  ("goal", goal_rows), ("goal_point", goal_point), ("grad", grad), ("cost_traj", cost_traj), ("info", info),
                  ("layer_poses", layer_poses)):
        _need(t, torch.float64, n_)
    if reach is not None:
        _need(reach, torch.float64, "reach")
    if cost_vector is not None:
        _need(cost_vector, torch.float64, "cost_vector")
    for n_, t in (("layer potentials", lp), ("layer grads", lg), ("layer collides", lc), ("goal_cost", cost), ("goal collides", col)):
        _need(t, torch.float32, n_)
    S, G, n = goals.shape[0], goals.shape[1], traj.shape[1]
    if traj.shape[0] != S or lp.numel() != S * n * 10 * P or lg.numel() != 3 * lp.numel() or lc.numel() != lp.numel() or layer_poses.numel() != S * n * 120:
        raise _lib.OmgHipError("layer outputs must be [S,n,10,P], [S,n,10,P,3], [S,n,10,P], layer_poses [S,n,10,12]")
    if cost.numel() < S * G or col.numel() < S * G:
        raise _lib.OmgHipError("goal_cost / collides must hold S * G elements")
    if goal_idx.dtype != torch.int32 or not goal_idx.is_cuda or goal_idx.numel() != S:
        raise _lib.OmgHipError("goal_idx must be an int32 device tensor [S]")
    _active(active, S); _active(goal_count, S); _eta(eta, S)
    K = len(iters)
    if not (d_iters.is_cuda and d_iters.dtype == torch.uint8 and d_iters.numel() >= K * C.sizeof(_lib.PlanIter)):
        raise _lib.OmgHipError("d_iters must be a uint8 device tensor holding the iteration table")
    l = _lib.lib()
    need = l.omgx_plan_persistent_workspace_bytes(S, n)
    if not (workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.numel() >= need):
        raise _lib.OmgHipError(f"workspace must be a uint8 device tensor of {need} bytes")
    with torch.cuda.device(traj.device):
        check(l.omgx_plan_persistent(_ptr(robot), int(P), _ptr(scenes.objects), _ptr(scenes.scene_begin), _ptr(scenes.pool), _ptr(goals), S, G,
                                     float(dt), int(bool(soften_fingers)), _ptr(cost), _ptr(col), _ptr(traj), n, int(bool(layer_soften_fingers)),
                                     _ptr(lp), _ptr(lg), _ptr(lc), _ptr(layer_poses), _ptr(active), _ptr(goal_count),
                                     C.byref(lparams), _ptr(goal_set), _ptr(reach), _ptr(state), _ptr(goal_idx), _ptr(cost_vector), _ptr(eta),
                                     C.byref(params), _ptr(start), _ptr(end), _ptr(goal_rows), _ptr(goal_point), _ptr(grad), _ptr(cost_traj), _ptr(info),
                                     iters, _ptr(d_iters), K, _ptr(workspace), workspace.numel(), int(max_workgroups), int(update_cus), _stream()),
              "omgx_plan_persistent")


def plan_persistent_status(workspace, num_scenes: int) -> dict:
    """{"failure", "scenes_finished", "scenes_planned", "activations"} of the last omgx_plan_persistent on `workspace` (synchronises)."""
    st = (C.c_int32 * 4)()
    with torch.cuda.device(workspace.device):
        check(_lib.lib().omgx_plan_persistent_status(_ptr(workspace), int(num_scenes), st, _stream()), "omgx_plan_persistent_status")
    return {"failure": int(st[0]), "scenes_finished": int(st[1]), "scenes_planned": int(st[2]), "activations": int(st[3])}


def point_cloud_sdf(points: torch.Tensor, grid_resolution: float = 0.02, margin: float = 0.24, out: "torch.Tensor | None" = None):
    """PointEnv.compute_sdf_from_points (omg/core.py:426-457) on the device: points [N,3] f64 (robot base frame) ->
    (grid float32 [X,Y,Z] of nearest-point distances, origin [3] float64 numpy, resolution).  The workspace bounds
    are the cloud's bounding box +- margin and the nodes np.arange(lo, hi, resolution), as in the reference.
    out: optional contiguous float32 device tensor with X*Y*Z elements (e.g. a slice of env.sdf_torch) written IN PLACE
    through its raw pointer; its autograd version counter is bumped so that caches keyed on it (Cost's object table and
    influence boxes) notice.  A caller that writes such a volume through the C ABI itself must do the same
    (torch.autograd.graph.increment_version) or call Cost.invalidate()."""
    _need(points, torch.float64, "points")
    lo = points.min(0).values.cpu().numpy() - margin
    hi = points.max(0).values.cpu().numpy() + margin
    dims = np.array([len(np.arange(lo[a], hi[a], grid_resolution)) for a in range(3)], np.int32)
    if out is None:
        out = torch.empty(tuple(int(d) for d in dims), dtype=torch.float32, device=points.device)
    else:
        _need(out, torch.float32, "out")
        if out.numel() != int(dims.prod()):
            raise _lib.OmgHipError(f"out must hold {int(dims.prod())} elements (grid {tuple(int(d) for d in dims)})")
        torch.autograd.graph.increment_version(out)
    origin = np.ascontiguousarray(lo, np.float64)
    with torch.cuda.device(points.device):
        check(_lib.lib().omgx_point_cloud_sdf(_ptr(points), points.shape[0], origin.ctypes.data_as(C.POINTER(C.c_double)),
                                              float(grid_resolution), dims.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(out),
                                              _stream()), "omgx_point_cloud_sdf")
    return out, origin, float(grid_resolution)
