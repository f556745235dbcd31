"""`DeviceLoop` — what makes the drop-in classes fast when they are used the way the reference's planner uses them
(omg/planner.py:612-653): per iteration `Learner.update_goal()` then `Optimizer.optimize(traj, force_update=True)`, then a look
at `traj.goal_idx`, `traj.data` and `info["terminate"]`.

Through the classes' stand-alone paths that loop costs an upload, two launches, a synchronisation and a download per CALL
(17 ms per 70-iteration plan, round 2).  Here the trajectory, the goal set, the learner's state and the layer outputs of ONE
(Learner, trajectory) pair stay on the device between the calls:

* `Learner.update_goal()` only notes that an update is due (the reference's parameters of that moment: t, window) and hands
  out PROMISES for what it would have produced — `traj.goal_idx` (LazyIndex), `traj.end` (LazyArray), its own return value
  (LazyBool);
* `Optimizer.optimize()` then issues the fused pair of launches the batched engine uses — omgx_goalset_cost_layer(_tiled)
  (goal-set batch + SDF layer of the current trajectory) and omgx_goal_update_optimize (learner + step) — with argument
  lists prepared once (ops.IterationCalls), and ONE pinned download: trajectory | gradient | cost_traj | info | end | goal
  rows | goal point | goal index.  The promises are filled from it;
* a promise that is looked at BEFORE optimize() (`int(traj.goal_idx)` right after update_goal) forces the update on its own
  — goal-set + layer launch, omgx_goal_update, a small download — and optimize() then runs the step alone on the layer that
  launch left;
* host-side changes between the calls (a new `traj.data`, start, goal index, goal set) are found by comparing against the
  mirrors of what the device holds (a few hundred doubles) and uploaded.

Same kernels, same arithmetic as the stand-alone paths (one scene in the latency-mode tiling: a goal's cost is the float32
sum of its parts' sums); there is no CPU path here either.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops


class _Promise:
    __slots__ = ("_loop", "_value")

    def __init__(self, loop):
        self._loop, self._value = loop, None

    def _get(self):
        if self._value is None:
            self._loop.force(self)
        return self._value

    def resolved(self) -> bool:
        return self._value is not None


class LazyIndex(_Promise):
    """`traj.goal_idx` after a deferred Learner.update_goal(): an int as soon as somebody needs the number."""
    __slots__ = ()

    def __int__(self):
        return int(self._get())

    __index__ = __int__

    def __eq__(self, other):
        return int(self) == other

    def __ne__(self, other):
        return int(self) != other

    def __lt__(self, other):
        return int(self) < other

    def __le__(self, other):
        return int(self) <= other

    def __gt__(self, other):
        return int(self) > other

    def __ge__(self, other):
        return int(self) >= other

    def __hash__(self):
        return hash(int(self))

    def __add__(self, other):
        return int(self) + other

    __radd__ = __add__

    def __sub__(self, other):
        return int(self) - other

    def __rsub__(self, other):
        return other - int(self)

    def __mul__(self, other):
        return int(self) * other

    __rmul__ = __mul__

    def __floordiv__(self, other):
        return int(self) // other

    def __mod__(self, other):
        return int(self) % other

    def __bool__(self):
        return int(self) != 0

    def __repr__(self):
        return repr(int(self))

    def __format__(self, spec):
        return format(int(self), spec)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(int(self), dtype=dtype)


class LazyBool(_Promise):
    """update_goal()'s return value ("the goal changed")."""
    __slots__ = ()

    def __bool__(self):
        return bool(self._get())

    def __repr__(self):
        return repr(bool(self))


class _LazyEnd:
    """`traj.end` after a deferred update: the chosen goal configuration [9], a numpy array on first use."""

    def __init__(self, loop, promise):
        self._loop, self._promise, self._value = loop, promise, None
        self.shape, self.dtype, self.ndim = (9,), np.dtype(np.float64), 1

    def _get(self):
        if self._value is None:
            self._value = np.array(self._loop.traj_obj.goal_set[int(self._promise)], np.float64)
        return self._value

    def __array__(self, dtype=None, copy=None):
        v = self._get()
        return v if dtype is None else v.astype(dtype)

    def __getitem__(self, idx):
        return self._get()[idx]

    def __len__(self):
        return 9

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._get(), name)

    def __sub__(self, o):
        return self._get() - o

    def __rsub__(self, o):
        return o - self._get()

    def __add__(self, o):
        return self._get() + o

    __radd__ = __add__


class _NoContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_CONTEXT = _NoContext()


class DeviceLoop:
    LAT_TILING = (4, 10, 4, 1)  # goal parts, layer link groups, layer block, spread: ChompEngine's latency-mode tiling

    def __init__(self, cost, learner):
        self.cost, self.learner = cost, learner
        self.cfg = cost.cfg
        self.device = cost.device
        traj = learner.traj
        self.traj_obj = traj
        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        goal_set = np.asarray(traj.goal_set, np.float64)
        G = goal_set.shape[0]
        n = int(self.cfg.timesteps)
        self.n, self.G = n, G
        model, self.robot = cost._robot_model()
        self.P = model.points_per_link
        reach = learner.env.objects[learner.env.target_idx].reach_grasps
        self.goal_set = torch.as_tensor(goal_set[None], **f64).contiguous()
        self.reach = None
        self.cv_goals = self.goal_set
        if self.cfg.use_standoff:
            r = np.asarray(reach, np.float64)
            if r.ndim != 3 or r.shape[0] != G:
                raise _lib.OmgHipError("cfg.use_standoff needs target_obj.reach_grasps [G,c,9]")
            self.reach = torch.as_tensor(r[None], **f64).contiguous()
            self.cv_goals = self.reach[:, :, -1, :].contiguous()  # online_learner.py:121-125
        self.c = int(self.reach.shape[2]) if self.reach is not None else 1
        c = self.c
        # private copies: an in-place edit of traj.goal_set / reach_grasps on the host is found by comparing bytes (matches())
        self._goal_set_host, self._reach_host = goal_set.copy(), (np.array(reach, np.float64) if self.reach is not None else None)
        # ---- one device buffer, one pinned mirror: [start | traj | grad | cost_traj | info | end | goal rows | goal point | goal index]
        slots = [("start", (1, 9)), ("traj", (1, n, 9)), ("grad", (1, n, 9)), ("cost_traj", (1, n)), ("info", (1, _lib.INFO_STRIDE)),
                 ("end", (1, 9)), ("rows", (1, c, 9)), ("gp", (1, 9)), ("idx", (2,))]
        off, self._slots = 0, {}
        for name, shape in slots:
            nb = int(np.prod(shape)) * (4 if name == "idx" else 8)
            self._slots[name] = (off, nb, shape)
            off += (nb + 7) & ~7
        self.nbytes = off
        self.dev = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.host = torch.zeros(off, dtype=torch.uint8).pin_memory()
        hnp = self.host.numpy()
        self._d, self._h = {}, {}
        for name, (o, nb, shape) in self._slots.items():
            tdt, ndt = (torch.int32, np.int32) if name == "idx" else (torch.float64, np.float64)
            self._d[name] = self.dev[o: o + nb].view(tdt).view(shape)
            self._h[name] = hnp[o: o + nb].view(ndt).reshape(shape)
        self._out_lo = self._slots["traj"][0]
        self._goal_lo = self._slots["end"][0]
        self._h_ptr, self._d_ptr, self._dl = self.host.data_ptr(), self.dev.data_ptr(), _lib.lib().omgx_download_sync
        self.state = ops.learner_state(1, G, dev)
        parts_max = ops.goalset_parts(n, self.LAT_TILING[0])
        self.gcost = torch.zeros((1, G * parts_max), dtype=torch.float32, device=dev)
        self.gcol = torch.zeros((1, G * parts_max), dtype=torch.float32, device=dev)
        self.cv = torch.zeros((1, G), **f64)
        self.pot = torch.zeros((1, n, 10, self.P), dtype=torch.float32, device=dev)
        self.pgrad = torch.zeros((1, n, 10, self.P, 3), dtype=torch.float32, device=dev)
        self.col = torch.zeros((1, n, 10, self.P), dtype=torch.float32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        # outputs of stand-alone evaluations (Learner.cost_vector: an FTC pass on a copy of the state) — never the step's inputs
        self.s_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        self.s_end, self.s_rows, self.s_gp = torch.zeros((1, 9), **f64), torch.zeros((1, c, 9), **f64), torch.zeros((1, 9), **f64)
        self._ticket = 0
        self._calls = None
        self._calls_key = None
        # mirrors of what the device holds (None: unknown / never uploaded)
        self._m_traj = self._m_start = self._m_rows = self._m_gp = self._m_end = None
        self._m_idx = None
        self._layer_valid = False
        self._pending = None       # (LearnerParams, LazyIndex, LazyBool, goal_idx_old) of a deferred update_goal
        self.state_dirty = False   # the device state is ahead of the learner's host attributes

    # ---- views ---------------------------------------------------------------------------------------------------------------
    def d(self, name):
        return self._d[name]

    def h(self, name):
        return self._h[name]

    def matches(self, traj) -> bool:
        """Is this loop still about `traj` with the goal arrays (and the robot points) it was built from?"""
        if traj is not self.traj_obj or int(self.cfg.timesteps) != self.n or self.cost._robot_model()[1] is not self.robot:
            return False
        gs = traj.goal_set
        if gs is not None and (len(gs) != self.G or not np.array_equal(np.asarray(gs, np.float64), self._goal_set_host)):
            return False
        if self._reach_host is not None:
            r = np.asarray(self.learner.env.objects[self.learner.env.target_idx].reach_grasps, np.float64)
            if r.shape != self._reach_host.shape or not np.array_equal(r, self._reach_host):
                return False
        return True

    # ---- launches ------------------------------------------------------------------------------------------------------------
    def _prepared(self):
        scenes = self.cost._scenes()
        key = (id(scenes), id(self.robot), float(self.cfg.time_interval), self.cfg.uncheck_finger_collision == -1)
        if self._calls is None or self._calls_key != key:
            self._calls = ops.IterationCalls(self.robot, self.P, scenes, self.cv_goals, self.cfg.time_interval, self.d("traj"),
                                             (self.pot, self.pgrad, self.col), (self.gcost, self.gcol), self.goal_set, self.reach,
                                             self.state, self.d("idx")[:1], self.d("start"), self.d("end"), self.d("rows"), self.d("gp"),
                                             (self.d("grad"), self.d("cost_traj"), self.d("info")), self.cv, None, None, None, self.flags,
                                             layer_soften_fingers=self.cfg.uncheck_finger_collision == -1, tiling=self.LAT_TILING)
            self._calls_key, self._scenes_ref = key, scenes
            self._layer_valid = False  # pot / pgrad / col on the device were computed against the OLD table
        return self._calls

    def _stream(self):
        """Raw handle of torch's current stream on the loop's device (a C call: torch.cuda.current_stream() builds a Stream object
        per call, several microseconds of a 90 us iteration)."""
        return torch._C._cuda_getCurrentRawStream(self._dev_index)

    def _on_device(self):
        """Context that makes the loop's device current — nothing at all when it already is (the usual case)."""
        if torch.cuda.current_device() == self._dev_index:
            return _NO_CONTEXT
        return torch.cuda.device(self.device)

    def _upload(self, name, value):
        self.h(name)[...] = np.asarray(value).reshape(self._slots[name][2])
        o, nb, _ = self._slots[name]
        self.dev[o: o + nb].copy_(self.host[o: o + nb], non_blocking=True)

    def sync_inputs(self, traj):
        """Bring the device's trajectory / start up to the host's if somebody changed them since the last download (the mirrors
        are the bytes of what the device holds: one memcmp each)."""
        data = traj.data
        if not (type(data) is np.ndarray and data.dtype == np.float64 and data.flags.c_contiguous):
            data = np.ascontiguousarray(data, np.float64)
        if data.shape != (self.n, 9):
            raise _lib.OmgHipError(f"trajectory has shape {data.shape}, cfg.timesteps is {self.n}")
        b = data.tobytes()
        if b != self._m_traj:
            self._upload("traj", data)
            self._m_traj = b
            self._layer_valid = False
        b = np.asarray(traj.start, np.float64).tobytes()
        if b != self._m_start:
            self._upload("start", np.frombuffer(b, np.float64))
            self._m_start = b

    def _sync_goal(self, traj, goal_rows, goal_point):
        """The step alone reads end / goal rows / goal point from the device: make them the host's."""
        for name, attr, val in (("end", "_m_end", traj.end), ("rows", "_m_rows", goal_rows), ("gp", "_m_gp", goal_point)):
            b = np.ascontiguousarray(val, np.float64).tobytes()
            if b != getattr(self, attr):
                self._upload(name, np.frombuffer(b, np.float64).reshape(self._slots[name][2]))
                setattr(self, attr, b)

    def _download(self, lo):
        _lib.check(self._dl(self._h_ptr + lo, self._d_ptr + lo, self.nbytes - lo, self._stream()), "omgx_download_sync")

    def _take_goal(self):
        """After a download that covers the goal block: the mirrors follow what the learner wrote on the device."""
        self._m_end, self._m_rows, self._m_gp = self.h("end").tobytes(), self.h("rows").tobytes(), self.h("gp").tobytes()
        self._m_idx = int(self.h("idx")[0])

    # ---- the learner's side --------------------------------------------------------------------------------------------------
    def defer_update(self, lprm):
        """Learner.update_goal(): note the update, hand out promises.  Returns (LazyIndex, LazyBool)."""
        if self._pending is not None:
            self.flush()
        self.sync_inputs(self.traj_obj)  # the update is about the trajectory of THIS moment
        old = self.traj_obj.goal_idx
        idx, changed = LazyIndex(self), LazyBool(self)
        self._pending = (lprm, idx, changed, old)
        return idx, changed

    def _resolve(self, pend):
        lprm, idx, changed, old = pend
        new = int(self.h("idx")[0])
        idx._value = new
        changed._value = bool(new != int(old))
        self.learner._goal_taken(new)
        # the trajectory keeps plain values once they are known (deepcopy / pickle / isinstance checks of a stored history)
        t = self.traj_obj
        if getattr(t, "goal_idx", None) is idx:
            t.goal_idx = new
        if isinstance(getattr(t, "end", None), _LazyEnd) and t.end._promise is idx:
            t.end = t.end._get()

    def force(self, promise):
        """A promise is needed before optimize() ran: do the learner's update on its own."""
        if self._pending is None or promise not in self._pending[1:3]:
            raise RuntimeError("this value belongs to an update that was already resolved")  # cannot happen: resolved promises hold a value
        self.flush()

    def flush(self):
        """Goal-set batch + layer launch, omgx_goal_update, a small download — the deferred update on its own."""
        pend, self._pending = self._pending, None
        if pend is None:
            return
        lprm = pend[0]
        calls, st = self._prepared(), self._stream()
        with self._on_device():
            if lprm.alg != _lib.ALG["Proj"]:
                calls.goalset_layer(lprm.start_idx, False, None, None, st)
                self._layer_valid = True
            calls.goal_update(lprm, st)
            self._download(self._goal_lo)
        self.state_dirty = True
        self._take_goal()
        self._resolve(pend)

    # ---- the optimiser's side ------------------------------------------------------------------------------------------------
    def optimize(self, traj, prm, goal_rows, goal_point):
        """One Optimizer.optimize call; the results are in the host views until the next call.  goal_rows / goal_point: callables
        giving the reference's host-side choice (optimizer.py:93-99), evaluated only when the step runs without the learner."""
        self.sync_inputs(traj)
        calls, st = self._prepared(), self._stream()
        pend = self._pending
        with self._on_device():
            if pend is not None and pend[0].alg != _lib.ALG["Proj"]:
                self._pending = None
                calls.goalset_layer(pend[0].start_idx, False, None, None, st)
                self._ticket = self._ticket + 1 if self._ticket < 0xfffff0 else 1
                calls.update(pend[0], prm, True, self._ticket, False, st)
                self.state_dirty = True
            else:
                if pend is not None:
                    self.flush()
                    pend = None
                self._sync_goal(traj, goal_rows(), goal_point())
                if not self._layer_valid:
                    calls.layer_only(st)
                calls.step(prm, False, st)
            self._download(self._out_lo)
        # what the launch left: the trajectory may have moved (then the layer is stale), the goal block follows the learner
        b = self.h("traj").tobytes()
        self._layer_valid = b == self._m_traj  # the layer on the device belongs to the trajectory the launch started from
        if pend is not None:
            self._take_goal()
            self._resolve(pend)
        self._m_traj = b
        return self

    def update_now(self, lprm):
        """Learner.update_goal_dist() called on its own: goal-set batch (+ layer) and omgx_goal_update at once; returns the goal index."""
        if self._pending is not None:
            self.flush()
        self.sync_inputs(self.traj_obj)
        calls, st = self._prepared(), self._stream()
        with self._on_device():
            if lprm.alg != _lib.ALG["Proj"]:
                calls.goalset_layer(lprm.start_idx, False, None, None, st)
                self._layer_valid = True
            calls.goal_update(lprm, st)
            self._download(self._goal_lo)
        self.state_dirty = True
        self._take_goal()
        return self._m_idx

    def pull_state(self):
        """Learner state -> host (sum_costs, p, experts_p, q, experts_costs), only when somebody reads them."""
        s = self.state[0].cpu().numpy()
        self.state_dirty = False
        return s
