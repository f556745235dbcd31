import sys, time, copy
sys.path.insert(0, '.')
import numpy as np, torch
dev = torch.device("cuda:0")
torch.from_numpy(np.zeros(1 << 18, np.float32)).to(dev); torch.cuda.synchronize()
def T(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter()-t0)*1e3:8.2f} ms"); return r
a = T("empty", lambda: torch.empty((100, 64, 9), dtype=torch.float64, device=dev))
z = T("zeros f64", lambda: torch.zeros((100, 16), dtype=torch.float64, device=dev))
z2 = T("zeros f32", lambda: torch.zeros((100, 64), dtype=torch.float32, device=dev))
zi = T("zeros i32", lambda: torch.zeros(100, dtype=torch.int32, device=dev))
o = T("ones i32", lambda: torch.ones(100, dtype=torch.int32, device=dev))
g = T("as_tensor h2d", lambda: torch.as_tensor(np.random.rand(100, 64, 9), dtype=torch.float64, device=dev))
c = T("contiguous().clone()", lambda: g[:, 0].contiguous().clone())
ar = T("arange", lambda: torch.arange(100, device=dev))
ix = T("index_select", lambda: torch.index_select(g.view(6400, 9), 0, ar * 64 + zi.long()))
cp = T("copy_ d2d", lambda: c.copy_(ix))
w = T("where/repeat", lambda: torch.where(torch.arange(64, device=dev)[None, :] < torch.full((100, 1), 60.0, device=dev, dtype=torch.float64), 1.0 / 60, 0.0).repeat(1, 6))
st = T("slice assign", lambda: z.__setitem__((slice(None), slice(0, 5)), 0.2))
from omg_planner_amd import ops, robot as rb
m = rb.PandaModel(seed=0)
T("robot_blob", lambda: ops.robot_blob(m, dev))
T("learner_state", lambda: ops.learner_state(100, 64, dev))
T("props", lambda: torch.cuda.get_device_properties(dev).multi_processor_count)
