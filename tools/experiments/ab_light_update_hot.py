"""ab_light_update.py through the engine's HOT path (prepared calls, the layout rule's pipeline): the update launch of every part replaced
by stand-in workgroups of tools/experiments/spin_update.hip that hold (threads, LDS bytes, VGPRs) for `usec`.  Timing only (the trajectories stay
where they are: compare the stand-in shapes with each other and with `none`).
    python tools/experiments/ab_light_update_hot.py --scenes 13 --goals 128 --shape none | real | THREADSxLDSxVGPRSxUSEC [--parts K]"""
import argparse, copy, ctypes, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="none")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--scenes", type=int, default=13)
ap.add_argument("--goals", type=int, default=128)
ap.add_argument("--parts", type=int, default=0)
ap.add_argument("--waypoints", type=int, default=30)
ap.add_argument("--objects", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
spin = ctypes.CDLL(str(ROOT / "tools" / "experiments" / "spin_update.so"))
spin.spin_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, a.waypoints, 64, 0, False, num_objects=a.objects)
eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=a.scenes, device=dev, ol_alg="MD")
if a.parts:
    eng.pipeline = a.parts
eng.pose_hand_over(True)

def step():
    eng.t = 0
    eng.iterate(0)

for _ in range(30):
    step()
eng.join(); torch.cuda.synchronize()
shape = None if a.shape in ("none", "real") else [int(x) for x in a.shape.split("x")]
if a.shape != "real":
    for p in (eng._parts or [eng]):
        calls = p._hot_calls()
        S = p.S
        if shape:
            calls.update = (lambda S: lambda prm, params, split, ticket, stop, stream: spin.spin_launch(2 * S, shape[0], shape[1], shape[2], shape[3], stream))(S)
        else:
            calls.update = lambda *args: None
out = []
for _ in range(3):
    eng.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    eng.join(); torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / a.iters * 1e3, 4))
print(json.dumps({"shape": a.shape, "scenes": a.scenes, "goals": a.goals, "parts": eng.pipeline, "ms_per_step": out}), flush=True)
