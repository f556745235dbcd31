import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
dev = torch.device("cuda:0")
t0 = time.perf_counter(); x = torch.zeros(1, device=dev); torch.cuda.synchronize(); print("context ms", round((time.perf_counter() - t0) * 1e3, 1))
vols = [np.random.rand(64, 64, 64).astype(np.float32) for _ in range(400)] + [np.random.rand(128, 96, 32).astype(np.float32) for _ in range(100)]
offs = np.concatenate([[0], np.cumsum([v.size for v in vols])])
t0 = time.perf_counter(); pool = torch.empty(int(offs[-1]), dtype=torch.float32, device=dev); torch.cuda.synchronize(); print("alloc pool ms", round((time.perf_counter() - t0) * 1e3, 1))
t0 = time.perf_counter(); y = torch.from_numpy(np.zeros(16, np.float32)).to(dev); torch.cuda.synchronize(); print("tiny h2d ms", round((time.perf_counter() - t0) * 1e3, 2))
t0 = time.perf_counter(); y = torch.from_numpy(np.zeros(1 << 18, np.float32)).to(dev); torch.cuda.synchronize(); print("1MB h2d ms", round((time.perf_counter() - t0) * 1e3, 2))
t0 = time.perf_counter(); y = torch.from_numpy(np.zeros(1 << 18, np.float32)).to(dev); torch.cuda.synchronize(); print("1MB h2d again ms", round((time.perf_counter() - t0) * 1e3, 2))
def serial(k=len(vols)):
    for o, v in list(zip(offs, vols))[:k]:
        pool[o:o + v.size].copy_(torch.from_numpy(v.reshape(-1)), non_blocking=True)
    torch.cuda.synchronize()
for k in (10, 50, 500, 500):
    t0 = time.perf_counter(); serial(k); print("serial", k, round((time.perf_counter() - t0) * 1e3, 1), "ms")
