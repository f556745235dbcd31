"""Statistics of the goal-set batch of bench.py's workload: how many (row, object) and (point, object) pairs survive
each culling level of k_goalset_queue (row test and point test against the objects' influence regions).  CPU only, uses the oracle's FK (a tool, not a product path).

    python tests/fuzz/pair_density.py [num_scenes] [num_goals]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT)]
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n = 30
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, True)
    blob = model.blob()
    P = model.collision_points.shape[1]
    D = 528 + 30 * P
    pts = blob[D + 246: D + 246 + 30 * P].reshape(10, P, 3)
    rad = blob[D + 306 + 30 * P: D + 316 + 30 * P]
    tot = dict(rows=0, rows_near=0, pairs=0, pairs_rowlive=0, pairs_box=0, pairs_inrange=0, pairs_contrib=0,
               wave_obj_iters=0, wave_obj_iters_live=0, compact_batches=0, link_tests=0, link_tests_needed=0,
               merged_batches=0, wave_lb_iters_live=0, box_in_wave_hist=np.zeros(9, np.int64))
    for s in range(S):
        t = (np.arange(1, n + 1) / (n + 1))[None, :, None]
        q = start[s][None, None, :] + t * (goals[s][:, None, :] - start[s][None, None, :])  # [G,n,9]
        pose, _, _ = orc.fk(blob, q.reshape(-1, 9))
        pose = pose.reshape(G, n, 10, 4, 4)
        # NB oracle poses have center_offset applied; PTS in the blob are relative to the pose used by the kernel
        # (close enough for statistics: recompute the points through orc.config_points for exactness if needed)
        R, T = pose[..., :3, :3], pose[..., :3, 3]
        x = np.einsum("gnlij,lpj->gnlpi", R, pts) + T[..., None, :]  # [G,n,10,P,3]
        ctr = T
        recs = batch.objects[batch.scene_begin[s]: batch.scene_begin[s + 1]]
        cnt_all = 0
        for r in recs:
            if r["disabled"] > 0:
                continue
            Ti = r["pose_inv"].reshape(3, 4).astype(np.float64)
            lo, hi, dim = r["lo"].astype(np.float64), r["hi"].astype(np.float64), r["dim"].astype(np.int64)
            rc, rh, rr = r["rb_c"].astype(np.float64), r["rb_h"].astype(np.float64), float(r["rb_r"])  # influence region (rounded box)
            uc = ctr @ Ti[:, :3].T + Ti[:, 3] - lo  # [G,n,10,3]
            near = (np.maximum(np.abs(uc - rc) - rh, 0.0) ** 2).sum(-1) <= (rr + rad[None, None, :]) ** 2
            u = x @ Ti[:, :3].T + Ti[:, 3] - lo  # [G,n,10,P,3]
            inbox = (np.maximum(np.abs(u - rc) - rh, 0.0) ** 2).sum(-1) <= float(r["rb_r2"])
            g = u / (hi - lo) * dim - 0.5
            i0 = np.trunc(g).astype(np.int64)
            inr = np.all((g > -1) & (i0 >= 0) & (i0 < dim - 1), axis=-1)
            grid = batch.pool[r["grid_offset"]: r["grid_offset"] + dim.prod()].reshape(dim)
            ic = np.clip(i0, 0, dim - 2)
            f = np.clip(g - ic, 0, 1)
            v = 0
            for dx in (0, 1):
                for dy in (0, 1):
                    for dz in (0, 1):
                        w = (f[..., 0] if dx else 1 - f[..., 0]) * (f[..., 1] if dy else 1 - f[..., 1]) * (f[..., 2] if dz else 1 - f[..., 2])
                        v = v + w * grid[ic[..., 0] + dx, ic[..., 1] + dy, ic[..., 2] + dz]
            contrib = inr & ((v <= r["epsilon"]) | (v < r["clearance"]))
            live = near[..., None] & inbox
            tot["rows"] += near.size
            tot["rows_near"] += int(near.sum())
            tot["pairs"] += inbox.size
            tot["pairs_rowlive"] += int(near.sum()) * P
            tot["pairs_box"] += int(live.sum())
            tot["pairs_inrange"] += int((live & inr).sum())
            tot["pairs_contrib"] += int((live & contrib).sum())
            # wave structure of k_goalset_compact<2>: a wave = 4 waypoints x 16 point lanes, 2 links per batch
            nb = -(-n // 16) * 16
            nearp = np.zeros((G, nb, 10), bool); nearp[:, :n] = near
            livep = np.zeros((G, nb, 10, P), bool); livep[:, :n] = live
            nw = nearp.reshape(G, nb // 4, 4, 5, 2)     # [G, wave, wp-in-wave, link-batch, k]
            lw = livep.reshape(G, nb // 4, 4, 5, 2, P)
            it = nw.any(axis=(2, 4))                    # wave executes this object iteration
            cnt = lw.sum(axis=(2, 4, 5))                # live lanes in it
            tot["wave_obj_iters"] += it.size
            tot["wave_obj_iters_live"] += int(it.sum())
            tot["compact_batches"] += int((-(-cnt // 64)).sum())
            cnt_all = cnt_all + cnt
            per_link = nw.any(axis=2)                   # [G, wave, lb, k]: link k has a row in reach of this object
            tot["link_tests"] += int(it.sum()) * 2      # pair_prepare calls per executed iteration today
            tot["link_tests_needed"] += int(per_link.sum())
            tot["box_in_wave_hist"] += np.bincount(np.minimum(cnt[it] // 16, 8), minlength=9)
            need = (grid <= r["epsilon"]) | (grid < r["clearance"])
            idx = np.argwhere(need)
            print(f"scene {s} obj eps={r['epsilon']:.2f} dim={tuple(dim)} rows_near={near.mean():.3f} box={live.mean():.3f} "
                  f"contrib={(live & contrib).mean():.3f} influence voxel box={idx.min(0)}..{idx.max(0)}")
        tot["merged_batches"] += int((-(-cnt_all // 64)).sum())
        tot["wave_lb_iters_live"] += int((cnt_all > 0).sum())
    for k, v in tot.items():
        print(f"{k:22s} {v}")
    print("row survive            %.3f" % (tot["rows_near"] / tot["rows"]))
    print("pair box | row live    %.3f" % (tot["pairs_box"] / max(tot["pairs_rowlive"], 1)))
    print("pair box / all         %.3f" % (tot["pairs_box"] / tot["pairs"]))
    print("contrib / box          %.3f" % (tot["pairs_contrib"] / max(tot["pairs_box"], 1)))
    print("lanes per exec'd iter  %.1f of 128" % (tot["pairs_box"] / max(tot["wave_obj_iters_live"], 1)))
    print("exact batches per exec'd iter %.2f" % (tot["compact_batches"] / max(tot["wave_obj_iters_live"], 1)))


    # what-if figures: far tests skipped per link, exact batches if the objects of a link batch shared one queue
    print("far tests needed / done %.3f" % (tot["link_tests_needed"] / max(tot["link_tests"], 1)))
    print("exact batches merged over objects / today %.3f" % (tot["merged_batches"] / max(tot["compact_batches"], 1)))


if __name__ == "__main__":
    main()
