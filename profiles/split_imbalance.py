#!/usr/bin/env python3
"""Load balance of the single-frame split (config 4: galleon 2048^3 octree, one 7680x4320 frame) on ONE GPU: every rank's
share rendered alone, kernel time by HIP events.  Contiguous row bands (shard.band_rows) against stripes of S rows dealt
round-robin (rank r renders stripes r, r + N, ...; one sub-rectangle each, all in one launch).
    python profiles/split_imbalance.py [--stripe rows] [--views 0,700,1400]"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--stripe", type=int, default=-1, help="-1: bench.py's rule (at most 64 stripes per rank, at least 16 rows)")
ap.add_argument("--views", default="0,700,1400,2100")
a = ap.parse_args()
vra = load_package()
p = B.PRESETS[4]
res, W, H = p["res"], p["width"], p["height"]
mesh = vra.HostMesh(os.path.join(ROOT, "assets", p["scene"]), 2)
gpu = vra.GpuScene.voxelize(mesh.desc, res, dense=False, svo=True, device=0)
mesh.close()
dev = torch.device("cuda:0")
planes = {q: torch.empty((H, W), dtype=torch.float32 if q == "depth" else torch.int32, device=dev) for q in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
ptrs = {q: t.data_ptr() for q, t in planes.items()}
gpu.set_profiling(True)


def timed(frames):
    ts = []
    for _ in range(5):
        gpu.render_device(frames, ptrs)
        torch.cuda.synchronize()
        ms, n = gpu.kernel_time()
        ts.append(ms)
    return statistics.median(ts)


for view in [int(v) for v in a.views.split(",")]:
    full = vra.orbit_frames([view], B.ORBIT_VIEWS, (res,) * 3, (res // 4,) * 3, W, H, radius=p["radius"])[0]
    t_full = timed(full)
    print(f"view {view}: whole frame {t_full:.3f} ms")
    for world in (2, 4, 8):
        bands = [timed(full.band(*vra.shard.band_rows(H, r, world))) for r in range(world)]
        S = a.stripe if a.stripe > 0 else max(16, -(-H // (64 * world * 8)) * 8)
        n_str = (H + S - 1) // S
        stripes = []
        for r in range(world):
            mine = list(range(r, n_str, world))
            fr = (vra.FrameDesc * len(mine))()
            for i, s in enumerate(mine):
                fr[i] = full.band(s * S, min(S, H - s * S))
            stripes.append(timed(fr))
        f = lambda t: f"max {max(t):.3f} ms, imbalance {max(t) / (sum(t) / len(t)) - 1:+.1%}, speed-up over one GPU {t_full / max(t):.2f}x of {len(t)}"
        print(f"  N = {world}: bands   {f(bands)}")
        print(f"         stripes {f(stripes)}  ({S}-row stripes)")
