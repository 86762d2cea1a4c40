#!/usr/bin/env python3
"""Per-ray fixed cost of the frame kernel: config 2's scene and frame shape, but every camera
looks AWAY from the volume, so both passes end at their first bounds test (0 steps).  What is
left is direction set-up, two normalisations, two make_ray, two enter_volume and the stores."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
vra = load_package()
RES = int(os.environ.get("FIXED_COST_RES", "1024"))      # 1024: the headline workload (4 x 4K); 512: config 2 (16 x 1080p)
scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", "dragon.off"), RES, dense=False, svo=True)
gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
W, H, nv = (3840, 2160, 4) if RES == 1024 else (1920, 1080, 16)
def frames(away):
    out = []
    for k in range(nv):
        pos, view = vra.orbit_camera(k * 37, 4096, 1.0)
        if away:
            c = np.array([0.5, 0.5, 0.5], np.float32)
            view = vra.look_at((0, 0, 0), c - pos, (0, 0, -1))     # the reference looks along -camDir (main.cpp:219,397)
        out.append(vra.frame_from_camera(view, pos, (RES,) * 3, (RES // 4,) * 3, W, H))
    return out
dev = torch.device("cuda:0")
planes = {q: torch.empty((nv, H, W), dtype=torch.float32 if q == "depth" else torch.int32, device=dev)
          for q in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
ptrs = {q: t.data_ptr() for q, t in planes.items()}
gpu.set_profiling(True)
def measure(fr, p):
    ts = []
    for _ in range(8):
        gpu.render_device(fr, p)
        torch.cuda.synchronize()
        ms, n = gpu.kernel_time()
        ts.append(ms / n)
    return statistics.median(ts)
for away in (False, True):
    fr = frames(away)
    t = measure(fr, ptrs)
    steps = planes["steps_lo"].sum(dtype=torch.int64).item() / (nv * W * H)
    print(f"{'looking away' if away else 'looking at the volume'}: kernel {t:.4f} ms, mean low-grid steps {steps:.2f}")
fr = frames(True)
print(f"looking away, rgba8 plane only: kernel {measure(fr, {'rgba8': ptrs['rgba8']}):.4f} ms")
for f in fr:
    f.flags = 1
print(f"looking away, naive frames (no low-grid pass set-up): kernel {measure(fr, ptrs):.4f} ms")
