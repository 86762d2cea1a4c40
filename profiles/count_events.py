#!/usr/bin/env python3
"""What the whole-traversal region loop does on a bench workload, counted by the `count` variant
build (make variants VARIANTS=count:-DVK_COUNT=1): per launch, waves that ran the loop, wave-steps,
steps that needed attention, region changes, grid exits, steps with a lane in a region / brick that
holds voxels, brick-word loads, hits.   python profiles/count_events.py [--config 0] [--mode beam]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import torch  # noqa
import ab_bench, bench as B
from __graft_entry__ import load_package

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=0)
ap.add_argument("--mode", default="beam")
ap.add_argument("--camera", choices=["orbit", "inside", "fill"], default="orbit")
a = ap.parse_args()
vra = load_package()
mod = ab_bench.load_variant("count", os.path.join(ROOT, "volume-renderer_amd", "lib", "variants", "libvolly_gpu_count.so"))
p = dict(B.PRESETS[a.config]); p["scene"] = p["scene"].split(",")[0]
res, W, H, nv = p["res"], p["width"], p["height"], p["views"]
sc = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", p["scene"]), res, dense=p["target"] == "dense", svo=p["target"] != "dense")
g = mod.GpuScene.upload_dense(sc.target, sc.low) if p["target"] == "dense" else mod.GpuScene.upload_svo(sc.svo, sc.low)
frames = vra.orbit_frames(list(range(nv)), 4096, (res,) * 3, (res // 4,) * 3, W, H, radius=B.CAMERA_RADIUS.get(a.camera, p["radius"]), naive=a.mode == "naive")
out = torch.empty((nv, H, W), dtype=torch.int32, device="cuda:0")
mod.lib.volly_gpu_debug_counters.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
mod.lib.volly_gpu_debug_counters(None, 1)
g.render_device(frames, {"steps_lo": out.data_ptr()})
torch.cuda.synchronize()
k = (C.c_ulonglong * 16)()
mod.lib.volly_gpu_debug_counters(k, 0)
names = ["waves in the loop", "wave-steps", "attention steps", "steps with a region change", "steps with a grid exit",
         "steps with a lane in a region with voxels", "steps with a lane in a brick with voxels", "steps with a brick-word load", "steps with a hit", "run-ahead blocks (4 steps each)"]
rays = nv * W * H
print(f"config {a.config} {a.mode} {a.camera}: {rays} rays, {rays // 64} tiles, lane-steps {int(out.sum(dtype=torch.int64).item())}")
ws = k[1]
for n, v in zip(names, k):
    print(f"  {n:45s} {v:12d}  {v / ws:.4f} per wave-step")
