#!/usr/bin/env python3
"""Repeatability under load: render the same batch of a bench config N times back to back and require
every plane of every run to equal the first run's byte for byte (an intermittent fault in the
kernels -- a missed wait state, a race -- shows up as a differing run).
    python profiles/repeat_check.py [--config 2] [--runs 200]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--runs", type=int, default=200)
    a = ap.parse_args()
    vra = load_package()
    p = dict(B.PRESETS[a.config])
    p["scene"] = p["scene"].split(",")[0]
    res, W, H, nv = p["res"], p["width"], p["height"], p["views"]
    dense = p["target"] == "dense"
    if p["scene"] == "noise":
        g = vra.GpuScene.noise(res, 1, 1 << 24)
    else:
        scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", p["scene"]), res, dense=dense, svo=not dense)
        g = vra.GpuScene.upload_dense(scene.target, scene.low) if dense else vra.GpuScene.upload_svo(scene.svo, scene.low)
    frames = vra.orbit_frames(list(range(nv)), 4096, (res,) * 3, (res // 4,) * 3, W, H, radius=p["radius"])
    dev = torch.device("cuda:0")
    names = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")
    def planes():
        return {q: torch.empty((nv, H, W), dtype=torch.int32, device=dev) for q in names}
    first = planes()
    g.render_device(frames, {q: t.data_ptr() for q, t in first.items()})
    torch.cuda.synchronize()
    cur = planes()
    ptrs = {q: t.data_ptr() for q, t in cur.items()}
    bad = 0
    for r in range(a.runs):
        g.render_device(frames, ptrs)
        torch.cuda.synchronize()
        for q in names:
            if not torch.equal(cur[q], first[q]):
                bad += 1
                print(f"run {r}: plane {q} differs in {(cur[q] != first[q]).sum().item()} values")
    print(f"config {a.config}: {a.runs} runs of {nv} x {W}x{H}, {bad} differing planes")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
