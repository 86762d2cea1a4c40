#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of libvolly_gpu.so in ONE process
(cdna_hip_programming.md rule 24): N variants x M rounds, HIP-event kernel time of
render_frames only, median and min per variant.

    python profiles/ab_bench.py [--config 2] [--mode beam] [--rounds 7] name=path/to/lib.so[@ENV=value...] ...
(@ENV=value: environment the library reads while the variant's scene is created, e.g. @VOLLY_EMPTY_CUBES=4)
"""
import argparse, importlib.util, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime for every variant)
from __graft_entry__ import load_package, PKG_DIR
import bench as B


def load_variant(name, path):
    os.environ["VOLLY_GPU_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location(f"volume_renderer_amd.gpu_{name}", os.path.join(PKG_DIR, "gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "volume_renderer_amd"
    spec.loader.exec_module(mod)
    os.environ.pop("VOLLY_GPU_LIB", None)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--mode", default="beam")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--res", type=int, default=None, help="voxelize at this side instead of the preset's (e.g. one that is no multiple of 16)")
    ap.add_argument("--target", choices=["dense", "svo"], default=None, help="target structure instead of the preset's")
    ap.add_argument("--noise-log2-inv-p", type=int, default=8, help="noise scene: a voxel is filled with probability 2^-n")
    ap.add_argument("--camera", choices=["orbit", "startup", "inside", "fill"], default="orbit", help="bench.py's camera regimes")
    ap.add_argument("--planes", default="rgba8,hit_index,steps_lo,steps_hi,depth", help="output planes written in the timed runs")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    vra = load_package()
    p = dict(B.PRESETS[a.config])
    p["scene"] = p["scene"].split(",")[0]
    if a.views:
        p["views"] = a.views
    if a.res:
        p["res"] = a.res
    if a.target:
        p["target"] = a.target
    res, W, H, nv = p["res"], p["width"], p["height"], p["views"]
    dense = p["target"] == "dense"
    scene = None
    if p["scene"] != "noise":
        scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", p["scene"]), res, dense=dense, svo=not dense)
    if a.camera == "startup":
        view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(0, 0), (0, 0, -1))
        one = (vra.FrameDesc * nv)()
        for i in range(nv):
            one[i] = vra.frame_from_camera(view, (0.0, 0.0, 0.0), (res,) * 3, (res // 4,) * 3, W, H, naive=a.mode == "naive")
        frames = [one] * 3
    else:
        radius = B.CAMERA_RADIUS.get(a.camera, p["radius"])
        frames = [vra.orbit_frames([(s * nv + i) % 4096 for i in range(nv)], 4096, (res,) * 3, (res // 4,) * 3, W, H,
                                   radius=radius, naive=a.mode == "naive") for s in range(3)]
    dev = torch.device("cuda:0")
    planes = {q: torch.empty((nv, H, W), dtype=torch.float32 if q == "depth" else torch.int32, device=dev)
              for q in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
    ptrs = {q: t.data_ptr() for q, t in planes.items()}
    gpus, sums = {}, {}
    for v in a.variants:
        name, path = v.split("=", 1)
        path, *envs = path.split("@")
        for e in envs:
            os.environ[e.split("=", 1)[0]] = e.split("=", 1)[1]
        mod = load_variant(name, path)
        if scene is None:
            g = mod.GpuScene.noise(res, 1, 1 << (32 - a.noise_log2_inv_p))
        elif dense:
            g = mod.GpuScene.upload_dense(scene.target, scene.low)
        else:
            g = mod.GpuScene.upload_svo(scene.svo, scene.low)
        for e in envs:
            os.environ.pop(e.split("=", 1)[0], None)
        g.render_device(frames[0], ptrs)
        torch.cuda.synchronize()
        sums[name] = tuple(int(planes[q].sum(dtype=torch.int64).item()) for q in ("hit_index", "steps_lo", "steps_hi", "rgba8"))
        g.set_profiling(True)
        gpus[name] = g
    ref = next(iter(sums.values()))
    for name, s in sums.items():
        print(f"{name}: plane sums {'match' if s == ref else 'DIFFER ' + str(s) + ' vs ' + str(ref)}")
    ptrs = {q: v for q, v in ptrs.items() if q in a.planes.split(",")}
    times = {n: [] for n in gpus}
    for r in range(a.rounds):
        for name, g in gpus.items():
            for f in frames:
                g.render_device(f, ptrs)
            torch.cuda.synchronize()
            ms, n = g.kernel_time()
            times[name].append(ms / n)
    rays = nv * W * H
    for name, t in times.items():
        print(f"{name:>12}: median {statistics.median(t):.4f} ms  min {min(t):.4f} ms  -> {rays / statistics.median(t) / 1e3:.0f} Mrays/s (kernel only)")


if __name__ == "__main__":
    main()
