#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host-buffer entry point (volume_gpu_render):
default bench workload, rgba8 only (what the reference's SDL texture receives) and all
five planes.  Never the bench `value`; quoted in DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
vra = load_package()
scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", "dragon.off"), 512, dense=False, svo=True)
gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
frames = vra.orbit_frames(range(16), 4096, (512,) * 3, (128,) * 3, 1920, 1080, radius=1.0)
for planes in (("rgba8",), ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")):
    gpu.render(frames, planes=planes)
    t = time.perf_counter()
    for _ in range(5):
        gpu.render(frames, planes=planes)
    dt = (time.perf_counter() - t) / 5
    print(f"{len(planes)} plane(s) to pageable host memory: {16 * 1920 * 1080 / dt / 1e6:.0f} Mrays/s ({dt * 1e3:.1f} ms per 16 views)")
    pins = {q: vra.gpu.PinnedArray((16, 1080, 1920), "float32" if q == "depth" else "uint32") for q in planes}
    arrays = {q: a.array for q, a in pins.items()}
    gpu.render_into(frames, arrays)
    ref = gpu.render(frames, planes=planes)
    same = all((arrays[q].view("uint32") == ref[q].view("uint32")).all() for q in planes)
    t = time.perf_counter()
    for _ in range(5):
        gpu.render_into(frames, arrays)
    dt = (time.perf_counter() - t) / 5
    print(f"{len(planes)} plane(s) to pinned host memory:   {16 * 1920 * 1080 / dt / 1e6:.0f} Mrays/s ({dt * 1e3:.1f} ms per 16 views), equal to the pageable result: {same}")
    for a in pins.values():
        a.close()
