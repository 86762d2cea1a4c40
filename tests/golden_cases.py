"""The golden cases (SURVEY 8c fixture list) shared by tests/golden/make_golden.py, the
CPU golden test (oracle + host pipeline reproduce the committed vectors) and the GPU
golden test (the HIP path reproduces them through the C-ABI)."""
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
F = np.float32


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def _look_from(vra, cam_pos):
    """camDir as the reference builds it (main.cpp:394-397), pointing from the volume centre to the camera."""
    p = np.asarray(cam_pos, F)
    d = (p - F(0.5)).astype(F)
    d = (d * F(F(1) / np.sqrt(F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])))).astype(F)
    d = (d + F(0.000003)).astype(F)
    d = (d * F(F(1) / np.sqrt(F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])))).astype(F)
    return vra.look_at((0, 0, 0), d, (0, 0, -1))


MOUSE_CAMS = [((0, 0), (0.0, 0.0, 0.0)),            # the reference's start-up state (main.h:96, mouse at 0,0)
              ((650, 390), (1.4, 0.3, 0.75)),
              ((-140, 520), (0.45, 1.6, 0.2))]

CASES = {}
for lod in (0, 2, 3):
    CASES[f"ico_lodset_lod{lod}"] = dict(scene_key="ico_lodset", kind="lodset", mesh="icosahedron.ply", cur_lod=lod,
                                         target="dense", w=64, h=48, cams=("mouse", MOUSE_CAMS), naive=False, store=True)
for tgt in ("dense", "svo"):
    CASES[f"sphere64_{tgt}"] = dict(scene_key="sphere64", kind="sphere", res=64, target=tgt, w=64, h=48,
                                    cams=("orbit", [(1, 12, 0.9), (7, 12, 1.3)]), naive=False, store=True)
    # BASELINE config 1: teapot.off at 128^3, 640x480, camera (0.5,-1.2,0.5) looking at the centre
    CASES[f"teapot128_{tgt}_beam"] = dict(scene_key="teapot128", kind="mesh", mesh="teapot.off", res=128, target=tgt,
                                          w=640, h=480, cams=("look", [(0.5, -1.2, 0.5), (1.1, 0.9, 0.8)]), naive=False,
                                          store=True, crop=10)
    CASES[f"teapot128_{tgt}_naive"] = dict(scene_key="teapot128", kind="mesh", mesh="teapot.off", res=128, target=tgt,
                                           w=640, h=480, cams=("look", [(0.5, -1.2, 0.5)]), naive=True, store=False)
# BASELINE config 2: dragon.off at 512^3 octree, 1920x1080 (CRCs only)
CASES["dragon512_svo_1080p"] = dict(scene_key="dragon512", kind="mesh", mesh="dragon.off", res=512, target="svo",
                                    w=1920, h=1080, cams=("orbit", [(5, 32, 1.0)]), naive=False, store=False, dense=False)
# BASELINE config 4: galleon.ply at 2048^3 octree, the whole 7680x4320 frame (33 M rays; CRCs only; hit_index mod 2^32)
CASES["galleon2048_svo_8k"] = dict(scene_key="galleon2048", kind="mesh", mesh="galleon.ply", res=2048, target="svo",
                                   w=7680, h=4320, cams=("orbit", [(300, 4096, 1.0)]), naive=False, store=False, dense=False, threads=16)

_scene_cache = {}


def _scene(vra, case):
    key = case["scene_key"]
    if key in _scene_cache:
        return _scene_cache[key]
    if case["kind"] == "lodset":
        s = vra.LodSet(os.path.join(ASSETS, case["mesh"]))
    elif case["kind"] == "sphere":
        s = vra.HostScene.sphere(case["res"])
    else:
        s = vra.HostScene.from_mesh(os.path.join(ASSETS, case["mesh"]), case["res"], dense=case.get("dense", True), svo=True)
    _scene_cache[key] = s
    return s


def scene_digest(vra, case, scene):
    """Content hashes that pin the host voxelizer / octree builder."""
    if case["kind"] == "lodset":
        grids = [vra.host.dense_view(scene.grid(i)) for i in range(4)]
        return {"filled": [int((g[..., 3] > 0).sum()) for g in grids], "crc": [crc(g) for g in grids]}
    d = {}
    if scene.target is not None:
        t = vra.host.dense_view(scene.target)
        d["target_filled"], d["target_crc"] = int((t[..., 3] > 0).sum()), crc(t)
    lo = vra.host.dense_view(scene.low)
    d["low_filled"], d["low_crc"] = int((lo[..., 3] > 0).sum()), crc(lo)
    nodes, leaves = scene.svo_arrays()
    d["svo_nodes"], d["svo_leaves"] = int(len(nodes)), int(len(leaves))
    d["svo_nodes_crc"], d["svo_leaves_crc"] = crc(nodes), crc(leaves)
    if case["kind"] == "mesh":
        d["lod_counts"] = [scene.lod_count(k) for k in range(6)]
    return d


def case_inputs(vra, case):
    """(frames, low desc, target dense desc or None, svo desc or None, scene)"""
    scene = _scene(vra, case)
    if case["kind"] == "lodset":
        target, low, svo = scene.grid(case["cur_lod"]), scene.grid(1), None
    else:
        low = scene.low
        target = scene.target if case["target"] == "dense" else None
        svo = scene.svo if case["target"] == "svo" else None
    hi = (target.sx, target.sy, target.sz) if target is not None else (svo.res,) * 3
    lo = (low.sx, low.sy, low.sz)
    kind, cams = case["cams"]
    frames = []
    for cam in cams:
        if kind == "mouse":
            (mx, my), pos = cam
            view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(mx, my), (0, 0, -1))
        elif kind == "orbit":
            pos, view = vra.orbit_camera(*cam)
        else:
            pos, view = cam, _look_from(vra, cam)
        frames.append(vra.frame_from_camera(view, pos, hi, lo, case["w"], case["h"], naive=case["naive"]))
    return frames, low, target, svo, scene


def render_case(vra, ora, case, backend):
    """dict plane -> (n_frames, h, w) array, and the scene digest."""
    frames, low, target, svo, scene = case_inputs(vra, case)
    if backend == "oracle":
        per = [ora.render_frame(f, low=low, target=target, svo=svo, threads=case.get("threads", 8)) for f in frames]
        planes = {p: np.stack([r[p] for r in per]) for p in per[0]}
    else:
        gpu = vra.GpuScene.upload_dense(target, low) if target is not None else vra.GpuScene.upload_svo(svo, low)
        planes = gpu.render(frames)
        gpu.close()
    return planes, scene_digest(vra, case, scene)
