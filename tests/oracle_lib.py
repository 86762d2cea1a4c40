"""ctypes binding of oracle/libvolly_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this;
the product package (volume-renderer_amd/) never does."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

_abi = load_package()._abi
DenseDesc, SvoDesc, FrameDesc, Outputs = _abi.DenseDesc, _abi.SvoDesc, _abi.FrameDesc, _abi.Outputs

_path = os.path.join(ROOT, "oracle", "libvolly_oracle.so")
if not os.path.exists(_path):
    raise ImportError(f"{_path} is missing: run `make oracle`")
lib = C.CDLL(_path)


class Report(C.Structure):
    _fields_ = [("found", C.c_int32), ("pt_end", C.c_float * 3), ("vox", C.c_uint8 * 8),
                ("hit_index", C.c_uint32), ("steps", C.c_uint32), ("capped", C.c_int32)]


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
lib.vo_raycast_dense.argtypes = [_fp, _fp, C.POINTER(DenseDesc), C.c_int, C.POINTER(Report)]
lib.vo_raycast_dense.restype = None
lib.vo_raycast_svo.argtypes = [_fp, _fp, C.POINTER(SvoDesc), C.POINTER(Report)]
lib.vo_raycast_svo.restype = None
lib.vo_raycast_beam.argtypes = [_fp, _fp, _fp, _fp, C.POINTER(DenseDesc), C.POINTER(DenseDesc), C.POINTER(SvoDesc),
                                C.POINTER(Report), C.POINTER(Report)]
lib.vo_raycast_beam.restype = None
lib.vo_raytrace.argtypes = [C.c_int, _fp, _fp, C.POINTER(DenseDesc), _fp, C.POINTER(Report)]
lib.vo_raytrace.restype = None
lib.vo_render_frame.argtypes = [C.POINTER(FrameDesc), C.POINTER(DenseDesc), C.POINTER(DenseDesc), C.POINTER(SvoDesc),
                                C.POINTER(Outputs), C.c_int]
lib.vo_render_frame.restype = C.c_int
lib.vo_frame_from_camera.argtypes = [_fp, _fp, _ip, _ip, C.c_int32, C.c_int32, C.POINTER(FrameDesc)]
lib.vo_frame_from_camera.restype = None
lib.vo_look_at.argtypes = [_fp, _fp, _fp, _fp]
lib.vo_look_at.restype = None
lib.vo_cam_dir_from_mouse.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _fp]
lib.vo_cam_dir_from_mouse.restype = None
lib.vo_shade.argtypes = [C.POINTER(C.c_uint8)]
lib.vo_shade.restype = C.c_uint32
lib.vo_fill_sphere.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_void_p]
lib.vo_fill_sphere.restype = None
lib.vo_step_cap.argtypes = [C.c_int32, C.c_int32, C.c_int32]
lib.vo_step_cap.restype = C.c_uint32


def _f3(v):
    a = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    return a, a.ctypes.data_as(_fp)


def _ref(x):
    return C.byref(x) if x is not None else None


def raycast_dense(start, direction, grid, solid=False):
    s, sp = _f3(start)
    d, dp = _f3(direction)
    r = Report()
    lib.vo_raycast_dense(sp, dp, C.byref(grid), int(solid), C.byref(r))
    return r


def raycast_svo(start, direction, svo):
    s, sp = _f3(start)
    d, dp = _f3(direction)
    r = Report()
    lib.vo_raycast_svo(sp, dp, C.byref(svo), C.byref(r))
    return r


def raytrace(iteration, start, direction4, grid):
    """raytrace_naiive (main.cpp:170-194): (colour xyzw, primary report)."""
    s, sp = _f3(start)
    d, dp = _f3(direction4)
    assert d.size == 4
    out = np.zeros(4, np.float32)
    rep = Report()
    lib.vo_raytrace(iteration, sp, dp, C.byref(grid), out.ctypes.data_as(_fp), C.byref(rep))
    return out, rep


_DT = {"pt_end": np.float32, "rgba8": np.uint32, "hit_index": np.uint32, "steps_lo": np.uint32, "steps_hi": np.uint32, "depth": np.float32}
PLANES = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")


def render_frame(frame, low=None, target=None, svo=None, threads=8, planes=PLANES):
    """Oracle render of one frame's sub-rectangle: dict plane -> (tile_h, tile_w) array."""
    res = {p: np.empty((frame.tile_h, frame.tile_w, 3) if p == "pt_end" else (frame.tile_h, frame.tile_w), _DT[p]) for p in planes}
    outs = Outputs(*[res[p].ctypes.data if p in res else None for p in PLANES + ("pt_end",)])
    rc = lib.vo_render_frame(C.byref(frame), _ref(low), _ref(target), _ref(svo), C.byref(outs), threads)
    if rc != 0:
        raise RuntimeError(f"vo_render_frame rc={rc}")
    return res


def frame_from_camera(view, cam_pos, size_hi, size_lo, width, height, naive=False):
    v, vp = _f3(view)
    p, pp = _f3(cam_pos)
    hi = (C.c_int32 * 3)(*[int(x) for x in size_hi])
    lo = (C.c_int32 * 3)(*[int(x) for x in size_lo])
    out = FrameDesc()
    lib.vo_frame_from_camera(vp, pp, hi, lo, width, height, C.byref(out))
    if naive:
        out.flags = 1
    return out


def look_at(eye, center, up):
    e, ep = _f3(eye)
    c, cp = _f3(center)
    u, up_ = _f3(up)
    out = np.zeros(16, np.float32)
    lib.vo_look_at(ep, cp, up_, out.ctypes.data_as(_fp))
    return out


def cam_dir_from_mouse(mx, my, win_x=1152, win_y=864):
    out = np.zeros(3, np.float32)
    lib.vo_cam_dir_from_mouse(mx, my, win_x, win_y, out.ctypes.data_as(_fp))
    return out


def shade(vox8):
    v = (C.c_uint8 * 8)(*[int(x) for x in vox8])
    return int(lib.vo_shade(v))


def fill_sphere(res, fill=(255, 255, 255, 255, 147, 147, 147, 0)):
    out = np.zeros((res, res, res, 8), np.uint8)
    f = (C.c_uint8 * 8)(*fill)
    lib.vo_fill_sphere(res, res, res, f, out.ctypes.data)
    return out


def step_cap(sx, sy, sz):
    return int(lib.vo_step_cap(sx, sy, sz))
