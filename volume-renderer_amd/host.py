"""ctypes binding of lib/libvolly_host.so (include/volly_host.h): the C++ host
scene pipeline -- mesh loading, voxelization, octree build, frame constants."""
import ctypes as C
import os

import numpy as np

from ._abi import (LIB_DIR, DenseDesc, SvoDesc, SvoNode, FrameDesc, MeshDesc, SCENE_DENSE, SCENE_SVO, VOLLY_OK,
                   VOLLY_E_UNSUPPORTED)

_path = os.path.join(LIB_DIR, "libvolly_host.so")
if not os.path.exists(_path):
    raise ImportError(f"{_path} is missing: run `make host` (or __graft_entry__.build()) first")
lib = C.CDLL(_path)

_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_sig = {
    "volly_host_scene_from_mesh": (_vp, [C.c_char_p, _i32, _i32, _i32]),
    "volly_host_scene_sphere": (_vp, [_i32, C.POINTER(C.c_uint8), _i32]),
    "volly_host_scene_from_dense": (_vp, [C.POINTER(DenseDesc), C.POINTER(DenseDesc), _i32]),
    "volly_host_scene_target": (C.c_int, [_vp, C.POINTER(DenseDesc)]),
    "volly_host_scene_low": (C.c_int, [_vp, C.POINTER(DenseDesc)]),
    "volly_host_scene_svo": (C.c_int, [_vp, C.POINTER(SvoDesc)]),
    "volly_host_scene_lod_count": (_i64, [_vp, _i32]),
    "volly_host_scene_free": (None, [_vp]),
    "volly_host_scene_save": (C.c_int, [_vp, C.c_char_p]),
    "volly_host_scene_load": (_vp, [C.c_char_p]),
    "volly_host_mesh_load": (_vp, [C.c_char_p, _i32]),
    "volly_host_mesh_view": (C.c_int, [_vp, C.POINTER(MeshDesc)]),
    "volly_host_mesh_free": (None, [_vp]),
    "volly_host_lod_set_load": (_vp, [C.c_char_p, _i32, _i32, _i32, _i32]),
    "volly_host_lod_set_grid": (C.c_int, [_vp, _i32, C.POINTER(DenseDesc)]),
    "volly_host_lod_set_free": (None, [_vp]),
    "volly_host_frame_from_camera": (None, [_fp, _fp, _ip, _ip, _i32, _i32, C.POINTER(FrameDesc)]),
    "volly_host_look_at": (None, [_fp, _fp, _fp, _fp]),
    "volly_host_cam_dir_from_mouse": (None, [_i32, _i32, _i32, _i32, _fp]),
    "volly_host_orbit_camera": (None, [_i32, _i32, C.c_float, _fp, _fp]),
    "volly_host_tricube": (C.c_int, [_fp]),
    "volly_host_last_error": (C.c_char_p, []),
}
for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args
EXPORTS = tuple(_sig)


def _err():
    return (lib.volly_host_last_error() or b"").decode()


def _f(arr, n):
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    assert a.size == n
    return a, a.ctypes.data_as(_fp)


def dense_view(desc):
    """numpy view (sz, sy, sx, 8) uint8 over the voxels of a dense descriptor (not owning)."""
    n = desc.sx * desc.sy * desc.sz * 8
    buf = (C.c_uint8 * n).from_address(desc.voxels)
    return np.frombuffer(buf, dtype=np.uint8).reshape(desc.sz, desc.sy, desc.sx, 8)


def dense_desc(voxels):
    """DenseDesc over a (sz, sy, sx, 8) uint8 array; the array must outlive the descriptor."""
    v = np.ascontiguousarray(voxels, dtype=np.uint8)
    assert v.ndim == 4 and v.shape[3] == 8
    d = DenseDesc(v.shape[2], v.shape[1], v.shape[0], v.ctypes.data)
    d._keep = v
    return d


class HostScene:
    """One target resolution worth of Vol_LOD_Set (main.cpp:23-50): target grid
    and/or octree plus the low beam grid."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("host scene: " + _err())
        self._h = handle

    @classmethod
    def from_mesh(cls, path, res, dense=True, svo=True, threads=0):
        flags = (SCENE_DENSE if dense else 0) | (SCENE_SVO if svo else 0)
        return cls(lib.volly_host_scene_from_mesh(os.fsencode(path), res, flags, threads))

    @classmethod
    def sphere(cls, res, fill=(255, 255, 255, 255, 147, 147, 147, 0), svo=True):
        f = (C.c_uint8 * 8)(*fill)
        return cls(lib.volly_host_scene_sphere(res, f, SCENE_DENSE | (SCENE_SVO if svo else 0)))

    @classmethod
    def from_dense(cls, voxels, low=None, svo=True):
        t = dense_desc(voxels)
        l = dense_desc(low) if low is not None else None
        return cls(lib.volly_host_scene_from_dense(C.byref(t), C.byref(l) if l is not None else None,
                                                   SCENE_DENSE | (SCENE_SVO if svo else 0)))

    def save(self, path):
        """Write the scene cache (low grid, dense target and/or octree) to `path`."""
        if lib.volly_host_scene_save(self._h, os.fsencode(path)) != VOLLY_OK:
            raise RuntimeError("scene cache: " + _err())

    @classmethod
    def load(cls, path):
        return cls(lib.volly_host_scene_load(os.fsencode(path)))

    def _get(self, fn, desc):
        rc = fn(self._h, C.byref(desc))
        if rc == VOLLY_E_UNSUPPORTED:
            return None
        if rc != VOLLY_OK:
            raise RuntimeError("host scene query failed")
        return desc

    @property
    def target(self):
        return self._get(lib.volly_host_scene_target, DenseDesc())

    @property
    def low(self):
        return self._get(lib.volly_host_scene_low, DenseDesc())

    @property
    def svo(self):
        return self._get(lib.volly_host_scene_svo, SvoDesc())

    def svo_arrays(self):
        """(nodes structured array, leaves (n,8) uint8) numpy views of the octree."""
        s = self.svo
        dt = np.dtype([("child_offset", "<u4"), ("valid_mask", "u1"), ("leaf_mask", "u1"), ("pad", "<u2")])
        nodes = np.frombuffer((C.c_uint8 * (s.n_nodes * 8)).from_address(s.nodes), dtype=dt)
        if s.n_leaves:
            leaves = np.frombuffer((C.c_uint8 * (s.n_leaves * 8)).from_address(s.leaves), dtype=np.uint8).reshape(-1, 8)
        else:
            leaves = np.zeros((0, 8), np.uint8)
        return nodes, leaves

    def lod_count(self, lod):
        return int(lib.volly_host_scene_lod_count(self._h, lod))

    def close(self):
        if self._h:
            lib.volly_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostMesh:
    """A mesh read and normalised the way rasterizeVoxelMapFromPoly sees it (volly_host_mesh_load)."""

    def __init__(self, path, n_normalize=2):
        self._h = lib.volly_host_mesh_load(os.fsencode(path), n_normalize)
        if not self._h:
            raise RuntimeError("host mesh: " + _err())
        self.desc = MeshDesc()
        if lib.volly_host_mesh_view(self._h, C.byref(self.desc)) != VOLLY_OK:
            raise RuntimeError("host mesh: descriptor")

    def close(self):
        if self._h:
            lib.volly_host_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LodSet:
    """The reference's four-grid Vol_LOD_Set (main.h:74-87), defaults 16/32/128/256."""

    def __init__(self, path, sLowR=16, lowR=32, medR=128, highR=256):
        self._h = lib.volly_host_lod_set_load(os.fsencode(path), sLowR, lowR, medR, highR)
        if not self._h:
            raise RuntimeError("lod set: " + _err())

    def grid(self, which):
        d = DenseDesc()
        rc = lib.volly_host_lod_set_grid(self._h, which, C.byref(d))
        if rc != VOLLY_OK:
            raise RuntimeError(f"lod set grid {which}: rc={rc}")
        return d

    def close(self):
        if self._h:
            lib.volly_host_lod_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def look_at(eye, center, up):
    e, ep = _f(eye, 3)
    c, cp = _f(center, 3)
    u, up_ = _f(up, 3)
    out = np.zeros(16, np.float32)
    lib.volly_host_look_at(ep, cp, up_, out.ctypes.data_as(_fp))
    return out


def cam_dir_from_mouse(mx, my, win_x=1152, win_y=864):
    out = np.zeros(3, np.float32)
    lib.volly_host_cam_dir_from_mouse(mx, my, win_x, win_y, out.ctypes.data_as(_fp))
    return out


def orbit_camera(k, n, radius=1.2):
    pos = np.zeros(3, np.float32)
    view = np.zeros(16, np.float32)
    lib.volly_host_orbit_camera(k, n, radius, pos.ctypes.data_as(_fp), view.ctypes.data_as(_fp))
    return pos, view


def frame_from_camera(view, cam_pos, size_hi, size_lo, width, height, naive=False, raytrace=False):
    """main.cpp:201-249 -> FrameDesc."""
    v, vp = _f(view, 16)
    p, pp = _f(cam_pos, 3)
    hi = (C.c_int32 * 3)(*[int(x) for x in size_hi])
    lo = (C.c_int32 * 3)(*[int(x) for x in size_lo])
    out = FrameDesc()
    lib.volly_host_frame_from_camera(vp, pp, hi, lo, width, height, C.byref(out))
    if naive:
        out.flags = 1
    if raytrace:
        out.flags = 2
    return out


def orbit_frames(ks, n, size_hi, size_lo, width, height, radius=1.2, naive=False):
    """FrameDesc array for orbit views ks (of n)."""
    ks = list(ks)
    arr = (FrameDesc * len(ks))()
    for i, k in enumerate(ks):
        pos, view = orbit_camera(k, n, radius)
        arr[i] = frame_from_camera(view, pos, size_hi, size_lo, width, height, naive)
    return arr


def tricube(tri9):
    t, tp = _f(tri9, 9)
    return int(lib.volly_host_tricube(tp))
