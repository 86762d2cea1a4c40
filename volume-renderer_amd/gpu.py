"""ctypes binding of lib/libvolly_gpu.so (include/volly_gpu.h): the HIP path.

There is no CPU fallback: if the library is missing the import fails, and every
entry point raises VollyError on a non-zero return code."""
import ctypes as C
import os

import numpy as np

from ._abi import (LIB_DIR, DenseDesc, SvoDesc, FrameDesc, MeshDesc, Outputs, PLANES, ALL_PLANES, VOLLY_OK, VOX_DENSE, VOX_SVO)

try:
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
    # process do not share the GPU ("No HIP GPUs are available" for the second one).
    # Loading torch first makes libvolly_gpu.so bind to the runtime torch already loaded
    # (same SONAME).  Without torch the library uses the system ROCm runtime.
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

# VOLLY_GPU_LIB: another build of the same library (kernel A/B runs, profiles/ab_bench.py)
_path = os.environ.get("VOLLY_GPU_LIB") or os.path.join(LIB_DIR, "libvolly_gpu.so")
if not os.path.exists(_path):
    raise ImportError(f"{_path} is missing: the HIP extension must be built (make gpu / __graft_entry__.build()); "
                      "there is no CPU fallback")
lib = C.CDLL(_path)

_vp = C.c_void_p
_sig = {
    "volume_gpu_upload": (C.c_int, [C.c_int, C.POINTER(DenseDesc), C.POINTER(DenseDesc), C.POINTER(_vp)]),
    "svo_gpu_upload": (C.c_int, [C.c_int, C.POINTER(SvoDesc), C.POINTER(DenseDesc), C.POINTER(_vp)]),
    "volume_gpu_generate_noise": (C.c_int, [C.c_int, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "volume_gpu_voxelize": (C.c_int, [C.c_int, C.POINTER(MeshDesc), C.c_int32, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "volume_gpu_voxelize_lod_set": (C.c_int, [C.c_int, C.POINTER(MeshDesc), C.POINTER(MeshDesc), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.POINTER(_vp)]),
    "volly_gpu_tricube": (C.c_int, [C.c_int, _vp, C.c_int64, _vp]),
    "volume_gpu_render": (C.c_int, [_vp, C.POINTER(FrameDesc), C.c_int, C.POINTER(Outputs)]),
    "volume_gpu_render_device": (C.c_int, [_vp, C.POINTER(FrameDesc), C.c_int, C.POINTER(Outputs), _vp]),
    "volume_gpu_render_device_noalloc": (C.c_int, [_vp, C.POINTER(FrameDesc), C.c_int, C.POINTER(Outputs), _vp]),
    "volly_gpu_host_alloc": (_vp, [C.c_size_t]),
    "volly_gpu_host_free": (None, [_vp]),
    "volly_gpu_set_profiling": (C.c_int, [_vp, C.c_int]),
    "volly_gpu_kernel_time": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "volly_gpu_read_voxels": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "volly_gpu_read_low_voxels": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "volly_gpu_read_svo": (C.c_int, [_vp, _vp, _vp]),
    "volly_gpu_scene_info": (C.c_int64, [_vp, C.c_int]),
    "volly_gpu_sky_rect": (C.c_int, [_vp, C.POINTER(FrameDesc), C.POINTER(C.c_int32)]),
    "volly_gpu_sky_rect_for": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(FrameDesc), C.POINTER(C.c_int32)]),
    "volly_gpu_sky_spans_for": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(FrameDesc), C.c_int32, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_float)]),
    "volly_gpu_free": (None, [_vp]),
    "volly_gpu_last_error": (C.c_char_p, []),
    "volly_gpu_abi_version": (C.c_int, []),
    "volly_gpu_build_info": (C.c_char_p, []),
}
for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args
EXPORTS = tuple(_sig)


GPU_SOURCES = ("volume-renderer_amd/csrc/volly_gpu.hip", "volume-renderer_amd/csrc/volly_voxelize.hip", "volume-renderer_amd/csrc/volly_kernels.h",
               "volume-renderer_amd/csrc/volly_internal.h", "include/volly_gpu.h")      # Makefile: GPU_HASHED, same order


def build_info():
    """volly_gpu_build_info as a dict: src_sha16, hipcc, ..."""
    text = (lib.volly_gpu_build_info() or b"").decode()
    out = {"text": text}
    for part in text.split(";"):
        if "=" in part:
            k, v = part.strip().split("=", 1)
            out[k] = v
    return out


def source_sha16(root):
    """The hash `make gpu` records in the library, recomputed from the tree under `root`."""
    import hashlib
    h = hashlib.sha256()
    for rel in GPU_SOURCES:
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()[:16]


class VollyError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = (lib.volly_gpu_last_error() or b"").decode()
        super().__init__(f"{where}: error {code}: {msg}")


def _check(rc, where):
    if rc != VOLLY_OK:
        raise VollyError(rc, where)


def _frames_array(frames):
    if isinstance(frames, FrameDesc):
        arr = (FrameDesc * 1)()
        arr[0] = frames
        return arr, 1
    if isinstance(frames, C.Array):
        return frames, len(frames)
    frames = list(frames)
    arr = (FrameDesc * len(frames))()
    for i, f in enumerate(frames):
        arr[i] = f
    return arr, len(frames)


def sky_spans_for(size_hi, size_lo, frame, block_h=8):
    """volly_gpu_sky_spans_for: per tile row of `frame` the (first, last) column that may hold a ray entering a grid
    (numpy int32 [rows, 2]; first > last: none), and the hull edges used (numpy float32 [5, 3]).  Host arithmetic, no GPU."""
    import numpy as np
    hi = (C.c_int32 * 3)(*[int(v) for v in size_hi])
    lo = (C.c_int32 * 3)(*[int(v) for v in size_lo]) if size_lo is not None else None
    rows = (frame.tile_h + block_h - 1) // block_h
    spans = np.zeros((rows, 2), np.int32)
    hull = np.zeros((5, 3), np.float32)
    _check(lib.volly_gpu_sky_spans_for(hi, lo, C.byref(frame), block_h, spans.ctypes.data_as(C.POINTER(C.c_int32)),
                                       hull.ctypes.data_as(C.POINTER(C.c_float))), "volly_gpu_sky_spans_for")
    return spans, hull


def sky_rect_for(size_hi, size_lo, frame):
    """volly_gpu_sky_rect_for: the sky rectangle of `frame` for grids of these sizes (host arithmetic, no GPU)."""
    hi = (C.c_int32 * 3)(*size_hi)
    lo = (C.c_int32 * 3)(*size_lo) if size_lo is not None else None
    r = (C.c_int32 * 4)()
    _check(lib.volly_gpu_sky_rect_for(hi, lo, C.byref(frame), r), "volly_gpu_sky_rect_for")
    return tuple(r)


def tricube(tri9, device=0):
    """volly_gpu_tricube: (n, 9) float32 triangles -> (n,) int32, 0 = touches the unit cube, 1 = outside."""
    t = np.ascontiguousarray(tri9, dtype=np.float32).reshape(-1, 9)
    out = np.empty(t.shape[0], np.int32)
    _check(lib.volly_gpu_tricube(device, t.ctypes.data, t.shape[0], out.ctypes.data), "volly_gpu_tricube")
    return out


class PinnedArray:
    """numpy view over page-locked host memory (volly_gpu_host_alloc); keep the object alive
    while `.array` is in use."""

    def __init__(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib.volly_gpu_host_alloc(n)
        if not self._p:
            raise VollyError(-3, "volly_gpu_host_alloc")
        self.array = np.frombuffer((C.c_uint8 * n).from_address(self._p), dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib.volly_gpu_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DT = {"pt_end": np.float32, "rgba8": np.uint32, "hit_index": np.uint32, "steps_lo": np.uint32, "steps_hi": np.uint32, "depth": np.float32}


class GpuScene:
    """A scene resident in HBM (volume_gpu_upload / svo_gpu_upload)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def upload_dense(cls, target, low=None, device=0):
        h = _vp()
        _check(lib.volume_gpu_upload(device, C.byref(target), C.byref(low) if low is not None else None, C.byref(h)),
               "volume_gpu_upload")
        return cls(h)

    @classmethod
    def upload_svo(cls, svo, low=None, device=0):
        h = _vp()
        _check(lib.svo_gpu_upload(device, C.byref(svo), C.byref(low) if low is not None else None, C.byref(h)),
               "svo_gpu_upload")
        return cls(h)

    @classmethod
    def noise(cls, res, seed=1, threshold=1 << 24, device=0):
        h = _vp()
        _check(lib.volume_gpu_generate_noise(device, res, seed, threshold, C.byref(h)), "volume_gpu_generate_noise")
        return cls(h)

    @classmethod
    def voxelize(cls, mesh, res, dense=False, svo=True, device=0):
        """volume_gpu_voxelize: `mesh` is a MeshDesc (host.HostMesh(...).desc).  Returns the scene;
        its .vox_stats holds the call's statistics."""
        h = _vp()
        stats = (C.c_int64 * 8)()
        flags = (VOX_DENSE if dense else 0) | (VOX_SVO if svo else 0)
        _check(lib.volume_gpu_voxelize(device, C.byref(mesh), res, flags, C.byref(h), stats), "volume_gpu_voxelize")
        out = cls(h)
        out.vox_stats = dict(zip(("triangles", "tests", "pairs", "levels", "cells", "microseconds", "scratch_bytes"), list(stats)))
        return out

    @classmethod
    def voxelize_lod_set(cls, mesh_med, mesh_high, sLowR=16, lowR=32, medR=128, highR=256, device=0):
        """volume_gpu_voxelize_lod_set: the four scenes of a Vol_LOD_Set (None for a missing high)."""
        hs = (_vp * 4)()
        _check(lib.volume_gpu_voxelize_lod_set(device, C.byref(mesh_med), C.byref(mesh_high) if mesh_high is not None else None,
                                               sLowR, lowR, medR, highR, hs), "volume_gpu_voxelize_lod_set")
        return [cls(_vp(h)) if h else None for h in hs]

    def info(self, what):
        return int(lib.volly_gpu_scene_info(self._h, what))

    def sky_rect(self, frame):
        """volly_gpu_sky_rect: (x_lo, x_hi, y_lo, y_hi), inclusive, relative to the frame's sub-rectangle."""
        r = (C.c_int32 * 4)()
        _check(lib.volly_gpu_sky_rect(self._h, C.byref(frame), r), "volly_gpu_sky_rect")
        return tuple(r)

    def dda_loops(self):
        """Which traversal loops this scene's frames run (volly_gpu_scene_info 14 / 15)."""
        names = {0: "none", 1: "asm-regions", 2: "asm-bricks", 3: "checked"}
        target = "octree" if self.info(3) else names[self.info(14)]
        return f"target {target}, low {names[self.info(15)]}"

    @property
    def size(self):
        return (self.info(0), self.info(1), self.info(2))

    @property
    def low_size(self):
        n = self.info(5)
        return (n, n, n)

    def render(self, frames, planes=PLANES):
        """Synchronous render to host memory: dict plane -> (n_frames, tile_h, tile_w) array."""
        arr, n = _frames_array(frames)
        th, tw = arr[0].tile_h, arr[0].tile_w
        res = {p: np.empty((n, th, tw, 3) if p == "pt_end" else (n, th, tw), _DT[p]) for p in planes}
        outs = Outputs(*[res[p].ctypes.data if p in res else None for p in ALL_PLANES])
        _check(lib.volume_gpu_render(self._h, arr, n, C.byref(outs)), "volume_gpu_render")
        return res

    def render_into(self, frames, arrays):
        """Synchronous render into caller-owned host arrays (dict plane -> C-contiguous array of
        n_frames * tile_h * tile_w 4-byte elements), e.g. PinnedArray(...).array."""
        arr, n = _frames_array(frames)
        for p, a in arrays.items():
            assert a.flags["C_CONTIGUOUS"] and a.size == n * arr[0].tile_h * arr[0].tile_w * (3 if p == "pt_end" else 1) and a.itemsize == 4, p
        outs = Outputs(*[arrays[p].ctypes.data if p in arrays else None for p in ALL_PLANES])
        _check(lib.volume_gpu_render(self._h, arr, n, C.byref(outs)), "volume_gpu_render")

    def render_device(self, frames, ptrs, stream=None):
        """Asynchronous render into device memory.  ptrs: dict plane -> device pointer (int)."""
        arr, n = _frames_array(frames)
        outs = Outputs(*[ptrs.get(p) for p in ALL_PLANES])
        _check(lib.volume_gpu_render_device(self._h, arr, n, C.byref(outs), stream), "volume_gpu_render_device")

    def render_device_noalloc(self, frames, ptrs, stream=None):
        """volume_gpu_render_device_noalloc: the same, scratch from the library's static device array."""
        arr, n = _frames_array(frames)
        outs = Outputs(*[ptrs.get(p) for p in ALL_PLANES])
        _check(lib.volume_gpu_render_device_noalloc(self._h, arr, n, C.byref(outs), stream), "volume_gpu_render_device_noalloc")

    def set_profiling(self, on=True):
        _check(lib.volly_gpu_set_profiling(self._h, int(on)), "volly_gpu_set_profiling")

    def kernel_time(self):
        ms, n = C.c_double(), C.c_int()
        _check(lib.volly_gpu_kernel_time(self._h, C.byref(ms), C.byref(n)), "volly_gpu_kernel_time")
        return ms.value, n.value

    def read_voxels(self, first, count):
        out = np.empty((count, 8), np.uint8)
        _check(lib.volly_gpu_read_voxels(self._h, first, count, out.ctypes.data), "volly_gpu_read_voxels")
        return out

    def read_low_voxels(self):
        n = self.info(5) ** 3
        out = np.empty((n, 8), np.uint8)
        _check(lib.volly_gpu_read_low_voxels(self._h, 0, n, out.ctypes.data), "volly_gpu_read_low_voxels")
        return out

    def read_svo(self):
        """(nodes structured array, leaves (n, 8) uint8) copied back from the device."""
        dt = np.dtype([("child_offset", "<u4"), ("valid_mask", "u1"), ("leaf_mask", "u1"), ("pad", "<u2")])
        nodes = np.empty(self.info(10), dt)
        leaves = np.empty((self.info(11), 8), np.uint8)
        _check(lib.volly_gpu_read_svo(self._h, nodes.ctypes.data, leaves.ctypes.data), "volly_gpu_read_svo")
        return nodes, leaves

    def close(self):
        if self._h:
            lib.volly_gpu_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
