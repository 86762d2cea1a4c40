"""ctypes mirrors of the PODs in include/volly_gpu.h (kept in lock-step by
tests/test_abi.py, which checks sizes and offsets against the C compiler)."""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")

VOLLY_OK = 0
VOLLY_E_ARG = -1
VOLLY_E_HIP = -2
VOLLY_E_NOMEM = -3
VOLLY_E_UNSUPPORTED = -4
MISS_INDEX = 0xFFFFFFFF
FRAME_BEAM = 0
FRAME_NAIVE = 1
SCENE_DENSE = 1
SCENE_SVO = 2
VOX_DENSE = 1
VOX_SVO = 2


class DenseDesc(C.Structure):
    _fields_ = [("sx", C.c_int32), ("sy", C.c_int32), ("sz", C.c_int32), ("voxels", C.c_void_p)]


class SvoNode(C.Structure):
    _fields_ = [("child_offset", C.c_uint32), ("valid_mask", C.c_uint8), ("leaf_mask", C.c_uint8),
                ("pad", C.c_uint16)]


class SvoDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("depth", C.c_int32), ("res", C.c_int32),
                ("nodes", C.c_void_p), ("leaves", C.c_void_p)]


class MeshDesc(C.Structure):
    _fields_ = [("n_verts", C.c_int64), ("verts", C.c_void_p), ("n_tris", C.c_int64), ("tris", C.c_void_p)]


class FrameDesc(C.Structure):
    _fields_ = [("row0_hi", C.c_float * 4), ("xdiff_hi", C.c_float * 4), ("ydiff_hi", C.c_float * 4),
                ("row0_lo", C.c_float * 4), ("xdiff_lo", C.c_float * 4), ("ydiff_lo", C.c_float * 4),
                ("origin_lo", C.c_float * 4), ("origin_hi", C.c_float * 4), ("scale_lo_to_hi", C.c_float * 4),
                ("width", C.c_int32), ("height", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]

    def copy(self):
        out = FrameDesc()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(FrameDesc))
        return out

    def band(self, y0, rows):
        """The same frame restricted to rows [y0, y0+rows) (multi-GPU row bands)."""
        out = self.copy()
        out.y0, out.tile_h = int(y0), int(rows)
        return out


class Outputs(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("hit_index", C.c_void_p), ("steps_lo", C.c_void_p),
                ("steps_hi", C.c_void_p), ("depth", C.c_void_p), ("pt_end", C.c_void_p)]


PLANES = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")
ALL_PLANES = PLANES + ("pt_end",)          # pt_end: three floats per ray, only on request
