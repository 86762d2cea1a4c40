"""The C-ABI libraries load without a GPU and export every symbol the headers
declare; the ctypes mirrors of the PODs match the C compiler's layout."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text.split("static inline")[0]
    names = re.findall(r"\b(vol\w+|svo_gpu_\w+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_desc") and not n.endswith("_node")))


def test_gpu_library_exports_every_declared_symbol(vra):
    names = _declared_functions("volly_gpu.h")
    assert "volume_gpu_render" in names and "svo_gpu_upload" in names and len(names) >= 12
    for n in names:
        assert hasattr(vra.gpu.lib, n), f"libvolly_gpu.so does not export {n}"
    assert set(names) == set(vra.gpu.EXPORTS)
    assert vra.gpu.lib.volly_gpu_abi_version() == 3


def test_the_prebuilt_gpu_library_matches_the_tree(vra):
    """The assembly loops keep state in fixed registers across asm statements; tools/check_kernel_contract.py checks
    that on the compiler's output when `make gpu` builds the library -- and only then.  The library records the hash
    of the sources it was built from: a .so that is older than the tree (or was built from other sources) fails here,
    on the CPU and on the GPU box, instead of rendering with unchecked loops."""
    info = vra.gpu.build_info()
    assert info.get("src_sha16") == vra.gpu.source_sha16(ROOT), (
        f"libvolly_gpu.so was built from other sources than this tree's ({info.get('text')}): run `make gpu`")
    assert info.get("hipcc", "unknown") != "unknown" and "contract" in info["text"]


def test_host_library_exports_every_declared_symbol(vra):
    names = _declared_functions("volly_host.h")
    assert len(names) >= 15
    for n in names:
        assert hasattr(vra.host.lib, n), f"libvolly_host.so does not export {n}"
    assert set(names) == set(vra.host.EXPORTS)


def test_struct_layouts_match_the_c_compiler(vra, tmp_path):
    abi = vra._abi
    structs = {
        "volly_dense_desc": (abi.DenseDesc, ["sx", "sy", "sz", "voxels"]),
        "volly_svo_node": (abi.SvoNode, ["child_offset", "valid_mask", "leaf_mask", "pad"]),
        "volly_svo_desc": (abi.SvoDesc, ["n_nodes", "n_leaves", "depth", "res", "nodes", "leaves"]),
        "volly_frame_desc": (abi.FrameDesc, ["row0_hi", "xdiff_hi", "ydiff_hi", "row0_lo", "xdiff_lo", "ydiff_lo",
                                             "origin_lo", "origin_hi", "scale_lo_to_hi", "width", "height", "x0",
                                             "y0", "tile_w", "tile_h", "flags", "reserved"]),
        "volly_outputs": (abi.Outputs, ["rgba8", "hit_index", "steps_lo", "steps_hi", "depth", "pt_end"]),
        "volly_mesh_desc": (abi.MeshDesc, ["n_verts", "verts", "n_tris", "tris"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "volly_gpu.h"', 'int main(void){']
    for cname, (_, fields) in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, (ct, fields) in structs.items():
        assert C.sizeof(ct) == int(out[cname]), cname
        for f in fields:
            assert getattr(ct, f).offset == int(out[f"{cname}.{f}"]), f"{cname}.{f}"
    assert C.sizeof(abi.SvoNode) == 8 and C.sizeof(abi.FrameDesc) == 176


def test_error_paths_without_a_gpu(vra):
    """Argument checks run before any HIP call: they must report VOLLY_E_ARG here too."""
    lib = vra.gpu.lib
    assert lib.volume_gpu_upload(0, None, None, None) == -1
    h = C.c_void_p()
    assert lib.volume_gpu_upload(0, None, None, C.byref(h)) == -1
    assert b"null grid" in lib.volly_gpu_last_error()
    assert lib.svo_gpu_upload(0, None, None, C.byref(h)) == -1
    assert lib.volume_gpu_render(None, None, 0, None) == -1
    assert lib.volly_gpu_scene_info(None, 0) == -1
    lib.volly_gpu_free(None)


def test_noise_hash_header_matches_python():
    """volly_noise_hash in the header (compiled with gcc) against the test suite's numpy restatement."""
    import numpy as np
    src = r'''
#include <stdio.h>
#include "volly_gpu.h"
int main(void){ for (unsigned i = 0; i < 5; ++i) printf("%u\n", volly_noise_hash(i*7u, i*i+3u, 1000u-i, 3u)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "h.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "h.c"), "-o", os.path.join(d, "h")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(d, "h")], check=True, capture_output=True, text=True).stdout.split()]
    want = []
    M = 0xFFFFFFFF
    for i in range(5):
        x, y, z, seed = i * 7, i * i + 3, 1000 - i, 3
        h = (seed * 0x9E3779B1 + x * 0x85EBCA77) & M
        h ^= h >> 15; h = (h * 0x2C1B3C6D) & M
        h = (h + y * 0xC2B2AE3D) & M
        h ^= h >> 13; h = (h * 0x297A2D39) & M
        h = (h + z * 0x27D4EB2F) & M
        h ^= h >> 16; h = (h * 0x165667B1) & M
        h ^= h >> 15
        want.append(h)
    assert got == want
