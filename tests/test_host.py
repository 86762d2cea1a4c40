"""Host scene pipeline (C++ product code) against independent restatements:
  * triangle/cube test vs the reference's own triangleCube.c (oracle/_ref, built from
    /root/reference where it lies; the prebuilt library travels to the GPU box);
  * the threaded voxelizer vs a literal, reference-order restatement of
    rasterizeVoxelMapFromPoly (PLYfileReader.cpp:218-321) in Python;
  * the octree vs the dense grid it was built from;
  * frame constants vs the oracle's restatement of main.cpp:201-249."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ASSETS, ROOT

F = np.float32
_REF = os.path.join(ROOT, "oracle", "_ref", "libtricube_ref.so")


def _ref_tricube():
    if not os.path.exists(_REF):
        return None
    lib = C.CDLL(_REF)
    lib.tricube_ref.argtypes = [C.POINTER(C.c_float)]
    lib.tricube_ref.restype = C.c_int
    return lib.tricube_ref


def test_tricube_matches_the_reference_source(vra):
    ref = _ref_tricube()
    if ref is None:
        pytest.skip("oracle/_ref/libtricube_ref.so not built (needs /root/reference)")
    rng = np.random.default_rng(11)
    n_in = 0
    tris = []
    for scale, n in ((0.4, 3000), (1.0, 12000), (3.0, 8000), (30.0, 2000)):
        tris.append(rng.uniform(-scale, scale, size=(n, 9)))
    # touching / degenerate cases: vertices on faces, edges, corners, flat triangles
    grid = np.array([-1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5])
    tris.append(rng.choice(grid, size=(15000, 9)))
    flat = rng.choice(grid, size=(2000, 9)); flat[:, 6:9] = flat[:, 0:3]
    tris.append(flat)
    for t in np.concatenate(tris).astype(np.float32):
        buf = (C.c_float * 9)(*t)
        a = ref(buf)
        b = vra.host.lib.volly_host_tricube(buf)
        assert a == b, f"triangle {t.tolist()}: reference {a}, host {b}"
        n_in += (a == 0)
    assert 5000 < n_in < 35000          # both outcomes well represented


# ---- literal restatement of the reference voxelizer ------------------------------------

def _normalize_poly(v):
    """PLYfileReader.cpp:144-171 in float32, one rounding per operation."""
    mn = v.min(axis=0).astype(F); mx = v.max(axis=0).astype(F)
    mul = (F(1) / (mx - mn)).astype(F)
    add = (-mn * mul).astype(F)
    out = (v * mul).astype(F)
    out = (out + add).astype(F)
    return np.minimum(np.maximum(out, F(0)), F(1)).astype(F)


def _cross(x, y):
    return np.array([x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]], F)


def _literal_rasterize(verts, inds, res, cube):
    """Reference order: std::map keyed (x,y,z,LoD), later inserts overwrite."""
    verts = _normalize_poly(verts)                                   # PLYfileReader.cpp:219
    resF = F(res - 1)
    out = {}
    for tri in inds:
        a, b, c = [(verts[i] * resF).astype(F) for i in tri]
        n = _cross((b - a).astype(F), (b - c).astype(F))
        d = F(F(n[0] * n[0]) + F(n[1] * n[1])); d = F(d + F(n[2] * n[2]))
        with np.errstate(divide="ignore", invalid="ignore"):
            n = (n * F(F(1) / np.sqrt(d))).astype(F)
        n = np.minimum(np.maximum(n, F(0)), F(1))
        nu8 = tuple(0 if np.isnan(x) else int(F(x * F(255))) for x in n) + (0,)
        seen, stack = set(), []
        for k in (a, b, c):
            s = (int(k[0]), int(k[1]), int(k[2]))
            stack.append(s); seen.add(s)
        while stack:
            cx, cy, cz = stack.pop()
            off = np.array([cx + 0.5, cy + 0.5, cz + 0.5], F)
            t = np.concatenate([(a - off).astype(F), (b - off).astype(F), (c - off).astype(F)])
            if cube((C.c_float * 9)(*t)) != 0:
                continue
            p, lod = [cx, cy, cz], 0
            while p[0] > 1:                                           # PLYfileReader.cpp:278-286
                r = res // (1 << lod)
                if not (min(p) < 0 or max(p) >= r):
                    out[(p[0], p[1], p[2], lod)] = nu8
                p = [int(q / 2) for q in p]                           # C++ division: -1/2 == 0
                lod += 1
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        nb = (cx + dx, cy + dy, cz + dz)
                        if nb not in seen:
                            seen.add(nb); stack.append(nb)
    return out


def _read_off(path):
    tok = open(path).read().split()
    assert tok[0] == "OFF"
    nv, nf = int(tok[1]), int(tok[2])
    v = np.array(tok[4:4 + 3 * nv], dtype=np.float64).astype(F).reshape(nv, 3)
    f = np.array(tok[4 + 3 * nv:4 + 3 * nv + 4 * nf], dtype=np.int64).reshape(nf, 4)[:, 1:]
    return v, f


def _dense_from_map(mp, res, lod):
    g = np.zeros((res, res, res, 8), np.uint8)
    for (x, y, z, w), n in mp.items():
        if w == lod and x < res and y < res and z < res:
            g[z, y, x] = (255, 255, 255, 255) + n
    return g


@pytest.mark.parametrize("mesh,res", [("icosahedron.off", 32), ("teapot.off", 32), ("dolphins.off", 24)])
def test_voxelizer_matches_literal_restatement(vra, mesh, res):
    cube = _ref_tricube() or vra.host.lib.volly_host_tricube
    verts, inds = _read_off(os.path.join(ASSETS, mesh))
    verts = _normalize_poly(verts)                                   # main.cpp:27
    want = _literal_rasterize(verts, inds, res, cube)
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, mesh), res, dense=True, svo=False, threads=3)
    for lod in range(6):
        assert scene.lod_count(lod) == sum(1 for k in want if k[3] == lod), f"LoD {lod} entry count"
    assert np.array_equal(vra.host.dense_view(scene.target), _dense_from_map(want, res, 0))
    assert np.array_equal(vra.host.dense_view(scene.low), _dense_from_map(want, res // 4, 2))
    assert scene.lod_count(0) > 100
    # quirk Q3: cells with x <= 1 are never stored
    assert not vra.host.dense_view(scene.target)[:, :, 0:2, 3].any()
    # thread count does not change the result
    again = vra.HostScene.from_mesh(os.path.join(ASSETS, mesh), res, dense=True, svo=False, threads=1)
    assert np.array_equal(vra.host.dense_view(scene.target), vra.host.dense_view(again.target))
    assert np.array_equal(vra.host.dense_view(scene.low), vra.host.dense_view(again.low))


def test_ply_reader_and_crlf(vra, tmp_path):
    ico_ply = vra.HostScene.from_mesh(os.path.join(ASSETS, "icosahedron.ply"), 32, svo=False)
    ico_off = vra.HostScene.from_mesh(os.path.join(ASSETS, "icosahedron.off"), 32, svo=False)
    a, b = vra.host.dense_view(ico_ply.target), vra.host.dense_view(ico_off.target)
    assert a[..., 3].sum() > 0
    # same solid, different face order/orientation in the two files: occupancy agrees
    assert np.array_equal(a[..., 3], b[..., 3])
    # CRLF copies (as the reference ships them) parse to the same grids
    for name in ("icosahedron.ply", "icosahedron.off"):
        crlf = tmp_path / name
        crlf.write_bytes(open(os.path.join(ASSETS, name), "rb").read().replace(b"\n", b"\r\n"))
        s = vra.HostScene.from_mesh(str(crlf), 32, svo=False)
        ref = ico_ply if name.endswith(".ply") else ico_off
        assert np.array_equal(vra.host.dense_view(s.target), vra.host.dense_view(ref.target))
    with pytest.raises(RuntimeError):
        vra.HostScene.from_mesh(str(tmp_path / "missing.ply"), 32)


def test_lod_set_matches_reference_defaults(vra):
    """Vol_LOD_Set (main.h:79): 16 / 32 / 128 / 256; arr[1] and arr[0] are LoD 2 and 3 of
    the 128 map, arr[3] comes from a second rasterization after one more normalizePoly."""
    ls = vra.LodSet(os.path.join(ASSETS, "icosahedron.ply"))
    sizes = [ls.grid(i).sx for i in range(4)]
    assert sizes == [16, 32, 128, 256]
    med = vra.host.dense_view(ls.grid(2))
    one = vra.HostScene.from_mesh(os.path.join(ASSETS, "icosahedron.ply"), 128, svo=False)
    assert np.array_equal(med, vra.host.dense_view(one.target))
    assert np.array_equal(vra.host.dense_view(ls.grid(1)), vra.host.dense_view(one.low))
    occ = [int((vra.host.dense_view(ls.grid(i))[..., 3] > 0).sum()) for i in range(4)]
    assert occ[0] < occ[1] < occ[2] < occ[3]


def _walk_octree(nodes, leaves, depth):
    """All (x, y, z) -> payload of the flattened octree, by explicit traversal."""
    out = {}
    stack = [(0, 0, 0, 0, depth)]
    while stack:
        i, x, y, z, level = stack.pop()
        n = nodes[i]
        k = 0
        for octant in range(8):
            if not (int(n["valid_mask"]) >> octant) & 1:
                continue
            h = 1 << (level - 1)
            cx, cy, cz = x + h * (octant & 1), y + h * ((octant >> 1) & 1), z + h * ((octant >> 2) & 1)
            child = int(n["child_offset"]) + k
            k += 1
            if (int(n["leaf_mask"]) >> octant) & 1:
                assert level == 1
                out[(cx, cy, cz)] = bytes(leaves[child])
            else:
                assert level > 1 and child > i
                stack.append((child, cx, cy, cz, level - 1))
    return out


@pytest.mark.parametrize("source", ["map", "dense"])
def test_octree_holds_exactly_the_dense_voxels(vra, source):
    if source == "map":
        scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64)
    else:
        rng = np.random.default_rng(5)
        vox = np.zeros((20, 24, 17, 8), np.uint8)          # not cubic, not a power of two
        m = rng.random((20, 24, 17)) < 0.03
        vox[m] = rng.integers(1, 256, size=(int(m.sum()), 8), dtype=np.uint8)
        scene = vra.HostScene.from_dense(vox)
    svo = scene.svo
    nodes, leaves = scene.svo_arrays()
    dense = vra.host.dense_view(scene.target)
    assert svo.depth == int(np.ceil(np.log2(svo.res))) and svo.res == max(dense.shape[:3])
    got = _walk_octree(nodes, leaves, svo.depth)
    zz, yy, xx = np.nonzero(dense[..., 3])
    assert len(got) == len(zz) == svo.n_leaves
    for x, y, z in zip(xx, yy, zz):
        assert got[(int(x), int(y), int(z))] == dense[z, y, x].tobytes()
    # breadth-first: root first, children always after their parent, pad zero
    assert not nodes["pad"].any()


def test_frame_constants_match_oracle(vra, ora):
    """Host C++ (camera.cpp) vs oracle C restatement of main.cpp:201-249: identical bytes."""
    for (mx, my), pos, hi, lo, wh in [((0, 0), (0, 0, 0), (16,) * 3, (32,) * 3, (400, 300)),
                                      ((300, 500), (0.3, -0.8, 0.2), (128,) * 3, (32,) * 3, (640, 480)),
                                      ((-977, 123), (1.5, 0.5, 0.9), (512, 256, 128), (128, 64, 32), (1920, 1080)),
                                      ((45, 864), (0.5, 0.5, 2.5), (1024,) * 3, (256,) * 3, (3840, 2160))]:
        d_h, d_o = vra.host.cam_dir_from_mouse(mx, my), ora.cam_dir_from_mouse(mx, my)
        assert d_h.tobytes() == d_o.tobytes()
        v_h = vra.look_at((0, 0, 0), d_h, (0, 0, -1))
        v_o = ora.look_at((0, 0, 0), d_o, (0, 0, -1))
        assert v_h.tobytes() == v_o.tobytes()
        f_h = vra.frame_from_camera(v_h, pos, hi, lo, *wh)
        f_o = ora.frame_from_camera(v_o, pos, hi, lo, *wh)
        assert bytes(f_h) == bytes(f_o)
        assert f_h.width == wh[0] and f_h.tile_h == wh[1] and f_h.flags == 0
        assert tuple(f_h.scale_lo_to_hi)[:3] == tuple(np.float32(h) / np.float32(l) for h, l in zip(hi, lo))
    # the reference's start-up state: camDir from mouse (0,0), 90 degree FOV
    assert np.allclose(vra.host.cam_dir_from_mouse(0, 0), np.array([1, 0, 2]) / np.sqrt(5), atol=1e-5)


def test_look_at_is_a_rotation(vra):
    pos, view = vra.orbit_camera(7, 64, 1.2)
    m = view.reshape(4, 4).T[:3, :3].astype(np.float64)
    assert np.allclose(m @ m.T, np.eye(3), atol=1e-5)
    assert np.allclose(np.linalg.norm(pos - 0.5), np.sqrt(1.2 ** 2 + (pos[2] - 0.5) ** 2), atol=1e-5)


def test_scene_cache_round_trip_and_rejection(vra, tmp_path):
    """On-disk scene cache: what comes back is byte for byte what was saved;
    missing, foreign, truncated and corrupted files are refused."""
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64)
    path = tmp_path / "teapot64.volly"
    scene.save(path)
    back = vra.HostScene.load(path)
    assert np.array_equal(vra.host.dense_view(back.target), vra.host.dense_view(scene.target))
    assert np.array_equal(vra.host.dense_view(back.low), vra.host.dense_view(scene.low))
    n0, l0 = scene.svo_arrays(); n1, l1 = back.svo_arrays()
    assert np.array_equal(n0, n1) and np.array_equal(l0, l1)
    assert (back.svo.depth, back.svo.res) == (scene.svo.depth, scene.svo.res)
    assert back.lod_count(0) == -1                       # the voxel map itself is not cached
    # octree-only scene
    svo_only = vra.HostScene.from_mesh(os.path.join(ASSETS, "icosahedron.off"), 32, dense=False, svo=True)
    p2 = tmp_path / "ico.volly"
    svo_only.save(p2)
    b2 = vra.HostScene.load(p2)
    assert b2.target is None and b2.svo.n_leaves == svo_only.svo.n_leaves
    raw = path.read_bytes()
    for name, data in (("trunc", raw[:len(raw) // 2]), ("flip", raw[:200] + bytes([raw[200] ^ 1]) + raw[201:]),
                       ("extra", raw + b"x"), ("foreign", b"OFF\n1 2 3\n")):
        bad = tmp_path / name
        bad.write_bytes(data)
        with pytest.raises(RuntimeError):
            vra.HostScene.load(bad)
    with pytest.raises(RuntimeError):
        vra.HostScene.load(tmp_path / "nope.volly")
    # a header that asks for 4096^3 voxels (512 GiB) and 2^32 - 1 leaves over a few bytes of payload:
    # refused on the file's size, before anything is allocated (the load returns at once)
    import struct, time
    magic = raw[:8]
    hostile = magic + struct.pack("<II3i3iIIiiQ", 1, 3, 4096, 4096, 4096, 4096, 4096, 4096, 1, 0xFFFFFFFF, 12, 4096, 0) + b"\0" * 64
    bad = tmp_path / "hostile"
    bad.write_bytes(hostile)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="size does not match"):
        vra.HostScene.load(bad)
    assert time.time() - t0 < 1.0


def test_host_mesh_view_is_what_the_rasteriser_sees(vra):
    """volly_host_mesh_load: vertices after n normalizePoly passes (PLYfileReader.cpp:141-171:
    into the unit cube), triangle indices untouched; the descriptor is what
    volume_gpu_voxelize takes.  The passes are not idempotent bit for bit, which is why the
    count matters (Vol_LOD_Set's high grid sees three, main.cpp:27 + PLYfileReader.cpp:219)."""
    import ctypes as C
    path = os.path.join(ASSETS, "galleon.ply")
    views = []
    for n in (0, 1, 2, 3):
        m = vra.HostMesh(path, n)
        d = m.desc
        assert d.n_verts == 2510 and d.n_tris == 4698
        v = np.frombuffer((C.c_float * (4 * d.n_verts)).from_address(d.verts), np.float32).reshape(-1, 4).copy()
        t = np.frombuffer((C.c_int32 * (3 * d.n_tris)).from_address(d.tris), np.int32).reshape(-1, 3).copy()
        assert t.min() >= 0 and t.max() < d.n_verts
        views.append((v, t))
        m.close()
    raw, once, twice, thrice = (v for v, _ in views)
    assert raw[:, :3].max() > 1.5                                   # file coordinates
    for v in (once, twice, thrice):
        assert v[:, :3].min() >= 0.0 and v[:, :3].max() <= 1.0
        assert np.isclose(v[:, :3].max(), 1.0) and np.isclose(v[:, :3].min(), 0.0, atol=1e-6)
    assert np.allclose(once, twice, atol=1e-5) and np.allclose(twice, thrice, atol=1e-5)
    assert all(np.array_equal(views[0][1], t) for _, t in views)
    with pytest.raises(RuntimeError):
        vra.HostMesh(os.path.join(ASSETS, "no_such_mesh.ply"), 2)
