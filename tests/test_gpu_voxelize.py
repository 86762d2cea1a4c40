"""volume_gpu_voxelize: scene preparation on the device against the host
pipeline, which test_host.py pins to a literal restatement of the reference's
rasterizeVoxelMapFromPoly and to the reference's own triangleCube.c.  Everything is compared
byte for byte: target grid, low beam grid, octree nodes and leaves, and a rendered frame."""
import os

import numpy as np
import pytest

from conftest import ASSETS, assert_planes_equal

pytestmark = pytest.mark.gpu


def _host_and_gpu(vra, mesh, res, dense, svo=True):
    path = os.path.join(ASSETS, mesh)
    host = vra.HostScene.from_mesh(path, res, dense=dense, svo=svo)
    hm = vra.HostMesh(path, 2)
    gpu = vra.GpuScene.voxelize(hm.desc, res, dense=dense, svo=svo)
    return host, hm, gpu


@pytest.mark.parametrize("mesh,res", [("icosahedron.ply", 32), ("icosahedron.off", 64), ("teapot.off", 128),
                                      ("galleon.ply", 128), ("stratocaster.ply", 256), ("dragon.off", 256)])
def test_grids_and_octree_equal_the_host_pipeline(vra, mesh, res):
    host, hm, gpu = _host_and_gpu(vra, mesh, res, dense=True)
    st = gpu.vox_stats
    assert st["triangles"] > 0 and st["tests"] >= st["pairs"] > 0 and st["levels"] > 0
    # target grid (volumeFromVoxelMap LoD 0)
    want = vra.host.dense_view(host.target).reshape(-1, 8)
    got = gpu.read_voxels(0, res ** 3)
    assert np.array_equal(got, want), f"{int((got != want).any(axis=1).sum())} target voxels differ"
    assert st["cells"] >= int((want[:, 3] > 0).sum())
    # low beam grid (LoD 2 at res/4, with the reference's x > 1 and -1/2 == 0 quirks)
    want_lo = vra.host.dense_view(host.low).reshape(-1, 8)
    got_lo = gpu.read_low_voxels()
    assert np.array_equal(got_lo, want_lo), f"{int((got_lo != want_lo).any(axis=1).sum())} low voxels differ"
    assert int((want_lo[:, 3] > 0).sum()) > 0
    # octree
    nodes_h, leaves_h = host.svo_arrays()
    nodes_g, leaves_g = gpu.read_svo()
    assert gpu.info(12) == host.svo.depth
    assert nodes_g.shape == nodes_h.shape and leaves_g.shape == leaves_h.shape
    assert np.array_equal(nodes_g.view(np.uint64), nodes_h.view(np.uint64))
    assert np.array_equal(leaves_g, leaves_h)
    gpu.close()
    hm.close()
    host.close()


def test_octree_only_scene_renders_like_the_uploaded_one(vra):
    res = 256
    host, hm, gpu = _host_and_gpu(vra, "dragon.off", res, dense=False)
    assert gpu.info(13) == 0 and gpu.info(3) == 1
    up = vra.GpuScene.upload_svo(host.svo, host.low)
    frames = vra.orbit_frames([100, 1500], 4096, (res,) * 3, (res // 4,) * 3, 320, 200)
    a, b = gpu.render(frames), up.render(frames)
    for p in a:
        assert np.array_equal(a[p].view(np.uint32), b[p].view(np.uint32)), p
    assert int((a["hit_index"] != 0xFFFFFFFF).sum()) > 1000
    for s in (gpu, up, hm, host):
        s.close()


def test_large_octree_matches_the_host(vra):
    """BASELINE config 2's scene (dragon 512^3) built on the device."""
    host, hm, gpu = _host_and_gpu(vra, "dragon.off", 512, dense=False)
    nodes_h, leaves_h = host.svo_arrays()
    nodes_g, leaves_g = gpu.read_svo()
    assert np.array_equal(nodes_g.view(np.uint64), nodes_h.view(np.uint64))
    assert np.array_equal(leaves_g, leaves_h)
    assert np.array_equal(gpu.read_low_voxels(), vra.host.dense_view(host.low).reshape(-1, 8))
    for s in (gpu, hm, host):
        s.close()


def test_argument_errors(vra):
    hm = vra.HostMesh(os.path.join(ASSETS, "icosahedron.ply"), 2)
    with pytest.raises(vra.VollyError):
        vra.GpuScene.voxelize(hm.desc, 30)                 # not a multiple of 4
    with pytest.raises(vra.VollyError):
        vra.GpuScene.voxelize(hm.desc, 64, dense=False, svo=False)
    with pytest.raises(vra.VollyError):
        vra.GpuScene.voxelize(hm.desc, 8192)
    hm.close()


def _write_off(path, verts, tris):
    with open(path, "w") as f:
        f.write(f"OFF\n{len(verts)} {len(tris)} 0\n")
        for v in verts:
            f.write("%r %r %r\n" % tuple(float(c) for c in v))
        for t in tris:
            f.write("3 %d %d %d\n" % tuple(t))


def _compare_everything(vra, path, res):
    host = vra.HostScene.from_mesh(path, res, dense=True, svo=True)
    hm = vra.HostMesh(path, 2)
    gpu = vra.GpuScene.voxelize(hm.desc, res, dense=True, svo=True)
    want = vra.host.dense_view(host.target).reshape(-1, 8)
    assert np.array_equal(gpu.read_voxels(0, res ** 3), want)
    assert np.array_equal(gpu.read_low_voxels(), vra.host.dense_view(host.low).reshape(-1, 8))
    nh, lh = host.svo_arrays()
    ng, lg = gpu.read_svo()
    assert np.array_equal(ng.view(np.uint64), nh.view(np.uint64)) and np.array_equal(lg, lh)
    filled = int((want[:, 3] > 0).sum())
    for s in (gpu, hm, host):
        s.close()
    return filled


def test_ragged_meshes_equal_the_host(vra, tmp_path):
    """Degenerate faces (repeated vertex: NaN normal -> 0 bytes), an index outside the vertex
    list (skipped), slivers, a triangle spanning the whole grid, coincident triangles (the later
    one owns the cell), random soup."""
    rng = np.random.default_rng(5)
    verts = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1), (0.3, 0.3, 0.3), (0.3001, 0.3, 0.3), (0.9, 0.1, 0.5)]
    tris = [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3),      # tetrahedron
            (4, 4, 5),                                       # degenerate: two equal vertices
            (0, 1, 99),                                      # vertex index out of range
            (5, 6, 7),                                       # sliver
            (0, 4, 7), (7, 4, 0),                            # same triangle twice, opposite winding: the later normal wins
            (2, 3, 4)]
    p = str(tmp_path / "ragged.off")
    _write_off(p, verts, tris)
    assert _compare_everything(vra, p, 64) > 500
    soup_v = rng.random((300, 3))
    soup_t = rng.integers(0, 300, (400, 3))
    p2 = str(tmp_path / "soup.off")
    _write_off(p2, soup_v, soup_t)
    assert _compare_everything(vra, p2, 96) > 5000


def test_flat_mesh_gives_the_same_empty_scene(vra, tmp_path):
    """All vertices in one plane: normalizePoly divides by a zero extent (PLYfileReader.cpp:
    160-168), the scaled vertices are not finite and no triangle is rasterised."""
    p = str(tmp_path / "flat.off")
    _write_off(p, [(0, 0, 0.5), (1, 0, 0.5), (0, 1, 0.5), (1, 1, 0.5)], [(0, 1, 2), (1, 3, 2)])
    assert _compare_everything(vra, p, 32) == 0


@pytest.mark.parametrize("mesh,res", [("galleon.ply", (16, 32, 128, 256)), ("icosahedron.ply", (16, 32, 128, 256)),
                                      ("stratocaster.ply", (8, 16, 64, 96)), ("teapot.off", (16, 32, 128, 0))])
def test_lod_set_on_the_device_equals_vol_lod_set(vra, ora, mesh, res):
    """volume_gpu_voxelize_lod_set against the host's Vol_LOD_Set (main.cpp:23-50): all four
    grids, the third normalizePoly pass that `high` sees, the beam grid every scene carries, and
    a frame rendered at each level of detail."""
    path = os.path.join(ASSETS, mesh)
    sLowR, lowR, medR, highR = res
    ls = vra.LodSet(path, sLowR, lowR, medR, highR)
    m2, m3 = vra.HostMesh(path, 2), vra.HostMesh(path, 3)
    scenes = vra.GpuScene.voxelize_lod_set(m2.desc, m3.desc if highR else None, sLowR, lowR, medR, highR)
    assert (scenes[3] is None) == (highR == 0)
    want_low = vra.host.dense_view(ls.grid(1)).reshape(-1, 8)
    for lod, gpu in enumerate(scenes):
        if gpu is None:
            continue
        tgt = ls.grid(lod)
        want = vra.host.dense_view(tgt).reshape(-1, 8)
        assert gpu.size == (tgt.sx, tgt.sy, tgt.sz)
        got = gpu.read_voxels(0, tgt.sx * tgt.sy * tgt.sz)
        assert np.array_equal(got, want), f"LoD {lod}: {int((got != want).any(axis=1).sum())} voxels differ"
        assert np.array_equal(gpu.read_low_voxels(), want_low)
        view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(314, 500), (0, 0, -1))
        f = vra.frame_from_camera(view, (0.9, -0.3, 0.6), (tgt.sx,) * 3, (lowR,) * 3, 100, 75)
        assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, ora.render_frame(f, low=ls.grid(1), target=tgt), f"LoD {lod}")
        gpu.close()
    assert int((vra.host.dense_view(ls.grid(2))[..., 3] > 0).sum()) > 100
    for o in (m2, m3, ls):
        o.close()


SOUPS = list(range(int(os.environ.get("VOLLY_RANDOM_SOUPS", "8"))))     # a soak run sets a larger number


@pytest.mark.parametrize("case", SOUPS)
def test_random_triangle_soups_equal_the_host(vra, tmp_path, case):
    """Random triangle soups -- a few huge triangles, many small ones, slivers, repeated vertices,
    triangles in axis planes -- at random resolutions: target grid, low grid and octree equal the
    host pipeline's."""
    rng = np.random.default_rng(4000 + case)
    nv = int(rng.integers(4, 200))
    verts = rng.random((nv, 3))
    if case % 3 == 0:
        verts[rng.integers(0, nv, nv // 3), int(rng.integers(0, 3))] = rng.choice([0.0, 0.5, 1.0])    # many vertices in one axis plane
    if case % 4 == 1:
        verts = np.round(verts * 8) / 8                                                              # vertices on cell corners
    nt = int(rng.integers(1, 300))
    tris = rng.integers(0, nv, (nt, 3))
    if case % 5 == 2:
        local = rng.integers(0, nv, nt)                                                              # small triangles: nearby vertices
        verts = np.vstack([verts, np.clip(verts[local] + rng.normal(0, 0.01, (nt, 3)), 0, 1),
                           np.clip(verts[local] + rng.normal(0, 0.01, (nt, 3)), 0, 1)])
        tris = np.stack([local, nv + np.arange(nt), nv + nt + np.arange(nt)], axis=1)
    res = int(4 * rng.integers(2, 41))
    p = str(tmp_path / "soup.off")
    _write_off(p, verts, tris)
    _compare_everything(vra, p, res)


def test_device_cube_test_against_the_reference_function(vra):
    """The device's t_c_intersection held directly against the reference's own triangleCube.c
    (oracle/_ref, built from /root/reference where it lies) and against the host restatement, on
    random, touching and degenerate triangles -- the same families as tests/test_host.py, more of
    them."""
    import ctypes as C
    from conftest import ROOT
    rng = np.random.default_rng(23)
    tris = [rng.uniform(-s, s, size=(n, 9)) for s, n in ((0.4, 20000), (1.0, 80000), (3.0, 60000), (30.0, 10000))]
    grid = np.array([-1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5])
    tris.append(rng.choice(grid, size=(100000, 9)))
    flat = rng.choice(grid, size=(10000, 9)); flat[:, 6:9] = flat[:, 0:3]
    tris.append(flat)
    near = rng.choice(grid, size=(40000, 9)) + rng.normal(0, 1e-4, (40000, 9))     # around the test's 1e-4 tolerance
    tris.append(near)
    t = np.concatenate(tris).astype(np.float32)
    got = vra.gpu.tricube(t)
    host = np.array([vra.host.lib.volly_host_tricube(row.ctypes.data_as(C.POINTER(C.c_float))) for row in t[:60000]], np.int32)
    assert np.array_equal(got[:60000], host)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libtricube_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/libtricube_ref.so not built (needs /root/reference at build time)")
    ref = C.CDLL(ref_path).tricube_ref
    ref.argtypes, ref.restype = [C.POINTER(C.c_float)], C.c_int
    want = np.array([ref(row.ctypes.data_as(C.POINTER(C.c_float))) for row in t], np.int32)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{bad.size} of {t.shape[0]} differ, first {t[bad[0]].tolist()}: device {got[bad[0]]}, reference {want[bad[0]]}"
    assert 30000 < int((want == 0).sum()) < 250000
