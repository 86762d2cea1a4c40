"""Known-answer tests that pin the CPU oracle to values derived by hand from the
reference source (main.cpp:88-161, 264-269; volume.h:95-111).  The reference has
no tests or fixtures of its own, so these and the committed
golden planes are what the oracle is pinned by: PARITY UNPINNED beyond them."""
import numpy as np
import pytest

from conftest import MISS


def _grid(vra, n, filled):
    vox = np.zeros((n, n, n, 8), np.uint8)
    for (x, y, z) in filled:
        vox[z, y, x] = (255, 255, 255, 255, 10, 20, 30, 0)
    return vra.host.dense_desc(vox)


def test_axis_ray_counts_cells_and_lands_on_the_cell_face(vra, ora):
    """+x ray from (-5, 10.5, 10.5) through a 32^3 grid with one voxel at (20,10,10).
    Entry clip (main.cpp:105-108): t0 = 5, cur = (0, 10.5, 10.5).  Every step moves
    exactly one cell (dn.x = 1, the y/z candidates are +inf), so the sample at
    main.cpp:137 runs for x = 0..20: 21 samples, ptEnd = (20, 10.5, 10.5)."""
    g = _grid(vra, 32, [(20, 10, 10)])
    r = ora.raycast_dense((-5, 10.5, 10.5), (1, 0, 0), g)
    assert r.found == 1 and r.steps == 21 and r.capped == 0
    assert tuple(r.pt_end) == (20.0, 10.5, 10.5)
    assert r.hit_index == 20 + 32 * (10 + 32 * 10)
    assert bytes(r.vox) == bytes([255, 255, 255, 255, 10, 20, 30, 0])


def test_empty_grid_miss_counts_every_cell_crossed(vra, ora):
    g = _grid(vra, 16, [])
    r = ora.raycast_dense((-2, 3.5, 4.5), (1, 0, 0), g)
    assert r.found == 0 and r.steps == 16 and r.hit_index == MISS
    assert tuple(r.pt_end) == (16.0, 3.5, 4.5)
    # Voxel(bgColor) = (0,0,0,255) with zero normal (main.cpp:159, voxel.h:23)
    assert bytes(r.vox) == bytes([0, 0, 0, 255, 0, 0, 0, 0])


def test_negative_exit_samples_cell_zero_twice(vra, ora):
    """glm::ivec4(vec4) truncates toward zero (main.cpp:124): after cell 0 the ray sits
    at x = 0, curPlusEpsilon = -0.0003 truncates to cell 0 again and passes the `< 0`
    test, so cell 0 is sampled a second time; floor(-0.0003) = -1 then sends the ray
    to x = -1 where -1.0003 truncates to -1 and the loop ends.  From x = 2.5 going -x:
    cells 2, 1, 0, 0 = 4 samples, exit point x = -1."""
    g = _grid(vra, 8, [])
    r = ora.raycast_dense((2.5, 3.5, 3.5), (-1, 0, 0), g)
    assert r.found == 0 and r.steps == 4
    assert tuple(r.pt_end) == (-1.0, 3.5, 3.5)


def test_start_inside_a_filled_voxel_hits_at_once(vra, ora):
    g = _grid(vra, 8, [(3, 3, 3)])
    r = ora.raycast_dense((3.25, 3.5, 3.75), (0.6, 0.0, 0.8), g)
    assert r.found == 1 and r.steps == 1 and tuple(r.pt_end) == (3.25, 3.5, 3.75)
    # solid mode (main.cpp:143-146) looks for the first EMPTY voxel instead
    s = ora.raycast_dense((3.25, 3.5, 3.75), (1, 0, 0), g, solid=True)
    assert s.found == 1 and s.steps == 2 and tuple(s.pt_end) == (4.0, 3.5, 3.75)


def test_ray_that_misses_the_volume_box(vra, ora):
    """Parallel to the x axis but outside the y slab: distToVolume.y*inv.y = +inf, so
    t0 = inf and cur = inf*dir + start has NaN where dir is 0; NaN converts to INT_MIN
    on x86 (cvttss2si), `< 0` holds, the loop breaks before any sample."""
    g = _grid(vra, 8, [(3, 3, 3)])
    r = ora.raycast_dense((-5, -2.0, 3.5), (1, 0, 0), g)
    assert r.found == 0 and r.steps == 0


def test_fill_sphere_known_first_hit(vra, ora):
    """fillSphere (volume.h:95-111): filled iff |4*(p-mid)/size|^2 < 1, i.e. |p - 32| < 16
    along an axis through the centre of a 64^3 grid: cells 17..47.  A +x ray along
    y = z = 32.5 samples cells 0..17: 18 samples, ptEnd.x = 17."""
    vox = ora.fill_sphere(64)
    assert vox[32, 32, 16, 3] == 0 and vox[32, 32, 17, 3] == 255 and vox[32, 32, 47, 3] == 255 and vox[32, 32, 48, 3] == 0
    g = vra.host.dense_desc(vox)
    r = ora.raycast_dense((-1, 32.5, 32.5), (1, 0, 0), g)
    assert r.found == 1 and r.steps == 18 and tuple(r.pt_end) == (17.0, 32.5, 32.5)
    # the host's VolumeStore::fillSphere builds the same grid
    host = vra.HostScene.sphere(64, svo=False)
    assert np.array_equal(vra.host.dense_view(host.target), vox)


def test_shade_known_values(ora):
    """main.cpp:264-269: L = clamp(dot(sun, n/255)/2 + 0.5, 0, 1), channel = (u8)(c*L)."""
    # zero normal: L = 0.5 -> 255*0.5 = 127.5 -> 127
    assert ora.shade([255, 255, 255, 255, 0, 0, 0, 0]) == 255 | (127 << 8) | (127 << 16) | (127 << 24)
    # full normal: dot = 3/sqrt(3) = 1.73 -> L clamps to 1
    assert ora.shade([255, 200, 100, 255, 255, 255, 255, 0]) == 255 | (255 << 8) | (200 << 16) | (100 << 24)
    # one axis: dot = 1/sqrt(3) -> L = 0.788675 -> 255*L = 201.1 -> 201
    assert ora.shade([255, 255, 255, 255, 255, 0, 0, 0]) == 255 | (201 << 8) | (201 << 16) | (201 << 24)


def test_beam_restarts_from_the_low_end_point_even_on_a_miss(vra, ora):
    """raycast_beamOpt (main.cpp:163-166) starts the target pass at scale*low.ptEnd
    whether or not the low pass hit: a low-grid miss leaves the start outside the
    grid, the target pass breaks at its first bounds check and reports a miss even
    though the target grid has a voxel on the ray."""
    import ctypes as C
    hi = _grid(vra, 32, [(20, 10, 10)])
    lo_empty = _grid(vra, 8, [])
    lo_hit = _grid(vra, 8, [(5, 2, 2)])
    st = np.array([-2, 2.6, 2.6], np.float32)
    d = np.array([1, 0, 0], np.float32)
    sc = np.array([4, 4, 4], np.float32)
    fp = C.POINTER(C.c_float)
    for low, found, steps_hi in ((lo_empty, 0, 0), (lo_hit, 1, 1)):
        a, b = ora.Report(), ora.Report()
        ora.lib.vo_raycast_beam(st.ctypes.data_as(fp), d.ctypes.data_as(fp), d.ctypes.data_as(fp), sc.ctypes.data_as(fp),
                                C.byref(low), C.byref(hi), None, C.byref(a), C.byref(b))
        assert b.found == found and b.steps == steps_hi
    # low hit at x = 5 -> target start x = 20 exactly: first sample is the voxel
    y = float(np.float32(2.6) * np.float32(4))
    assert tuple(b.pt_end) == (20.0, y, y)


def test_step_cap(ora):
    assert ora.step_cap(128, 128, 128) == 4 * 384 + 8


def test_octree_with_unit_cells_equals_the_dense_dda(vra, ora):
    """In a 2^3 grid every empty octree cell is one voxel, so the octree DDA takes the
    reference's unit steps and must agree bit for bit with the dense DDA."""
    rng = np.random.default_rng(3)
    vox = np.zeros((2, 2, 2, 8), np.uint8)
    vox[1, 0, 1] = (255, 255, 255, 255, 1, 2, 3, 0)
    vox[0, 1, 0] = (255, 255, 255, 255, 4, 5, 6, 0)
    scene = vra.HostScene.from_dense(vox, None, svo=True)
    for _ in range(300):
        start = rng.uniform(-3, 5, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        a = ora.raycast_dense(start, d, scene.target)
        b = ora.raycast_svo(start, d, scene.svo)
        assert (a.found, a.steps, a.hit_index, tuple(a.pt_end), bytes(a.vox)) == \
               (b.found, b.steps, b.hit_index, tuple(b.pt_end), bytes(b.vox))


def test_octree_and_dense_agree_on_hits_for_a_real_scene(vra, ora, teapot64):
    """Not a gate of the design (big empty steps accumulate `cur` differently), but the
    two traversals must see the same scene: hit/miss and hit voxel agree on nearly all rays."""
    pos, view = vra.orbit_camera(2, 16, 1.0)
    f = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, 160, 120, naive=True)
    a = ora.render_frame(f, target=teapot64.target)
    b = ora.render_frame(f, svo=teapot64.svo)
    assert (a["hit_index"] != MISS).sum() > 500
    assert (a["hit_index"] == b["hit_index"]).mean() > 0.995
    assert b["steps_hi"].sum() < a["steps_hi"].sum() / 3      # the octree skips empty space


def test_raytrace_known_values(vra, ora):
    """raytrace_naiive (main.cpp:170-194) on a scene small enough to follow by hand.
    One white voxel at (5,3,3) with normal bytes (254,127,127,0): the decoded normal is
    n = b/127 - 1 = (1, 0, 0, -1) (w lane: 0/127 - 1).  A -x ray from (7.5,3.5,3.5) hits it.
      iteration 1: colour = rgba/255 = (1,1,1,1).
      iteration 2: d = dot(n, I) = -1; reflect = I - n*d*2 = (+1,0,0,-2): the reflected ray
      leaves the grid (miss -> (0,0,0,1)); k = 1 - 4*(1 - 1) = 1, refract = 2I - (2d + 1)n =
      (-1,0,0,-1): the solid pass walks out of the voxel; the continuation ray misses too.
      colour = 0.35*(0,0,0,1) + 0*(0,0,0,1) + 0.65*(1,1,1,1) = (0.65, 0.65, 0.65, 1.0)."""
    vox = np.zeros((8, 8, 8, 8), np.uint8)
    vox[3, 3, 5] = (255, 255, 255, 255, 254, 127, 127, 0)
    g = vra.host.dense_desc(vox)
    c1, rep = ora.raytrace(1, (7.5, 3.5, 3.5), (-1, 0, 0, 0), g)
    assert rep.found == 1 and tuple(rep.pt_end) == (6.0, 3.5, 3.5)
    assert c1.tolist() == [1.0, 1.0, 1.0, 1.0]
    c2, _ = ora.raytrace(2, (7.5, 3.5, 3.5), (-1, 0, 0, 0), g)
    f = np.float32
    diffuse = f(f(f(0.65) * f(255)) * f(f(1) / f(255)))
    assert c2[:3].tolist() == [float(diffuse)] * 3
    assert c2[3] == f(f(f(0.35) * f(1)) + f(0)) + diffuse
    # a ray that hits nothing returns the background voxel's colour (0,0,0,255)/255
    c0, rep0 = ora.raytrace(3, (7.5, 6.5, 6.5), (-1, 0, 0, 0), g)
    assert rep0.found == 0 and c0.tolist() == [0.0, 0.0, 0.0, 1.0]


def test_raytrace_frame_differs_from_plain_shading_only_in_colour(vra, ora, teapot64):
    pos, view = vra.orbit_camera(2, 16, 1.0)
    fr = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, 96, 64, raytrace=True)
    fn = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, 96, 64, naive=True)
    a = ora.render_frame(fr, target=teapot64.target)
    b = ora.render_frame(fn, target=teapot64.target)
    for p in ("hit_index", "steps_hi", "depth"):
        assert np.array_equal(a[p].view(np.uint32), b[p].view(np.uint32))
    hit = a["hit_index"] != MISS
    assert hit.sum() > 300
    assert (a["rgba8"][~hit] == 255).all()                      # miss: (0,0,0,1) -> u8vec4(255,0,0,0)
    grey = (a["rgba8"][hit] >> 8) & 0xFF
    assert grey.min() >= int(0.65 * 255) - 1                    # at least the diffuse share of a white voxel
