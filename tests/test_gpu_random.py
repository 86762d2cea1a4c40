"""Seeded differential test: random scenes x random cameras x random frame shapes, GPU (through
the C-ABI) against the oracle, all five planes bit for bit.  The cases are drawn to reach every
traversal path: grids made of whole 16^3 regions and grids that are not (the assembly loops run on the
grid padded to whole regions, whose rim cells read as filled: a sample there is where the ray left the
grid), with and without the region level, octree and dense targets, beam / naive /
raytrace frames, cameras outside, inside and on cell boundaries, sub-rectangles, axis-parallel
views (zero direction components) and several frames per call."""
import numpy as np
import pytest

from conftest import MISS, assert_planes_equal

pytestmark = pytest.mark.gpu

ALL = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth", "pt_end")     # pt_end: RaycastReport::ptEnd, 3 floats per ray


def _random_grid(rng, shape, density, blobs):
    sz, sy, sx = shape
    v = np.zeros((sz, sy, sx, 8), np.uint8)
    fill = rng.random((sz, sy, sx)) < density
    z, y, x = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    for _ in range(blobs):
        c = rng.random(3) * (sz, sy, sx)
        r = rng.uniform(2, max(3, min(shape) / 3))
        fill |= (z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2 < r * r
    n = int(fill.sum())
    v[fill, 0:3] = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    v[fill, 3] = rng.integers(1, 256, n, dtype=np.uint8)
    v[fill, 4:7] = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    return v


def _low_of(v, f):
    sz, sy, sx = v.shape[:3]
    lz, ly, lx = -(-sz // f), -(-sy // f), -(-sx // f)
    pad = np.zeros((lz * f, ly * f, lx * f), bool)
    pad[:sz, :sy, :sx] = v[..., 3] > 0
    any_ = pad.reshape(lz, f, ly, f, lx, f).any(axis=(1, 3, 5))
    low = np.zeros((lz, ly, lx, 8), np.uint8)
    low[any_, 0:4] = 255
    return low


import os
CASES = list(range(int(os.environ.get("VOLLY_RANDOM_CASES", "36"))))   # a soak run sets a larger number


@pytest.mark.parametrize("case", CASES)
def test_random_scene_camera_frame(vra, ora, case):
    rng = np.random.default_rng(1000 + case)
    regular = case % 3 != 2
    if regular:
        shape = tuple(int(16 * rng.integers(1, 5)) for _ in range(3))
    else:
        shape = tuple(int(rng.integers(5, 50)) for _ in range(3))
    cubic_pow2 = case % 4 == 0
    if cubic_pow2:
        shape = (int(2 ** rng.integers(4, 7)),) * 3
    density = [0.0, 0.002, 0.02, 0.3][case % 4]
    vox = _random_grid(rng, shape, density, blobs=int(rng.integers(0, 4)))
    low = _low_of(vox, int(rng.choice([2, 4, 8])))
    use_svo = cubic_pow2 and case % 8 == 0
    scene = vra.HostScene.from_dense(vox, low, svo=use_svo)
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low) if use_svo else vra.GpuScene.upload_dense(scene.target, scene.low)
    assert gpu.info(15) in (1, 2) and (use_svo or gpu.info(14) in (1, 2))      # no grid of these sizes is left to the checked loop
    hi = (shape[2], shape[1], shape[0])
    lo = (low.shape[2], low.shape[1], low.shape[0])
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    frames = []
    for k in range(int(rng.integers(1, 4))):
        kind = (case + k) % 5
        if kind == 0:                                   # inside the volume
            pos = rng.random(3)
        elif kind == 1:                                 # on a cell boundary of the target grid
            pos = np.round(rng.random(3) * np.array(hi)) / np.array(hi)
        else:                                           # outside, anywhere around
            pos = 0.5 + (rng.random(3) - 0.5) * rng.uniform(1.2, 4.0)
        if kind == 3:                                   # axis-parallel view: zero direction components
            d = np.zeros(3)
            d[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])
            center = pos + d
        else:
            center = rng.random(3)
        if np.allclose(center, pos):
            center = pos + np.array([0.3, 0.2, 0.1])
        up = (0, 0, -1) if abs((center - pos)[2]) < 0.9 * np.linalg.norm(center - pos) else (0, 1, 0)
        view = vra.look_at(pos.astype(np.float32), center.astype(np.float32), up)
        mode = int(rng.integers(0, 3)) if not use_svo else int(rng.integers(0, 2))
        f = vra.frame_from_camera(view, pos.astype(np.float32), hi, lo, w, h, naive=mode == 1, raytrace=mode == 2)
        frames.append(f)
    if rng.random() < 0.4 and w > 3 and h > 3:          # a sub-rectangle of the frame(s)
        x0, y0 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
        tw, th = int(rng.integers(1, w - x0 + 1)), int(rng.integers(1, h - y0 + 1))
        for f in frames:
            f.x0, f.y0, f.tile_w, f.tile_h = x0, y0, tw, th
    got = gpu.render(frames, planes=ALL)
    five = gpu.render(frames)          # without pt_end the frame kernel may skip the set-up of sky tiles (sky_rect, volly_gpu.hip)
    for i, f in enumerate(frames):
        want = (ora.render_frame(f, low=scene.low, svo=scene.svo, planes=ALL) if use_svo
                else ora.render_frame(f, low=scene.low, target=scene.target, planes=ALL))
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"case {case} frame {i} shape {shape} flags {f.flags}")
        assert_planes_equal({k: v[i] for k, v in five.items()}, want, f"case {case} frame {i} shape {shape} flags {f.flags}, five planes")
    gpu.close()
    scene.close()


SVO_CASES = list(range(int(os.environ.get("VOLLY_RANDOM_SVO_CASES", "24"))))


@pytest.mark.parametrize("case", SVO_CASES)
def test_random_octree(vra, ora, case):
    """Octree targets only, drawn to reach both traversals: cubic grids whose side is a multiple of 8 -- powers of two and
    not (res < 2^depth: cells of the tree beyond the grid) -- take the flattened tree (table + occupancy blocks, the
    assembly loop; scene_info 16), other sides the node chain; sparse to dense fillings, cameras inside, outside, on cell
    boundaries and axis-parallel (irregular rays: the node chain whatever the tree), beam and naive frames."""
    rng = np.random.default_rng(5000 + case)
    res = int([8, 16, 24, 32, 40, 56, 64, 72, 100, 128, 20, 96][case % 12])
    density = [0.0005, 0.004, 0.05, 0.4][(case // 3) % 4]
    vox = _random_grid(rng, (res,) * 3, density, blobs=int(rng.integers(0, 5)))
    f_low = int(rng.choice([2, 4, 8]))
    low = _low_of(vox, f_low)
    scene = vra.HostScene.from_dense(vox, low, svo=True)
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
    # (an empty tree deeper than three levels has no level-3 node: it takes the node chain)
    assert gpu.info(16) == (1 if res % 8 == 0 and (gpu.info(11) > 0 or res == 8) else 0)
    assert gpu.info(18) == gpu.info(16)       # the host builder stores leaves in tree order: a hit's leaf is found by rank
    hi, lo = (res,) * 3, (low.shape[2], low.shape[1], low.shape[0])
    w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
    frames = []
    for k in range(3):
        kind = (case + k) % 5
        if kind == 0:
            pos = rng.random(3)
        elif kind == 1:
            pos = np.round(rng.random(3) * res) / res
        else:
            pos = 0.5 + (rng.random(3) - 0.5) * rng.uniform(1.2, 3.0)
        if kind == 3:
            d = np.zeros(3)
            d[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])
            center = pos + d
        else:
            center = 0.5 + (rng.random(3) - 0.5) * 0.6
        if np.allclose(center, pos):
            center = pos + np.array([0.3, 0.2, 0.1])
        up = (0, 0, -1) if abs((center - pos)[2]) < 0.9 * np.linalg.norm(center - pos) else (0, 1, 0)
        view = vra.look_at(pos.astype(np.float32), center.astype(np.float32), up)
        frames.append(vra.frame_from_camera(view, pos.astype(np.float32), hi, lo, w, h, naive=bool(rng.integers(0, 2))))
    got = gpu.render(frames, planes=ALL)
    five = gpu.render(frames)
    hits = 0
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=scene.low, svo=scene.svo, planes=ALL)
        hits += int((want["hit_index"] != MISS).sum())
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"octree case {case} frame {i} res {res} flags {f.flags}")
        assert_planes_equal({k: v[i] for k, v in five.items()}, want, f"octree case {case} frame {i} res {res} flags {f.flags}, five planes")
    gpu.close()
    scene.close()


def test_octree_with_leaves_out_of_tree_order(vra, ora):
    """A valid tree whose leaf array is NOT in tree order (the leaf groups of the level-1 nodes stored back to front):
    the flattened traversal still applies, but a hit's leaf cannot be found by its rank among the voxels of its
    occupancy word -- the library must notice at upload (scene_info 18 = 0) and walk down from the level-3 node."""
    import ctypes as C
    rng = np.random.default_rng(77)
    res = 32
    vox = _random_grid(rng, (res,) * 3, 0.05, blobs=3)
    low = _low_of(vox, 4)
    scene = vra.HostScene.from_dense(vox, low, svo=True)
    nodes, leaves = scene.svo_arrays()
    nodes, leaves = nodes.copy(), leaves.copy()
    l1 = np.flatnonzero(nodes["leaf_mask"] != 0)                      # level-1 nodes, in tree order
    counts = np.array([bin(int(m)).count("1") for m in nodes["leaf_mask"][l1]])
    old_start = nodes["child_offset"][l1].astype(np.int64)
    new_start = len(leaves) - np.cumsum(counts)                       # the first group goes last
    shuffled = np.empty_like(leaves)
    for a, b, k in zip(old_start, new_start, counts):
        shuffled[b:b + k] = leaves[a:a + k]
    nodes["child_offset"][l1] = new_start.astype(np.uint32)
    d = scene.svo
    desc = type(d)(d.n_nodes, d.n_leaves, d.depth, d.res, nodes.ctypes.data, shuffled.ctypes.data)
    gpu = vra.GpuScene.upload_svo(desc, scene.low)
    assert gpu.info(16) == 1 and gpu.info(18) == 0
    ordered = vra.GpuScene.upload_svo(scene.svo, scene.low)
    assert ordered.info(16) == 1 and ordered.info(18) == 1
    hi, lo = (res,) * 3, (low.shape[2], low.shape[1], low.shape[0])
    frames = list(vra.orbit_frames(range(4), 4, hi, lo, 96, 64, radius=0.9)) + list(vra.orbit_frames(range(2), 4, hi, lo, 96, 64, radius=0.3, naive=True))
    got, ref = gpu.render(frames, planes=ALL), ordered.render(frames, planes=ALL)
    hits = 0
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=scene.low, svo=desc, planes=ALL)
        hits += int((want["hit_index"] != MISS).sum())
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"shuffled leaves, frame {i}")
        assert_planes_equal({k: v[i] for k, v in ref.items()}, want, f"ordered leaves against the shuffled tree's oracle, frame {i}")
    assert hits > 1000
    gpu.close(); ordered.close(); scene.close()


PACKED_CASES = list(range(int(os.environ.get("VOLLY_RANDOM_PACKED_CASES", "30"))))


@pytest.mark.parametrize("case", PACKED_CASES)
def test_random_dense_packed_kernel(vra, ora, case, monkeypatch):
    """vk::render_frames_packed (dense targets without the region level: blocks of four tiles, the rays still under way
    packed into full wavefronts between rounds) forced onto random grids of every filling and of sides that are and are not
    multiples of 16, with round lengths
    from 1 iteration (a compaction after every step) to more than any ray takes (none at all): beam and naive frames,
    several per call, odd sizes and sub-rectangles (blocks that hang over the edge), cameras inside and outside, all six
    planes bit for bit against the oracle."""
    rng = np.random.default_rng(9000 + case)
    budget = [1, 2, 5, 16, 32, 100000][case % 6]
    monkeypatch.setenv("VOLLY_PACKED", "1")
    monkeypatch.setenv("VOLLY_PACK_BUDGET", str(budget))
    passes, pass_len = [(1, 96), (3, 1), (6, 7), (32, 3), (6, 96)][case % 5]      # rays parked and pooled between passes: never, after every few iterations, as shipped
    monkeypatch.setenv("VOLLY_PACK_PASSES", str(passes))
    monkeypatch.setenv("VOLLY_PACK_PASS_LEN", str(pass_len))
    monkeypatch.setenv("VOLLY_PACK_LOOKAHEAD_FROM", str([1, 2, 9][case % 3]))      # which passes take the look-ahead loop
    shape = tuple(int(16 * rng.integers(1, 5)) for _ in range(3))
    if case % 4 == 1:                                   # sides that are no multiple of 16, 4 or 2: rays leave through rim cells
        shape = tuple(int(rng.integers(5, 60)) for _ in range(3))
    density = [0.0, 0.001, 0.01, 0.1][(case // 2) % 4]
    vox = _random_grid(rng, shape, density, blobs=int(rng.integers(0, 3)))
    low = _low_of(vox, int(rng.choice([2, 4, 8])))
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    gpu = vra.GpuScene.upload_dense(scene.target, scene.low)
    assert gpu.info(8) == 3
    hi, lo = (shape[2], shape[1], shape[0]), (low.shape[2], low.shape[1], low.shape[0])
    w, h = int(rng.integers(1, 150)), int(rng.integers(1, 70))
    frames = []
    for k in range(int(rng.integers(1, 4))):
        kind = (case + k) % 4
        pos = rng.random(3) if kind == 0 else 0.5 + (rng.random(3) - 0.5) * rng.uniform(1.2, 3.0)
        if kind == 3:                                   # axis-parallel view: irregular rays take the checked loop inside the kernel
            d = np.zeros(3)
            d[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])
            center = pos + d
        else:
            center = 0.5 + (rng.random(3) - 0.5) * 0.5
        up = (0, 0, -1) if abs((center - pos)[2]) < 0.9 * np.linalg.norm(center - pos) else (0, 1, 0)
        view = vra.look_at(pos.astype(np.float32), center.astype(np.float32), up)
        frames.append(vra.frame_from_camera(view, pos.astype(np.float32), hi, lo, w, h, naive=bool(case % 2)))
    if case % 3 == 0 and w > 3 and h > 3:
        x0, y0 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
        tw, th = int(rng.integers(1, w - x0 + 1)), int(rng.integers(1, h - y0 + 1))
        for f in frames:
            f.x0, f.y0, f.tile_w, f.tile_h = x0, y0, tw, th
    got = gpu.render(frames, planes=ALL)
    five = gpu.render(frames)
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=scene.low, target=scene.target, planes=ALL)
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"packed case {case} frame {i} shape {shape} budget {budget} passes {passes} x {pass_len}")
        assert_planes_equal({k: v[i] for k, v in five.items()}, want, f"packed case {case} frame {i} shape {shape} budget {budget}, five planes")
    gpu.close()
    scene.close()


def test_the_random_cases_reach_hits_and_misses(vra, ora):
    """Guard against a vacuous suite: over a few of the cases above there must be hits, misses
    and rays that take steps in both grids."""
    rng = np.random.default_rng(7)
    vox = _random_grid(rng, (32, 48, 64), 0.01, blobs=3)
    low = _low_of(vox, 4)
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    pos, view = vra.orbit_camera(3, 16, 1.1)
    f = vra.frame_from_camera(view, pos, (64, 48, 32), (16, 12, 8), 80, 60)
    want = ora.render_frame(f, low=scene.low, target=scene.target)
    assert (want["hit_index"] != MISS).sum() > 200 and (want["hit_index"] == MISS).sum() > 200
    assert (want["steps_lo"] > 0).sum() > 1000 and (want["steps_hi"] > 0).sum() > 200
    scene.close()


def test_large_empty_cubes_and_run_ahead(vra, ora):
    """The whole-traversal loop keeps a lane quiet across the largest aligned empty cube around its
    region (32^3, 64^3, 128^3) and runs four steps ahead without looking when every lane is deep inside
    one: a 256^3 grid that is almost empty -- a few voxels in a corner, a thin slab, a lone voxel in
    the middle of a 128^3 void -- traversed from the camera (naive) and behind its 64^3 beam grid, from
    outside, from inside a void, along the axes and diagonally across cube corners."""
    rng = np.random.default_rng(5)
    n = 256
    vox = np.zeros((n, n, n, 8), np.uint8)
    fill = np.zeros((n, n, n), bool)
    fill[3:9, 2:5, 250:255] = True                       # a clump in one corner
    fill[200, 40:220:3, 30:230:5] = True                 # a perforated slab
    fill[64, 64, 192] = True                             # one voxel in the middle of an otherwise empty 128^3 octant
    fill[rng.integers(130, 250, 40), rng.integers(130, 250, 40), rng.integers(0, 120, 40)] = True
    k = int(fill.sum())
    vox[fill, 0:3] = rng.integers(0, 256, (k, 3), dtype=np.uint8)
    vox[fill, 3] = 255
    vox[fill, 4:7] = rng.integers(0, 256, (k, 3), dtype=np.uint8)
    low = _low_of(vox, 4)
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    gpu = vra.GpuScene.upload_dense(scene.target, scene.low)
    assert gpu.dda_loops() == "target asm-regions, low asm-regions"
    hi, lo = (n,) * 3, (n // 4,) * 3
    cams = [((1.6, -0.4, 1.3), (0.2, 0.6, 0.4)), ((0.25, 0.25, 0.75), (0.9, 0.9, 0.1)), ((-0.5, 0.5, 0.5), (1.0, 0.5, 0.5)),
            ((0.5, 0.5, -0.7), (0.5, 0.5, 1.0)), ((-0.3, -0.3, -0.3), (1.0, 1.0, 1.0)), ((0.75, 0.26, 0.24), (0.0, 0.3, 0.8))]
    total_hits = longest = 0
    for i, (pos, center) in enumerate(cams):
        eye, at = np.array(pos, np.float32), np.array(center, np.float32)
        up = (0, 0, -1) if abs((at - eye)[2]) < 0.9 * np.linalg.norm(at - eye) else (0, 1, 0)     # lookAt needs up not parallel to the view
        view = vra.look_at(eye, 2 * eye - at, up)             # the reference looks along -camDir (main.cpp:219,397)
        for naive in (True, False):
            f = vra.frame_from_camera(view, np.array(pos, np.float32), hi, lo, 120, 88, naive=naive)
            want = ora.render_frame(f, low=scene.low, target=scene.target, planes=ALL, threads=16)
            got = gpu.render(f, planes=ALL)
            assert_planes_equal({p: v[0] for p, v in got.items()}, want, f"camera {i} naive={naive}")
            assert_planes_equal({p: v[0] for p, v in gpu.render(f).items()}, want, f"camera {i} naive={naive}, five planes")
            total_hits += int((want["hit_index"] != MISS).sum())
            longest = max(longest, int(want["steps_hi"].max() if naive else want["steps_lo"].max()))
    assert total_hits > 500 and longest > 200
    gpu.close()
    scene.close()


def test_sky_rectangle_is_safe_and_not_vacuous(vra, ora, teapot64, sphere64):
    """volly_gpu_sky_rect: outside the rectangle the ORACLE must report a miss with zero steps in both
    passes for every pixel (that is what culled tiles write without tracing); from outside the volume
    the rectangle must actually exclude a good part of the frame; from inside, for raytrace frames and
    for frames whose low and target constants disagree it must be the whole frame."""
    rng = np.random.default_rng(11)
    hi, lo = (64,) * 3, (16,) * 3
    gpu = vra.GpuScene.upload_dense(teapot64.target, teapot64.low)
    culled = total = 0
    for case in range(60):
        naive = case % 3 == 0
        pos = (0.5 + (rng.random(3) - 0.5) * rng.uniform(0.2, 6.0)).astype(np.float32)
        at = rng.random(3).astype(np.float32)
        up = (0, 0, -1) if abs((at - pos)[2]) < 0.9 * np.linalg.norm(at - pos) else (0, 1, 0)
        view = vra.look_at(pos, 2 * pos - at, up)
        w, h = int(rng.integers(16, 200)), int(rng.integers(16, 150))
        f = vra.frame_from_camera(view, pos, hi, lo, w, h, naive=naive)
        if case % 4 == 1 and w > 20 and h > 20:
            f.x0, f.y0, f.tile_w, f.tile_h = 5, 7, w - 9, h - 11
        x_lo, x_hi, y_lo, y_hi = gpu.sky_rect(f)
        want = ora.render_frame(f, low=teapot64.low, target=teapot64.target)
        inside = np.zeros((f.tile_h, f.tile_w), bool)
        if x_lo <= x_hi and y_lo <= y_hi:
            inside[y_lo:y_hi + 1, x_lo:x_hi + 1] = True
        out = ~inside
        assert (want["hit_index"][out] == MISS).all() and (want["steps_lo"][out] == 0).all() and (want["steps_hi"][out] == 0).all(), \
            f"case {case}: a pixel outside the sky rectangle {x_lo, x_hi, y_lo, y_hi} is not sky"
        assert np.isinf(want["depth"][out]).all() and (want["rgba8"][out] == 255).all()
        assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, want, f"case {case}")
        culled += int(out.sum()); total += out.size
        if np.all((pos > 0.05) & (pos < 0.95)):           # camera inside the volume: corners behind it
            assert (x_lo, x_hi, y_lo, y_hi) == (0, f.tile_w - 1, 0, f.tile_h - 1)
    assert culled > 0.2 * total, (culled, total)
    # raytrace frames and inconsistent constants: no cull
    pos, view = vra.orbit_camera(3, 16, 3.0)
    f = vra.frame_from_camera(view, pos, hi, lo, 96, 64, raytrace=True)
    assert gpu.sky_rect(f) == (0, 95, 0, 63)
    f = vra.frame_from_camera(view, pos, hi, lo, 96, 64)
    assert gpu.sky_rect(f) != (0, 95, 0, 63)
    f.xdiff_hi[0] = f.xdiff_hi[0] * 1.01 + 0.001          # the target constants no longer describe the low rays
    assert gpu.sky_rect(f) == (0, 95, 0, 63)
    assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, ora.render_frame(f, low=teapot64.low, target=teapot64.target), "inconsistent")
    gpu.close()
