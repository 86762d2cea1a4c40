"""GPU parity: the HIP path, called through the C-ABI (ctypes over libvolly_gpu.so),
against the CPU oracle on the same inputs.  Bar: hit index, step counts, depth
(bit pattern) and rgba8 all bit-exact -- tighter than the 1/255 the north star
allows for the colour."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, ASSETS, MISS, assert_planes_equal

pytestmark = pytest.mark.gpu


def _oracle(ora, frame, scene, kind):
    if kind == "dense":
        return ora.render_frame(frame, low=scene.low, target=scene.target)
    return ora.render_frame(frame, low=scene.low, svo=scene.svo)


def _upload(vra, scene, kind):
    if kind == "dense":
        return vra.GpuScene.upload_dense(scene.target, scene.low)
    return vra.GpuScene.upload_svo(scene.svo, scene.low)


def _sizes(scene):
    lo = scene.low
    if scene.target is not None:
        t = scene.target
        hi = (t.sx, t.sy, t.sz)
    else:
        hi = (scene.svo.res,) * 3
    return hi, (lo.sx, lo.sy, lo.sz)


@pytest.mark.parametrize("kind", ["dense", "svo"])
@pytest.mark.parametrize("naive", [False, True])
def test_teapot64_views(vra, ora, teapot64, kind, naive):
    hi, lo = _sizes(teapot64)
    gpu = _upload(vra, teapot64, kind)
    frames = vra.orbit_frames(range(0, 16, 3), 16, hi, lo, 160, 120, radius=1.0, naive=naive)
    got = gpu.render(frames)
    total_hits = 0
    for i, f in enumerate(frames):
        want = _oracle(ora, f, teapot64, kind)
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"{kind} naive={naive} view {i}")
        total_hits += int((want["hit_index"] != MISS).sum())
    assert total_hits > 1000
    gpu.close()


@pytest.mark.parametrize("kind", ["dense", "svo"])
def test_sphere_inside_and_odd_sizes(vra, ora, sphere64, kind):
    """fillSphere scene (volume.h:95-111); frame sizes that are not multiples of
    the 32x8 block or the 8x8 tile; camera inside the volume."""
    hi, lo = _sizes(sphere64)
    gpu = _upload(vra, sphere64, kind)
    for (w, h), pos in [((37, 19), (0.1, 0.1, 0.1)), ((8, 8), (1.4, 0.3, 0.6)), ((1, 1), (0.5, -0.7, 0.5)),
                        ((131, 7), (0.5, 0.5, 0.5))]:
        view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(300, 500), (0, 0, -1))
        f = vra.frame_from_camera(view, pos, hi, lo, w, h)
        assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, _oracle(ora, f, sphere64, kind),
                            f"{kind} {w}x{h} from {pos}")
    gpu.close()


@pytest.mark.parametrize("kind", ["dense", "svo"])
def test_row_bands_and_columns(vra, ora, teapot64, kind):
    """Sub-rectangles (multi-GPU row bands): the band render equals the same rows of
    the full frame, which needs the serial direction sums to be restarted exactly."""
    hi, lo = _sizes(teapot64)
    gpu = _upload(vra, teapot64, kind)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, hi, lo, 200, 96)
    want = _oracle(ora, full, teapot64, kind)
    for (x0, y0, tw, th) in [(0, 24, 200, 24), (0, 72, 200, 24), (40, 13, 91, 50), (199, 95, 1, 1)]:
        sub = full.copy()
        sub.x0, sub.y0, sub.tile_w, sub.tile_h = x0, y0, tw, th
        got = gpu.render(sub)
        assert_planes_equal({k: v[0] for k, v in got.items()},
                            {k: v[y0:y0 + th, x0:x0 + tw] for k, v in want.items()}, f"{kind} sub {x0},{y0},{tw},{th}")
        assert_planes_equal({k: v[0] for k, v in got.items()}, _oracle(ora, sub, teapot64, kind), f"{kind} oracle sub")
    gpu.close()


def test_batch_crosses_launch_chunk(vra, ora, teapot64):
    """More frames than one pre-pass launch carries (16) and than one frame-kernel launch takes (64):
    frame-major output planes stay in order, through the host-buffer and the device entry point."""
    import torch
    hi, lo = _sizes(teapot64)
    gpu = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    n = 140
    frames = vra.orbit_frames(range(n), n, hi, lo, 48, 40, radius=0.9)
    got = gpu.render(frames, planes=("hit_index", "depth", "steps_lo"))
    for i in (0, 15, 16, 17, 31, 32, 47, 48, 63, 64, 65, 127, 128, 139):
        want = ora.render_frame(frames[i], low=teapot64.low, svo=teapot64.svo)
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"frame {i}")
    dev = {p: torch.full((n, 40, 48), 7, dtype=torch.float32 if p == "depth" else torch.int32, device="cuda:0") for p in got}
    gpu.render_device(frames, {p: t.data_ptr() for p, t in dev.items()})
    torch.cuda.synchronize()
    for p, t in dev.items():
        assert np.array_equal(t.cpu().numpy().view(np.uint32), np.asarray(got[p]).view(np.uint32)), p
    gpu.close()


def _manual_frame(vra, w, h, row0, xdiff, ydiff, origin, scale=1.0, naive=True):
    f = vra._abi.FrameDesc()
    for name, v in (("row0_hi", row0), ("xdiff_hi", xdiff), ("ydiff_hi", ydiff),
                    ("row0_lo", row0), ("xdiff_lo", xdiff), ("ydiff_lo", ydiff),
                    ("origin_hi", origin), ("origin_lo", origin)):
        arr = getattr(f, name)
        for c in range(3):
            arr[c] = float(v[c])
    for c in range(3):
        f.scale_lo_to_hi[c] = scale
    f.width, f.height, f.x0, f.y0, f.tile_w, f.tile_h = w, h, 0, 0, w, h
    f.flags = 1 if naive else 0
    return f


@pytest.mark.parametrize("kind", ["dense", "svo"])
def test_degenerate_rays(vra, ora, sphere64, kind):
    """Direction components that are exactly zero (1/0 = inf, 0*inf = NaN inside the
    DDA), rays parallel to a face outside the slab (NaN positions), rays that never
    touch the volume, and an all-zero direction (normalize gives NaN)."""
    gpu = _upload(vra, sphere64, kind)
    cases = [
        ((1, 0, 0), (0, 0, 0), (0, 0, 0), (-5, 32.5, 32.5)),        # +x through the centre
        ((-1, 0, 0), (0, 0, 0), (0, 0, 0), (70, 32.5, 32.5)),       # -x
        ((0, 0, 1), (0, 0, 0), (0, 0, 0), (32.5, 32.5, -3)),        # +z
        ((1, 0, 0), (0, 0, 0), (0, 0, 0), (-5, -2.0, 32.5)),        # parallel, outside the slab: NaN
        ((1, 0, 0), (0, 0.25, 0), (0, 0, 0.25), (-5, 30, 30)),      # fan with zero components in row 0 / col 0
        ((0, 0, 0), (0, 0, 0), (0, 0, 0), (10, 10, 10)),            # zero direction
        ((0, -1, 0), (0, 0, 0), (0, 0, 0), (0.0, 40, 0.0)),         # along an edge of the volume
        ((1, 1, 1), (0, 0, 0), (0, 0, 0), (-1, -1, -1)),            # exact diagonal through cell corners
    ]
    for row0, xd, yd, org in cases:
        f = _manual_frame(vra, 16, 8, row0, xd, yd, org)
        assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, _oracle(ora, f, sphere64, kind),
                            f"{kind} row0={row0} origin={org}")
    gpu.close()


def test_random_noncubic_dense(vra, ora):
    """Random sparse occupancy in a grid whose sides are not multiples of the 4^3
    brick, low grid with different aspect: exercises brick edges and payload fetch."""
    rng = np.random.default_rng(7)
    sz, sy, sx = 37, 50, 41
    vox = np.zeros((sz, sy, sx, 8), np.uint8)
    mask = rng.random((sz, sy, sx)) < 0.004
    vox[mask] = rng.integers(1, 256, size=(int(mask.sum()), 8), dtype=np.uint8)
    low = np.zeros((9, 13, 11, 8), np.uint8)
    lmask = rng.random((9, 13, 11)) < 0.05
    low[lmask] = 255
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    gpu = vra.GpuScene.upload_dense(scene.target, scene.low)
    for k in range(4):
        pos, view = vra.orbit_camera(k, 4, 0.8)
        for naive in (False, True):
            f = vra.frame_from_camera(view, pos, (sx, sy, sz), (11, 13, 9), 96, 64, naive=naive)
            want = ora.render_frame(f, low=scene.low, target=scene.target)
            assert_planes_equal({k2: v[0] for k2, v in gpu.render(f).items()}, want, f"noncubic view {k} naive={naive}")
    gpu.close()


def test_empty_scene_and_errors(vra, ora):
    vox = np.zeros((8, 8, 8, 8), np.uint8)
    scene = vra.HostScene.from_dense(vox, None, svo=True)
    for gpu in (vra.GpuScene.upload_dense(scene.target), vra.GpuScene.upload_svo(scene.svo)):
        pos, view = vra.orbit_camera(1, 8, 0.9)
        f = vra.frame_from_camera(view, pos, (8, 8, 8), (8, 8, 8), 40, 24, naive=True)
        got = gpu.render(f)
        assert (got["hit_index"] == MISS).all() and (got["rgba8"] == 255).all() and np.isinf(got["depth"]).all()
        # a beam frame needs the low grid (VOLLY_E_UNSUPPORTED)
        fb = vra.frame_from_camera(view, pos, (8, 8, 8), (8, 8, 8), 40, 24)
        with pytest.raises(vra.VollyError) as e:
            gpu.render(fb)
        assert e.value.code == -4
        bad = f.copy()
        bad.tile_w = 41
        with pytest.raises(vra.VollyError) as e:
            gpu.render(bad)
        assert e.value.code == -1
        gpu.close()


@pytest.mark.parametrize("res", [64, 100])
def test_noise_scene_matches_host_hash(vra, ora, res):
    """volume_gpu_generate_noise: voxels equal the documented hash on the host, the
    render of the generated scene equals the oracle on a host rebuild of it.  (100: a side that is no multiple of 16 --
    the target and its 25^3 beam grid are traversed on grids padded to whole regions, and the beam grid must not see the
    padding's cells.)"""
    seed, thr = 3, 1 << 25
    gpu = vra.GpuScene.noise(res, seed, thr)
    assert gpu.info(8) == 3 and gpu.info(9) == 0      # every region holds voxels: no region level, frames through render_frames_packed
    assert gpu.info(14) == 2 and gpu.info(15) == 2    # both grids through the assembly brick loop
    dev = gpu.read_voxels(0, res ** 3).reshape(res, res, res, 8)
    z, y, x = np.meshgrid(np.arange(res, dtype=np.uint32), np.arange(res, dtype=np.uint32),
                          np.arange(res, dtype=np.uint32), indexing="ij")
    with np.errstate(over="ignore"):
        h = np.uint32(seed) * np.uint32(0x9E3779B1) + x * np.uint32(0x85EBCA77)
        h ^= h >> np.uint32(15); h *= np.uint32(0x2C1B3C6D)
        h += y * np.uint32(0xC2B2AE3D)
        h ^= h >> np.uint32(13); h *= np.uint32(0x297A2D39)
        h += z * np.uint32(0x27D4EB2F)
        h ^= h >> np.uint32(16); h *= np.uint32(0x165667B1)
        h ^= h >> np.uint32(15)
    host = np.zeros((res, res, res, 8), np.uint8)
    fill = h < thr
    host[fill, 0:4] = 255
    host[fill, 4] = ((h >> 8) & 0xFF)[fill]
    host[fill, 5] = ((h >> 16) & 0xFF)[fill]
    host[fill, 6] = ((h >> 24) & 0xFF)[fill]
    assert np.array_equal(dev, host)
    assert gpu.info(17) == int(fill.sum())            # filled voxels as the library counts them (less the padding's cells)
    low = np.zeros((res // 4,) * 3 + (8,), np.uint8)
    lfill = fill.reshape(res // 4, 4, res // 4, 4, res // 4, 4).any(axis=(1, 3, 5))
    low[lfill, 0:4] = 255
    scene = vra.HostScene.from_dense(host, low, svo=False)
    pos, view = vra.orbit_camera(2, 9, 0.8)
    f = vra.frame_from_camera(view, pos, (res,) * 3, (res // 4,) * 3, 128, 72)
    assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()},
                        ora.render_frame(f, low=scene.low, target=scene.target), "noise")
    gpu.close()


def test_device_entry_point_and_timing(vra, ora, teapot64):
    """volume_gpu_render_device: device pointers (torch tensors), a non-default
    stream, event timing."""
    import torch
    hi, lo = _sizes(teapot64)
    gpu = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    frames = vra.orbit_frames(range(3), 8, hi, lo, 96, 64, radius=0.9)
    dev = torch.device("cuda:0")
    planes = {p: torch.zeros((3, 64, 96), dtype=torch.float32 if p == "depth" else torch.int32, device=dev)
              for p in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
    stream = torch.cuda.Stream(device=dev)
    gpu.set_profiling(True)
    with torch.cuda.stream(stream):
        gpu.render_device(frames, {p: t.data_ptr() for p, t in planes.items()}, stream=stream.cuda_stream)
    stream.synchronize()
    ms, n = gpu.kernel_time()
    assert n == 1 and ms > 0
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=teapot64.low, svo=teapot64.svo)
        got = {p: t[i].cpu().numpy().view(np.float32 if p == "depth" else np.uint32) for p, t in planes.items()}
        assert_planes_equal(got, want, f"device frame {i}")
    gpu.close()


def test_calls_on_alternating_streams_share_the_scratch_safely(vra, teapot128):
    """A scene's scratch (frame descriptors, direction checkpoints) is shared by its calls.  Calls
    queued back to back on two different streams, with different cameras and nothing waited for in
    between, must each produce what the same call produces alone: the second stream is made to
    wait for the first (volume_gpu_render_device's contract)."""
    import torch
    hi, lo = _sizes(teapot128)
    gpu = vra.GpuScene.upload_svo(teapot128.svo, teapot128.low)
    dev = torch.device("cuda:0")
    batches = [vra.orbit_frames(range(8 * b, 8 * b + 8), 64, hi, lo, 640, 360, radius=1.0) for b in range(6)]
    want = []
    for fr in batches:                                     # one at a time, fully synchronised
        t = torch.empty((8, 360, 640), dtype=torch.int32, device=dev)
        gpu.render_device(fr, {"hit_index": t.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want.append(t)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for _ in range(5):
        got = [torch.empty((8, 360, 640), dtype=torch.int32, device=dev) for _ in batches]
        for i, fr in enumerate(batches):                   # no synchronisation between the calls
            gpu.render_device(fr, {"hit_index": got[i].data_ptr()}, stream=streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(batches)):
            assert torch.equal(got[i], want[i]), f"batch {i} rendered on stream {i % 2}"
    gpu.close()


def test_config2_dragon512_full_frame(vra, ora):
    """BASELINE config 2: dragon.off at 512^3, octree target, 1920x1080, three
    cameras; all five planes bit-exact against the oracle on the full frame."""
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "dragon.off"), 512, dense=False, svo=True)
    hi, lo = (512,) * 3, (128,) * 3
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
    frames = vra.orbit_frames((0, 5, 11), 32, hi, lo, 1920, 1080, radius=1.0)
    got = gpu.render(frames)
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=scene.low, svo=scene.svo, threads=16)
        assert (want["hit_index"] != MISS).sum() > 50000
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"dragon512 view {i}")
    gpu.close()


def test_north_star_dragon1024_octree_full_4k_frame(vra, ora):
    """BASELINE.json's north-star target (bench.py's default workload): dragon.off at 1024^3,
    octree target, 256^3 beam grid, one full 3840x2160 frame, in beam mode (raycast_beamOpt,
    main.cpp:260 -> main.cpp:163-166) and in naive mode (main.cpp:261: the pure octree traversal
    from the camera).  All five planes of all 8.3 M rays bit-exact against the oracle."""
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "dragon.off"), 1024, dense=False, svo=True)
    hi, lo = (1024,) * 3, (256,) * 3
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
    assert gpu.dda_loops() == "target octree, low asm-regions"
    for naive, view in ((False, 900), (True, 2900)):
        f = vra.orbit_frames((view,), 4096, hi, lo, 3840, 2160, radius=1.0, naive=naive)[0]
        got = gpu.render(f)
        want = ora.render_frame(f, low=scene.low, svo=scene.svo, threads=16)
        assert (want["hit_index"] != MISS).sum() > 400000
        assert_planes_equal({k: v[0] for k, v in got.items()}, want, f"dragon1024 4K naive={naive}")
        if naive:
            assert int(want["steps_lo"].max()) == 0 and int(want["steps_hi"].max()) > 20
        else:
            assert int(want["steps_lo"].max()) > 200
    gpu.close()
    scene.close()


def test_camera_regimes_dragon1024_full_4k_frames(vra, ora):
    """The regimes bench.py's headline orbit avoids (bench.py --camera), each as one full 3840x2160 frame of the
    north-star scene against the oracle, all five planes bit for bit:
      startup  the reference's own start-up state, camPos = 0 (a corner of the grid, main.h:96) and camDir from the
               mouse at (0, 0) (main.cpp:393-397): every ray starts at the grid, nothing is sky, the cull is off;
      inside   the orbit at radius 0.25: the camera inside the grid (= the model's bounding box);
      fill     the orbit at radius 0.5: the model fills the frame (hit fraction >= 0.5).
    Beam mode (the live path) for all three, naive mode (pure octree traversal) for inside and fill."""
    import bench
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "dragon.off"), 1024, dense=False, svo=True)
    hi, lo = (1024,) * 3, (256,) * 3
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low)
    assert gpu.info(16) == 1
    W, H = 3840, 2160
    view0 = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(0, 0), (0, 0, -1))
    cases = [("startup", False, vra.frame_from_camera(view0, (0.0, 0.0, 0.0), hi, lo, W, H)),
             ("startup", True, vra.frame_from_camera(view0, (0.0, 0.0, 0.0), hi, lo, W, H, naive=True))]
    for cam, view in (("inside", 1300), ("fill", 2048)):
        for naive in (False, True):
            cases.append((cam, naive, vra.orbit_frames((view,), 4096, hi, lo, W, H, radius=bench.CAMERA_RADIUS[cam], naive=naive)[0]))
    for cam, naive, f in cases:
        want = ora.render_frame(f, low=scene.low, svo=scene.svo, threads=16)
        got = gpu.render(f)
        assert_planes_equal({k: v[0] for k, v in got.items()}, want, f"dragon1024 4K camera={cam} naive={naive}")
        hit = float((want["hit_index"] != MISS).mean())
        if cam == "startup":
            assert gpu.sky_rect(f) == (0, W - 1, 0, H - 1)                      # no cull from inside / at the grid
            assert int(want["steps_hi" if naive else "steps_lo"].min()) >= 1      # every ray samples: none is sky
        else:
            assert hit >= 0.5, (cam, hit)
    gpu.close()
    scene.close()


def test_config5_dragon1024_from_the_scene_cache(vra, ora, tmp_path):
    """BASELINE config 5's dragon half, and the on-disk scene cache on the GPU path (the
    reference re-voxelizes every model at every start, main.cpp:369-387): dragon.off is voxelized
    at 1024^3 on the device, the host pipeline's scene is saved and loaded back, the LOADED scene
    is uploaded, and two 1920x1080 orbit views of it must equal both the oracle (on the loaded
    scene) and the device-built scene's render."""
    path = os.path.join(ASSETS, "dragon.off")
    built = vra.HostScene.from_mesh(path, 1024, dense=False, svo=True)
    cache = str(tmp_path / "dragon1024.volly")
    built.save(cache)
    assert os.path.getsize(cache) > 10 << 20
    loaded = vra.HostScene.load(cache)
    nb, lb = built.svo_arrays()
    nl, ll = loaded.svo_arrays()
    assert np.array_equal(nb.view(np.uint64), nl.view(np.uint64)) and np.array_equal(lb, ll)
    built.close()
    mesh = vra.HostMesh(path, 2)
    gpu_dev = vra.GpuScene.voxelize(mesh.desc, 1024, dense=False, svo=True)
    gpu = vra.GpuScene.upload_svo(loaded.svo, loaded.low)
    ng, lg = gpu_dev.read_svo()
    assert np.array_equal(ng.view(np.uint64), nl.view(np.uint64)) and np.array_equal(lg, ll)
    frames = vra.orbit_frames((400, 3000), 4096, (1024,) * 3, (256,) * 3, 1920, 1080, radius=1.2)
    got, got_dev = gpu.render(frames), gpu_dev.render(frames)
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=loaded.low, svo=loaded.svo, threads=16)
        assert (want["hit_index"] != MISS).sum() > 50000
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"dragon1024 from the cache, view {i}")
        assert_planes_equal({k: v[i] for k, v in got_dev.items()}, want, f"dragon1024 built on the device, view {i}")
    for s_ in (gpu, gpu_dev, mesh, loaded):
        s_.close()


def test_config3_noise1024_rays_of_the_4k_frame(vra, ora):
    """BASELINE config 3: the device-generated 1024^3 noise volume (8 GiB of voxels) and its
    256^3 low grid, 3840x2160.  The volume is read back and the oracle traces two 256x144
    sub-rectangles of the 4K frame on it, beam and naive; the generator itself is pinned to the
    host hash by test_noise_scene_matches_host_hash."""
    res = 1024
    gpu = vra.GpuScene.noise(res, 1, 1 << 24)
    assert gpu.info(9) == 0
    vox = gpu.read_voxels(0, res ** 3).reshape(res, res, res, 8)
    low = gpu.read_low_voxels().reshape(res // 4, res // 4, res // 4, 8)
    filled = int((vox[::8, ::8, ::8, 3] > 0).sum())
    assert 0.5 < filled / (128 ** 3 / 256.0) < 2.0        # p = 2^24 / 2^32
    t_desc, l_desc = vra.host.dense_desc(vox), vra.host.dense_desc(low)
    for naive in (False, True):
        full = vra.orbit_frames((700,), 4096, (res,) * 3, (res // 4,) * 3, 3840, 2160, radius=1.1, naive=naive)[0]
        for x0, y0 in ((1800, 1000), (2500, 600)):
            f = full.copy()
            f.x0, f.y0, f.tile_w, f.tile_h = x0, y0, 256, 144
            want = ora.render_frame(f, low=l_desc, target=t_desc, threads=16)
            got = gpu.render(f)
            assert_planes_equal({k: v[0] for k, v in got.items()}, want, f"noise1024 naive={naive} at {x0},{y0}")
            assert (want["steps_hi"] > 0).sum() > 1000 and (want["hit_index"] != MISS).sum() > 1000
    gpu.close()


def test_sparse_1024_dense_grid_takes_the_region_loop_to_its_field_limits(vra, ora):
    """A 1024^3 dense target that is almost empty (device-generated noise, p = 2^-26: ~16 voxels): the
    region loop with its empty cubes and run-ahead blocks on a grid whose packed cell uses all ten bits
    per axis -- cells next to x, y, z = 0 and 1023, where `cell + ahead` wraps between fields."""
    res = 1024
    gpu = vra.GpuScene.noise(res, 3, 1 << 6)
    assert gpu.info(9) == 1 and gpu.dda_loops() == "target asm-regions, low asm-regions"
    vox = gpu.read_voxels(0, res ** 3).reshape(res, res, res, 8)
    low = gpu.read_low_voxels().reshape(res // 4, res // 4, res // 4, 8)
    assert 0 < int((vox[..., 3] > 0).sum()) < 200
    t_desc, l_desc = vra.host.dense_desc(vox), vra.host.dense_desc(low)
    longest = 0
    for naive in (True, False):
        for view, (x0, y0) in ((300, (1700, 900)), (1900, (2300, 1300)), (3500, (900, 400))):
            full = vra.orbit_frames((view,), 4096, (res,) * 3, (res // 4,) * 3, 3840, 2160, radius=1.05, naive=naive)[0]
            f = full.copy()
            f.x0, f.y0, f.tile_w, f.tile_h = x0, y0, 192, 96
            want = ora.render_frame(f, low=l_desc, target=t_desc, threads=16)
            assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, want, f"sparse1024 naive={naive} view {view}")
            longest = max(longest, int(want["steps_hi"].max()), int(want["steps_lo"].max()))
    assert longest > 1000                                  # rays cross the whole grid
    # along the grid's edges and faces: cells with coordinates 0 and 1023
    for org, d in (((-3.0, 0.5, 0.5), (1, 0, 0)), ((1030.0, 1023.5, 1023.5), (-1, 0, 0)), ((0.5, 1023.5, -2.0), (0, 0, 1)), ((1023.5, 1030.0, 0.5), (0, -1, 0))):
        f = _manual_frame(vra, 16, 8, d, (0.0, 1e-4, 2e-4) if d[0] else (1e-4, 0.0, 2e-4) if d[1] else (1e-4, 2e-4, 0.0), (2e-4, 1e-4, 1e-4), org)
        want = ora.render_frame(f, low=l_desc, target=t_desc)
        assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, want, f"sparse1024 edge ray from {org}")
    gpu.close()


def test_config5_stratocaster1024_full_frame_from_a_device_built_scene(vra, ora):
    """BASELINE config 5: stratocaster.ply at 1024^3 (octree), 1920x1080 orbit views.  The GPU
    scene is built on the device (volume_gpu_voxelize); the oracle traverses the host pipeline's
    octree and low grid, so one comparison covers scene preparation and traversal."""
    path = os.path.join(ASSETS, "stratocaster.ply")
    scene = vra.HostScene.from_mesh(path, 1024, dense=False, svo=True)
    mesh = vra.HostMesh(path, 2)
    gpu = vra.GpuScene.voxelize(mesh.desc, 1024, dense=False, svo=True)
    frames = vra.orbit_frames((100, 2200), 4096, (1024,) * 3, (256,) * 3, 1920, 1080, radius=1.2)
    got = gpu.render(frames)
    for i, f in enumerate(frames):
        want = ora.render_frame(f, low=scene.low, svo=scene.svo, threads=16)
        assert (want["hit_index"] != MISS).sum() > 20000
        assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"stratocaster1024 view {i}")
    for s_ in (gpu, mesh, scene):
        s_.close()


def test_config4_galleon2048_full_8k_frame_and_its_row_bands(vra, ora):
    """BASELINE config 4: galleon.ply at 2048^3 (octree only; hit_index is taken mod 2^32 there), the WHOLE
    7680x4320 frame -- 33 M rays, all five planes bit for bit against the oracle (16 threads), with the sky
    cull active (pt_end not requested: at this size the serial direction sums drift by pixels, which the cull's
    margin must cover, sky_drift_margin in volly_gpu.hip) -- and its plane CRCs against tests/golden/golden.json
    (the assembled-frame CRC).  Then row bands rendered as sub-rectangles, the way one GPU of eight
    would, against the same oracle rows: 64-row bands at the model's silhouette and a 540-row eighth."""
    import json
    from golden_cases import CASES, crc
    case = CASES["galleon2048_svo_8k"]
    path = os.path.join(ASSETS, "galleon.ply")
    scene = vra.HostScene.from_mesh(path, 2048, dense=False, svo=True)
    mesh = vra.HostMesh(path, 2)
    gpu = vra.GpuScene.voxelize(mesh.desc, 2048, dense=False, svo=True)
    assert gpu.info(16) == 1                              # the flattened tree: 256^3-entry table + blocks
    full = vra.orbit_frames((case["cams"][1][0][0],), 4096, (2048,) * 3, (512,) * 3, 7680, 4320, radius=1.0)[0]
    want = ora.render_frame(full, low=scene.low, svo=scene.svo, threads=16)
    got = gpu.render(full)
    assert_planes_equal({k: v[0] for k, v in got.items()}, want, "galleon2048 full 8K frame")
    assert (want["hit_index"] != MISS).sum() > 1_000_000
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]["galleon2048_svo_8k"]
    for p, a in got.items():
        assert crc(a) == meta["crc"][p], f"galleon2048 8K: CRC of plane {p} differs from the golden file"
    rect = gpu.sky_rect(full)
    assert rect[0] > 0 or rect[1] < 7679 or rect[2] > 0 or rect[3] < 4319      # the cull is active on this frame
    for y0, rows in ((rect[2], 64), (max(rect[3] - 63, 0), 64), (1500, 64), (4 * 540, 540)):
        band = full.band(y0, rows)
        part = gpu.render(band)
        assert_planes_equal({k: v[0] for k, v in part.items()}, {k: v[y0:y0 + rows] for k, v in want.items()}, f"galleon2048 rows {y0}..{y0 + rows - 1}")
    for s_ in (gpu, mesh, scene):
        s_.close()


def test_reference_style_state_example(vra, ora, tmp_path):
    """examples/volly_headless: the reference's main loop on volly::State (C++ adapter over
    the C-ABI): galleon.ply, the default 16/32/128/256 LoD set, 400x300 target (main.h:112).
    Its last frame must equal the oracle's render of the same State."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "volly_headless")
    if not os.path.exists(exe):
        pytest.fail("examples/volly_headless is not built (make examples)")
    mesh = os.path.join(ASSETS, "galleon.ply")
    for lod, cam in ((2, (0.9, -0.3, 0.6)), (0, (0.0, 0.0, 0.0))):
        ppm = tmp_path / f"out{lod}.ppm"
        r = subprocess.run([exe, mesh, str(lod), "3", str(ppm), "300", "500"] + [str(c) for c in cam],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "raycasts per second" in r.stdout
        raw = ppm.read_bytes()
        # the same State with its Vol_LOD_Set built on the GPU (State::initAllVolumesOnDevice)
        ppm_dev = tmp_path / f"out{lod}_dev.ppm"
        r = subprocess.run([exe, mesh, str(lod), "3", str(ppm_dev), "300", "500"] + [str(c) for c in cam], capture_output=True,
                           text=True, timeout=120, env=dict(os.environ, VOLLY_DEVICE_VOXELIZE="1"))
        assert r.returncode == 0, r.stderr
        assert ppm_dev.read_bytes() == raw
        assert raw.startswith(b"P6\n400 300\n255\n")
        got = np.frombuffer(raw[len(b"P6\n400 300\n255\n"):], np.uint8).reshape(300, 400, 3)
        ls = vra.LodSet(mesh)
        tgt, low = ls.grid(lod), ls.grid(1)
        view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(300 + 14, 500), (0, 0, -1))
        f = vra.frame_from_camera(view, cam, (tgt.sx,) * 3, (low.sx,) * 3, 400, 300)
        want = ora.render_frame(f, low=low, target=tgt)["rgba8"].view(np.uint8).reshape(300, 400, 4)
        assert want[..., 0].min() == 255
        assert np.array_equal(got, want[..., 1:4])


def test_pt_end_plane(vra, ora, teapot64):
    """The optional sixth plane, RaycastReport::ptEnd (main.cpp:81-86): three floats per ray, hits
    and misses, for dense and octree targets, beam / naive / raytrace frames; alone and together
    with the other planes; into pageable and into pinned memory.  (The experimental frame kernels
    of the `experimental` variant build: tests/test_gpu_variants.py.)"""
    ALL = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth", "pt_end")
    hi, lo = _sizes(teapot64)
    frames = list(vra.orbit_frames((1, 9), 16, hi, lo, 101, 67, radius=0.9)) + list(vra.orbit_frames((4,), 16, hi, lo, 101, 67, naive=True))
    for kind in ("dense", "svo"):
        gpu = _upload(vra, teapot64, kind)
        got = gpu.render(frames, planes=ALL)
        only = gpu.render(frames, planes=("pt_end",))
        for i, f in enumerate(frames):
            want = (ora.render_frame(f, low=teapot64.low, target=teapot64.target, planes=ALL) if kind == "dense"
                    else ora.render_frame(f, low=teapot64.low, svo=teapot64.svo, planes=ALL))
            assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"pt_end {kind} frame {i}")
            assert np.array_equal(only["pt_end"][i].view(np.uint32), want["pt_end"].view(np.uint32))
            assert np.isfinite(want["pt_end"]).all()
        if kind == "dense":
            pos, view = vra.orbit_camera(6, 16, 0.7)
            f = vra.frame_from_camera(view, pos, hi, lo, 64, 40, raytrace=True)
            assert_planes_equal({p: v[0] for p, v in gpu.render(f, planes=ALL).items()},
                                ora.render_frame(f, target=teapot64.target, planes=ALL), "pt_end raytrace")
        gpu.close()
    gpu = _upload(vra, teapot64, "svo")
    pin = vra.gpu.PinnedArray((len(frames), 67, 101, 3), np.float32)
    pin_rgba = vra.gpu.PinnedArray((len(frames), 67, 101), np.uint32)
    gpu.render_into(frames, {"pt_end": pin.array, "rgba8": pin_rgba.array})
    ref = gpu.render(frames, planes=("pt_end", "rgba8"))
    assert np.array_equal(pin.array.view(np.uint32), ref["pt_end"].view(np.uint32)) and np.array_equal(pin_rgba.array, ref["rgba8"])
    pin.close(); pin_rgba.close(); gpu.close()


def test_raytrace_frames(vra, ora, teapot64, sphere64):
    """VOLLY_FRAME_RAYTRACE: raytrace_naiive (main.cpp:170-194) with the solid-mode traversal
    raycast_naiive_vec4<true>; colour and primary-ray planes bit-exact against the oracle."""
    for scene in (teapot64, sphere64):
        hi, lo = _sizes(scene)
        gpu = vra.GpuScene.upload_dense(scene.target, scene.low)
        for k, radius in ((1, 1.0), (6, 0.7), (11, 0.45)):
            pos, view = vra.orbit_camera(k, 16, radius)
            f = vra.frame_from_camera(view, pos, hi, lo, 120, 67, raytrace=True)
            want = ora.render_frame(f, target=scene.target)
            assert_planes_equal({p: v[0] for p, v in gpu.render(f).items()}, want, f"raytrace view {k}")
            assert (want["hit_index"] != MISS).sum() > 200
        gpu.close()
    svo = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    with pytest.raises(vra.VollyError) as e:
        svo.render(f)
    assert e.value.code == -4
    svo.close()


def test_short_division_sequences_are_exact():
    """The kernels replace the compiler's IEEE expansions of 1/d, sqrt(x) and n/255 by short
    sequences for operands in [2^-100, 2^100] (volly_kernels.h: rcp_in_range, sqrt_in_range,
    over_255).  tools/exhaustive_math runs all 2^32 binary32 patterns through both on the
    device: the sequences in use must not differ once inside that range."""
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "exhaustive_math")
    if not os.path.exists(exe):
        pytest.fail("tools/exhaustive_math is not built (make tools)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    bad = {m.group(1): int(m.group(2)) for m in re.finditer(r"^(\S+)\s+mismatches in range (\d+)", r.stdout, re.M)}
    assert bad["rcp_a"] == 0 and bad["sqrt_candidates"] == 0, r.stdout
    m = re.search(r"n/255 refined: (\d+) mismatches", r.stdout)
    assert m and int(m.group(1)) == 0, r.stdout


def test_pinned_output_buffers_take_the_overlapped_path_and_agree(vra, ora):
    """volume_gpu_render into page-locked buffers (volly_gpu_host_alloc) renders in batches and
    copies back on a second stream; the planes must equal the pageable path's, for a call that
    spans several batches and an odd frame size."""
    host = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 128, dense=True, svo=False)
    gpu = vra.GpuScene.upload_dense(host.target, host.low)
    n, w, h = 37, 333, 251
    frames = vra.orbit_frames(range(0, 37 * 100, 100), 4096, (128,) * 3, (32,) * 3, w, h)
    ref = gpu.render(frames)
    pins = {p: vra.gpu.PinnedArray((n, h, w), np.float32 if p == "depth" else np.uint32) for p in ref}
    for a in pins.values():
        a.array.view(np.uint32)[...] = 0xDEADBEEF
    gpu.render_into(frames, {p: a.array for p, a in pins.items()})
    for p in ref:
        assert np.array_equal(pins[p].array.view(np.uint32), ref[p].view(np.uint32)), p
    # a subset of planes, half of them pinned: falls back to the plain path, same result
    mixed = {"rgba8": pins["rgba8"].array, "depth": np.empty((n, h, w), np.float32)}
    mixed["rgba8"][...] = 0
    gpu.render_into(frames, mixed)
    assert np.array_equal(mixed["rgba8"], ref["rgba8"]) and np.array_equal(mixed["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    for a in pins.values():
        a.close()
    gpu.close()
    host.close()
