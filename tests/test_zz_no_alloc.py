"""volume_gpu_render_device_noalloc (include/volly_gpu.h): the entry point for callers that
cannot allocate, called through the C-ABI and compared with volume_gpu_render_device -- the
entry point it is a variant of -- bit for bit on every plane, on two calls in a row, the second
with a smaller input (the static scratch keeps the first call's contents).  Also against the CPU
oracle, and the size limit's error.  Collected last (test_zz_*): the round's end runs pytest -x."""
import numpy as np
import pytest

from conftest import assert_planes_equal

pytestmark = pytest.mark.gpu

PLANES = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth", "pt_end")


def _device_planes(torch, n, h, w):
    out = {}
    for p in PLANES:
        shape = (n, h, w, 3) if p == "pt_end" else (n, h, w)
        out[p] = torch.full(shape, -7, dtype=torch.float32 if p in ("depth", "pt_end") else torch.int32, device="cuda:0")
    return out


def _host(planes):
    res = {}
    for p, t in planes.items():
        a = t.cpu().numpy()
        res[p] = a if p in ("depth", "pt_end") else a.view(np.uint32)
    return res


def _both(vra, torch, gpu, frames, planes=PLANES):
    """render `frames` through both entry points; returns (existing, noalloc) host planes"""
    n, h, w = len(frames), frames[0].tile_h, frames[0].tile_w
    a, b = _device_planes(torch, n, h, w), _device_planes(torch, n, h, w)
    stream = torch.cuda.current_stream().cuda_stream
    gpu.render_device(frames, {p: t.data_ptr() for p, t in a.items() if p in planes}, stream=stream)
    gpu.render_device_noalloc(frames, {p: t.data_ptr() for p, t in b.items() if p in planes}, stream=stream)
    torch.cuda.synchronize()
    return ({p: v for p, v in _host(a).items() if p in planes}, {p: v for p, v in _host(b).items() if p in planes})


@pytest.mark.parametrize("kind", ["svo", "dense"])
def test_noalloc_equals_render_device_twice(vra, ora, teapot64, kind):
    import torch
    hi, lo = (64, 64, 64), (16, 16, 16)
    gpu = (vra.GpuScene.upload_svo(teapot64.svo, teapot64.low) if kind == "svo"
           else vra.GpuScene.upload_dense(teapot64.target, teapot64.low))
    # first call: 21 frames (more than one launch of 16), odd size; second call: fewer, smaller frames
    for views, (w, h), naive in ((range(21), (203, 117), False), (range(3, 8), (64, 40), True), (range(2), (203, 117), False)):
        frames = vra.orbit_frames(list(views), 24, hi, lo, w, h, radius=1.0, naive=naive)
        want, got = _both(vra, torch, gpu, frames)
        for i in range(len(frames)):
            assert_planes_equal({k: v[i] for k, v in got.items()}, {k: v[i] for k, v in want.items()},
                                f"{kind} noalloc vs render_device, {w}x{h} frame {i}")
        kw = dict(low=teapot64.low, svo=teapot64.svo) if kind == "svo" else dict(low=teapot64.low, target=teapot64.target)
        ref = ora.render_frame(frames[-1], **kw)
        assert_planes_equal({k: v[-1] for k, v in got.items() if k != "pt_end"}, ref, f"{kind} noalloc vs oracle")
        assert int((ref["hit_index"] != 0xFFFFFFFF).sum()) > 50
    gpu.close()


def test_noalloc_sub_rectangle_and_limit(vra, ora, teapot64):
    import torch
    hi, lo = (64, 64, 64), (16, 16, 16)
    gpu = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, hi, lo, 200, 96)
    sub = full.copy()
    sub.x0, sub.y0, sub.tile_w, sub.tile_h = 40, 13, 91, 50
    want, got = _both(vra, torch, gpu, [sub])
    assert_planes_equal({k: v[0] for k, v in got.items()}, {k: v[0] for k, v in want.items()}, "sub-rectangle")
    # beyond the static scratch: 6 * ceil(w / 8) * h floats > 32 MiB is rejected, nothing is launched
    big = vra.frame_from_camera(view, pos, hi, lo, 7680, 4320)
    with pytest.raises(vra.gpu.VollyError) as e:
        gpu.render_device_noalloc([big], {"rgba8": 0})
    assert e.value.code == -1 and "static scratch" in str(e.value)
    # a row band of the same frame that fits is accepted (rendered into a real buffer)
    band = big.band(0, 64)
    out = torch.empty((1, 64, 7680), dtype=torch.int32, device="cuda:0")
    ref = torch.empty_like(out)
    gpu.render_device_noalloc([band], {"rgba8": out.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
    gpu.render_device([band], {"rgba8": ref.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    gpu.close()


@pytest.mark.parametrize("kind", ["svo", "dense"])
def test_noalloc_five_planes_with_the_sky_cull_and_no_allocation(vra, ora, teapot64, kind):
    """Without pt_end the sky cull is active: the spans part of the static scratch and the span-limited checkpoint
    stores are in use.  21 frames (two launches), beam and naive, against volume_gpu_render_device and the oracle;
    and the calls allocate nothing: free device memory and the scene's own byte count stay what they were."""
    import torch
    five = PLANES[:5]
    hi, lo = (64, 64, 64), (16, 16, 16)
    gpu = (vra.GpuScene.upload_svo(teapot64.svo, teapot64.low) if kind == "svo"
           else vra.GpuScene.upload_dense(teapot64.target, teapot64.low))
    kw = dict(low=teapot64.low, svo=teapot64.svo) if kind == "svo" else dict(low=teapot64.low, target=teapot64.target)
    for naive in (False, True):
        frames = vra.orbit_frames(list(range(21)), 24, hi, lo, 203, 117, radius=1.6, naive=naive)     # far enough for sky tiles
        rect = gpu.sky_rect(frames[0])
        assert rect[0] > 8 or rect[1] < 203 - 9 or rect[2] > 8 or rect[3] < 117 - 9                   # the cull is not vacuous
        want, got = _both(vra, torch, gpu, frames, planes=five)
        for i in range(len(frames)):
            assert_planes_equal({k: v[i] for k, v in got.items()}, {k: v[i] for k, v in want.items()},
                                f"{kind} noalloc vs render_device, five planes, naive={naive}, frame {i}")
        for i in (0, 13, 20):
            assert_planes_equal({k: v[i] for k, v in got.items()}, ora.render_frame(frames[i], **kw), f"{kind} noalloc vs oracle, frame {i}")
    # nothing is allocated by the calls themselves
    frames = vra.orbit_frames(list(range(21)), 24, hi, lo, 203, 117, radius=1.6)
    out = _device_planes(torch, 21, 117, 203)
    ptrs = {p: t.data_ptr() for p, t in out.items() if p in five}
    stream = torch.cuda.current_stream().cuda_stream
    gpu.render_device_noalloc(frames, ptrs, stream=stream)
    torch.cuda.synchronize()
    free0, held0 = torch.cuda.mem_get_info()[0], gpu.info(7)
    for _ in range(5):
        gpu.render_device_noalloc(frames, ptrs, stream=stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0 and gpu.info(7) == held0
    gpu.close()


def test_noalloc_call_under_hip_graph_capture(vra, ora, teapot64):
    """include/volly_gpu.h names code under HIP graph capture among the callers this entry point is for: capture one
    call (pre-pass + frame kernel, the frame descriptors travel as kernel arguments) into a graph, replay it into
    cleared planes, and compare with the directly rendered planes and the oracle."""
    import torch
    five = PLANES[:5]
    hi, lo = (64, 64, 64), (16, 16, 16)
    gpu = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    frames = vra.orbit_frames([1, 5, 9], 24, hi, lo, 160, 96, radius=1.0)
    direct, replayed = _device_planes(torch, 3, 96, 160), _device_planes(torch, 3, 96, 160)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gpu.render_device_noalloc(frames, {p: t.data_ptr() for p, t in direct.items() if p in five}, stream=side.cuda_stream)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gpu.render_device_noalloc(frames, {p: t.data_ptr() for p, t in replayed.items() if p in five}, stream=side.cuda_stream)
    torch.cuda.synchronize()
    for p in five:
        assert int((replayed[p] != -7).sum()) == 0          # capture recorded the work, it did not run it
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    a = {p: v for p, v in _host(direct).items() if p in five}
    b = {p: v for p, v in _host(replayed).items() if p in five}
    for i in range(3):
        assert_planes_equal({k: v[i] for k, v in b.items()}, {k: v[i] for k, v in a.items()}, f"graph replay vs direct call, frame {i}")
    assert_planes_equal({k: v[2] for k, v in b.items()}, ora.render_frame(frames[2], low=teapot64.low, svo=teapot64.svo), "graph replay vs oracle")
    gpu.close()


def test_noalloc_on_a_scene_of_the_packed_kernel(vra, ora):
    """A dense scene without the region level (noise) renders through vk::render_frames_packed; the ordinary entry point
    parks rays in a queue it allocates between passes, the one that allocates nothing runs one pass to the end: same planes."""
    import torch
    gpu = vra.GpuScene.noise(64, 5, 1 << 25)
    assert gpu.info(8) == 3
    frames = vra.orbit_frames([0, 3, 7, 11, 13], 24, (64,) * 3, (16,) * 3, 203, 117, radius=1.1)
    for naive in (False, True):
        for f in frames:
            f.flags = 1 if naive else 0
        want, got = _both(vra, torch, gpu, frames)
        for i in range(len(frames)):
            assert_planes_equal({k: v[i] for k, v in got.items()}, {k: v[i] for k, v in want.items()}, f"packed scene, noalloc vs render_device, naive={naive}, frame {i}")
        assert int((want["hit_index"] != 0xFFFFFFFF).sum()) > 1000 and int(want["steps_hi"].max()) > 100     # rays long enough to be parked
    gpu.close()
