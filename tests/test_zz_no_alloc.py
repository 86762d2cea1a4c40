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


def _both(vra, torch, gpu, frames):
    """render `frames` through both entry points; returns (existing, noalloc) host planes"""
    n, h, w = len(frames), frames[0].tile_h, frames[0].tile_w
    a, b = _device_planes(torch, n, h, w), _device_planes(torch, n, h, w)
    stream = torch.cuda.current_stream().cuda_stream
    gpu.render_device(frames, {p: t.data_ptr() for p, t in a.items()}, stream=stream)
    gpu.render_device_noalloc(frames, {p: t.data_ptr() for p, t in b.items()}, stream=stream)
    torch.cuda.synchronize()
    return _host(a), _host(b)


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
