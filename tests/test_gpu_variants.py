"""The two frame kernels round 1 measured and did not adopt -- ray refill (volly_refill.h,
VOLLY_REFILL=1) and budgeted rounds with parked rays (volly_requeue.h, VOLLY_REQUEUE=1) -- are not
part of the shipped libvolly_gpu.so any more; they live in the `experimental` build of
`make variants` (volume-renderer_amd/lib/variants/libvolly_gpu_experimental.so).  When that build
is present these tests keep them bit-exact against the oracle; otherwise they are skipped."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT, assert_planes_equal
from test_gpu_parity import _manual_frame, _oracle, _sizes

pytestmark = pytest.mark.gpu

EXPERIMENTAL = os.path.join(ROOT, "volume-renderer_amd", "lib", "variants", "libvolly_gpu_experimental.so")


@pytest.fixture(scope="module")
def vexp(vra):
    """volume-renderer_amd/gpu.py bound to the experimental build (a second module object)"""
    if not os.path.exists(EXPERIMENTAL):
        pytest.skip("the experimental variant is not built (make variants VARIANTS=experimental:-DVK_EXPERIMENTAL=1)")
    os.environ["VOLLY_GPU_LIB"] = EXPERIMENTAL
    try:
        spec = importlib.util.spec_from_file_location("volume_renderer_amd.gpu_experimental", os.path.join(ROOT, "volume-renderer_amd", "gpu.py"))
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "volume_renderer_amd"
        spec.loader.exec_module(mod)
    finally:
        os.environ.pop("VOLLY_GPU_LIB", None)
    return mod


def test_the_shipped_library_has_no_experimental_kernel(vra, teapot64, monkeypatch):
    monkeypatch.setenv("VOLLY_REFILL", "1")
    monkeypatch.setenv("VOLLY_REQUEUE", "1")
    gpu = vra.GpuScene.upload_dense(teapot64.target, teapot64.low)
    assert gpu.info(8) == 0
    gpu.close()


def test_ray_refill_kernel(vra, vexp, ora, teapot64, sphere64, monkeypatch):
    """The ray-refill frame kernel (opt-in, VOLLY_REFILL=1), on scenes where it meets every
    case the tile kernel meets: beam and naive
    frames, odd sizes, sub-rectangles, a batch, degenerate rays.  Same planes, bit for bit."""
    monkeypatch.setenv("VOLLY_REFILL", "1")
    for scene in (teapot64, sphere64):
        hi, lo = _sizes(scene)
        gpu = vexp.GpuScene.upload_dense(scene.target, scene.low)
        assert gpu.info(8) == 1
        frames = list(vra.orbit_frames(range(0, 16, 5), 16, hi, lo, 150, 77, radius=0.9))
        frames += list(vra.orbit_frames((3,), 16, hi, lo, 150, 77, radius=0.6, naive=True))
        got = gpu.render(frames)
        for i, f in enumerate(frames):
            assert_planes_equal({k: v[i] for k, v in got.items()}, _oracle(ora, f, scene, "dense"), f"refill frame {i}")
        sub = frames[1].copy()
        sub.x0, sub.y0, sub.tile_w, sub.tile_h = 33, 9, 70, 41
        assert_planes_equal({k: v[0] for k, v in gpu.render(sub).items()}, _oracle(ora, sub, scene, "dense"), "refill sub")
        for row0, org in (((1, 0, 0), (-5, 32.5, 32.5)), ((0, 0, 0), (10, 10, 10)), ((1, 0, 0), (-5, -2.0, 32.5)),
                          ((1, 1, 1), (-1, -1, -1))):
            f = _manual_frame(vra, 16, 8, row0, (0, 0.25, 0), (0, 0, 0.25), org)
            assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, _oracle(ora, f, scene, "dense"),
                                f"refill degenerate {row0} {org}")
        gpu.close()
    monkeypatch.setenv("VOLLY_REFILL", "0")
    gpu = vexp.GpuScene.upload_dense(teapot64.target, teapot64.low)
    assert gpu.info(8) == 0
    gpu.close()


@pytest.mark.parametrize("budget", ["1", "7", "40"])
def test_budgeted_rounds_with_parked_rays(vra, vexp, ora, teapot64, sphere64, monkeypatch, budget):
    """The frame kernels of volly_requeue.h (VOLLY_REQUEUE=1): rays are traced up to an iteration
    budget, parked, and continued by follow-up launches.  Tiny budgets force many rounds and
    rays parked at every stage of their life; the planes must equal the oracle's bit for bit,
    for beam and naive frames, odd sizes, sub-rectangles, a batch and degenerate rays."""
    monkeypatch.setenv("VOLLY_REQUEUE", "1")
    monkeypatch.setenv("VOLLY_REQUEUE_BUDGET", budget)
    for scene in (teapot64, sphere64):
        hi, lo = _sizes(scene)
        gpu = vexp.GpuScene.upload_dense(scene.target, scene.low)
        assert gpu.info(8) == 2
        frames = list(vra.orbit_frames(range(0, 16, 5), 16, hi, lo, 150, 77, radius=0.9))
        frames += list(vra.orbit_frames((3,), 16, hi, lo, 150, 77, radius=0.6, naive=True))
        got = gpu.render(frames)
        for i, f in enumerate(frames):
            assert_planes_equal({k: v[i] for k, v in got.items()}, _oracle(ora, f, scene, "dense"), f"requeue frame {i}")
        sub = frames[1].copy()
        sub.x0, sub.y0, sub.tile_w, sub.tile_h = 33, 9, 70, 41
        assert_planes_equal({k: v[0] for k, v in gpu.render(sub).items()}, _oracle(ora, sub, scene, "dense"), "requeue sub")
        for row0, org in (((1, 0, 0), (-5, 32.5, 32.5)), ((0, 0, 0), (10, 10, 10)), ((1, 0, 0), (-5, -2.0, 32.5)),
                          ((1, 1, 1), (-1, -1, -1))):
            f = _manual_frame(vra, 16, 8, row0, (0, 0.25, 0), (0, 0, 0.25), org)
            assert_planes_equal({k: v[0] for k, v in gpu.render(f).items()}, _oracle(ora, f, scene, "dense"),
                                f"requeue degenerate {row0} {org}")
        # raytrace frames take the tile kernel of the raytrace instantiation, as before
        pos, view = vra.orbit_camera(6, 16, 0.7)
        f = vra.frame_from_camera(view, pos, hi, lo, 64, 40, raytrace=True)
        assert_planes_equal({p: v[0] for p, v in gpu.render(f).items()}, ora.render_frame(f, target=scene.target), "requeue raytrace")
        gpu.close()
    monkeypatch.setenv("VOLLY_REQUEUE", "0")
    gpu = vexp.GpuScene.upload_dense(teapot64.target, teapot64.low)
    assert gpu.info(8) == 0
    gpu.close()


def test_pt_end_plane_on_the_experimental_kernels(vra, vexp, ora, teapot64, monkeypatch):
    ALL = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth", "pt_end")
    hi, lo = _sizes(teapot64)
    frames = list(vra.orbit_frames((1, 9), 16, hi, lo, 101, 67, radius=0.9)) + list(vra.orbit_frames((4,), 16, hi, lo, 101, 67, naive=True))
    for env in ({"VOLLY_REFILL": "1"}, {"VOLLY_REQUEUE": "1", "VOLLY_REQUEUE_BUDGET": "9"}):
        for k in ("VOLLY_REFILL", "VOLLY_REQUEUE", "VOLLY_REQUEUE_BUDGET"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu = vexp.GpuScene.upload_dense(teapot64.target, teapot64.low)
        assert gpu.info(8) != 0
        got = gpu.render(frames, planes=ALL)
        for i, f in enumerate(frames):
            want = ora.render_frame(f, low=teapot64.low, target=teapot64.target, planes=ALL)
            assert_planes_equal({k: v[i] for k, v in got.items()}, want, f"pt_end {env} frame {i}")
        gpu.close()
