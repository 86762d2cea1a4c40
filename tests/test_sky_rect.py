"""The sky rectangle (volly_gpu_sky_rect_for: host arithmetic of libvolly_gpu.so, no GPU needed) against the
CPU oracle: for every pixel outside it the oracle must report zero steps in both passes and a miss -- the
values culled tiles write without tracing their rays -- for cameras all around, near, far, inside, for
sub-rectangles, non-cubic grids and grids whose low grid is not the target's quarter; and the rectangle
must actually exclude sky when the volume is small in the frame."""
import numpy as np
import pytest

from conftest import MISS


def _grid(rng, shape, n):
    v = np.zeros(shape + (8,), np.uint8)
    z, y, x = (rng.integers(0, s, n) for s in shape)
    v[z, y, x, 0:4] = 255
    v[z, y, x, 4:7] = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    v[0, 0, 0, 0:4] = 255; v[-1, -1, -1, 0:4] = 255        # the corners: the box's silhouette is really there
    v[0, -1, 0, 0:4] = 255; v[-1, 0, -1, 0:4] = 255; v[0, 0, -1, 0:4] = 255; v[-1, -1, 0, 0:4] = 255; v[0, -1, -1, 0:4] = 255; v[-1, 0, 0, 0:4] = 255
    return v


def _low_of(v, f):
    sz, sy, sx = v.shape[:3]
    lz, ly, lx = -(-sz // f), -(-sy // f), -(-sx // f)
    pad = np.zeros((lz * f, ly * f, lx * f), bool)
    pad[:sz, :sy, :sx] = v[..., 3] > 0
    low = np.zeros((lz, ly, lx, 8), np.uint8)
    low[pad.reshape(lz, f, ly, f, lx, f).any(axis=(1, 3, 5)), 0:4] = 255
    return low


@pytest.mark.parametrize("seed", range(6))
def test_pixels_outside_the_sky_rectangle_are_sky(vra, ora, seed):
    rng = np.random.default_rng(100 + seed)
    shape = [(32, 32, 32), (16, 48, 32), (20, 33, 47), (64, 64, 64), (8, 8, 8), (32, 16, 64)][seed]
    vox = _grid(rng, shape, 40)
    low = _low_of(vox, [4, 4, 8, 4, 2, 4][seed])
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    hi = (shape[2], shape[1], shape[0]); lo = (low.shape[2], low.shape[1], low.shape[0])
    culled = total = trivial = 0
    for case in range(40):
        naive = case % 3 == 0
        pos = (0.5 + (rng.random(3) - 0.5) * rng.uniform(0.1, 8.0)).astype(np.float32)
        at = rng.random(3).astype(np.float32)
        if case % 7 == 3:                                   # looking past the volume
            at = (pos + rng.normal(size=3)).astype(np.float32)
        if np.allclose(at, pos):
            at = pos + np.float32(0.3)
        up = (0, 0, -1) if abs((at - pos)[2]) < 0.9 * np.linalg.norm(at - pos) else (0, 1, 0)
        view = vra.look_at(pos, 2 * pos - at, up)           # the reference looks along -camDir
        w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
        f = vra.frame_from_camera(view, pos, hi, lo, w, h, naive=naive)
        if case % 4 == 1 and w > 20 and h > 20:
            f.x0, f.y0, f.tile_w, f.tile_h = 3, 6, w - 8, h - 13
        x_lo, x_hi, y_lo, y_hi = vra.gpu.sky_rect_for(hi, lo, f)
        want = ora.render_frame(f, low=scene.low, target=scene.target)
        inside = np.zeros((f.tile_h, f.tile_w), bool)
        if x_lo <= x_hi and y_lo <= y_hi:
            assert 0 <= x_lo and x_hi < f.tile_w and 0 <= y_lo and y_hi < f.tile_h
            inside[y_lo:y_hi + 1, x_lo:x_hi + 1] = True
        out = ~inside
        bad = out & ((want["hit_index"] != MISS) | (want["steps_lo"] != 0) | (want["steps_hi"] != 0))
        assert not bad.any(), f"seed {seed} case {case}: {int(bad.sum())} pixels outside {x_lo, x_hi, y_lo, y_hi} are not sky (camera {pos})"
        assert np.isinf(want["depth"][out]).all() and (want["rgba8"][out] == 255).all()
        culled += int(out.sum()); total += out.size
        trivial += (x_lo, x_hi, y_lo, y_hi) == (0, f.tile_w - 1, 0, f.tile_h - 1)
    assert culled > 0.15 * total, (culled, total)
    assert 0 < trivial < 40                                  # cameras inside / close: no cull; far ones: cull
    scene.close()


def test_sky_rectangle_switches_itself_off(vra):
    hi, lo = (64, 64, 64), (16, 16, 16)
    pos, view = vra.orbit_camera(3, 16, 3.0)
    whole = (0, 95, 0, 63)
    f = vra.frame_from_camera(view, pos, hi, lo, 96, 64)
    assert vra.gpu.sky_rect_for(hi, lo, f) != whole
    assert vra.gpu.sky_rect_for(hi, lo, vra.frame_from_camera(view, pos, hi, lo, 96, 64, raytrace=True)) == whole
    inside = vra.frame_from_camera(view, np.array([0.5, 0.4, 0.6], np.float32), hi, lo, 96, 64)
    assert vra.gpu.sky_rect_for(hi, lo, inside) == whole
    g = f.copy(); g.xdiff_hi[1] = g.xdiff_hi[1] + 0.002     # target constants that no longer describe the low rays
    assert vra.gpu.sky_rect_for(hi, lo, g) == whole
    g = f.copy(); g.row0_lo[0] = float("nan")
    assert vra.gpu.sky_rect_for(hi, lo, g) == whole
    g = f.copy(); g.xdiff_lo[0] = g.xdiff_lo[1] = g.xdiff_lo[2] = 0.0      # degenerate: no projection
    assert vra.gpu.sky_rect_for(hi, lo, g) == whole
    n = vra.frame_from_camera(view, pos, hi, lo, 96, 64, naive=True)
    assert vra.gpu.sky_rect_for(hi, None, n) != whole        # naive frames need no low grid
    with pytest.raises(vra.gpu.VollyError):
        vra.gpu.sky_rect_for(hi, None, f)
