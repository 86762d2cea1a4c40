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
    culled = total = trivial = hull_culled = rect_area = 0
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
        # the spans the frame kernel culls with (rectangle cut by the hull's slanted edges), per tile row
        for bh in (8, 2):
            spans, hull = vra.gpu.sky_spans_for(hi, lo, f, bh)
            assert spans.shape == ((f.tile_h + bh - 1) // bh, 2)
            keep = np.zeros((f.tile_h, f.tile_w), bool)
            for r, (a, b) in enumerate(spans):
                if a <= b:
                    assert x_lo <= a and b <= x_hi, (spans[r], x_lo, x_hi)
                    keep[r * bh:(r + 1) * bh, a:b + 1] = True
            bad = ~keep & ((want["hit_index"] != MISS) | (want["steps_lo"] != 0) | (want["steps_hi"] != 0))
            assert not bad.any(), f"seed {seed} case {case} block_h {bh}: {int(bad.sum())} pixels outside their span are not sky (camera {pos}, hull {hull.tolist()})"
            if bh == 8:
                hull_culled += int((inside & ~keep).sum()); rect_area += int(inside.sum())
    assert culled > 0.15 * total, (culled, total)
    assert hull_culled > 0, (hull_culled, rect_area)          # the slanted edges cut something off even these small rectangles (6 px of margins)
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


def test_hull_edges_cut_the_rectangle_on_oblique_views(vra):
    """At frame sizes where the margins are small, an oblique view's spans hold clearly fewer tiles than its rectangle."""
    hi, lo = (1024,) * 3, (256,) * 3
    f = vra.orbit_frames([300], 4096, hi, lo, 960, 540, radius=1.0)[0]
    x_lo, x_hi, y_lo, y_hi = vra.gpu.sky_rect_for(hi, lo, f)
    spans, hull = vra.gpu.sky_spans_for(hi, lo, f, 8)
    rect_tiles = (x_hi // 8 - x_lo // 8 + 1) * (y_hi // 8 - y_lo // 8 + 1)
    span_tiles = sum(b // 8 - a // 8 + 1 for a, b in spans if a <= b)
    assert (hull[:, 2] != -1).sum() >= 3                    # slanted edges were found
    assert span_tiles < 0.85 * rect_tiles, (span_tiles, rect_tiles)
