"""CPU soak of the sky rectangle and spans (volly_gpu_sky_rect_for, volly_gpu_sky_spans_for) against the oracle: random grids, cameras from 0.05 to 40 volume\nsizes away, looking at and past the volume; no pixel outside the rectangle, or outside its tile row's span, may take a step.  python profiles/sky_rect_soak.py"""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import oracle_lib as ora
from test_sky_rect import _grid, _low_of
MISS = 0xFFFFFFFF
vra = load_package()
rng = np.random.default_rng(2024)
bad_total = cases = culled = total = span_culled = 0
for s in range(30):
    shape = tuple(int(rng.integers(4, 70)) for _ in range(3))
    vox = _grid(rng, shape, 30)
    fdiv = int(rng.choice([2, 4, 8]))
    low = _low_of(vox, fdiv)
    scene = vra.HostScene.from_dense(vox, low, svo=False)
    hi = (shape[2], shape[1], shape[0]); lo = (low.shape[2], low.shape[1], low.shape[0])
    for case in range(100):
        naive = rng.random() < 0.3
        pos = (0.5 + (rng.random(3) - 0.5) * rng.uniform(0.05, 40.0)).astype(np.float32)
        at = rng.random(3).astype(np.float32) if rng.random() < 0.7 else (pos + rng.normal(size=3)).astype(np.float32)
        up = (0, 0, -1) if abs((at - pos)[2]) < 0.9 * np.linalg.norm(at - pos) else (0, 1, 0)
        view = vra.look_at(pos, 2 * pos - at, up)
        w, h = int(rng.integers(4, 100)), int(rng.integers(4, 80))
        f = vra.frame_from_camera(view, pos, hi, lo, w, h, naive=naive)
        r = vra.gpu.sky_rect_for(hi, lo, f)
        want = ora.render_frame(f, low=scene.low, target=scene.target)
        inside = np.zeros((h, w), bool)
        if r[0] <= r[1] and r[2] <= r[3]:
            inside[r[2]:r[3] + 1, r[0]:r[1] + 1] = True
        out = ~inside
        bad = out & ((want["hit_index"] != MISS) | (want["steps_lo"] != 0) | (want["steps_hi"] != 0))
        if bad.any():
            bad_total += 1; print("BAD", s, case, shape, pos, r)
        cases += 1; culled += int(out.sum()); total += out.size
        taken = (want["hit_index"] != MISS) | (want["steps_lo"] != 0) | (want["steps_hi"] != 0)
        for bh in (8, 2):                                   # the spans the frame kernel culls with (rectangle cut by hull edges)
            spans, hull = vra.gpu.sky_spans_for(hi, lo, f, bh)
            keep = np.zeros((h, w), bool)
            for rr, (a, b) in enumerate(spans):
                if a <= b:
                    keep[rr * bh:(rr + 1) * bh, a:b + 1] = True
            if (~keep & taken).any():
                bad_total += 1; print("BAD SPAN", s, case, shape, pos, bh, hull.tolist())
            if bh == 8:
                span_culled += int((~keep).sum())
    scene.close()
print("cases", cases, "bad", bad_total, "culled fraction: rectangle", culled / total, "spans (8-row tiles)", span_culled / total)
