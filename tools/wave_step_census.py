#!/usr/bin/env python3
"""CPU census of what the 64 rays of an 8x8 tile do together in the low-grid pass of the headline
workload (numpy, float32, the reference's step; statistics only -- not bit-exact, not product code).
For every wave-step it classifies the live lanes: E = in an empty 16^3 region, B = in an empty 4^3
brick of a region with voxels, F = in a brick with voxels; and asks which wave-level shortcuts
would apply.  Used to size loop designs before writing them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

def main():
    import importlib
    sys.modules.setdefault("torch", None) if False else None
    scene_name, res, W, H, radius = "dragon.off", 1024, 3840, 2160, 1.0
    if len(sys.argv) > 1 and sys.argv[1] == "c2":
        res, W, H = 512, 1920, 1080
    os.environ.setdefault("VOLLY_GPU_LIB", "")
    # host library only (no GPU here)
    import importlib.util
    pkg = os.path.join(ROOT, "volume-renderer_amd")
    spec = importlib.util.spec_from_file_location("vhost_abi", os.path.join(pkg, "_abi.py"))
    vra = None
    try:
        vra = load_package()
    except Exception as e:
        print("package import failed:", e); return
    sc = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", scene_name), res, dense=False, svo=False)
    low = vra.host.dense_view(sc.low)           # (z,y,x,8) ?
    n = res // 4
    occ = (np.asarray(low).reshape(n, n, n, 8)[..., 3] > 0)     # [z][y][x]
    brick = occ.reshape(n // 4, 4, n // 4, 4, n // 4, 4).any(axis=(1, 3, 5))
    region = brick.reshape(n // 16, 4, n // 16, 4, n // 16, 4).any(axis=(1, 3, 5))
    # octant skip distances: k[s][R] = largest k such that the cube of (k+1)^3 regions with R at its corner, extending
    # in octant s, is all empty and inside the grid (0 for a non-empty R is irrelevant: encoded as -1)
    nr = n // 16
    K = np.zeros((8, nr, nr, nr), np.int64)
    emp = ~region
    for s_ in range(8):
        sx, sy, sz = (1 if s_ & 1 else -1), (1 if s_ & 2 else -1), (1 if s_ & 4 else -1)
        # flip so that the octant is (+,+,+), then DP from the far corner
        e = emp[::sz, ::sy, ::sx]
        k = np.zeros(e.shape, np.int64) - 1
        for zz in range(nr - 1, -1, -1):
            for yy in range(nr - 1, -1, -1):
                for xx in range(nr - 1, -1, -1):
                    if not e[zz, yy, xx]:
                        continue
                    if zz == nr - 1 or yy == nr - 1 or xx == nr - 1:
                        k[zz, yy, xx] = 0
                    else:
                        k[zz, yy, xx] = 1 + min(k[zz + 1, yy, xx], k[zz, yy + 1, xx], k[zz, yy, xx + 1], k[zz + 1, yy + 1, xx], k[zz + 1, yy, xx + 1],
                                                k[zz, yy + 1, xx + 1], k[zz + 1, yy + 1, xx + 1])
        K[s_] = k[::sz, ::sy, ::sx]
    print("regions", nr ** 3, "empty", int(emp.sum()), "mean k over empty (per octant)", [round(float(K[s_][emp].mean()), 2) for s_ in range(8)])
    # adaptive empty cubes: level[R] = log2 (in regions) of the largest aligned all-empty cube of regions that holds R (0..3)
    lvl = np.zeros((nr, nr, nr), np.int64)
    for L in (1, 2, 3):
        b = 1 << L
        if nr % b:
            break
        e = emp.reshape(nr // b, b, nr // b, b, nr // b, b).all(axis=(1, 3, 5))
        lvl = np.where(np.repeat(np.repeat(np.repeat(e, b, 0), b, 1), b, 2) & emp, L, lvl)
    print("empty regions by cube level", [int(((lvl == L) & emp).sum()) for L in range(4)])
    cube = dict(old=0, new=0)
    KS = (4, 8, 16)
    deepc = {k: dict(cov=0, tests=0, fails=0) for k in KS}
    blk = dict(covered=0, blocks=0, evals=0)
    rng = np.random.default_rng(1)
    tot = dict(steps=0, quiet=0, allE=0, deep4=0, deep2=0, lanes=0, live=0)
    greedy4 = 0; tests4 = 0; fails4 = 0
    for view in (0, 1000, 2100, 3300):
        f = vra.orbit_frames([view], 4096, (res,) * 3, (n,) * 3, W, H, radius=radius)[0]
        row0 = np.array(f.row0_lo[:3], np.float32); xd = np.array(f.xdiff_lo[:3], np.float32); yd = np.array(f.ydiff_lo[:3], np.float32)
        org = np.array(f.origin_lo[:3], np.float32)
        tx = rng.integers(0, W // 8, 6000); ty = rng.integers(0, H // 8, 6000)
        lx = np.arange(64) % 8; ly = np.arange(64) // 8
        px = (tx[:, None] * 8 + lx[None, :]).astype(np.float32); py = (ty[:, None] * 8 + ly[None, :]).astype(np.float32)
        d = row0[None, None, :] + px[..., None] * xd + py[..., None] * yd
        d = (d / np.sqrt((d * d).sum(-1, keepdims=True))).astype(np.float32)
        inv = (1.0 / d).astype(np.float32); h = (d >= 0).astype(np.float32); e = np.copysign(np.float32(3e-4), d)
        size = np.float32(n)
        nd = (-(size * (h - 1) + org)) * inv
        t0 = np.maximum(nd.max(-1), 0).astype(np.float32)
        cur = (t0[..., None] * d + org).astype(np.float32)
        live = np.ones(cur.shape[:2], bool)
        prev_reg = np.full(cur.shape[:2], -1, np.int64); prev_brk = np.full(cur.shape[:2], -1, np.int64)
        octant = (d[..., 0] > 0).astype(np.int64) | ((d[..., 1] > 0).astype(np.int64) << 1) | ((d[..., 2] > 0).astype(np.int64) << 2)
        w_shift = np.full(cur.shape[:2], 4, np.int64); w_key = np.full(cur.shape[:2], -1, np.int64)
        w_valid = np.zeros(cur.shape[:2], bool)
        d_left = {k: np.zeros(cur.shape[0], np.int64) for k in KS}; d_try = {k: np.ones(cur.shape[0], bool) for k in KS}
        free_until = np.zeros(cur.shape[:2], np.int64)
        skip_left = np.zeros(cur.shape[0], np.int64)
        MINBLOCK = int(os.environ.get("MINBLOCK", "8"))
        run_left = np.zeros(cur.shape[0], np.int64)     # remaining unchecked steps of a greedy 4-block per wave
        tried_ok = np.ones(cur.shape[0], bool)
        for it in range(4 * 3 * n + 8):
            p = cur + e
            c = p.astype(np.int32)                      # truncation
            inside = ((p > -1) & (p < size)).all(-1)
            live &= inside
            if not live.any():
                break
            x, y, z = c[..., 0].clip(0, n - 1), c[..., 1].clip(0, n - 1), c[..., 2].clip(0, n - 1)
            reg_id = (x >> 4) + (n >> 4) * ((y >> 4) + (n >> 4) * (z >> 4)); brk_id = (x >> 2) + (n >> 2) * ((y >> 2) + (n >> 2) * (z >> 2))
            rfull = region[z >> 4, y >> 4, x >> 4]; bfull = brick[z >> 2, y >> 2, x >> 2]
            hit = occ[z, y, x] & live
            E = ~rfull; B = rfull & ~bfull
            quiet_lane = (E & (reg_id == prev_reg)) | (B & (brk_id == prev_brk))
            wl = live.any(1)
            wq = (quiet_lane | ~live).all(1) & wl
            allE = (E | ~live).all(1) & wl
            def deep(k):
                ok = np.ones(live.shape, bool)
                for a, v in enumerate((x, y, z)):
                    lo_d = v & 15
                    ok &= np.where(d[..., a] > 0, (15 - lo_d) >= k, lo_d >= k)
                return ((ok & E) | ~live).all(1) & wl
            d4 = deep(3); d2 = deep(1)
            tot["steps"] += int(wl.sum()); tot["quiet"] += int(wq.sum()); tot["allE"] += int(allE.sum())
            tot["deep4"] += int(d4.sum()); tot["deep2"] += int(d2.sum()); tot["live"] += int(live.sum()); tot["lanes"] += int(wl.sum()) * 64
            # greedy 4-blocks: policy "test when no block is running and (the last test passed or this wave-step is all-quiet)"
            in_run = run_left > 0
            can_try = ~in_run & wl & (tried_ok | wq)
            passed = can_try & d4
            tests4 += int(can_try.sum()); fails4 += int((can_try & ~d4).sum())
            tried_ok = np.where(can_try, passed, tried_ok)
            run_left = np.where(passed, 4, run_left)
            greedy4 += int((run_left > 0).sum())
            run_left = np.maximum(run_left - 1, 0)
            # adaptive cubes: a lane in an empty region watches the largest empty aligned cube around it
            def ckey(sh):
                return (x >> sh) | ((y >> sh) << 10) | ((z >> sh) << 20)
            k_now = np.where(w_shift == 4, ckey(4), np.where(w_shift == 5, ckey(5), np.where(w_shift == 6, ckey(6), ckey(7))))
            left = live & (k_now != w_key)
            cube["new"] += int((left.any(1) & wl).sum())
            cube["old"] += int(((live & (reg_id != prev_reg)).any(1) & wl).sum())
            nl = 4 + np.where(E, lvl[z >> 4, y >> 4, x >> 4], 0)
            w_shift = np.where(left, nl, w_shift)
            k_new = np.where(w_shift == 4, ckey(4), np.where(w_shift == 5, ckey(5), np.where(w_shift == 6, ckey(6), ckey(7))))
            w_key = np.where(left, k_new, w_key)
            w_valid = np.where(left, E, w_valid)          # the watch is a cube only in an empty region
            # unchecked blocks of K steps: all live lanes stay inside their watch cube for K more steps
            sgn = np.where(d > 0, 1, -1)
            for K_ in KS:
                ok = w_valid.copy()
                xs = (x + sgn[..., 0] * K_, y + sgn[..., 1] * K_, z + sgn[..., 2] * K_)
                k_far = sum(((v >> w_shift) & 1023) << (10 * a) for a, v in enumerate(xs))
                inb = (xs[0] >= 0) & (xs[0] < n) & (xs[1] >= 0) & (xs[1] < n) & (xs[2] >= 0) & (xs[2] < n)
                ok &= inb & (k_far == w_key)
                wave_ok = (ok | ~live).all(1) & wl
                running = d_left[K_] > 0
                attempt = ~running & wl & d_try[K_] & ~left.any(1)
                deepc[K_]["tests"] += int(attempt.sum()); deepc[K_]["fails"] += int((attempt & ~wave_ok).sum())
                startb = attempt & wave_ok
                d_try[K_] = np.where(attempt, wave_ok, d_try[K_]) | left.any(1)
                d_left[K_] = np.where(startb, K_, d_left[K_])
                deepc[K_]["cov"] += int(((d_left[K_] > 0) & wl).sum())
                d_left[K_] = np.maximum(d_left[K_] - 1, 0)
            # skip blocks: a lane entering a region learns for how many steps it is surely in empty space inside the grid
            changed = live & (reg_id != prev_reg)
            kk = K[octant, z >> 4, y >> 4, x >> 4]
            free_until = np.where(changed, np.where(kk >= 1, it + 16 * kk, 0), free_until)
            in_skip = skip_left > 0
            ev = changed.any(1) & ~in_skip & wl
            T = np.where(live, free_until, 1 << 40).min(1) - it
            start = ev & (T >= MINBLOCK)
            blk["evals"] += int(ev.sum()); blk["blocks"] += int(start.sum())
            skip_left = np.where(start, T, skip_left)
            blk["covered"] += int(((skip_left > 0) & wl).sum())
            skip_left = np.maximum(skip_left - 1, 0)
            prev_reg = np.where(live, reg_id, prev_reg); prev_brk = np.where(live, brk_id, prev_brk)
            live &= ~hit
            fl = np.floor(p)
            ndv = ((fl + h) - cur) * inv
            m = ndv.min(-1)
            cur = np.where(live[..., None], (m[..., None] * d + cur), cur).astype(np.float32)
    s = tot["steps"]
    print(f"wave-steps {s}  all-quiet {tot['quiet']/s:.3f}  all-in-empty-regions {tot['allE']/s:.3f}  deep>=3 {tot['deep4']/s:.3f}  deep>=1 {tot['deep2']/s:.3f}  live lanes/step {tot['live']/s:.1f}")
    print(f"octant skip blocks (min {os.environ.get('MINBLOCK', '8')}) cover {blk['covered']/s:.3f} of the wave-steps in {blk['blocks']} blocks (mean {blk['covered']/max(1,blk['blocks']):.1f} steps), wave-min evaluations {blk['evals']/s:.3f} per wave-step")
    print(f"wave-steps with a region change: {cube['old']/s:.3f}; with adaptive empty cubes (32^3..128^3): {cube['new']/s:.3f}")
    for K_ in KS:
        print(f"unchecked blocks of {K_} steps inside the watch cubes cover {deepc[K_]['cov']/s:.3f} of the wave-steps; tests {deepc[K_]['tests']/s:.3f}, failed {deepc[K_]['fails']/s:.3f} per wave-step")
    print(f"greedy 4-blocks cover {greedy4/s:.3f} of the wave-steps; deep tests {tests4/s:.3f} per wave-step, failed {fails4/s:.3f} per wave-step")

main()
