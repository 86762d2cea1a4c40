#!/usr/bin/env python3
"""Scene preparation, host pipeline vs volume_gpu_voxelize, on the bench scenes.
    python profiles/vox_bench.py [--check] mesh:res ...
Prints per scene: host seconds (threads used by the host voxelizer = min(cores, 16)), device
milliseconds (whole call, incl. mesh upload and octree build; best of 3), statistics of the
search, and -- with --check -- whether octree and low grid equal the host's byte for byte."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--skip-host", action="store_true")
    ap.add_argument("scenes", nargs="*", default=["dragon.off:512", "stratocaster.ply:1024", "galleon.ply:2048"])
    a = ap.parse_args()
    vra = load_package()
    for sc in a.scenes:
        mesh, res = sc.split(":")
        res = int(res)
        path = os.path.join(ROOT, "assets", mesh)
        host = None
        th = float("nan")
        if not a.skip_host:
            t = time.time()
            host = vra.HostScene.from_mesh(path, res, dense=False, svo=True)
            th = time.time() - t
        t = time.time()
        hm = vra.HostMesh(path, 2)
        tm = time.time() - t
        best, stats = None, None
        for _ in range(3):
            t = time.time()
            g = vra.GpuScene.voxelize(hm.desc, res, dense=False, svo=True)
            dt = time.time() - t
            if best is None or dt < best:
                best, stats = dt, g.vox_stats
            last = g
            if _ < 2:
                g.close()
        line = (f"{mesh} {res}^3: host {th:.2f} s | mesh load+normalise {tm * 1e3:.1f} ms | device {best * 1e3:.1f} ms "
                f"(x{th / best:.0f}) | triangles {stats['triangles']} tests {stats['tests']} pairs {stats['pairs']} "
                f"levels {stats['levels']} cells {stats['cells']} scratch {stats['scratch_bytes'] / 2**20:.0f} MiB "
                f"| nodes {last.info(10)} leaves {last.info(11)}")
        if a.check and host is not None:
            nh, lh = host.svo_arrays()
            ng, lg = last.read_svo()
            same = (np.array_equal(ng.view(np.uint64), nh.view(np.uint64)) and np.array_equal(lg, lh)
                    and np.array_equal(last.read_low_voxels(), vra.host.dense_view(host.low).reshape(-1, 8)))
            line += f" | equal to host: {same}"
        print(line, flush=True)
        last.close()
        hm.close()
        if host is not None:
            host.close()


if __name__ == "__main__":
    main()
