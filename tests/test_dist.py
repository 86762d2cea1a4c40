"""The N>1 paths on the CPU with the gloo backend, world_size 2:
  * single-frame split: each rank renders its row band (with the oracle standing in
    for the GPU: same C-ABI frame descriptors, same sub-rectangle semantics), the
    bands are gathered to rank 0 and must equal the full frame;
  * the same with stripes of rows dealt round-robin (the default of bench.py's split: balanced load), including a
    shorter last stripe and a rank left without one;
  * view-parallel: the round-robin partition covers every view exactly once."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, ASSETS


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package, load_oracle
    vra, ora = load_package(), load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64, threads=2)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, 96, height)
    band = vra.shard.band_frame(full, rank, world)
    y0, rows = vra.shard.band_rows(height, rank, world)
    assert (band.y0, band.tile_h, band.tile_w, band.x0) == (y0, rows, 96, 0)
    got = ora.render_frame(band, low=scene.low, svo=scene.svo, threads=1)
    frames = {}
    for p in ("rgba8", "hit_index", "depth"):
        t = torch.from_numpy(got[p].view(np.int32).copy())
        whole = vra.shard.gather_bands(t, height, 96, rank, world, dist, dst=0)
        if rank == 0:
            frames[p] = whole.numpy().view(np.uint32)
        else:
            assert whole is None
    if rank == 0:
        want = ora.render_frame(full, low=scene.low, svo=scene.svo, threads=1)
        ok = all(np.array_equal(frames[p], want[p].view(np.uint32)) for p in frames)
        hits = int((want["hit_index"] != 0xFFFFFFFF).sum())
        open(out_path, "w").write(f"{int(ok)} {hits}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [64, 37])          # even split, and bands that differ by one row
def test_row_band_split_and_gather_gloo(tmp_path, height):
    import torch.multiprocessing as mp
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), height, str(out)), nprocs=2, join=True)
    ok, hits = out.read_text().split()
    assert ok == "1" and int(hits) > 100


def _stripe_worker(rank, world, port, height, stripe, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package, load_oracle
    vra, ora = load_package(), load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64, threads=2)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, 96, height)
    whole_stripes, tail = vra.shard.stripe_frames(full, rank, world, stripe)
    mine = vra.shard.stripe_rows(height, rank, world, stripe)
    assert len(whole_stripes) + (tail is not None) == len(mine)
    ok = True
    for p in ("rgba8", "hit_index", "depth"):
        g = vra.shard.StripeGatherer(height, 96, rank, world, torch.int32, "cpu", stripe)
        for i, f in enumerate(list(whole_stripes) + ([tail] if tail is not None else [])):      # slot i = the rank's i-th stripe
            assert (f.y0, f.tile_h) == mine[i]
            got = ora.render_frame(f, low=scene.low, svo=scene.svo, threads=1)
            g.send[i, :f.tile_h] = torch.from_numpy(got[p].view(np.int32).copy())
        frame = g.gather(dist)
        if rank == 0:
            want = ora.render_frame(full, low=scene.low, svo=scene.svo, threads=1)
            ok = ok and np.array_equal(frame.numpy().view(np.uint32), want[p].view(np.uint32))
        else:
            assert frame is None
    if rank == 0:
        open(out_path, "w").write(str(int(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,stripe", [(64, 16), (37, 8), (20, 16)])     # whole stripes; a 5-row last stripe on rank 0; one stripe and a tail
def test_stripe_split_and_gather_gloo(tmp_path, height, stripe):
    import torch.multiprocessing as mp
    out = tmp_path / "result.txt"
    mp.spawn(_stripe_worker, args=(2, _free_port(), height, stripe, str(out)), nprocs=2, join=True)
    assert out.read_text() == "1"


def test_view_partition_and_bands(vra):
    sh = vra.shard
    for world in (1, 2, 3, 8):
        seen = sorted(k for r in range(world) for k in sh.views_for_rank(4096, r, world))
        assert seen == list(range(4096))
        for h in (1080, 4320, 37, 8):
            rows = [sh.band_rows(h, r, world) for r in range(world)]
            assert rows[0][0] == 0 and sum(n for _, n in rows) == h
            assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert max(n for _, n in rows) - min(n for _, n in rows) <= 1
            for stripe in (8, 16):
                cover = sorted(sum((sh.stripe_rows(h, r, world, stripe) for r in range(world)), []))
                assert cover[0][0] == 0 and sum(n for _, n in cover) == h
                assert all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
                assert all(n == stripe for _, n in cover[:-1]) and 1 <= cover[-1][1] <= stripe
                counts = [len(sh.stripe_rows(h, r, world, stripe)) for r in range(world)]
                assert counts[0] == max(counts) and max(counts) - min(counts) <= 1
