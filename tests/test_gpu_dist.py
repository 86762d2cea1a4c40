"""The single-frame split with the GPU doing the rendering:
  * in one process: two row bands rendered through volume_gpu_render_device, assembled with
    shard.gather_bands' own band arithmetic, equal the full frame one call renders, and the oracle;
  * two ranks (gloo, both on the one GPU of a test box): each renders its band on the device,
    shard.gather_bands assembles the frame on rank 0 -- the path bench.py --config 4 takes.
The nccl (RCCL) gather between two real GPUs needs a multi-GPU node; the driver's scaling run is
where it executes."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, ASSETS, assert_planes_equal

pytestmark = pytest.mark.gpu

W, H = 200, 101


def _render_device(torch, gpu, frame):
    planes = {p: torch.empty((1, frame.tile_h, frame.tile_w), dtype=torch.float32 if p == "depth" else torch.int32, device="cuda:0")
              for p in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
    gpu.render_device([frame], {p: t.data_ptr() for p, t in planes.items()}, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return planes


@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_in_one_process_equal_the_full_frame(vra, ora, teapot64, world):
    import torch
    gpu = vra.GpuScene.upload_svo(teapot64.svo, teapot64.low)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, W, H)
    whole = _render_device(torch, gpu, full)
    bands = [_render_device(torch, gpu, vra.shard.band_frame(full, r, world)) for r in range(world)]
    want = ora.render_frame(full, low=teapot64.low, svo=teapot64.svo)
    for p in whole:
        assembled = torch.cat([b[p][0] for b in bands], dim=0)
        assert assembled.shape == (H, W) and torch.equal(assembled, whole[p][0]), p
    got = {p: t[0].cpu().numpy().view(np.float32 if p == "depth" else np.uint32) for p, t in whole.items()}
    assert_planes_equal(got, want, "full frame through render_device")
    gpu.close()


def _rank(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    vra = load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = vra.HostScene.from_mesh(os.path.join(ASSETS, "teapot.off"), 64, threads=2)
    gpu = vra.GpuScene.upload_svo(scene.svo, scene.low, device=0)
    pos, view = vra.orbit_camera(3, 16, 1.0)
    full = vra.frame_from_camera(view, pos, (64,) * 3, (16,) * 3, W, H)
    band = vra.shard.band_frame(full, rank, world)
    mine = torch.empty((1, band.tile_h, W), dtype=torch.int32, device="cuda:0")
    gpu.render_device([band], {"rgba8": mine.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    frame = vra.shard.gather_bands(mine[0].cpu(), H, W, rank, world, dist, dst=0)
    if rank == 0:
        alone = torch.empty((1, H, W), dtype=torch.int32, device="cuda:0")
        gpu.render_device([full], {"rgba8": alone.data_ptr()}, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok = bool(torch.equal(frame, alone[0].cpu()))
        lit = int((alone[0] != 255).sum().item())
        open(out_path, "w").write(f"{int(ok)} {lit}")
    else:
        assert frame is None
    dist.barrier()
    gpu.close()
    dist.destroy_process_group()


def test_two_ranks_render_bands_on_the_gpu_and_gather(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.txt"
    mp.spawn(_rank, args=(2, port, str(out)), nprocs=2, join=True)
    ok, lit = out.read_text().split()
    assert ok == "1" and int(lit) > 500


@pytest.mark.parametrize("split_rows", [-1, 8, 0])      # stripes chosen by bench.py (16 rows here), 8-row stripes with a 1-row last one, row bands
def test_bench_split_rehearsal_reports_per_rank_figures(split_rows):
    """bench.py's single-frame split (config 4's path) with two gloo ranks sharing this box's GPU: every rank builds its
    replica with volume_gpu_voxelize, renders its share of the rows -- stripes dealt round-robin, or one contiguous
    band -- the shares are gathered through pre-allocated buffers, and the JSON line carries what SURVEY 8e asks the
    split to report -- per-rank kernel times, load imbalance, the gather alone -- and says that the assembled frame
    equals the frame one GPU renders."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--split",
                        "--split-rows", str(split_rows),
                        "--scene", "teapot.off", "--res", "128", "--width", "640", "--height", "361", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["assembled_frame_equals_single_gpu_frame"] is True
    assert out["config"]["rays_per_step"] == 640 * 361
    assert len(out["per_rank_kernel_ms"]) == 2 and all(ms > 0 for ms in out["per_rank_kernel_ms"])
    assert out["load_imbalance"] >= 0 and out["gather_ms"] > 0
    # what travels per peer: the tallest band, or rank 0's stripes (12 of 16 rows, 23 of 8 rows)
    assert out["gather_bytes_per_peer"] == {0: 181, -1: 12 * 16, 8: 23 * 8}[split_rows] * 640 * 4
    assert ("stripes" in out["config"]["parallelism"]) == (split_rows != 0)
    assert "device" in out["data"]                      # replicas built by volume_gpu_voxelize on every rank
