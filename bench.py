#!/usr/bin/env python3
"""bench.py -- primary Mrays/s of the MI355X traversal path on BASELINE.json's
config 2 (dragon.off voxelized at 512^3, octree target, 1920x1080 frames).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--views` orbit views on
every rank (weak scaling: per-GPU work is fixed, views are split round-robin over
the ranks, scene replicated, no collective on the data path).  Inputs (octree,
low grid) are resident in HBM before the timed region; outputs (rgba8, hit
index, step counts, depth: 20 B/ray) stay in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ORBIT_VIEWS = 4096             # SURVEY 8d config 5 orbit


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=16, help="orbit views per step per GPU")
    ap.add_argument("--scene", default="dragon.off")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--radius", type=float, default=1.0)
    ap.add_argument("--target", choices=["svo", "dense"], default="svo")
    ap.add_argument("--mode", choices=["beam", "naive"], default="beam",
                    help="beam = raycast_beamOpt (the reference's live path); naive = traversal from the camera")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the workload timed on the CPU oracle (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    vra = load_package()

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- scene: host voxelization + octree build, then one upload to HBM ---------------------
    t0 = time.time()
    want_dense = args.target == "dense"
    scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", args.scene), args.res, dense=want_dense,
                                    svo=not want_dense, threads=max(1, 16 // max(1, min(world, 8))) if world > 1 else 0)
    size_hi, size_lo = (args.res,) * 3, (args.res // 4,) * 3
    if want_dense:
        gpu = vra.GpuScene.upload_dense(scene.target, scene.low, device=local_rank)
    else:
        gpu = vra.GpuScene.upload_svo(scene.svo, scene.low, device=local_rank)
    setup_s = time.time() - t0

    W, H, B = args.width, args.height, args.views
    naive = args.mode == "naive"
    planes = {p: torch.empty((B, H, W), dtype=torch.float32 if p == "depth" else torch.int32, device=dev)
              for p in ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")}
    ptrs = {p: t.data_ptr() for p, t in planes.items()}
    stream = torch.cuda.current_stream(dev)

    def step_frames(step):
        # global view index: steps advance through the orbit, ranks interleave (k = rank mod world)
        ks = [((step * B + i) * world + rank) % ORBIT_VIEWS for i in range(B)]
        return vra.orbit_frames(ks, ORBIT_VIEWS, size_hi, size_lo, W, H, radius=args.radius, naive=naive)

    all_frames = [step_frames(s) for s in range(args.warmup + args.steps)]

    def run(frames):
        gpu.render_device(frames, ptrs, stream=stream.cuda_stream)

    for s in range(args.warmup):
        run(all_frames[s])
    torch.cuda.synchronize(dev)
    gpu.kernel_time()                       # drop warm-up events
    gpu.set_profiling(True)                 # HIP events around every traversal-kernel launch, on `stream`

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        run(all_frames[s])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    kern_ms, launches = gpu.kernel_time()
    gpu.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic bytes of the timed launches (untimed re-render, exact integer sums) -----
    # B_ray = 8*(steps_lo + steps_hi) + 8*[hit] + 4 (pixel) + 16 (aux planes)
    alg_bytes = 0
    sum_lo = sum_hi = hits = 0
    rays_per_step = B * H * W
    for s in range(args.warmup, args.warmup + args.steps):
        run(all_frames[s])
        torch.cuda.synchronize(dev)
        lo = int(planes["steps_lo"].sum(dtype=torch.int64).item())
        hi = int(planes["steps_hi"].sum(dtype=torch.int64).item())
        h = int((planes["hit_index"] != -1).sum().item())
        sum_lo += lo; sum_hi += hi; hits += h
        alg_bytes += 8 * (lo + hi) + 8 * h + 20 * rays_per_step

    total_rays = rays_per_step * args.steps * world
    mrays = total_rays / elapsed / 1e6
    out = {
        "metric": "primary Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "bundled asset (dragon.off) voxelized on the host, synthetic orbit cameras",
        "config": {"workload": f"{args.scene} at {args.res}^3 {args.target} target, low grid {args.res // 4}^3, "
                               f"{W}x{H} frames, {args.mode} mode (BASELINE config 2)",
                   "views_per_step_per_gpu": B, "rays_per_step_per_gpu": rays_per_step,
                   "orbit_radius": args.radius, "planes_written": 5, "parallelism": f"view-parallel x{world}, scene replicated",
                   "mean_steps_lo": round(sum_lo / (rays_per_step * args.steps), 3),
                   "mean_steps_hi": round(sum_hi / (rays_per_step * args.steps), 3),
                   "hit_fraction": round(hits / (rays_per_step * args.steps), 4),
                   "scene_setup_s": round(setup_s, 2)},
    }
    if launches:
        avg_ms = kern_ms / launches
        alg_per_launch = alg_bytes / launches
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "kernel": "vk::render_frames<%s>" % ("true" if args.target == "svo" else "false"),
                           "kernel_avg_ms": round(avg_ms, 4), "launches": launches,
                           "algorithmic_bytes_per_launch": int(alg_per_launch)}

    # ---- CPU baseline: the oracle, one thread (the reference is single threaded), bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from __graft_entry__ import load_oracle
        ora = load_oracle()
        n_cpu = args.cpu_frames or max(1, min(8, int(round(16e6 / (W * H)))))
        frames = all_frames[args.warmup]
        kw = dict(low=scene.low, target=scene.target) if want_dense else dict(low=scene.low, svo=scene.svo)
        t1 = time.perf_counter()
        for i in range(n_cpu):
            ora.render_frame(frames[i % B], threads=1, **kw)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n_cpu * W * H / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} of the {B} frames of the first timed step ({n_cpu * W * H} rays), "
                                         f"scalar C oracle, gcc -O3 -mavx -ffp-contract=off, {cpu_s:.1f} s"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    gpu.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
