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

# BASELINE.json configs -> workload parameters (scene "noise" = volume_gpu_generate_noise, p = 1/256)
PRESETS = {
    1: dict(scene="teapot.off", res=128, target="svo", width=640, height=480, views=64, radius=1.0, split=False),
    2: dict(scene="dragon.off", res=512, target="svo", width=1920, height=1080, views=16, radius=1.0, split=False),
    3: dict(scene="noise", res=1024, target="dense", width=3840, height=2160, views=2, radius=1.1, split=False),
    4: dict(scene="galleon.ply", res=2048, target="svo", width=7680, height=4320, views=1, radius=1.0, split=True),
    5: dict(scene="stratocaster.ply", res=1024, target="svo", width=1920, height=1080, views=16, radius=1.2, split=False),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(PRESETS),
                    help="BASELINE.json config number (default 2, the config the metric is quoted on)")
    ap.add_argument("--views", type=int, default=None, help="orbit views per step per GPU")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--target", choices=["svo", "dense"], default=None)
    ap.add_argument("--split", action="store_true", default=None,
                    help="one frame per step split into row bands over the ranks + gather to rank 0 (config 4)")
    ap.add_argument("--mode", choices=["beam", "naive"], default="beam",
                    help="beam = raycast_beamOpt (the reference's live path); naive = traversal from the camera")
    ap.add_argument("--no-device-prepare", action="store_true",
                    help="skip the (untimed) run of volume_gpu_voxelize on the same mesh and its comparison with the host scene")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo + --force-device rehearses the N>1 paths on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=None, help="use this GPU on every rank (rehearsal only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the workload timed on the CPU oracle (0 = auto)")
    args = ap.parse_args()
    for k, v in PRESETS[args.config].items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    return args


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    vra = load_package()

    if args.force_device is not None:
        local_rank = args.force_device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective operands live

    # ---- scene: host voxelization + octree build (or device-side noise), one upload to HBM ----
    t0 = time.time()
    want_dense = args.target == "dense"
    scene = None
    if args.scene == "noise":
        gpu = vra.GpuScene.noise(args.res, seed=1, threshold=1 << 24, device=local_rank)      # p = 2^24 / 2^32 = 1/256
        data = f"synthetic {args.res}^3 noise volume generated on the device (p = 1/256), synthetic orbit cameras"
    else:
        threads = 0 if world == 1 else max(2, 16 // min(world, 8))
        scene = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", args.scene), args.res, dense=want_dense,
                                        svo=not want_dense, threads=threads)
        if want_dense:
            gpu = vra.GpuScene.upload_dense(scene.target, scene.low, device=local_rank)
        else:
            gpu = vra.GpuScene.upload_svo(scene.svo, scene.low, device=local_rank)
        data = f"bundled asset ({args.scene}) voxelized on the host, synthetic orbit cameras"
    size_hi, size_lo = (args.res,) * 3, (args.res // 4,) * 3
    setup_s = time.time() - t0

    # the same scene prepared on the device (volume_gpu_voxelize), outside the timed region: how
    # long it takes and whether octree / grids equal the host pipeline's (reported, rank 0)
    prepare = None
    if scene is not None and rank == 0 and not args.no_device_prepare:
        try:
            import numpy as np
            mesh = vra.HostMesh(os.path.join(ROOT, "assets", args.scene), 2)
            best = None
            for _ in range(2):
                t1 = time.time()
                g2 = vra.GpuScene.voxelize(mesh.desc, args.res, dense=want_dense, svo=not want_dense, device=local_rank)
                dt = time.time() - t1
                best = dt if best is None else min(best, dt)
                if _ == 0:
                    g2.close()
            same = bool(np.array_equal(g2.read_low_voxels(), vra.host.dense_view(scene.low).reshape(-1, 8)))
            if want_dense:
                same = same and bool(np.array_equal(g2.read_voxels(0, args.res ** 3), vra.host.dense_view(scene.target).reshape(-1, 8)))
            else:
                nh, lh = scene.svo_arrays()
                ng, lg = g2.read_svo()
                same = same and bool(np.array_equal(ng.view(np.uint64), nh.view(np.uint64)) and np.array_equal(lg, lh))
            prepare = {"device_ms": round(best * 1e3, 1), "host_s": round(setup_s, 2), "equal_to_host": same,
                       "cube_tests": g2.vox_stats["tests"], "search_levels": g2.vox_stats["levels"]}
            g2.close()
            mesh.close()
        except Exception as e:      # reported, never fatal: this is not the measured path
            prepare = {"error": str(e)[:200]}

    W, H, B = args.width, args.height, args.views
    naive = args.mode == "naive"
    split = bool(args.split)
    if split:
        B = 1
        y0, rows = vra.shard.band_rows(H, rank, world)
    else:
        y0, rows = 0, H
    names = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")
    planes = {p: torch.empty((B, rows, W), dtype=torch.float32 if p == "depth" else torch.int32, device=dev)
              for p in names}
    ptrs = {p: t.data_ptr() for p, t in planes.items()}
    stream = torch.cuda.current_stream(dev)

    def step_frames(step):
        if split:      # every rank renders its row band of the SAME view
            ks = [step % ORBIT_VIEWS]
        else:          # ranks interleave through the orbit: view k goes to rank k mod world
            ks = [((step * B + i) * world + rank) % ORBIT_VIEWS for i in range(B)]
        frames = vra.orbit_frames(ks, ORBIT_VIEWS, size_hi, size_lo, W, H, radius=args.radius, naive=naive)
        if split:
            frames[0] = frames[0].band(y0, rows)
        return frames

    all_frames = [step_frames(s) for s in range(args.warmup + args.steps)]
    assembled = [None]

    def run(frames):
        gpu.render_device(frames, ptrs, stream=stream.cuda_stream)
        if split:      # frame assembly on rank 0: the one collective of the path (RCCL gather over xGMI)
            assembled[0] = vra.shard.gather_bands(planes["rgba8"][0].to(coll_dev), H, W, rank, world, dist, dst=0)

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
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic bytes of the timed launches (untimed re-render, exact integer sums) -----
    # B_ray = 8*(steps_lo + steps_hi) + 8*[hit] + 4 (pixel) + 16 (aux planes)
    alg_bytes = 0
    sum_lo = sum_hi = hits = 0
    rays_per_launch = B * rows * W
    for s in range(args.warmup, args.warmup + args.steps):
        gpu.render_device(all_frames[s], ptrs, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        lo = int(planes["steps_lo"].sum(dtype=torch.int64).item())
        hi = int(planes["steps_hi"].sum(dtype=torch.int64).item())
        h = int((planes["hit_index"] != -1).sum().item())
        sum_lo += lo; sum_hi += hi; hits += h
        alg_bytes += 8 * (lo + hi) + 8 * h + 20 * rays_per_launch

    assembled_ok = None
    if split and rank == 0:     # the assembled frame equals the frame one GPU renders alone
        last = args.warmup + args.steps - 1
        full = vra.orbit_frames([last % ORBIT_VIEWS], ORBIT_VIEWS, size_hi, size_lo, W, H, radius=args.radius, naive=naive)
        alone = torch.empty((1, H, W), dtype=torch.int32, device=dev)
        gpu.render_device(full, {"rgba8": alone.data_ptr()}, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        assembled_ok = bool(torch.equal(assembled[0].to(dev), alone[0]))

    rays_per_step = (H * W) if split else (B * H * W * world)       # whole job, all ranks
    mrays = rays_per_step * args.steps / elapsed / 1e6
    par = (f"one frame split into {world} row bands + gather of rgba8 to rank 0" if split
           else f"view-parallel x{world}, scene replicated, no collective")
    out = {
        "metric": "primary Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None,
        "dtype": "f32", "data": data,
        "config": {"workload": f"{args.scene} at {args.res}^3 {args.target} target, low grid {args.res // 4}^3, "
                               f"{W}x{H} frames, {args.mode} mode (BASELINE config {args.config})",
                   "views_per_step_per_gpu": B, "rays_per_step": rays_per_step,
                   "orbit_radius": args.radius, "planes_written": 5, "parallelism": par,
                   "mean_steps_lo": round(sum_lo / (rays_per_launch * args.steps), 3),
                   "mean_steps_hi": round(sum_hi / (rays_per_launch * args.steps), 3),
                   "hit_fraction": round(hits / (rays_per_launch * args.steps), 4),
                   "scene_setup_s": round(setup_s, 2), "scene_device_mib": gpu.info(7)},
    }
    if assembled_ok is not None:
        out["config"]["assembled_frame_equals_single_gpu_frame"] = assembled_ok
    if prepare is not None:
        out["config"]["scene_prepared_on_device"] = prepare
    if launches:
        avg_ms = kern_ms / launches
        alg_per_launch = alg_bytes / launches
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == 2 and args.mode == "beam" and not split:
            try:     # measured with rocprofv3 --pmc on this same default workload (see the file's comment)
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "kernel": {1: "vk::render_frames_refill", 2: "vk::render_frames_budget + vk::continue_rays"}.get(
                               gpu.info(8), "vk::render_frames<%s>" % ("true" if args.target == "svo" else "false")),
                           "kernel_avg_ms": round(avg_ms, 4), "launches": launches,
                           "algorithmic_bytes_per_launch": int(alg_per_launch)}

    # ---- CPU baseline: the oracle, one thread (the reference is single threaded), bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline and scene is not None:
        from __graft_entry__ import load_oracle
        ora = load_oracle()
        n_cpu = args.cpu_frames or max(1, min(8, int(round(16e6 / (W * H)))))
        frames = all_frames[args.warmup]
        kw = dict(low=scene.low, target=scene.target) if want_dense else dict(low=scene.low, svo=scene.svo)
        t1 = time.perf_counter()
        for i in range(n_cpu):
            ora.render_frame(frames[i % B], threads=1, **kw)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n_cpu * rows * W / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} frame(s) of the first timed step ({n_cpu * rows * W} rays), "
                                         f"scalar C oracle, gcc -O3 -mavx -ffp-contract=off, {cpu_s:.1f} s"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    gpu.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
