#!/usr/bin/env python3
"""bench.py -- primary Mrays/s of the MI355X traversal path.

    python bench.py --gpus N --steps K --warmup W [--config 0..5]

Default workload (--config 0) is BASELINE.json's north-star target: dragon.off voxelized at
1024^3, octree target, 256^3 beam grid, 3840x2160 frames, the reference's live per-pixel call
(raycast_beamOpt, main.cpp:260 -> main.cpp:163-166).  --config 1..5 are BASELINE.json's configs.

One "step" = one pass of the hot path over one batch of `--views` orbit views on every rank,
one launch of the frame kernel per rank for up to 64 views (16 x 4K = 133 M rays by default: launches
of this size run at the kernel's steady rate).  Weak scaling: per-GPU work is
fixed, views are dealt round-robin to the ranks, scene replicated, no collective on the data path
(--config 4 / --split is the single-frame row-band split with one gather).  Inputs (octree, low grid) are resident in HBM before the timed region; outputs (rgba8,
hit index, step counts, depth: 20 B/ray) stay in HBM.  Rank 0 prints ONE JSON line.

--gpus N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts N ranks
itself, one per GPU, before it touches HIP or torch, relays rank 0's JSON line and exits with the
worst exit code.  Under torch.distributed.run it is one of the ranks; WORLD_SIZE != --gpus is an error.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMDS = 1024                 # 256 CUs x 4 SIMDs
# The bound that can bind this path: unfused FP32 vector arithmetic.  MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector
# peak = 1024 SIMDs x 64 lanes / 2 cycles x 2.4 GHz x 2 flops per FMA.  The parity contract forbids FMA
# (reference Makefile:3: -mavx without -mfma, so every glm::fma is a multiply and an add), which leaves one
# flop per lane-op: 78.65 T lane-ops/s.
VALU_PEAK_TLANEOPS = 157.3 / 2
# the reference's own float operations: per DDA / octree step (main.cpp:122-153: 3 adds for the sample point,
# 3 floor, 3 add, 3 sub, 3 mul, min3, 3 mul, 3 add) and per ray outside the loops (direction sums, two
# normalisations, two ray set-ups, two entry clips, shading, depth: SURVEY 8d's "~ 40 * steps + 60" flops)
OPS_PER_STEP, OPS_PER_RAY = 22, 60
ORBIT_VIEWS = 4096             # SURVEY 8d config 5 orbit
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_r03.json")   # written by profiles/summarize.py from rocprofv3 --pmc passes
LIB_PATH = os.path.join(ROOT, "volume-renderer_amd", "lib", "libvolly_gpu.so")

# camera regimes (--camera): where the views of a step are taken from
#   orbit    the orbit of SURVEY 8d at the preset's radius, looking at the volume's centre (the headline: the
#            model seen whole from outside; most of a 16:9 frame is sky)
#   startup  the reference's start-up state: camPos = 0, a corner of the grid (main.h:96), camDir from the mouse
#            at (0, 0) (main.cpp:393-397); every ray starts at the grid, nothing is sky, the cull is off
#   inside   the same orbit at radius 0.25: the camera is inside the grid (= the model's bounding box: normalizePoly
#            stretches every mesh to the unit cube, PLYfileReader.cpp:144-171)
#   fill     the same orbit at radius 0.5: the camera sits at the grid's faces and the model fills the frame
#            (hit fraction 0.5 - 1.0 along the orbit)
CAMERA_RADIUS = {"inside": 0.25, "fill": 0.5}

# workload parameters (scene "noise" = volume_gpu_generate_noise; several scenes = alternate per step)
PRESETS = {
    0: dict(scene="dragon.off", res=1024, target="svo", width=3840, height=2160, views=16, radius=1.0, split=False),  # north-star target
    1: dict(scene="teapot.off", res=128, target="svo", width=640, height=480, views=64, radius=1.0, split=False),
    2: dict(scene="dragon.off", res=512, target="svo", width=1920, height=1080, views=64, radius=1.0, split=False),
    3: dict(scene="noise", res=1024, target="dense", width=3840, height=2160, views=2, radius=1.1, split=False),
    4: dict(scene="galleon.ply", res=2048, target="svo", width=7680, height=4320, views=1, radius=1.0, split=True),
    5: dict(scene="stratocaster.ply,dragon.off", res=1024, target="svo", width=1920, height=1080, views=64, radius=1.2, split=False),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, choices=sorted(PRESETS),
                    help="0 = the north-star target workload (default); 1..5 = BASELINE.json's configs")
    ap.add_argument("--views", type=int, default=None, help="orbit views per step per GPU")
    ap.add_argument("--scene", default=None, help="asset file name(s), comma separated (alternate per step), or 'noise'")
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--target", choices=["svo", "dense"], default=None)
    ap.add_argument("--noise-log2-inv-p", type=int, default=8,
                    help="noise scene: a voxel is filled with probability 2^-n (config 3: 8 = 1/256; sweeps 6 and 10)")
    ap.add_argument("--split", action="store_true", default=None,
                    help="one frame per step split over the ranks (stripes of rows, see --split-rows) + gather to rank 0 (config 4)")
    ap.add_argument("--split-rows", type=int, default=-1,
                    help="single-frame split: stripes of this many rows dealt round-robin over the ranks; -1 = the smallest multiple "
                         "of 8 (at least 16) that gives a rank at most 64 stripes, i.e. one launch per step; 0 = contiguous row bands "
                         "(SURVEY 8e's first form: 50-65 %% load imbalance at 4-8 ranks, profiles/split_imbalance.py)")
    ap.add_argument("--mode", choices=["beam", "naive"], default="beam",
                    help="beam = raycast_beamOpt (the reference's live path); naive = traversal from the camera")
    ap.add_argument("--camera", choices=["orbit", "startup", "inside", "fill"], default="orbit",
                    help="camera regime of the step's views (see CAMERA_RADIUS above)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the untimed end-to-end block (host-buffer entry point)")
    ap.add_argument("--no-device-prepare", action="store_true",
                    help="skip the (untimed) run of volume_gpu_voxelize on the same mesh and its comparison with the host scene")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo + --force-device rehearses the N>1 paths on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=None, help="use this GPU on every rank (rehearsal only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal without a GPU: ranks meet (use --backend gloo), run the barrier and "
                         "the max-over-ranks reduction, rank 0 prints the JSON line with value null; nothing is rendered")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the workload timed on the CPU oracle (0 = auto)")
    args = ap.parse_args()
    for k, v in PRESETS[args.config].items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    if args.camera in CAMERA_RADIUS:
        args.radius = CAMERA_RADIUS[args.camera]
    return args


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the library this run loads: ties counter files to a build."""
    import hashlib
    try:
        return hashlib.sha256(open(os.environ.get("VOLLY_GPU_LIB") or LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def launch_ranks(n):
    """Start n fresh ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment).  The parent has made no HIP or torch call; it only waits and relays."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    for p in procs:
        rc = p.wait()
        worst = max(worst, abs(rc))
    return worst


def init_ranks(dist, torch, backend, rank, world, local_rank):
    """The process group of the N ranks: RCCL (torch's "nccl" backend) as the contract asks, probed with one
    all-reduce.  If RCCL cannot be brought up on this node -- an exception from the set-up or from the probe, on
    every rank alike, since the probe needs all of them; a rank that fails alone leaves the others waiting for RCCL's
    own timeout first -- the ranks fall back to gloo rather than lose the run:
    the default workload has no collective on its data path (barrier and max-over-ranks timing only), and the
    split mode's gather then goes through host memory, which the JSON line says ("backend").  Returns the
    backend in use and a note for the JSON line (None when nothing went wrong)."""
    import datetime
    note = None
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=120))
            probe = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"probe all-reduce returned {probe.item()} on {world} ranks")
            return "nccl", None
        except Exception as e:
            note = f"nccl failed ({type(e).__name__}: {str(e)[:160]}), control collectives on gloo"
            print(f"bench.py rank {rank}: {note}", file=sys.stderr, flush=True)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    return "gloo", note


def pmc_entry(key, rays_per_launch, src_sha):
    """Counter figures of this workload's dominant kernel from a separate rocprofv3 --pmc run
    (profiles/run_pmc.sh + profiles/summarize.py), or None when that run was not made, had another
    shape, or was taken on a library built from other sources.  The tie is the hash of the kernel
    sources that the library records at build time (volly_gpu_build_info, src_sha16): the hash of the
    .so file itself changes from one build of the same sources to the next (hipcc's temporary names)."""
    try:
        e = json.load(open(PMC_FILE)).get(key)
    except Exception:
        return None
    if not e or e.get("rays_per_launch") != rays_per_launch or not src_sha or e.get("src_sha16") != src_sha:
        return None
    return e


def binding_resource(pmc):
    """Name what the counters say binds the kernel: the busiest of the resources they cover, or none."""
    shares = {"vector-ALU issue": pmc.get("valu_issue_frac") or 0.0, "HBM": pmc.get("hbm_real_frac") or 0.0,
              "texture addresser (vector memory issue)": pmc.get("ta_busy_frac") or 0.0}
    name, top = max(shares.items(), key=lambda kv: kv[1])
    if top < 0.6:
        return f"none above 0.6 (busiest: {name} {top:.2f}): dependent-load latency and lane divergence"
    return name


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))          # before any torch / HIP call in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus N` "
              f"or under torch.distributed.run with --nproc-per-node equal to --gpus", file=sys.stderr)
        sys.exit(2)
    # stdout carries ONE line, the JSON line: RCCL prints a version banner and gloo its connection lines to file
    # descriptor 1 from native code, so from here on descriptor 1 is stderr and the JSON line goes to the saved one
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    if args.force_device is not None:
        local_rank = args.force_device
    backend_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not args.dry_run:
            torch.cuda.set_device(local_rank)
        args.backend, backend_note = init_ranks(dist, torch, args.backend, rank, world, local_rank)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

    if args.dry_run:                               # plumbing only: the N>1 launch path without a GPU
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
        if rank == 0:
            line = {"metric": "primary Mrays/s", "value": None, "unit": "Mrays/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                    "backend": args.backend if world > 1 else None, "backend_note": backend_note,
                    "max_over_ranks_probe": float(t.item())}
            print(json.dumps(line), file=json_out, flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from __graft_entry__ import load_package
    vra = load_package()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective operands live

    # ---- scenes: host voxelization + octree build (or device-side noise), one upload to HBM ----
    t0 = time.time()
    want_dense = args.target == "dense"
    names = args.scene.split(",")
    host_scenes, gpus = [], []
    for name in names:
        if name == "noise":
            gpus.append(vra.GpuScene.noise(args.res, seed=1, threshold=1 << (32 - args.noise_log2_inv_p), device=local_rank))
            host_scenes.append(None)
        elif world > 1:
            # N ranks: every rank builds its replica on its own GPU (volume_gpu_voxelize: 13-80 ms, byte-identical to
            # the host pipeline -- tests/test_gpu_voxelize.py) instead of N host voxelizations sharing the node's cores
            mesh = vra.HostMesh(os.path.join(ROOT, "assets", name), 2)
            gpus.append(vra.GpuScene.voxelize(mesh.desc, args.res, dense=want_dense, svo=not want_dense, device=local_rank))
            mesh.close()
            host_scenes.append(None)
        else:
            sc = vra.HostScene.from_mesh(os.path.join(ROOT, "assets", name), args.res, dense=want_dense, svo=not want_dense, threads=0)
            gpus.append(vra.GpuScene.upload_dense(sc.target, sc.low, device=local_rank) if want_dense
                        else vra.GpuScene.upload_svo(sc.svo, sc.low, device=local_rank))
            host_scenes.append(sc)
    if names == ["noise"]:
        data = f"synthetic {args.res}^3 noise volume generated on the device (p = 2^-{args.noise_log2_inv_p}), synthetic orbit cameras"
    else:
        data = (f"bundled asset(s) {args.scene} voxelized on the " + ("device (volume_gpu_voxelize, one replica per rank)" if world > 1 else "host")
                + f", synthetic cameras ({args.camera})")
    size_hi, size_lo = (args.res,) * 3, (args.res // 4,) * 3
    setup_s = time.time() - t0
    scene, gpu = host_scenes[0], gpus[0]

    # the same scene prepared on the device (volume_gpu_voxelize), outside the timed region: how
    # long it takes and whether octree / grids equal the host pipeline's (reported, rank 0)
    prepare = None
    if scene is not None and rank == 0 and not args.no_device_prepare:
        try:
            import numpy as np
            mesh = vra.HostMesh(os.path.join(ROOT, "assets", names[0]), 2)
            best = None
            for i in range(2):
                t1 = time.time()
                g2 = vra.GpuScene.voxelize(mesh.desc, args.res, dense=want_dense, svo=not want_dense, device=local_rank)
                dt = time.time() - t1
                best = dt if best is None else min(best, dt)
                if i == 0:
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
    if args.split_rows < 0:      # one launch takes 64 sub-rectangles
        args.split_rows = max(16, -(-args.height // (64 * world * 8)) * 8)
    stripes = split and args.split_rows > 0 and world > 1       # (one rank renders the whole frame in one piece)
    gatherer = None
    if stripes:
        # one frame, stripes of --split-rows rows dealt round-robin: a rank's planes are its stripes stacked (the
        # frame-major layout of ONE call over its sub-rectangles); rgba8 is the gatherer's own send buffer
        B = 1
        S = args.split_rows
        mine = vra.shard.stripe_rows(H, rank, world, S)
        rows = sum(r for _, r in mine)
        gatherer = vra.shard.StripeGatherer(H, W, rank, world, torch.int32, coll_dev, S)
        slots = gatherer.send.shape[0]
        plane_shape = (slots, S, W)
    elif split:
        B = 1
        y0, rows = vra.shard.band_rows(H, rank, world)
        plane_shape = (B, rows, W)
    else:
        y0, rows = 0, H
        plane_shape = (B, rows, W)
    plane_names = ("rgba8", "hit_index", "steps_lo", "steps_hi", "depth")
    planes = {p: torch.empty(plane_shape, dtype=torch.float32 if p == "depth" else torch.int32, device=dev)
              for p in plane_names}
    if stripes:       # the planes of rows nobody renders (unused slots, the rest of a shorter last stripe) read as misses with no steps
        for p in ("steps_lo", "steps_hi"):
            planes[p].zero_()
        planes["hit_index"].fill_(-1)
        if coll_dev == dev:
            planes["rgba8"] = gatherer.send     # rendered in place: no copy in front of the collective
    ptrs = {p: t.data_ptr() for p, t in planes.items()}
    stream = torch.cuda.current_stream(dev)

    # Weak scaling wants the same work per rank at every N.  Work per view changes along the orbit, so the
    # north-star workload samples the orbit `world` times finer at `world` ranks: the job's K steps then cover
    # the same arc at every N (views j of a 4096 * world orbit, dealt round-robin), and each rank's views span
    # it evenly.  Configs 1-5 keep SURVEY 8d's 4096-view orbit.
    orbit_n = ORBIT_VIEWS * world if (args.config == 0 and not split) else ORBIT_VIEWS

    def step_frames(step):
        if split:      # every rank renders its row band of the SAME view
            ks = [step % ORBIT_VIEWS]
        else:          # ranks interleave through the orbit: view k goes to rank k mod world
            ks = [((step * B + i) * world + rank) % orbit_n for i in range(B)]
        if args.camera == "startup":     # main.h:96 + main.cpp:393-397 with the mouse at (0, 0): the same frame B times
            view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(0, 0), (0, 0, -1))
            frames = (vra.FrameDesc * len(ks))()
            for i in range(len(ks)):
                frames[i] = vra.frame_from_camera(view, (0.0, 0.0, 0.0), size_hi, size_lo, W, H, naive=naive)
        else:
            frames = vra.orbit_frames(ks, orbit_n, size_hi, size_lo, W, H, radius=args.radius, naive=naive)
        if stripes:   # (full, tail): one call over the stripes of full height, one for a shorter last stripe
            return vra.shard.stripe_frames(frames[0], rank, world, args.split_rows)
        if split:
            frames[0] = frames[0].band(y0, rows)
        return frames

    all_frames = [step_frames(s) for s in range(args.warmup + args.steps)]
    assembled = [None]
    # send buffer, receive list and the assembled frame are allocated once, outside the timed loop
    if split and not stripes:
        gatherer = vra.shard.BandGatherer(H, W, rank, world, torch.int32, coll_dev)

    def render(s, into):
        """one step's launches: planes at the device pointers `into`"""
        g = gpus[s % len(gpus)]
        if not stripes:
            g.render_device(all_frames[s], into, stream=stream.cuda_stream)
            return
        full, tail = all_frames[s]
        if len(full):
            g.render_device(full, into, stream=stream.cuda_stream)
        if tail is not None:      # behind the full stripes' slots
            off = len(full) * args.split_rows * W * 4
            g.render_device(tail, {p: a + off for p, a in into.items()}, stream=stream.cuda_stream)

    def gather_frame():
        if stripes:
            if coll_dev != dev:
                gatherer.send.copy_(planes["rgba8"])
            return gatherer.gather(dist)
        band = planes["rgba8"][0]
        return gatherer.gather(band if coll_dev == dev else band.cpu(), dist)

    def run(s):
        render(s, ptrs)
        if split:      # frame assembly on rank 0: the one collective of the path (RCCL gather over xGMI)
            assembled[0] = gather_frame()

    for s in range(args.warmup):
        run(s)
    torch.cuda.synchronize(dev)
    for g in gpus:
        g.kernel_time()                     # drop warm-up events
        g.set_profiling(True)               # HIP events around every traversal-kernel launch, on `stream`

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        run(s)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    kern_ms = launches = 0
    for g in gpus:
        ms, n = g.kernel_time()
        kern_ms += ms
        launches += n
        g.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-rank figures: kernel time of every rank, and for the split the gather alone ----
    per_rank_ms, gather_ms = None, None
    if world > 1:
        # per step: a rank that also holds the frame's shorter last stripe launches twice per step
        mine = torch.tensor([kern_ms / args.steps if split else kern_ms / max(launches, 1)], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(v.item()), 4) for v in every]
    if split and world > 1:            # untimed: the collective by itself, bands already rendered
        for _ in range(2):
            gather_frame()
        torch.cuda.synchronize(dev)
        dist.barrier()
        tg = time.perf_counter()
        reps = 10
        for _ in range(reps):
            gather_frame()
        torch.cuda.synchronize(dev)
        dist.barrier()
        tgm = torch.tensor([(time.perf_counter() - tg) / reps * 1e3], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tgm, op=dist.ReduceOp.MAX)
        gather_ms = round(float(tgm.item()), 4)

    # ---- algorithmic bytes of the timed launches (untimed re-render, exact integer sums) -----
    # B_ray = 8*(steps_lo + steps_hi) + 8*[hit] + 4 (pixel) + 16 (aux planes)
    alg_bytes = 0
    sum_lo = sum_hi = hits = 0
    rays_per_launch = B * rows * W
    # The re-renders run back to back like the timed launches (same five planes; the three planes the sums need
    # go to per-step buffers, up to 48 GB of them at a time) and the sums are taken afterwards: reductions
    # between the launches would sweep the caches and slow the next launch, and rocprofv3's per-kernel average
    # over the whole run would then disagree with the timed launches' HIP-event average.
    acc = torch.zeros(3, dtype=torch.int64, device=dev)
    need = ("steps_lo", "steps_hi", "hit_index")
    group = max(1, min(args.steps, int(48e9 // (12 * rays_per_launch))))
    for first in range(args.warmup, args.warmup + args.steps, group):
        steps_here = range(first, min(first + group, args.warmup + args.steps))
        if stripes:      # slots and rows nobody renders must read as misses with no steps
            keep = [{p: (torch.full_like(planes[p], -1) if p == "hit_index" else torch.zeros_like(planes[p])) for p in need} for _ in steps_here]
        else:
            keep = [{p: torch.empty_like(planes[p]) for p in need} for _ in steps_here]
        for s, k in zip(steps_here, keep):
            render(s, dict(ptrs, **{p: t.data_ptr() for p, t in k.items()}))
        for k in keep:
            acc[0] += k["steps_lo"].sum(dtype=torch.int64)
            acc[1] += k["steps_hi"].sum(dtype=torch.int64)
            acc[2] += (k["hit_index"] != -1).sum()
        torch.cuda.synchronize(dev)
        del keep
    sum_lo, sum_hi, hits = (int(v) for v in acc.tolist())
    alg_bytes = 8 * (sum_lo + sum_hi) + 8 * hits + 20 * rays_per_launch * args.steps

    assembled_ok = None
    if split and rank == 0:     # the assembled frame equals the frame one GPU renders alone
        last = args.warmup + args.steps - 1
        full = vra.orbit_frames([last % ORBIT_VIEWS], ORBIT_VIEWS, size_hi, size_lo, W, H, radius=args.radius, naive=naive)
        alone = torch.empty((1, H, W), dtype=torch.int32, device=dev)
        gpus[last % len(gpus)].render_device(full, {"rgba8": alone.data_ptr()}, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        assembled_ok = bool(torch.equal(assembled[0].to(dev), alone[0]))

    rays_per_step = (H * W) if split else (B * H * W * world)       # whole job, all ranks
    mrays = rays_per_step * args.steps / elapsed / 1e6
    par = (f"one frame split into stripes of {args.split_rows} rows dealt round-robin over {world} ranks + gather of rgba8 to rank 0" if stripes
           else f"one frame split into {world} row bands + gather of rgba8 to rank 0" if split
           else f"view-parallel x{world}, scene replicated, no collective")
    label = "north-star target" if args.config == 0 else f"BASELINE config {args.config}"
    if args.scene == "noise" and args.noise_log2_inv_p != 8:
        label += f", p = 2^-{args.noise_log2_inv_p} sweep"
    out = {
        "metric": "primary Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if split else "weak", "vs_baseline": None,
        "dtype": "f32", "data": data,
        "config": {"workload": f"{args.scene} at {args.res}^3 {args.target} target, low grid {args.res // 4}^3, "
                               f"{W}x{H} frames, {args.mode} mode ({label})",
                   "views_per_step_per_gpu": B, "rays_per_step": rays_per_step,
                   "orbit_radius": args.radius, "planes_written": 5, "parallelism": par,
                   "mean_steps_lo": round(sum_lo / (rays_per_launch * args.steps), 3),
                   "mean_steps_hi": round(sum_hi / (rays_per_launch * args.steps), 3),
                   "hit_fraction": round(hits / (rays_per_launch * args.steps), 4),
                   "dda_loop": ", ".join(g.dda_loops() for g in gpus),
                   "scene_setup_s": round(setup_s, 2), "scene_device_mib": sum(g.info(7) for g in gpus)},
    }
    out["config"]["camera"] = args.camera
    out["lib_sha16"] = lib_sha16()
    out["src_sha16"] = vra.gpu.build_info().get("src_sha16")
    if assembled_ok is not None:
        out["config"]["assembled_frame_equals_single_gpu_frame"] = assembled_ok
    if world > 1:
        out["backend"] = "rccl (torch nccl)" if args.backend == "nccl" else "gloo"
        if backend_note:
            out["backend_note"] = backend_note
    if per_rank_ms is not None:
        mean_ms = sum(per_rank_ms) / len(per_rank_ms)
        out["per_rank_kernel_ms"] = per_rank_ms
        out["load_imbalance"] = round(max(per_rank_ms) / mean_ms - 1.0, 4) if mean_ms > 0 else None
    if gather_ms is not None:
        out["gather_ms"] = gather_ms
        out["gather_bytes_per_peer"] = int(gatherer.send.numel() * 4)
    if prepare is not None:
        out["config"]["scene_prepared_on_device"] = prepare
    if launches:
        avg_ms = kern_ms / launches
        alg_per_launch = alg_bytes / launches
        alg_gbs = alg_per_launch / (avg_ms * 1e-3) / 1e9
        kernel = "vk::render_frames<%s>" % ("true" if args.target == "svo" else "false")
        # `roofline`: the bound that can bind this path is unfused FP32 vector arithmetic -- the kernel walks
        # 1-bit occupancy words that stay in cache (real HBM traffic is ~6 % of the algorithmic bytes), while every
        # step's floats must be the reference's own 22 operations.  work = 22 x (exact sum of both step planes of the
        # timed launches) + 60 x rays lane-ops; achieved = work / HIP-event kernel time; peak = 78.65 T lane-ops/s
        # (157.3 TFLOP/s FP32 vector peak / 2: no FMA under the parity contract); frac <= 1 by construction of a
        # kernel that executes those operations.  SURVEY 8d's algorithmic-byte figure (8 B per executed sample of
        # the reference layout) stays beside it as hbm_algorithmic_*: it is not a utilisation and passes 1.
        ops_per_launch = (OPS_PER_STEP * (sum_lo + sum_hi) + OPS_PER_RAY * rays_per_launch * args.steps) / launches
        achieved = ops_per_launch / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "valu", "achieved": round(achieved, 3), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s (unfused FP32)",
                "frac": round(achieved / VALU_PEAK_TLANEOPS, 5), "traffic": None, "kernel": kernel,
                "kernel_avg_ms": round(avg_ms, 4), "launches": launches,
                "lane_ops_per_launch": int(ops_per_launch), "ops_per_step": OPS_PER_STEP, "ops_per_ray": OPS_PER_RAY,
                "hbm_algorithmic_bytes_per_launch": int(alg_per_launch), "hbm_algorithmic_gbs": round(alg_gbs, 2),
                "hbm_algorithmic_frac_of_8tbs": round(alg_gbs / HBM_PEAK_GBS, 5)}
        sha = out["src_sha16"]
        key = f"config{args.config}:{args.mode}" + ("" if args.camera == "orbit" else f":{args.camera}") + \
              ("" if args.noise_log2_inv_p == 8 else f":p{args.noise_log2_inv_p}")
        pmc = pmc_entry(key, rays_per_launch, sha) if not split or world == 1 else None
        if pmc:
            roof["traffic"] = pmc["hbm_bytes_per_launch"]
            roof["traffic_source"] = pmc["source"] + " (rocprofv3 --pmc, separate run of this workload on this build; 2*FETCH_SIZE + WRITE_SIZE)"
            real_gbs = pmc["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
            roof["hbm_real_gbs"] = round(real_gbs, 1)
            roof["hbm_real_frac"] = round(real_gbs / HBM_PEAK_GBS, 4)
            pmc = dict(pmc, hbm_real_frac=real_gbs / HBM_PEAK_GBS)
            out["measured"] = {
                "binding_resource": binding_resource(pmc),
                "valu_issue_frac": pmc["valu_issue_frac"],
                "valu_issue_frac_def": "SQ_INSTS_VALU / (1024 SIMDs * kernel cycles / 2): share of the one-wave64-instruction-per-2-cycles issue peak",
                "valu_insts_per_launch": pmc.get("valu_insts_per_launch"),
                "hbm_real_gbs": round(real_gbs, 1), "hbm_real_frac": round(real_gbs / HBM_PEAK_GBS, 4),
                "l2_read_hit_rate": pmc["l2_read_hit_rate"], "waves_per_simd": pmc["waves_per_simd"],
                "active_lanes_avg": pmc["active_lanes_avg"], "ta_busy_frac": pmc.get("ta_busy_frac"),
                "pmc_kernel_avg_ms": pmc["kernel_avg_ms"],
                "rocprof_kernel_avg_ms_timed_steps": pmc.get("kernel_timed_avg_ms"),   # the trace's launches of the timed steps: agrees with roofline.kernel_avg_ms
                "src_sha16": pmc.get("src_sha16"), "lib_sha16": pmc.get("lib_sha16"), "git_head": pmc.get("git_head"), "source": pmc["source"]}
        out["roofline"] = roof

    # ---- end to end (untimed relative to `value`): the synchronous host-buffer entry point
    # volume_gpu_render into pinned memory -- the reference's contract is pixels in the locked texture
    # (main.h:38-42, main.cpp:269-271,290) -- for one step's frames, and the reference's own call shape:
    # one 400x300 frame per call (main.h:112, main.cpp:314)
    if rank == 0 and world == 1 and not args.no_end_to_end:
        try:
            import numpy as np
            import statistics
            e2e = {}
            frames = all_frames[args.warmup]
            g0 = gpus[args.warmup % len(gpus)]
            for label, names_ in (("rgba8_only", ("rgba8",)), ("five_planes", plane_names)):
                pinned = {p: vra.PinnedArray((B, rows, W), np.float32 if p == "depth" else np.uint32) for p in names_}
                arrays = {p: a.array for p, a in pinned.items()}
                ts = []
                for _ in range(4):
                    t1 = time.perf_counter()
                    g0.render_into(frames, arrays)
                    ts.append(time.perf_counter() - t1)
                ms = statistics.median(ts[1:]) * 1e3
                e2e[label] = {"ms_per_call": round(ms, 3), "mrays_per_s": round(B * rows * W / ms / 1e3, 1),
                              "pcie_gbs": round(len(names_) * 4 * B * rows * W / ms / 1e6, 2)}
                for a in pinned.values():
                    a.close()
            w1, h1 = 400, 300
            if args.camera == "startup":
                view = vra.look_at((0, 0, 0), vra.host.cam_dir_from_mouse(0, 0), (0, 0, -1))
                one = [vra.frame_from_camera(view, (0.0, 0.0, 0.0), size_hi, size_lo, w1, h1, naive=naive)]
            else:
                one = vra.orbit_frames([args.warmup * B % orbit_n], orbit_n, size_hi, size_lo, w1, h1, radius=args.radius, naive=naive)
            px = vra.PinnedArray((1, h1, w1), np.uint32)
            ts = []
            for _ in range(220):
                t1 = time.perf_counter()
                g0.render_into(one, {"rgba8": px.array})
                ts.append(time.perf_counter() - t1)
            ts = sorted(ts[20:])
            e2e["reference_call_shape_400x300"] = {"latency_us_median": round(ts[len(ts) // 2] * 1e6, 1), "latency_us_p90": round(ts[int(len(ts) * 0.9)] * 1e6, 1),
                                                   "mrays_per_s": round(w1 * h1 / ts[len(ts) // 2] / 1e6, 1), "calls": len(ts),
                                                   "what": "one 400x300 frame per volume_gpu_render call, rgba8 into pinned host memory, call to return"}
            px.close()
            e2e["entry_point"] = "volume_gpu_render (synchronous, outputs in pinned host memory)"
            e2e["frames_per_call"] = B
            out["end_to_end"] = e2e
        except Exception as e:      # reported, never fatal: this is not the measured path
            out["end_to_end"] = {"error": str(e)[:200]}

    # ---- CPU baseline: the oracle, one thread (the reference is single threaded), bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline and scene is not None:
        from __graft_entry__ import load_oracle
        ora = load_oracle()
        n_cpu = args.cpu_frames or max(1, min(B, int(round(40e6 / (W * H)))))      # ~40 M rays: 10-20 s of one core
        s0 = args.warmup + (-args.warmup) % len(gpus)          # a timed step that rendered names[0]
        frames = all_frames[s0 if s0 < len(all_frames) else args.warmup]
        kw = dict(low=scene.low, target=scene.target) if want_dense else dict(low=scene.low, svo=scene.svo)
        t1 = time.perf_counter()
        for i in range(n_cpu):
            ora.render_frame(frames[i % B], threads=1, **kw)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n_cpu * rows * W / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} frame(s) of a timed step of {names[0]} ({n_cpu * rows * W} rays), "
                                         f"scalar C oracle, gcc -O3 -mavx -ffp-contract=off, {cpu_s:.1f} s"}
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    for g in gpus:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
