"""bench.py's multi-GPU launch path on the CPU (no GPU needed: --dry-run renders nothing).
`python bench.py --gpus N` must start N ranks itself when no launcher did, must run as one of N
ranks under a torch.distributed.run-style environment, and must refuse WORLD_SIZE != --gpus."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_by_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["steps"] == 2 and out["warmup"] == 1
    assert out["max_over_ranks_probe"] == 2.0          # the MAX all-reduce saw rank 1's value


def test_as_ranks_of_a_launcher():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True,
                              env=_clean_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert _json_line(outs[0][0])["n_gpus"] == 2
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints the JSON line


def test_world_size_must_equal_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=_clean_env(WORLD_SIZE="1", RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=_clean_env(WORLD_SIZE="4", RANK="0"))
    assert r.returncode == 2


def test_single_rank_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], capture_output=True, text=True, timeout=120, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_ranks_fall_back_to_gloo_when_rccl_cannot_start():
    """Without a GPU the "nccl" (RCCL) process group cannot be brought up: every rank sees the failure, the
    ranks meet again on gloo, and the JSON line names the backend that ran and why."""
    # (GPUs hidden, so that the failure is the same on every rank wherever this runs)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and "nccl failed" in out["backend_note"]
    assert out["max_over_ranks_probe"] == 2.0
