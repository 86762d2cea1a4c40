"""The region loop keeps its iteration number in m0 (volly_kernels.h, trace_dense_regular_asm): the
compiler cannot be told, so this test reads the gfx950 assembly of the kernels and checks that no
other instruction in them touches m0."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = (re.compile(r"^s_mov_b32 m0, 0$"), re.compile(r"^s_mov_b32 s\d+, m0$"), re.compile(r"^s_add_u32 m0, m0, 1$"),
        re.compile(r"^s_cmp_ge_u32 m0, s\d+$"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_nothing_but_the_loop_counter_uses_m0(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "volly_gpu.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(out),
           os.path.join(ROOT, "volume-renderer_amd", "csrc", "volly_gpu.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    uses, counters = [], 0
    for line in out.read_text().splitlines():
        t = line.split(";")[0].strip()
        if not re.search(r"\bm0\b", t) or t.startswith("."):
            continue
        if any(p.match(t) for p in OURS):
            counters += 1
        else:
            uses.append(t)
    assert counters > 0, "the assembly loop is gone: drop this test with it"
    assert not uses, "m0 is used by code the loop does not know about: " + "; ".join(uses[:5])
