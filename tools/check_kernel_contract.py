#!/usr/bin/env python3
"""Build-time contract of the hand-written loops in volly_kernels.h, checked on the gfx950
assembly the compiler actually produced (hipcc -S --cuda-device-only, same flags as the library):

  1. m0 belongs to the DDA loops: they keep their iteration number there across separate `asm`
     statements and the compiler is not told, so no other instruction of the translation unit may
     touch m0 (indexed registers, LDS-direct, messages, spills through m0 would corrupt the step
     counts and the iteration cap): every instruction that names m0 must lie between ASMSTART / ASMEND and be one
     of the loops' own forms;
  2. occupancy: the live frame kernels -- render_frames<true,false,TW> (octree) and <false,false,TW>
     (dense), TW = 8 and 32 -- stay at <= 64 vector and <= 96 scalar registers, i.e. 8 wavefronts per SIMD, and
     spill nothing.  The loops' nine temporaries are the fixed registers v47-v55 (declared as
     clobbers, so the compiler keeps nothing live in them across the loops); if its own allocation
     grew past v55 the kernel would still be correct, and this check says when it passes 64;
  3. the loops are still there (their `v47` .. `v55` text shows between ASMSTART / ASMEND).

usage: check_kernel_contract.py [extra hipcc flags ...]      exit status 0 = contract holds
Run by `make gpu`, by every build of `make variants`, and by tests/test_kernel_contract.py."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "volume-renderer_amd", "csrc", "volly_gpu.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include")]
OURS = (re.compile(r"^s_mov_b32 m0, 0$"), re.compile(r"^s_mov_b32 s\d+, m0$"), re.compile(r"^s_add_u32 m0, m0, 1$"),
        re.compile(r"^s_cmp_ge_u32 m0, s\d+$"), re.compile(r"^s_add_u32 s\d+, m0, \d+$"))
LIVE = ("_ZN2vk13render_framesILb1ELb0ELi8EEE", "_ZN2vk13render_framesILb0ELb0ELi8EEE",       # beam frames: 8 x 8 tiles
        "_ZN2vk13render_framesILb1ELb0ELi32EEE", "_ZN2vk13render_framesILb0ELb0ELi32EEE")     # naive-only launches: 32 x 2 tiles
MAX_VGPR, MAX_SGPR = 64, 96


def device_assembly(extra=()):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "volly_gpu.s")
        r = subprocess.run([hipcc] + FLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", out, SRC],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stderr[-3000:])
        return open(out).read()


def check(text, probe_waves=False, expect_loops=True):
    """returns (problems, facts)"""
    problems, facts = [], {}
    uses, counters = [], 0
    in_asm, loop_temps = False, 0
    for line in text.splitlines():
        t = line.split(";")[0].strip()
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        if in_asm and re.search(r"\bv(4[7-9]|5[0-5])\b", t):
            loop_temps += 1
        if not re.search(r"\bm0\b", t) or t.startswith("."):
            continue
        if in_asm and (any(p.match(t) for p in OURS) or re.match(r"^s_mov_b32 m0, s\d+$", t)):
            counters += 1               # (s_mov_b32 m0, sN: a loop re-entered at iteration sN, bricks_loop_asm)
        else:
            uses.append(t)
    facts["m0_counter_instructions"] = counters
    facts["loop_instructions_on_v47_v55"] = loop_temps
    if expect_loops and (counters == 0 or loop_temps == 0):
        problems.append("the assembly loops are gone: drop this check with them")
    if uses:
        problems.append("m0 is used by code the loops do not know about: " + "; ".join(uses[:5]))
    # register counts from the code-object metadata
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        name, sgpr, vgpr, spill = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        for live in LIVE:
            if name.startswith(live):
                facts[live] = {"vgpr": vgpr, "sgpr": sgpr, "vgpr_spills": spill}
                if probe_waves:
                    continue
                if vgpr > MAX_VGPR:
                    problems.append(f"{name}: {vgpr} vector registers > {MAX_VGPR}: fewer than 8 wavefronts per SIMD")
                if sgpr > MAX_SGPR:
                    problems.append(f"{name}: {sgpr} scalar registers > {MAX_SGPR}: fewer than 8 wavefronts per SIMD")
                if spill:
                    problems.append(f"{name}: {spill} vector registers spilled")
    for live in LIVE:
        if live not in facts:
            problems.append(f"kernel {live}... not found in the assembly")
    return problems, facts


def main():
    extra = sys.argv[1:]
    problems, facts = check(device_assembly(extra), probe_waves=any("VK_COUNT" in f for f in extra),
                            expect_loops=not any(f.replace(" ", "") == "-DVK_FAST_DDA=0" for f in extra))
    print("kernel contract:", facts)
    for p in problems:
        print("CONTRACT BROKEN:", p, file=sys.stderr)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
