#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/<workload key>/ (written by profiles/run_pmc.sh) into
profiles/<tag>/:
  <key>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of that workload
  pmc_summary.csv          per-launch mean of every counter of the frame kernel, one block per workload,
                           followed by the derived figures
  and profiles/pmc_<tag>.json, which bench.py attaches to its JSON line (roofline.traffic, `measured`) --
  each entry stamped with the source hash recorded in the library the counters were taken on (src_sha16, from
  the bench line of the trace run; lib_sha16 = that .so file's own hash) and the git head at the time of condensing; bench.py ignores entries of another build.

Derived figures (per launch of vk::render_frames<...>; a launch of vk::render_frames_packed includes the
vk::continue_rays_packed passes that follow it):
  cycles            = GRBM_GUI_ACTIVE / 8                       (the counter sums the 8 XCDs)
  hbm_bytes         = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      (MI355X_MICROARCH.md, HBM: gfx950 tallies a
                      128-B read request as 64 B -- doubled; WRITE_SIZE exact; both are taken at the L2's
                      memory side, so Infinity-Cache hits are included: an upper bound on HBM bytes)
  valu_issue_frac   = SQ_INSTS_VALU / (1024 SIMDs * cycles / 2) (share of one wave64 VALU instruction per 2 cycles)
  l2_read_hit_rate  = 1 - TCC_EA0_RDREQ_sum / TCC_READ_sum      (reads only: every read miss leaves the L2 as one
                      memory-side read request; the store segments, which dominate TCC_REQ, are left out)
  waves_per_simd    = 4 * SQ_WAVE_CYCLES / (1024 * cycles)      (SQ_WAVE_CYCLES counts in units of 4 cycles)
  active_lanes_avg  = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU
usage: python profiles/summarize.py <tag>"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
summary, table = {}, []
import subprocess, re
try:
    git_head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    if subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "volume-renderer_amd", "include"], capture_output=True, text=True).stdout.strip():
        git_head += "+uncommitted"
except Exception:
    git_head = None
for kdir in sorted(glob.glob(os.path.join(src, "*", ""))):
    key = os.path.basename(kdir.rstrip("/"))
    stats = glob.glob(os.path.join(kdir, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(dst, f"{key}_kernel_stats.csv"))
    # a launch = one dispatch of the frame kernel plus, for the packed kernel, the passes that continue its parked rays
    kern_ms, total_ns, first_calls = None, 0.0, 0
    for r in csv.DictReader(open(stats[0])):
        if "render_frames" in r["Name"]:
            if kern_ms is None:
                kname = r["Name"].split("(")[0]
                first_calls = int(r["Calls"])
                kern_ms = 0.0
            total_ns += float(r["TotalDurationNs"])
        elif "continue_rays_packed" in r["Name"]:
            total_ns += float(r["TotalDurationNs"])
    if kern_ms is not None and first_calls:
        kern_ms = total_ns / first_calls / 1e6
    # the same kernel launch by launch (rocprofv3's kernel trace): the stats file's average includes the warm-up launches,
    # which run before the clocks have ramped; bench.py's HIP events bracket the timed steps only, so the figure that
    # must agree with `roofline.kernel_avg_ms` is the average over the launches of the timed steps
    timed_ms, per_launch = None, []
    traces = glob.glob(os.path.join(kdir, "trace", "**", "*kernel_trace.csv"), recursive=True)
    args_txt = open(os.path.join(kdir, "bench_args.txt")).read() if os.path.exists(os.path.join(kdir, "bench_args.txt")) else ""
    m_steps, m_warm = re.search(r"--steps (\d+)", args_txt), re.search(r"--warmup (\d+)", args_txt)
    n_steps, n_warm = int(m_steps.group(1)) if m_steps else 20, int(m_warm.group(1)) if m_warm else 5
    if traces:
        rows = sorted(csv.DictReader(open(traces[0])), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if "render_frames" in r["Kernel_Name"]:
                per_launch.append(ms)
            elif "continue_rays_packed" in r["Kernel_Name"] and per_launch:
                per_launch[-1] += ms
        timed = per_launch[n_warm:n_warm + n_steps]
        if len(timed) == n_steps:
            timed_ms = sum(timed) / n_steps
        with open(os.path.join(dst, f"{key}_kernel_launches.csv"), "w") as o:
            o.write("launch,kernel_ms,what\n")
            for i, ms in enumerate(per_launch):
                what = "warm-up" if i < n_warm else "timed step" if i < n_warm + n_steps else "after the timed region (plane sums for the bench line)"
                o.write(f"{i + 1},{ms:.4f},{what}\n")
    bench = {}
    try:
        bench = json.loads(open(os.path.join(kdir, "bench_line.json")).readline())
        json.dump(bench, open(os.path.join(dst, f"{key}_bench_under_rocprof.json"), "w"))
    except Exception:
        pass
    c = {}
    for f in sorted(glob.glob(os.path.join(kdir, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        acc, launches = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "render_frames" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                launches[r["Counter_Name"]] += 1
            elif "continue_rays_packed" in r["Kernel_Name"]:      # the later passes of a packed launch count towards that launch
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
        for name, v in acc.items():
            if launches[name]:
                c[name] = (v / launches[name], launches[name])
    if not c:
        continue
    g = lambda n: c[n][0] if n in c else None
    d = {}
    cycles = g("GRBM_GUI_ACTIVE") / 8 if g("GRBM_GUI_ACTIVE") else None
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["hbm_bytes_per_launch"] = int((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
    if cycles and g("SQ_INSTS_VALU"):
        d["kernel_cycles"] = round(cycles)
        d["valu_issue_frac"] = round(g("SQ_INSTS_VALU") / (1024 * cycles / 2), 4)
    if g("TCC_READ_sum") and g("TCC_EA0_RDREQ_sum") is not None:
        d["l2_read_hit_rate"] = round(max(0.0, 1 - g("TCC_EA0_RDREQ_sum") / g("TCC_READ_sum")), 4)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        d["l2_hit_rate_all_requests"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if cycles and g("SQ_WAVE_CYCLES"):
        d["waves_per_simd"] = round(4 * g("SQ_WAVE_CYCLES") / (1024 * cycles), 2)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
        d["active_lanes_avg"] = round(g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU"), 1)
    if cycles and g("TA_BUSY_avr"):
        d["ta_busy_frac"] = round(g("TA_BUSY_avr") / cycles, 3)
    d["kernel_avg_ms"] = round(kern_ms, 4) if kern_ms else None
    d["kernel_timed_avg_ms"] = round(timed_ms, 4) if timed_ms else None      # launches of the timed steps only
    d["hip_event_kernel_avg_ms_same_run"] = bench.get("roofline", {}).get("kernel_avg_ms")
    d["kernel"] = kname if kern_ms else None
    if kern_ms and "hbm_bytes_per_launch" in d:
        d["hbm_real_gbs_under_rocprof"] = round(d["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, 1)
    cfg = bench.get("config", {})
    if g("SQ_INSTS_VALU"):
        d["valu_insts_per_launch"] = int(g("SQ_INSTS_VALU"))
    # the build the counters were taken on: bench.py drops `measured` when the library it loads has another hash
    d["src_sha16"] = bench.get("src_sha16")      # hash of the kernel sources the library records: what bench.py matches
    d["lib_sha16"] = bench.get("lib_sha16")      # hash of the .so file of that run (differs between builds of one source)
    d["git_head"] = git_head
    d["rays_per_launch"] = cfg.get("rays_per_step")
    d["workload"] = cfg.get("workload")
    d["source"] = f"profiles/{tag}/pmc_summary.csv [{key}]"
    d["bench_args"] = open(os.path.join(kdir, "bench_args.txt")).read().strip() if os.path.exists(os.path.join(kdir, "bench_args.txt")) else None
    # key of bench.py's lookup: config<N>:<mode>[:p<n>]
    summary[key.replace("_", ":")] = d
    for name in sorted(c):
        table.append((key, name, c[name][1], f"{c[name][0]:.1f}"))
    for name in sorted(d):
        if isinstance(d[name], (int, float)):
            table.append((key, "derived:" + name, "", d[name]))
with open(os.path.join(dst, "pmc_summary.csv"), "w") as o:
    o.write("workload,counter,launches,mean_per_launch\n")
    for r in table:
        o.write(",".join(str(x) for x in r) + "\n")
json.dump(summary, open(os.path.join(root, "profiles", f"pmc_{tag}.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, {n: v[n] for n in v if n not in ("workload", "source", "bench_args")})
