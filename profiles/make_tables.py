#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 5 from profiles/r03/bench_*.json (profiles/r03_sweep.sh) and profiles/pmc_r03.json
(profiles/r03_pmc_all.sh + profiles/summarize.py).  usage: python profiles/make_tables.py [dir of bench lines]"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
R = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "r03")
R2 = os.path.join(HERE, "r02")
rows = [("c0", "**headline**: dragon 1024³ octree, 16 × 4K, beam (orbit, radius 1.0)"), ("c0_naive", "headline, naive (pure octree traversal from the camera)"),
        ("c0_startup", "headline scene, the reference's start-up camera (camPos = 0, mouse 0,0), beam"), ("c0_startup_naive", "… naive"),
        ("c0_inside", "headline scene, camera inside the grid (orbit radius 0.25), beam"), ("c0_inside_naive", "… naive"),
        ("c0_fill", "headline scene, model fills the frame (orbit radius 0.5), beam"), ("c0_fill_naive", "… naive"),
        ("c1", "config 1: teapot 128³, 64 × 640×480"), ("c2", "config 2: dragon 512³, 64 × 1080p, beam"), ("c2_naive", "config 2, naive"),
        ("c3", "config 3: noise 1024³ dense, 2 × 4K, p = 1/256"), ("c3_naive", "config 3, naive"), ("c3_v16", "config 3, 16 × 4K per launch (the headline's launch size)"), ("c3_p64", "config 3, p = 1/64"), ("c3_p1024", "config 3, p = 1/1024"),
        ("c4", "config 4: galleon 2048³ octree, one 8K frame, 1 GPU"), ("c5", "config 5: stratocaster / dragon 1024³ alternating, 64 × 1080p")]
print("| Workload | Mrays/s (1 GPU, wall) | round 2 | kernel ms / launch | hit fraction | steps per ray (low + target) | `roofline.frac` (VALU) | algorithmic GB/s (of 8 TB/s) | CPU oracle, 1 core |")
print("|---|---|---|---|---|---|---|---|---|")
for k, name in rows:
    f = os.path.join(R, f"bench_{k}.json")
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    c, r = d["config"], d["roofline"]
    cpu = d.get("cpu_baseline", {}).get("value")
    old = os.path.join(R2, f"bench_{k}.json")
    prev = f"{json.load(open(old))['value']:,.0f}" if os.path.exists(old) else "—"
    print(f"| {name} | {d['value']:,.0f} | {prev} | {r['kernel_avg_ms']:.3f} | {c['hit_fraction']:.3f} | {c['mean_steps_lo']:.1f} + {c['mean_steps_hi']:.2f} | "
          f"{r['frac']:.3f} | {r['hbm_algorithmic_gbs']:,.0f} ({r['hbm_algorithmic_frac_of_8tbs']:.2f}) | {cpu if cpu is not None else '—'} |")
print()
pmc = json.load(open(os.path.join(HERE, "pmc_r03.json")))
print("| Workload | kernel | VALU issue | vector instructions / launch | real HBM bytes / launch | real HBM GB/s (of 8 TB/s) | L2 read hit rate | waves / SIMD | active lanes | TA busy |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, name in (("config0:beam", "headline, beam"), ("config0:naive", "headline, naive"), ("config0:beam:startup", "start-up camera, beam"),
                  ("config0:beam:inside", "inside the grid, beam"), ("config0:beam:fill", "model fills the frame, beam"), ("config0:naive:fill", "model fills the frame, naive"),
                  ("config2:beam", "config 2, beam"), ("config3:beam", "config 3 (noise)"), ("config4:beam", "config 4 (8K, 1 GPU)"), ("config5:beam", "config 5")):
    e = pmc.get(key)
    if not e:
        continue
    gbs = e["hbm_bytes_per_launch"] / (e["kernel_avg_ms"] * 1e-3) / 1e9
    print(f"| {name} | `{(e['kernel'] or '').replace('void vk::', '')}` | {e['valu_issue_frac']:.2f} | {e.get('valu_insts_per_launch', 0) / 1e6:,.0f} M | {e['hbm_bytes_per_launch'] / 1e9:.2f} GB | "
          f"{gbs:,.0f} ({gbs / 8000:.3f}) | {e['l2_read_hit_rate']:.2f} | {e['waves_per_simd']:.1f} | {e['active_lanes_avg']:.1f} | {e.get('ta_busy_frac', 0):.2f} |")
