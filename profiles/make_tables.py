#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 5a from profiles/r02/bench_*.json and profiles/pmc_r02.json."""
import json, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r02")
rows = [("c0", "**headline**: dragon 1024³ octree, 16 × 4K, beam"), ("c0_naive", "headline, naive (pure octree traversal from the camera)"),
        ("c1", "config 1: teapot 128³, 64 × 640×480"), ("c2", "config 2: dragon 512³, 64 × 1080p, beam"), ("c2_naive", "config 2, naive"),
        ("c3", "config 3: noise 1024³ dense, 2 × 4K, p = 1/256"), ("c3_p64", "config 3, p = 1/64"), ("c3_p1024", "config 3, p = 1/1024"),
        ("c4", "config 4: galleon 2048³ octree, one 8K frame, 1 GPU"), ("c5", "config 5: stratocaster / dragon 1024³ alternating, 64 × 1080p")]
print("| Workload | Mrays/s (1 GPU, wall) | ms per step | kernel ms / launch | steps per ray (low + target) | algorithmic GB/s (frac of 8 TB/s) | CPU oracle, 1 core |")
print("|---|---|---|---|---|---|---|")
for k, name in rows:
    d = json.load(open(os.path.join(R, f"bench_{k}.json")))
    c, r = d["config"], d["roofline"]
    cpu = d.get("cpu_baseline", {}).get("value")
    print(f"| {name} | {d['value']:,.0f} | {d['ms_per_step']:.3f} | {r['kernel_avg_ms']:.3f} | {c['mean_steps_lo']:.1f} + {c['mean_steps_hi']:.2f} | "
          f"{r['achieved']:,.0f} ({r['frac']:.2f}) | {cpu if cpu is not None else '—'} |")
print()
pmc = json.load(open(os.path.join(os.path.dirname(R), "pmc_r02.json")))
print("| Workload | VALU issue | real HBM bytes / launch | real HBM GB/s (of 8 TB/s) | L2 read hit rate | waves / SIMD | active lanes | TA busy |")
print("|---|---|---|---|---|---|---|---|")
for key, bk, name in (("config0:beam", "c0", "headline, beam"), ("config0:naive", "c0_naive", "headline, naive"), ("config2:beam", "c2", "config 2, beam"),
                      ("config2:naive", "c2_naive", "config 2, naive"), ("config3:beam", "c3", "config 3 (noise)"), ("config4:beam", "c4", "config 4 (8K, 1 GPU)"),
                      ("config5:beam", "c5", "config 5")):
    e = pmc[key]
    m = json.load(open(os.path.join(R, f"bench_{bk}.json"))).get("measured", {})
    print(f"| {name} | {e['valu_issue_frac']:.2f} | {e['hbm_bytes_per_launch'] / 1e9:.2f} GB | {m.get('hbm_real_gbs', 0):,.0f} ({m.get('hbm_real_frac', 0):.3f}) | "
          f"{e['l2_read_hit_rate']:.2f} | {e['waves_per_simd']:.1f} | {e['active_lanes_avg']:.1f} | {e.get('ta_busy_frac', 0):.2f} |")
