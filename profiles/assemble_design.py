"""Assemble DESIGN.md from profiles/design_src/*.md and the measured tables.

The text pieces are hand-written; the three tables of section 5 come from the bench lines and
PMC summaries under profiles/<round>/ (make_tables.py + the end_to_end blocks), so that the
numbers in DESIGN.md are the files' numbers.  Usage: python profiles/assemble_design.py [profiles/r03]
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(HERE, "r03")
SRC = os.path.join(HERE, "design_src")

E2E_ROWS = (
    ("c0", "headline (16 × 4K per call)"),
    ("c0_fill", "model fills the frame"),
    ("c0_startup", "start-up camera"),
    ("c1", "config 1 (64 × 640×480)"),
    ("c2", "config 2 (64 × 1080p)"),
    ("c4", "config 4 (one 8K frame)"),
    ("c5", "config 5"),
)


def end_to_end_table():
    rows = []
    for key, name in E2E_ROWS:
        path = os.path.join(R, f"bench_{key}.json")
        if not os.path.exists(path):
            continue
        e = json.load(open(path)).get("end_to_end")
        if not e or "error" in e:
            continue
        a, b, c = e["rgba8_only"], e["five_planes"], e["reference_call_shape_400x300"]
        rows.append(
            f"| {name} | {a['mrays_per_s']:,.0f} Mrays/s ({a['ms_per_call']:.1f} ms, {a['pcie_gbs']:.0f} GB/s) "
            f"| {b['mrays_per_s']:,.0f} Mrays/s ({b['ms_per_call']:.1f} ms, {b['pcie_gbs']:.0f} GB/s) "
            f"| {c['latency_us_median']:.0f} µs (p90 {c['latency_us_p90']:.0f}) = {c['mrays_per_s']:,.0f} Mrays/s |")
    head = ("| Workload | rgba8 only, one step's frames per call | all five planes "
            "| one 400×300 frame per call, rgba8 (call to return) |\n|---|---|---|---|\n")
    return head + "\n".join(rows)


def main():
    pieces = [open(os.path.join(SRC, f)).read() for f in sorted(os.listdir(SRC)) if f.endswith(".md")]
    head, sec4, launch, sec4b, sec5, sec678, sec9 = pieces
    tabs = subprocess.run([sys.executable, os.path.join(HERE, "make_tables.py"), R],
                          capture_output=True, text=True, check=True).stdout.split("\n\n")
    t1 = tabs[0].strip()
    t2 = tabs[1].strip() if len(tabs) > 1 else ""
    sec5 = sec5.replace("@@TABLE1@@", t1).replace("@@TABLE2@@", t2).replace("@@TABLE3@@", end_to_end_table())
    doc = head + "\n" + sec4 + launch + "\n" + sec4b + "\n" + sec5 + "\n" + sec678 + "\n" + sec9
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(doc)
    print(len(doc.split("\n")), "lines")


if __name__ == "__main__":
    main()
