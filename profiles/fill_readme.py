#!/usr/bin/env python3
"""Fill the README's round-3 table from profiles/r03/bench_*.json (placeholders HEAD_BEAM ...)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def v(k):
    return f"{json.load(open(os.path.join(ROOT, 'profiles', 'r03', f'bench_{k}.json')))['value'] / 1e3:.1f}"
keys = {"HEAD_BEAM": "c0", "HEAD_NAIVE": "c0_naive", "START_BEAM": "c0_startup", "START_NAIVE": "c0_startup_naive", "INSIDE_BEAM": "c0_inside",
        "INSIDE_NAIVE": "c0_inside_naive", "FILL_BEAM": "c0_fill", "FILL_NAIVE": "c0_fill_naive", "C1": "c1", "C2": "c2", "C3": "c3", "C4": "c4", "C5": "c5"}
tmpl = os.path.join(ROOT, "profiles", "README.tmpl.md")
s = open(tmpl).read()
for name in sorted(keys, key=len, reverse=True):
    s = re.sub(r"\b%s\b" % name, v(keys[name]), s)
open(os.path.join(ROOT, "README.md"), "w").write(s)
