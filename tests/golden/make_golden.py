#!/usr/bin/env python3
"""Generates tests/golden/golden.npz and tests/golden/golden.json.

The reference has no tests, golden vectors or fixtures and cannot be
built here, so these vectors come from THIS repository's CPU oracle and host
pipeline (PARITY UNPINNED: they pin the oracle against regressions and give the
GPU path a fixed target; they are not reference outputs).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package, load_oracle  # noqa: E402
from golden_cases import CASES, render_case, scene_digest, crc  # noqa: E402


def main():
    vra, ora = load_package(), load_oracle()
    arrays, meta = {}, {"cases": {}, "scenes": {}}
    for name, case in CASES.items():
        planes, digest = render_case(vra, ora, case, backend="oracle")
        meta["cases"][name] = {"crc": {p: crc(a) for p, a in planes.items()},
                               "hits": int((planes["hit_index"] != 0xFFFFFFFF).sum()),
                               "steps_lo": int(planes["steps_lo"].sum()), "steps_hi": int(planes["steps_hi"].sum())}
        meta["scenes"][case["scene_key"]] = digest
        if case.get("store"):
            crop = case.get("crop")
            for p, a in planes.items():
                arrays[f"{name}/{p}"] = a[..., ::crop, ::crop] if crop else a
        print(name, meta["cases"][name])
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays")


if __name__ == "__main__":
    main()
