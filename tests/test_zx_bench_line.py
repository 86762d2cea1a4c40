"""bench.py's JSON line on a small workload: the keys the driver's contract names, a roofline that is a bound
(0 < frac <= 1, achieved / peak), a CPU baseline from the oracle on one core, and nothing else on stdout.
(Collected late: it starts bench.py as a child process.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_carries_the_contract():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1", "--steps", "3", "--warmup", "1", "--cpu-frames", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end"):
        assert key in d, key
    assert d["metric"] == "primary Mrays/s" and d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    # value is rays of the timed steps over their wall time
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) <= 1e-3 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "valu" and 0.0 < roof["frac"] <= 1.0 and roof["peak"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["launches"] == 3 and roof["kernel_avg_ms"] > 0 and roof["kernel_avg_ms"] <= d["ms_per_step"] * 1.001
    assert "traffic" in roof                              # HBM bytes per launch from the counter file of this build, or null
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["unit"] == "Mrays/s" and cpu["sample"]
    assert d["value"] > 100 * cpu["value"]
    e2e = d["end_to_end"]
    assert "error" not in e2e and e2e["rgba8_only"]["mrays_per_s"] > 0 and e2e["reference_call_shape_400x300"]["latency_us_median"] > 0
