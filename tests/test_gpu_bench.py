"""GPU test of bench.py's contract: one JSON line with the fields the driver reads, the roofline and cpu_baseline
objects, on a reduced run (the sizes the driver uses are the defaults; this checks the plumbing, not the numbers)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, timeout=300):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py prints exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contracted_fields():
    r = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--instances", "16384", "--cpu-seconds", "1", "--no-queens128"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["warmup"] == 2 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "int32" and r["data"] == "synthetic"
    assert "workload" in r["config"] and r["config"]["forbidden_sets_precomputed"] is False
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and "cs_propagate_ne_shave" in roof["kernel"]
    assert roof["bytes_per_node_instance"] == 16 * 64 + 32
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "reference" and cpu["cores"] == 1 and cpu["value"] > 0 and "bit for bit" in cpu["sample"]
    assert r["value"] > 1000 * cpu["value"]


def test_bench_refuses_to_measure_one_gpu_under_the_label_of_two():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
