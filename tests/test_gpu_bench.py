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
    r = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--instances", "16384", "--cpu-seconds", "1", "--no-queens128", "--no-search"])
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


SMALL_SEARCH = ["--search-queens", "11", "--search-steps", "1", "--search-time-limit", "0.5", "--search-record-schedule", "6"]


def _check_search_record(rec, ranks):
    """the `search` sub-record: the sharded search next to the propagation headline"""
    assert rec["ranks_seen"] == ranks
    q = rec["queens11_all"]
    assert q["solutions"] == 2680 and q["timeout"] is False and q["seconds_per_search"] > 0 and q["nodes_per_s"] > 0
    assert len(q["ranks"]) == ranks and len(q["nodes_per_rank"]) == ranks and sum(q["nodes_per_rank"]) == q["nodes"]
    for r in q["ranks"]:
        for k in ("idle_fraction", "exchange_seconds", "states_moved", "busy_seconds", "seed_seconds"):
            assert k in r, k
    big = rec["queens128_all"]
    assert big["timeout"] is True and big["nodes"] > 0 and big["time_limit_s"] == 0.5
    m = rec["schedule6_min"]
    assert m["best"] == 22 and m["timeout"] is False
    return q


def test_the_default_line_carries_the_search_record_at_one_gpu():
    r = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--instances", "8192", "--no-cpu"] + SMALL_SEARCH)
    assert r["n_gpus"] == 1 and r["scaling"] == "weak" and "roofline" in r
    rec = r["search"]
    assert rec["process_group"] is None and rec["launched_by"].startswith("a single process")
    _check_search_record(rec, 1)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two rank processes before
    anything touches the GPU and relays rank 0's single line (here both ranks share cuda:0 over gloo: the box has one GPU).
    The sharded search of the two ranks walks the one-rank tree."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--instances", "8192",
           "--comm", "gloo", "--same-device"] + SMALL_SEARCH
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["instances_per_gpu"] == 8192
    rec = r["search"]
    assert rec["process_group"] == "gloo" and rec["launched_by"].startswith("bench.py itself") and rec["same_device"] is True
    two = _check_search_record(rec, 2)
    one = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--instances", "8192", "--no-cpu"] + SMALL_SEARCH)["search"]["queens11_all"]
    assert (two["nodes"], two["cuts"], two["solutions"]) == (one["nodes"], one["cuts"], one["solutions"])


def test_a_failing_rank_fails_the_launcher():
    """rank 1 cannot select cuda:1 on a one-GPU box: the launcher must not report success (and must not hang)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a one-GPU box")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--instances", "4096", "--no-search", "--comm", "gloo"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode != 0


def test_bench_refuses_a_launchers_world_that_differs_from_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}, WORLD_SIZE="1"))
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
