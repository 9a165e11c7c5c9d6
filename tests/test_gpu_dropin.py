"""GPU test of the drop-in boundary: the reference's OWN search driver (csolve.c, strategy.c,
objective.c, util.c, normalize.c, parser_support.c ... compiled from the reference's sources in the
authoring container, shipped as oracle/_ref/csolve_ref_dropin) linked against libcsolve_dropin.so,
so that its propagate()/propagate_clauses()/eval_*() calls run on the GPU.  The search traces
(CALLS, CUTS, solutions, incumbent) must equal the goldens of the all-CPU reference."""
import json
import os
import re
import subprocess

import pytest

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "oracle", "_ref", "csolve_ref_dropin")

DET = ["-c", "false", "-f", "false", "-r", "0"]
CASES = [("queens4", DET), ("queens8", DET), ("queens8_all", DET), ("queens16", DET),
         ("ref_schedule", ["-c", "false"]), ("schedule6_s1", ["-c", "false", "-f", "false"])]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/csolve_ref_dropin not built (needs the reference tree)")
@pytest.mark.parametrize("name,flags", CASES, ids=[c[0] for c in CASES])
def test_reference_driver_on_gpu_propagator(name, flags):
    stats = json.load(open(golden("solve_stats.json")))
    want = next(r for r in stats if r["problem"] == name and r["flags"] == flags)
    p = subprocess.run([BIN, "solve", golden("problems", name + ".txt")] + flags, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr
    got = json.loads(re.search(r"@STATS (\{.*\})", p.stdout).group(1))
    used = json.loads(re.search(r"@DROPIN (\{.*\})", p.stdout).group(1))
    assert used["propagate_clauses"] >= got["calls"] and used["propagate"] >= 2
    for k in ("calls", "cuts", "solutions", "best"):
        assert got[k] == want[k], (k, got[k], want[k])
    sols = re.findall(r"SOLUTION: (.*?)BEST: (-?\d+)", p.stdout)
    assert len(sols) == want["n_solution_lines"]
    if sols and "last_solution" in want:
        last = {k.strip(): int(v) for k, v in (kv.split(" = ") for kv in sols[-1][0].rstrip(", ").split(", "))}
        assert last == want["last_solution"]
