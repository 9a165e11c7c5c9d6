"""Latency of heavy single nodes (assignments on a bound of the root domains of queens-N: every other variable moves)
through the plain and the tracing single-node entries.  usage: time_heavy_one.py [N]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from csolve_amd import problems
from csolve_amd.solver import solve_root
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = solve_root(problems.queens(nq))
root = np.ascontiguousarray(model.domains())
cases = [(v, x) for v in (0, nq // 2, nq - 1) for x in (1, nq)]
for name, fn in (("propagate_one", lambda v, x: model.propagate_one(root, v, x, x)),
                 ("propagate_one_causes", lambda v, x: model.propagate_one_causes(root, v, x, x, 16384)),
                 ("propagate_one_traced", lambda v, x: model.propagate_one_traced(root, v, x, x, 16384))):
    for v, x in cases: r = fn(v, x)
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        for v, x in cases: r = fn(v, x)
    dt = (time.perf_counter() - t0) / (reps * len(cases))
    print(f"queens-{nq} {name:24s} {dt * 1e6:8.1f} us per call (status {r[0]}, props {r[1]})")
for k in (7, 4, 3, 2, 1):
    if not model.qualifies(k): continue
    model.set_kernel(k)
    for v, x in cases: model.propagate_one(root, v, x, x)
    t0 = time.perf_counter()
    for _ in range(50):
        for v, x in cases: r = model.propagate_one(root, v, x, x)
    print(f"queens-{nq} propagate_one, kernel {k}: {(time.perf_counter() - t0) / (50 * len(cases)) * 1e6:8.1f} us (props {r[1]})")
