"""Latency of one node through the three host-buffer entries (plain, with clause trail, with cause trail) on states of
a queens walk.  usage: time_traced_one.py [N]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from csolve_amd import problems
from csolve_amd.solver import solve_root
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = solve_root(problems.queens(nq))
rng = np.random.default_rng(3)
# a walk: states of increasing depth
states, dom = [], np.ascontiguousarray(model.domains())
for depth in range(40):
    open_vars = np.flatnonzero(dom[:, 0] != dom[:, 1])
    if len(open_vars) == 0: break
    v = int(rng.choice(open_vars)); val = int(rng.integers(dom[v, 0], dom[v, 1] + 1))
    st, props, out = model.propagate_one(dom, v, val, val)
    states.append((dom.copy(), v, val, st))
    if st < 0: continue
    dom = out
print(f"queens-{nq}: {len(states)} nodes, {sum(1 for s in states if s[3] < 0)} inconsistent")
for name, fn in (("propagate_one", lambda d, v, x: model.propagate_one(d, v, x, x)),
                 ("propagate_one_causes", lambda d, v, x: model.propagate_one_causes(d, v, x, x, 16384)),
                 ("propagate_one_traced", lambda d, v, x: model.propagate_one_traced(d, v, x, x, 16384))):
    for d, v, x, _ in states[:5]: fn(d, v, x)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        for d, v, x, _ in states: fn(d, v, x)
    dt = (time.perf_counter() - t0) / (reps * len(states))
    print(f"  {name:24s} {dt * 1e6:7.1f} us per call")
