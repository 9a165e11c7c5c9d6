import sys, time, numpy as np
sys.path.insert(0, ".")
from csolve_amd import problems
from csolve_amd.solver import solve_root
for nq in (16, 128):
    model = solve_root(problems.queens(nq))
    root = model.domains()
    for _ in range(50): model.propagate_one(root, 0, 1, 1)
    t0 = time.perf_counter()
    N = 2000
    for i in range(N): st, props, out = model.propagate_one(root, i % nq, 1 + i % nq, 1 + i % nq)
    dt = time.perf_counter() - t0
    print(f"queens-{nq}: {dt / N * 1e6:.1f} us per csgpu_propagate_one (through ctypes)")
