"""Latency of the drop-in's host-buffer entries: one node (csgpu_propagate_one) and every value of a variable
(csgpu_propagate_values) on the root state, through ctypes.  usage: time_propagate_one.py"""
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, ".")
from csolve_amd import problems, _lib
from csolve_amd.solver import solve_root
lib = _lib.load_library()
for nq in (16, 64, 128):
    model = solve_root(problems.queens(nq))
    root = np.ascontiguousarray(model.domains())
    for _ in range(50): model.propagate_one(root, 0, 1, 1)
    t0 = time.perf_counter()
    N = 2000
    for i in range(N): st, props, out = model.propagate_one(root, i % nq, 1 + i % nq, 1 + i % nq)
    dt = time.perf_counter() - t0
    outs = np.empty((nq, nq, 2), dtype=np.int32)
    res = np.empty((nq, 4), dtype=np.int32)
    for width in (2, 8, nq):
        vals = np.arange(1, width + 1, dtype=np.int32)
        for _ in range(20):
            assert lib.csgpu_propagate_values(model._h, root.ctypes.data, 0, vals.ctypes.data, width, outs.ctypes.data, res.ctypes.data) == 0
        t1 = time.perf_counter()
        M = 500
        for i in range(M):
            lib.csgpu_propagate_values(model._h, root.ctypes.data, i % nq, vals.ctypes.data, width, outs.ctypes.data, res.ctypes.data)
        dv = time.perf_counter() - t1
        print(f"queens-{nq}: csgpu_propagate_values over {width} values: {dv / M * 1e6:.1f} us")
    print(f"queens-{nq}: {dt / N * 1e6:.1f} us per csgpu_propagate_one (through ctypes)")
