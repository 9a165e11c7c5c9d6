"""Latency of the drop-in's single-node calls (csgpu_propagate_one_causes on the states of queens walks), with the
resident server and, in a second process with CSGPU_SERVER=0, as one launch per call; the library's own decomposition
of the host's time (csgpu_debug_one_timing).     usage: time_dropin_call.py [N] [--child]"""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")


def measure(nq):
    from csolve_amd import problems, _lib
    from csolve_amd.solver import solve_root
    model = solve_root(problems.queens(nq))
    rng = np.random.default_rng(3)
    states, dom = [], np.ascontiguousarray(model.domains())
    for walk in range(6):  # several dives: shallow and deep states
        dom = np.ascontiguousarray(model.domains())
        for depth in range(nq):
            open_vars = np.flatnonzero(dom[:, 0] != dom[:, 1])
            if len(open_vars) == 0:
                break
            v = int(rng.choice(open_vars)); val = int(rng.integers(dom[v, 0], dom[v, 1] + 1))
            st, props, out, _ = model.propagate_one_causes(dom, v, val, val, 2048)
            states.append((dom.copy(), v, val))
            if st < 0:
                break
            dom = out
    lat = []
    for _ in range(3):
        for d, v, x in states:
            t0 = time.perf_counter()
            model.propagate_one_causes(d, v, x, x, 2048)
            lat.append(time.perf_counter() - t0)
    L = _lib.load_library()
    sec = (C.c_double * 4)(); calls = C.c_uint64(); starts = C.c_uint64()
    L.csgpu_debug_one_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.csgpu_debug_one_timing(model._h, sec, C.byref(calls), C.byref(starts))
    lat = np.sort(np.array(lat)) * 1e6
    n = max(1, calls.value)
    return {"queens": nq, "nodes": len(states), "calls_timed": len(lat), "median_us": float(lat[len(lat) // 2]),
            "p10_us": float(lat[len(lat) // 10]), "p90_us": float(lat[len(lat) * 9 // 10]), "mean_us": float(lat.mean()),
            "library_us_per_call": [round(1e6 * s / n, 3) for s in sec], "library_calls": calls.value, "server_starts": starts.value,
            "server": os.environ.get("CSGPU_SERVER", "1") != "0"}


if __name__ == "__main__":
    nq = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    if "--child" in sys.argv:
        print("@R " + json.dumps(measure(nq)))
    else:
        for server in ("1", "0"):
            p = subprocess.run([sys.executable, __file__, str(nq), "--child"], env=dict(os.environ, CSGPU_SERVER=server),
                               capture_output=True, text=True)
            rec = [ln[3:] for ln in p.stdout.splitlines() if ln.startswith("@R ")]
            print(rec[0] if rec else "FAILED: " + p.stderr[-600:])
