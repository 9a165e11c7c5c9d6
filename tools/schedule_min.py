"""MIN search of a schedule.txt-style model on the device engine: optimum, time, node counts; the best solution is
checked against the oracle (every clause true, objective value = the optimum).
usage: schedule_min.py TASKS [SEED] [LANES]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from csolve_amd import problems
from csolve_amd.solver import Search, solve_root
from csolve_amd.parallel import LaneSearch

T = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1; lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
text = problems.schedule(T, seed)
model = solve_root(text)
t0 = time.perf_counter()
if lanes > 1:
    engines = [Search(model, 1 << 21, 1 << 16) for _ in range(lanes)]
    ls = LaneSearch(engines, model.objective)
    st = ls.run(model.root_state())
    row = ls.best_solution()
else:
    s = Search(model, 1 << 22, 1 << 16)
    s.put(model.root_state())
    st = s.run(0)
    while not st["done"]:
        st = s.run(4096)
        print(f"  {time.perf_counter() - t0:7.1f}s nodes {st['nodes']:,} best {st['best']} pool {st['pool']}", flush=True)
    row = s.best_solution()
dt = time.perf_counter() - t0
ok = None
if row is not None:
    from oracle.cs_oracle import Model as OModel, Oracle
    om = OModel.parse(text)
    dom = np.stack([row, row], axis=1).astype(np.int32)
    om.set_domains(dom)
    om.index()
    v = Oracle(om).eval(om.root)
    ok = tuple(v) == (1, 1) and int(row[model.objective_var]) == st["best"]
print(json.dumps({"tasks": T, "seed": seed, "lanes": lanes, "best": st["best"], "done": st["done"], "seconds": round(dt, 3),
                  "nodes": st["nodes"], "cuts": st["cuts"], "solution_checked_by_oracle": ok}))
