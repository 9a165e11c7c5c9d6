"""SURVEY 8f-1 measured: the search below a prefix state with the model as it is and with the model specialised for
that prefix (csgpu_model_specialize + normalize + finalize): same results, shorter lists -- what does it buy?
usage: time_specialize.py [TASKS] [STEPS ...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from csolve_amd import problems
from csolve_amd.solver import Search, solve_root

tasks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
depths = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
model = solve_root(problems.schedule(tasks, 1))


def prefix_state(steps, seed=5):
    rng = np.random.default_rng(seed)
    state = model.root_state()
    for _ in range(steps):
        dom = state[0].cpu().numpy()
        open_vars = [v for v in range(model.n_vars) if dom[v, 0] != dom[v, 1] and v != model.objective_var]
        if not open_vars:
            break
        v = int(rng.choice(open_vars))
        for value in range(int(dom[v, 0]), int(dom[v, 1]) + 1):
            nodes = torch.tensor([[v, value, value, 0]], dtype=torch.int32, device="cuda")
            out, res = model.propagate(state, nodes)
            if int(res[0, 0]) >= 0:
                state = out[:1].contiguous()
                break
    return state


for steps in depths:
    prefix = prefix_state(steps)
    t0 = time.perf_counter()
    special = model.specialize(prefix)
    t_spec = time.perf_counter() - t0
    row = {"tasks": tasks, "prefix_assignments": steps, "specialize_seconds": round(t_spec, 4),
           "adjacency_entries": [model.device_info()["adjacency_entries"], special.device_info()["adjacency_entries"]]}
    for name, m in (("model", model), ("specialised", special)):
        s = Search(m, 1 << 22, 1 << 17)
        s.put(prefix)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = s.run(1 << 40)
        torch.cuda.synchronize()
        row[name] = {"seconds": round(time.perf_counter() - t0, 4), "nodes": st["nodes"], "best": st["best"], "iterations": st["iterations"]}
        s.close()
    print(json.dumps(row), flush=True)
