"""schedule-N MIN on the device engine under the reference's strategy options (-o order, -f prefer failing): time and nodes.
usage: time_strategies.py [TASKS]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csolve_amd import problems
from csolve_amd.solver import Search, solve_root
tasks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model = solve_root(problems.schedule(tasks, 1))
names = {0: "none", 1: "smallest-domain", 2: "largest-domain", 3: "smallest-value", 4: "largest-value"}
for order in [int(a) for a in sys.argv[2:]] or (1, 0, 2, 3, 4):
    for prefer in (0, 1):
        s = Search(model, 1 << 24, 1 << 21)
        s.set_strategy(order, bool(prefer))
        s.put(model.root_state())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = s.run(1 << 40)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"tasks": tasks, "order": names[order], "prefer_failing": bool(prefer), "seconds": round(dt, 3), "best": st["best"],
                          "nodes": st["nodes"], "iterations": st["iterations"], "done": st["done"]}), flush=True)
        s.close()
