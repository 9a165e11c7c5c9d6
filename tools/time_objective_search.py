"""Whole MIN / MAX searches on one GPU: examples/wcet.txt, examples/schedule.txt and schedule.txt-style models.
usage: time_objective_search.py [lanes ...]   (default 1: one engine; N > 1: csolve_amd.parallel.LaneSearch)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from csolve_amd import problems
from csolve_amd.parallel import LaneSearch
from csolve_amd.solver import Search, solve_root
lane_counts = [int(a) for a in sys.argv[1:]] or [1]
for name, text in (("ref_wcet", open("tests/golden/problems/ref_wcet.txt").read()),
                   ("ref_schedule", open("tests/golden/problems/ref_schedule.txt").read()),
                   ("schedule6", problems.schedule(6, 1)), ("schedule8", problems.schedule(8, 1)), ("schedule10", problems.schedule(10, 1))):
    model = solve_root(text)
    for lanes in lane_counts:
        engines = [Search(model, 1 << 20, 1 << 16) for _ in range(lanes)]
        torch.cuda.synchronize(); t0 = time.time()
        if lanes == 1:
            engines[0].put(model.root_state())
            st = engines[0].run()
        else:
            st = LaneSearch(engines, model.objective).run(model.root_state())
        torch.cuda.synchronize(); dt = time.time() - t0
        print(name, "lanes", lanes, "best", st["best"], "nodes", st["nodes"], "iterations", st["iterations"], f"{dt*1e3:.1f} ms",
              f"{st['nodes']/dt:.3g} nodes/s", "tree clauses", model.device_info()["tree_clauses"], flush=True)
