import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from csolve_amd import problems
from csolve_amd.solver import Search, solve_root
for name, text in (("ref_wcet", open("tests/golden/problems/ref_wcet.txt").read()),
                   ("ref_schedule", open("tests/golden/problems/ref_schedule.txt").read()),
                   ("schedule6", problems.schedule(6, 1)), ("schedule8", problems.schedule(8, 1)), ("schedule10", problems.schedule(10, 1))):
    model = solve_root(text)
    s = Search(model, 1 << 20, 1 << 16)
    s.put(model.root_state())
    torch.cuda.synchronize(); t0 = time.time()
    st = s.run()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(name, "best", st["best"], "nodes", st["nodes"], "iterations", st["iterations"], f"{dt*1e3:.1f} ms", f"{st['nodes']/dt:.3g} nodes/s", "tree clauses", model.device_info()["tree_clauses"])
