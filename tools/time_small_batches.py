"""Duration of the general kernel on small batches of a schedule.txt-style model: the latency floor that bounds
an iteration of a MIN / MAX search (usage: time_small_batches.py [tasks])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
import bench
from csolve_amd import problems
from csolve_amd.solver import solve_root
tasks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = solve_root(problems.schedule(tasks, 1))
states_in, nodes, _ = bench.make_instances(model, 1 << 16, seed=5, walks=2048)
for kernel, count in [(k, c) for k in (1, 6) if model.qualifies(k) for c in (64, 256, 1024, 4096, 16384, 65536)]:
    model.set_kernel(kernel)
    nd = nodes[:count].contiguous()
    out, res = model.propagate(states_in, nd)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        model.propagate(states_in, nd, states_out=out, results=res)
    b.record()
    torch.cuda.synchronize()
    r = res.cpu().numpy()
    print(f"schedule-{tasks} kernel {kernel}: {count:6d} nodes {a.elapsed_time(b) / 50 * 1e3:8.1f} us per launch, rounds max {r[:, 3].max()} mean {r[:, 3].mean():.1f}, "
          f"revisions mean {r[:, 2].mean():.0f}, failed {float((r[:, 0] < 0).mean()):.2f}")
