"""Rebuild mode (no incoming sets, every valued variable pushes) of the register-resident forbidden-set
kernel on a full-size batch: states must come back unchanged with zero PROPS, sets equal to the LDS
kernel's."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from csolve_amd import problems
from csolve_amd.solver import solve_root
model = solve_root(problems.queens(64))
count = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
states_in, nodes, _ = bench.make_instances(model, count, seed=12345, with_sets=False, restore_kernel=0)
rebuild = torch.zeros((count, 4), dtype=torch.int32, device="cuda")
rebuild[:, 0] = -1
rebuild[:, 3] = torch.arange(count, dtype=torch.int32, device="cuda")
model.set_kernel(3)
s3, f3, r3 = model.propagate_fb(states_in, rebuild)
torch.cuda.synchronize()
assert torch.equal(s3, states_in) and bool((r3[:, 1] == 0).all())
model.set_kernel(4)
for rep in range(5):
    s4, f4, r4 = model.propagate_fb(states_in, rebuild)
    torch.cuda.synchronize()
    bad = (r4[:, 0] < 0) | (s4 != states_in).flatten(1).any(1) | (f4 != f3).flatten(1).any(1)
    nb = torch.nonzero(bad)[:, 0]
    print("rep", rep, "revision mismatches", int((r4[:, 2] != r3[:, 2]).sum()), "props!=0", int((r4[:, 1] != 0).sum()), "bad nodes", len(nb), "positions", np.bincount((nb % 16).cpu().numpy(), minlength=16).tolist())
