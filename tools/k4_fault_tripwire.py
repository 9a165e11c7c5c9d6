"""Reads the tripwire of variant VT (tools/k4_fault_isa_variants.py): run in tools/fault_wt with CSOLVE_HIP_LIB set.
Revision counts >= 2^20 mark nodes in which the duplicated table read returned something else than the
compiled one."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from csolve_amd import problems
from csolve_amd.solver import solve_root
nq, count = 64, 1 << 18
model = solve_root(problems.queens(nq))
states_in, nodes, forb_in = bench.make_instances(model, count, seed=4242, with_sets=True, restore_kernel=0)
model.set_kernel(3)
o3, f3, r3 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
model.set_kernel(4)
for rep in range(3):
    o4, f4, r4 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
    torch.cuda.synchronize()
    trips = r4[:, 2] >> 20
    rev_ok = (r4[:, 2] & 0xFFFFF) == r3[:, 2]
    ok = (r3[:, 0] >= 0) & (r4[:, 0] >= 0)
    wrong = ((f4 != f3).flatten(1).any(1) & ok) | ((r3[:, 0] >= 0) != (r4[:, 0] >= 0))
    tripped = trips > 0
    print(f"launch {rep}: wrong nodes {int(wrong.sum())}, tripped nodes {int(tripped.sum())}, both {int((wrong & tripped).sum())}, "
          f"wrong only {int((wrong & ~tripped).sum())}, tripped only {int((tripped & ~wrong).sum())}, "
          f"revision counts otherwise equal: {bool(rev_ok.all())}, trips per tripped node max {int(trips.max())}")
    odd = torch.arange(count, device=wrong.device) % 2 == 1
    print(f"          wrong nodes at even chunk positions (first copy): {int((wrong & ~odd).sum())}, odd (second copy): {int((wrong & odd).sum())}")
