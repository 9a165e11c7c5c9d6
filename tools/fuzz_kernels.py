"""Differential run of every kernel a model qualifies for against the general kernel on seeded irregular
!= networks (csolve_amd.problems.offsets): verdicts, fixpoints and PROPS of every node.
usage: fuzz_kernels.py [models] [instances per model]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np, torch
import bench
from csolve_amd import problems
from csolve_amd.solver import solve_root

models = int(sys.argv[1]) if len(sys.argv) > 1 else 40
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rng = np.random.default_rng(11)
checked = 0
for seed in range(models):
    n = int(rng.integers(3, 70))
    values = int(rng.integers(3, 65))
    try:
        model = solve_root(problems.offsets(n, values, seed + 1))
    except Exception as e:  # the root itself may be inconsistent
        print(f"seed {seed}: n={n} values={values}: skipped ({str(e)[:60]})")
        continue
    states_in, nodes, forb_in = bench.make_instances(model, count, seed=seed, walks=1024, with_sets=model.forbidden_words() > 0,
                                                     restore_kernel=0)
    model.set_kernel(1)
    o1, r1 = model.propagate(states_in, nodes)
    torch.cuda.synchronize()
    ok = r1[:, 0] >= 0
    ran = [1]
    for k in (2, 3, 4, 5, 6, 7):
        if not model.qualifies(k):
            continue
        model.set_kernel(k)
        if k in (3, 4, 5) and forb_in is not None:
            o, f, r = model.propagate_fb(states_in, nodes, forb_in=forb_in)
        else:
            o, r = model.propagate(states_in, nodes)
        torch.cuda.synchronize()
        assert torch.equal(r[:, 0] >= 0, ok), (seed, n, values, k, "verdicts")
        assert torch.equal(o[ok], o1[ok]), (seed, n, values, k, "fixpoints")
        assert torch.equal(r[ok][:, :2], r1[ok][:, :2]), (seed, n, values, k, "status / PROPS")
        ran.append(k)
    checked += 1
    print(f"seed {seed}: n={n} values={values} inconsistent {1 - float(ok.float().mean()):.2f} kernels {ran} ok", flush=True)
print(f"{checked} models, {count} nodes each: all kernels agree")
