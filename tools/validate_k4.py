"""Repeated full-batch comparison of the register-resident forbidden-set kernels (4, and 5 where the model
qualifies) with the LDS one (3) and the general kernel (1) on the bench's instance set: verdicts, states, sets,
PROPS of every node.   usage: validate_k4.py [N | offsetsV:N] [instances] [launches]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from csolve_amd import problems
from csolve_amd.solver import solve_root
what = sys.argv[1] if len(sys.argv) > 1 else "64"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if what.startswith("offsets"):
    values, nq = (int(x) for x in what[7:].split(":"))
    model = solve_root(problems.offsets(nq, values, 1))
    nq = f"{nq} offsets{values}"
else:
    nq = int(what)
    model = solve_root(problems.queens(nq))
states_in, nodes, forb_in = bench.make_instances(model, count, seed=4242, with_sets=True, restore_kernel=0)
model.set_kernel(1)
o1, r1 = model.propagate(states_in, nodes)
model.set_kernel(3)
o3, f3, r3 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
torch.cuda.synchronize()
ok = r1[:, 0] >= 0
assert torch.equal(r3[:, 0] >= 0, ok) and torch.equal(o3[ok], o1[ok]) and torch.equal(r3[ok][:, :2], r1[ok][:, :2])
model.set_kernel(4)
bad_total = 0
t0 = time.time()
for rep in range(reps):
    o4, f4, r4 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
    torch.cuda.synchronize()
    bad = int((~torch.equal(r4[:, 0] >= 0, ok)))
    v = (r4[:, 0] >= 0) != ok
    s = ok & ~v & ((o4 != o3).flatten(1).any(1) | (f4 != f3).flatten(1).any(1) | (r4[:, :2] != r3[:, :2]).any(1))
    nb = int(v.sum()) + int(s.sum())
    bad_total += nb
    if nb:
        print("rep", rep, "verdict mismatches", int(v.sum()), "state/set/props mismatches", int(s.sum()))
print(f"queens-{nq}: {reps} launches x {count} nodes, mismatching nodes: {bad_total}  ({time.time()-t0:.1f} s)")

if model.qualifies(5):
    # several nodes per wave; set bits outside the root domains are unspecified (compare on the domains)
    dom = model.domains()
    fw = model.forbidden_words()
    mask = np.zeros((model.n_vars, fw), dtype=np.uint64)
    for v in range(model.n_vars):
        for q in range(fw):
            bits = int(min(max(int(dom[v, 1]) - int(dom[v, 0]) + 1 - 64 * q, 0), 64))
            mask[v, q] = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    d_mask = torch.from_numpy(mask.view(np.int64)).cuda()
    model.set_kernel(5)
    bad5 = 0
    for rep in range(reps):
        o5, f5, r5 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
        torch.cuda.synchronize()
        v = (r5[:, 0] >= 0) != ok
        s = ok & ~v & ((o5 != o3).flatten(1).any(1) | ((f5 & d_mask) != (f3 & d_mask)).flatten(1).any(1) |
                       (r5[:, :2] != r3[:, :2]).any(1))
        bad5 += int(v.sum()) + int(s.sum())
    print(f"queens-{nq} kernel 5: {reps} launches x {count} nodes, mismatching nodes: {bad5}")
    model.set_kernel(4)

# the sets-only layout of the same kernel: unpacked outputs against the same reference
sets_in = model.pack_sets(states_in)
bad_sets = 0
for rep in range(reps):
    so, rs = model.propagate_sets(sets_in, nodes)
    torch.cuda.synchronize()
    v = (rs[:, 0] >= 0) != ok
    good = ok & ~v
    un = model.unpack_sets(so)
    s = good & ((un != o3).flatten(1).any(1) | (rs[:, :2] != r3[:, :2]).any(1))
    bad_sets += int(v.sum()) + int(s.sum())
print(f"queens-{nq} sets-only layout: {reps} launches x {count} nodes, mismatching nodes: {bad_sets}")

