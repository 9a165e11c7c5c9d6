"""What do the wrong forbidden-set bits of the kernel-4 fault look like?  Run in tools/fault_wt with
CSOLVE_HIP_LIB pointing at a failing variant (tools/k4_fault_isa_variants.py).  For nodes whose sets differ
from kernel 3's: extra and missing bits per word, and for single-assignment nodes the relation of every
differing bit to the three bits the assignment is expected to push (queens: value, value +- distance)."""
import sys, collections, numpy as np, torch
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
o4, f4, r4 = model.propagate_fb(states_in, nodes, forb_in=forb_in)
torch.cuda.synchronize()
ok = (r3[:, 0] >= 0) & (r4[:, 0] >= 0)
diff = ((f4 != f3).flatten(1).any(1)) & ok
idx = torch.nonzero(diff).flatten().cpu().numpy()
print("nodes with differing sets:", len(idx), "of", int(ok.sum()), "consistent; verdict mismatches:",
      int(((r3[:, 0] >= 0) != (r4[:, 0] >= 0)).sum()))
F3 = f3.cpu().numpy().view(np.uint64).reshape(count, nq)
F4 = f4.cpu().numpy().view(np.uint64).reshape(count, nq)
FI = forb_in.cpu().numpy().view(np.uint64).reshape(-1, nq)
N = nodes.cpu().numpy()
R3 = r3.cpu().numpy()
extra_hist, missing_hist = collections.Counter(), collections.Counter()
rel = collections.Counter()
words_per_node = collections.Counter()
shown = 0
for i in idx[:20000]:
    d = np.nonzero(F3[i] != F4[i])[0]
    words_per_node[len(d)] += 1
    var, lo, hi, parent = (int(x) for x in N[i])
    for w in d:
        a, b = int(F3[i, w]), int(F4[i, w])
        ex, mi = b & ~a, a & ~b
        extra_hist[bin(ex).count("1")] += 1
        missing_hist[bin(mi).count("1")] += 1
        if lo == hi and int(R3[i, 3]) == 0:  # one round: only the assignment pushed
            dist = abs(w - var)
            want = {lo - 1: "col", lo - 1 + dist: "up", lo - 1 - dist: "down"}
            for bit in range(64):
                if (ex >> bit) & 1:
                    rel[("extra", want.get(bit, "other"), "hi" if bit >= 32 else "lo")] += 1
                if (mi >> bit) & 1:
                    rel[("missing", want.get(bit, "other"), "hi" if bit >= 32 else "lo")] += 1
            if shown < 12:
                shown += 1
                print(f"node {i}: X{var}={lo} onto X{w}: parent set {int(FI[parent, w]):016x} want {a:016x} got {b:016x} "
                      f"extra {ex:016x} missing {mi:016x} expected bits {sorted(k for k in want if 0 <= k < 64)}")
print("differing words per node:", sorted(words_per_node.items()))
print("extra bits per differing word:", sorted(extra_hist.items()))
print("missing bits per differing word:", sorted(missing_hist.items()))
print("single-round nodes, differing bits by (kind, which expected bit, half):")
for k, v in sorted(rel.items(), key=lambda kv: -kv[1]):
    print("  ", k, v)
