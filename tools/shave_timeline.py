"""Diagnostic: per-wave start / end times of kernel 7 (library built with -DCS_SHAVE_TIMELINE as
csolve_amd/libcsolve_hip_tl.so; run with CSOLVE_HIP_LIB pointing at it).  usage: shave_timeline.py [queens N] [instances]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from csolve_amd import problems, _lib
from csolve_amd.solver import solve_root
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
model = solve_root(problems.queens(nq))
states_in, nodes, _ = bench.make_instances(model, count, seed=12345, with_sets=False, restore_kernel=7)
model.set_kernel(7)
for _ in range(3):
    out, res = model.propagate(states_in, nodes)
torch.cuda.synchronize()
lib = _lib.load_library()
waves = 65536
buf = (C.c_ulonglong * (3 * waves))()
assert lib.csgpu_debug_shave_timeline(buf, waves) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(waves, 3).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start, end, nn = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]  # microseconds
print(f"waves {len(a)}, kernel span {end.max():.1f} us; starts: min {start.min():.1f} median {np.median(start):.1f} max {start.max():.1f}")
print(f"ends: p1 {np.percentile(end,1):.1f} p10 {np.percentile(end,10):.1f} median {np.median(end):.1f} p90 {np.percentile(end,90):.1f} p99 {np.percentile(end,99):.1f} max {end.max():.1f}")
life = end - start
print(f"lifetime: median {np.median(life):.1f} mean {life.mean():.1f}; nodes per wave: min {nn.min()} median {int(np.median(nn))} max {nn.max()}")
print(f"mean residency = sum(lifetime) / (span x waves) = {life.sum() / (end.max() * len(a)):.2f}")
hist, edges = np.histogram(end, bins=12)
print("end-time histogram:", [(f"{edges[i]:.0f}", int(h)) for i, h in enumerate(hist)])
ops = res[:, 2].cpu().numpy().astype(np.int64) / (3 * (nq - 1))
print("table-row operations per node: mean %.2f median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f; rounds max %d" % (
    ops.mean(), np.median(ops), np.percentile(ops, 90), np.percentile(ops, 99), np.percentile(ops, 99.9), ops.max(), int(res[:, 3].max())))
