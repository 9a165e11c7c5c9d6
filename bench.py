#!/usr/bin/env python3
"""bench.py -- constraint propagations/s and nodes/s of the HIP propagation fixpoint.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): queens-64 (text identical to what the reference's
scripts/gen_queens.sh writes), propagation-only.  The node instances are produced by
seeded random assignment walks: from the root fixpoint pick a random open
variable and a random value of its interval, propagate, continue from the result; a walk
restarts at the first inconsistent node or when every variable is assigned.  Every step
of every walk is one instance (state-before, variable, value) -> (state-after | FAIL).
The instance set is generated once with the device path (untimed) and then stays resident
in HBM.  One "step" = one launch of the batched fixpoint kernel over the whole instance
set.  With N ranks every rank owns its own instance set of the same size (weak scaling);
nodes are independent, so the data path has no collective.

Prints ONE JSON line (rank 0): value = narrowing events ("propagations", the reference's
PROPS counter, propagate.c:77-78) per second over all ranks; `roofline` for the fixpoint
kernel from HIP-event timing; `cpu_baseline` = the compiled reference itself
(oracle/_ref/csolve_ref, kind "reference"; the oracle restatement, kind "port", when that
binary is absent) replaying the same instances on one host core -- and, as `all_cores`,
as independent processes on all of them -- which also re-checks every device result of the
sample bit for bit.

Other workloads: --queens N, --sudoku N, --schedule N (propagation-only on other model
shapes), --layout sets (the states carried as bit vectors only), --workload search (the
sharded search engine, strong scaling).
"""
import argparse
import json
import os
import sys
import time


def _launch_ranks_if_needed(argv, script=None):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself -- fresh child
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the same command line -- relays rank 0's one JSON
    line and exits non-zero if any rank does.  It runs before torch or the HIP library is imported: the parent never
    touches a GPU and nothing is re-executed (the reference forks its workers from one process, csolve.c:105-152;
    HIP state does not survive a fork, so the ranks are separate processes from the start)."""
    import socket
    import subprocess
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ or "-h" in argv or "--help" in argv:
        return
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CSOLVE_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0 = None
    rc = 0
    pending = set(range(n))
    deadline_after_failure = None
    while pending:
        for r in sorted(pending):
            p = procs[r]
            if r == 0 and out0 is None and p.poll() is not None:
                out0 = p.stdout.read()
            if p.poll() is not None:
                pending.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    if deadline_after_failure is None:
                        deadline_after_failure = time.time() + float(os.environ.get("CSOLVE_BENCH_RANK_GRACE", "20"))
        if pending and deadline_after_failure is not None and time.time() > deadline_after_failure:
            for r in pending:  # a rank died: the others may wait for it in a collective for ever
                procs[r].kill()
        if pending:
            if 0 in pending and out0 is None:
                try:  # drain rank 0's pipe so that it never blocks on a full one
                    out0, _ = procs[0].communicate(timeout=0.2)
                except subprocess.TimeoutExpired:
                    pass
            else:
                time.sleep(0.05)
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"bench: rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    _launch_ranks_if_needed(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from csolve_amd import problems  # noqa: E402
from csolve_amd.solver import solve_root  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
UNTIMED_REPLAY_MS = 25.0  # time_steps: load on the device right before the timed graph replay (see there)
STREAMING_CEILING_GBS = 5900.0  # measured copy-kernel ceiling of this part for row-shaped traffic


def measured_traffic(kernel_name, workload_key, instances):
    """HBM bytes per launch from the committed PMC profile of this very kernel and workload
    (profiles/*_pmc_*.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own passes, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950), or (None, None) when there is no such profile.
    workload_key: "queens-64", "sudoku-25x25", ... (the first words of the profile's "workload").
    -> (bytes, source): the figure is NOT measured in this run (counters need the profiler's own passes); `source` names
    the profile file it was taken from, the commit that profile was made at and the command, so that a reader can tell
    a stale profile from a current one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_*.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if kernel_name in rec.get("kernel", "") and rec.get("workload", "").startswith(workload_key + " ") and \
                f"{instances} instances" in rec.get("workload", "") and "hbm_traffic_bytes_per_launch" in rec:
            return rec["hbm_traffic_bytes_per_launch"], {
                "measured_in_this_run": False, "profile": os.path.relpath(path, ROOT), "profile_commit": rec.get("commit"),
                "profile_command": rec.get("command"), "workload": rec.get("workload")}
    return None, None


def make_instances(model, count, seed, walks=8192, with_sets=False, restore_kernel=0):
    """Seeded random walks on the device path (untimed setup).
    -> states_in [count,n,2], nodes [count,4], forb_in [count,n,FW] or None   (all on the device)
    The walks run through the general kernel so that the profile of the timed kernel only contains
    the full-size launches; the forbidden sets of the collected states are then produced by ONE
    full-size launch of the forbidden-set kernel in rebuild mode (tests show they equal the sets a
    search would have carried down to these states)."""
    n = model.n_vars
    root = model.domains()
    rng = np.random.default_rng(seed)
    walks = min(walks, count)
    cur = np.repeat(root[None], walks, 0)
    model.set_kernel(1)
    states, nodes = [], []
    have = 0
    while have < count:
        open_mask = cur[:, :, 0] < cur[:, :, 1]
        done = ~open_mask.any(1)
        cur[done] = root
        open_mask[done] = root[:, 0] < root[:, 1]
        # random open variable per walk, random value of its interval
        keys = rng.random(open_mask.shape)
        keys[~open_mask] = -1.0
        var = keys.argmax(1)
        lo = cur[np.arange(walks), var, 0]
        hi = cur[np.arange(walks), var, 1]
        val = lo + (rng.random(walks) * (hi - lo + 1)).astype(np.int64).clip(0, hi - lo)
        nd = np.stack([var, val, val, np.arange(walks)], 1).astype(np.int32)
        d_cur = torch.from_numpy(cur).cuda()
        out, res = model.propagate(d_cur, torch.from_numpy(nd).cuda())
        torch.cuda.synchronize()
        take = min(walks, count - have)
        states.append(d_cur[:take].clone())
        nd_take = nd[:take].copy()
        nd_take[:, 3] = np.arange(have, have + take)
        nodes.append(torch.from_numpy(nd_take).cuda())
        have += take
        ok = (res[:, 0] >= 0).cpu().numpy()
        nxt = out.cpu().numpy()
        nxt[~ok] = root
        cur = nxt
    model.set_kernel(restore_kernel)
    states_in, nodes = torch.cat(states).contiguous(), torch.cat(nodes).contiguous()
    forb_in = None
    if with_sets:
        rebuild = torch.zeros((count, 4), dtype=torch.int32, device="cuda")
        rebuild[:, 0] = -1
        rebuild[:, 3] = torch.arange(count, dtype=torch.int32, device="cuda")
        same, forb_in, res = model.propagate_fb(states_in, rebuild)
        torch.cuda.synchronize()
        assert torch.equal(same, states_in) and bool((res[:, 1] == 0).all())
    return states_in, nodes, forb_in


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "csolve_ref")


def reference_replay(text, n, states_in_np, nodes_np, count=None):
    """The compiled reference (oracle/_ref/csolve_ref bench) on the first `count` instances: bind + propagate_clauses
    per instance on one host core.  -> (stats dict, status [count] (-1 or PROPS), after [count, n, 2])"""
    import subprocess
    import tempfile
    count = nodes_np.shape[0] if count is None else count
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        prob = os.path.join(tmp, "problem.txt")
        open(prob, "w").write(text)
        rec = np.concatenate([nodes_np[:count, 0:2], states_in_np[nodes_np[:count, 3]].reshape(count, 2 * n)], 1).astype(np.int32)
        fin, fout = os.path.join(tmp, "inst.in"), os.path.join(tmp, "res.out")
        with open(fin, "wb") as f:
            np.array([0x4E495343, n, count], dtype=np.int32).tofile(f)
            rec.tofile(f)
        p = subprocess.run([REF_BIN, "bench", prob, fin, fout, "-c", "false"], capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("csolve_ref bench failed: " + p.stderr)
        stats = json.loads(p.stdout.split("@BENCH ", 1)[1])
        raw = np.fromfile(fout, dtype=np.int32)[3:].reshape(count, 1 + 2 * n)
        return stats, raw[:, 0].astype(np.int64), raw[:, 1:].reshape(count, n, 2)


def cpu_baseline(args, text, model, states_in, nodes, states_out, res_h):
    """The reference CPU propagator on one host core over a bounded sample of the SAME instances,
    used at the same time as the checker of the device results on that sample.
    kind "reference": the compiled reference itself (oracle/_ref/csolve_ref, built from the
    reference's own sources in the authoring container and shipped as a binary);
    kind "port": the oracle restatement, when that binary is not there."""
    import subprocess
    import tempfile
    n = model.n_vars
    B = nodes.shape[0]
    si = states_in.cpu().numpy()
    nd = nodes.cpu().numpy()
    so = states_out.cpu().numpy()

    def check(sel, status, after):
        fail = status < 0
        assert (fail == (res_h[sel, 0] < 0)).all(), "device verdicts differ from the CPU reference"
        assert (so[sel][~fail] == after[~fail]).all(), "device fixpoints differ from the CPU reference"
        # PROPS is an order-independent quantity only on pure != networks; with other clauses the
        # Gauss-Seidel reference and the round-based device count different numbers of narrowing events
        if model.qualifies(2):
            assert (res_h[sel, 1][~fail] == status[~fail]).all(), "device PROPS differ from the CPU reference"

    if os.path.exists(REF_BIN):
        # pilot chunk to size the sample for the time budget, then one timed run
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            prob = os.path.join(tmp, "problem.txt")
            open(prob, "w").write(text)

            def run(count):
                rec = np.concatenate([nd[:count, 0:2], si[nd[:count, 3]].reshape(count, 2 * n)], 1).astype(np.int32)
                fin, fout = os.path.join(tmp, "inst.in"), os.path.join(tmp, "res.out")
                with open(fin, "wb") as f:
                    np.array([0x4E495343, n, count], dtype=np.int32).tofile(f)
                    rec.tofile(f)
                p = subprocess.run([REF_BIN, "bench", prob, fin, fout, "-c", "false"], capture_output=True, text=True)
                if p.returncode != 0:
                    raise RuntimeError("csolve_ref bench failed: " + p.stderr)
                stats = json.loads(p.stdout.split("@BENCH ", 1)[1])
                raw = np.fromfile(fout, dtype=np.int32)[3:].reshape(count, 1 + 2 * n)
                return stats, raw[:, 0].astype(np.int64), raw[:, 1:].reshape(count, n, 2)

            pilot = min(B, 4096)
            stats, st, after = run(pilot)
            count = int(min(B, max(pilot, pilot * args.cpu_seconds / max(stats["seconds"], 1e-6))))
            if count > pilot:
                stats, st, after = run(count)
            check(slice(0, count), st, after)
            # the whole batch is cheaper than the budget: repeat the pass for a steadier clock
            passes, binds, seconds = 1, stats["binds"], stats["seconds"]
            while count == B and seconds + stats["seconds"] <= args.cpu_seconds and passes < 8:
                more, _, _ = run(count)
                passes, binds, seconds = passes + 1, binds + more["binds"], seconds + more["seconds"]
            # process-parallel over disjoint slices of the same instances on the box's host cores (the
            # reference's own -j mode does not scale, independent processes are its multi-core baseline)
            cores = max(1, min(len(os.sched_getaffinity(0)), 16))
            all_cores = None
            if cores > 1 and count >= 64 * cores:
                per = count // cores
                procs = []
                for c in range(cores):
                    lo_i = c * per
                    rec = np.concatenate([nd[lo_i:lo_i + per, 0:2], si[nd[lo_i:lo_i + per, 3]].reshape(per, 2 * n)], 1).astype(np.int32)
                    fin, fout = os.path.join(tmp, f"inst{c}.in"), os.path.join(tmp, f"res{c}.out")
                    with open(fin, "wb") as f:
                        np.array([0x4E495343, n, per], dtype=np.int32).tofile(f)
                        rec.tofile(f)
                    procs.append((fin, fout))
                w0 = time.perf_counter()
                running = [subprocess.Popen([REF_BIN, "bench", prob, fin, fout, "-c", "false"], stdout=subprocess.PIPE,
                                            stderr=subprocess.DEVNULL, text=True) for fin, fout in procs]
                outs = [p.communicate()[0] for p in running]
                wall = time.perf_counter() - w0
                if all(p.returncode == 0 for p in running):
                    pb = sum(json.loads(o.split("@BENCH ", 1)[1])["binds"] for o in outs)
                    all_cores = {"value": pb / wall, "nodes_per_s": cores * per / wall, "cores": cores,
                                 "note": "independent reference processes on disjoint slices, wall time incl. start-up"}
            return {"value": binds / seconds, "unit": "propagations/s", "cores": 1, "all_cores": all_cores,
                    "kind": "reference", "nodes_per_s": passes * count / seconds,
                    "sample": f"first {count} of the {B} instances of this run, {passes} pass(es) through the compiled "
                              f"reference's propagate_clauses() (gcc -O3, conflict learning off), {seconds:.1f} s on one "
                              f"host core; device verdicts, fixpoints and PROPS re-checked against it bit for bit"}

    from oracle.cs_oracle import Model as OModel, Oracle
    omodel = OModel.parse(text)
    omodel.set_domains(model.domains())
    omodel.index()
    orc = Oracle(omodel)
    chunk, done, binds, spent = 2048, 0, 0, 0.0
    while done < B and spent < args.cpu_seconds:
        sel = slice(done, min(B, done + chunk))
        nd_c = nd[sel].copy()
        nd_c[:, 3] -= done
        c0 = time.perf_counter()
        st, exp, b = orc.instances_nodes(si[sel], nd_c)
        spent += time.perf_counter() - c0
        binds += b
        check(sel, st, exp)
        done += nd_c.shape[0]
    return {"value": binds / spent, "unit": "propagations/s", "cores": 1, "kind": "port",
            "nodes_per_s": done / spent,
            "sample": f"first {done} of the {B} instances of this run through the oracle restatement, {spent:.1f} s "
                      f"on one host core; device results re-checked against it bit for bit"}


def run_sharded_search(args, rank, world, dist, text, steps, warmup, time_limit=None, children=0, pool=0, slice_iterations=0,
                       max_slices=0):
    """One sharded search (csolve_amd/parallel.py: every-rank seeding, stealing of open states, shared incumbent)
    run `warmup` + `steps` times; the timed runs sit between two barriers, the time is the maximum over the ranks.
    -> dict (the same on every rank): totals of the last run, wall seconds per search, per-rank clocks."""
    from csolve_amd.parallel import ShardedSearch
    from csolve_amd.solver import Search
    model = solve_root(text)
    n = model.n_vars
    comm = "cpu" if args.comm == "gloo" else "cuda"
    max_children = children or (1 << 23 if n <= 32 else 1 << 19)  # 2^23: a quarter of the iterations of 2^21 (queens-16 ALL 82.7 -> 74.7 ms)
    eng = Search(model, pool or 8 * max_children, max_children)  # buffers are allocated once, outside the timed region

    def once():
        eng.reset()
        sh = ShardedSearch(eng, model.objective, n, rank, world, dist, engine_device="cuda", comm_device=comm,
                           # a dry rank calls the exchange earlier.  (MIN / MAX used 256 while an iteration was 45 us and 10,000
                           # nodes; at 137 us and 139,000 nodes 64 iterations are 9 ms: two ranks on one GPU 2.29 -> 1.97 s)
                           slice_iterations=slice_iterations or 64,
                           seed_states_per_rank=256, low_water=4096, poll_iterations=args.poll,
                           time_limit=time_limit if time_limit else None)
        local_stats, totals = sh.run(model.root_state(), max_slices if max_slices > 0 else 1 << 40)
        return local_stats, totals, sh

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        once()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        local_stats, totals, sh = once()
    barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=comm)
    moved = torch.tensor([sh.states_moved], dtype=torch.int64, device=comm)
    share = torch.tensor([local_stats["nodes"]], dtype=torch.int64, device=comm)
    # the last step's clocks of every rank: where a rank's time went (diagnosis of the scaling runs)
    put_s, put_n = eng.put_cost()
    clocks = torch.tensor([sh.idle_fraction(), sh.seconds["total"], sh.seconds["seed"], sh.seconds["busy"],
                           sh.seconds["exchange"], put_s, float(put_n), float(sh.exchanges), float(sh.early_exchanges),
                           float(sh.states_moved)],
                          dtype=torch.float64, device=comm).unsqueeze(0)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(moved, op=dist.ReduceOp.SUM)
        gathered = torch.empty(world, dtype=torch.int64, device=comm)
        dist.all_gather_into_tensor(gathered, share)
        share = gathered
        allc = torch.empty((world, clocks.shape[1]), dtype=torch.float64, device=comm)
        dist.all_gather_into_tensor(allc, clocks)
        clocks = allc
    clocks = clocks.cpu().tolist()
    elapsed = float(el.item())
    eng.close()
    del eng
    return {
        "n_vars": n, "objective": int(model.objective), "seconds_per_search": elapsed / steps, "steps": steps, "warmup": warmup,
        "totals": totals, "states_moved_between_ranks": int(moved.item()), "nodes_per_rank": share.cpu().tolist(),
        "seeded_alike": sh.seeded_alike, "status_page": sh.page_used,
        # one entry per rank, last step: idle = share of the time after seeding not spent inside the engine;
        # put = csgpu_search_put (copy of seeded / stolen states into the pool)
        "ranks": [{"idle_fraction": round(c[0], 4), "seconds": round(c[1], 6), "seed_seconds": round(c[2], 6),
                   "busy_seconds": round(c[3], 6), "exchange_seconds": round(c[4], 6),
                   "put_seconds": round(c[5], 6), "put_states": int(c[6]),
                   "put_fraction": round(c[5] / c[1], 5) if c[1] > 0 else None,
                   "transfers": int(c[7]), "early_exchanges": int(c[8]), "states_moved": int(c[9])} for c in clocks]}


def group_info(args, world, dist):
    """what the ranks of this run are and how they talk: the backend of the process group, the world size the group
    itself reports (not the command line's), and how the ranks were started"""
    return {"process_group": None if dist is None else dist.get_backend(),
            "ranks_seen": 1 if dist is None else dist.get_world_size(), "comm": args.comm if dist is not None else None,
            "launched_by": "bench.py itself (child processes)" if os.environ.get("CSOLVE_BENCH_SELF_LAUNCHED") else
                           ("an external launcher (torch.distributed.run)" if world > 1 else "a single process"),
            "same_device": bool(args.same_device)}


def search_workload(args, rank, world, local, dist):
    """BASELINE configs[3]-style run as the headline: queens-N ALL (or a schedule.txt-style MIN model), the whole
    search tree sharded over the ranks.  Strong scaling: the tree is fixed, every step is one complete search."""
    text = problems.queens(args.search_queens, args.search_objective)
    what = f"queens-{args.search_queens} {args.search_objective}"
    if args.search_schedule:
        text = problems.schedule(args.search_schedule, 1)
        what = f"schedule-{args.search_schedule} MIN (examples/schedule.txt style)"
    r = run_sharded_search(args, rank, world, dist, text, args.steps, args.warmup, time_limit=args.time_limit,
                           children=args.children or (1 << 21 if args.search_schedule else 0), pool=args.pool,
                           slice_iterations=args.slice,
                           max_slices=args.search_slices)
    totals = r["totals"]
    if rank == 0:
        sec = r["seconds_per_search"]
        print(json.dumps({
            "metric": "constraint propagations/sec + nodes/sec, queens-N, 1/2/4/8 MI355X",
            "value": totals["props"] / sec, "unit": "propagations/s",
            "nodes_per_s": totals["nodes"] / sec, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": dict({"workload": f"{what}, full search sharded GPU-per-subtree "
                                        f"(BASELINE configs[{4 if args.search_schedule else 3}] shape)", "solutions": totals["solutions"],
                            "best": totals.get("best") if args.search_schedule else None,
                            "nodes": totals["nodes"], "cuts": totals["cuts"], "props": totals["props"],
                            "iterations": totals["iterations"], "states_moved_between_ranks": r["states_moved_between_ranks"],
                            "nodes_per_rank": r["nodes_per_rank"], "timeout": bool(totals.get("timeout")),
                            "stopped_after_slices": args.search_slices or None}, **group_info(args, world, dist)),
            "ranks": r["ranks"]}))


def search_record(args, rank, world, dist):
    """The sharded search next to the propagation headline, in EVERY line (N = 1 included, so that the driver's 1 / 2 / 4 / 8
    runs form a curve with a base): BASELINE configs[3] -- queens-17 ALL, a fixed tree (95,815,104 solutions): strong scaling
    (seconds per search must fall with N); queens-128 ALL under a time limit (the north-star instance: the tree is not
    finished by anyone, nodes per second is the figure) -- and configs[4]: a schedule.txt-style MIN model, whose incumbent
    travels between the ranks.  All ranks take part; every rank returns the same dict."""
    out = dict(group_info(args, world, dist))
    out["note"] = ("strong scaling of the sharded search (one process per GPU, subtrees stolen between ranks, incumbent shared): "
                   "seconds_per_search of the fixed trees must fall with n_gpus; nodes_per_s is the whole job's")

    def rec(name, text, steps, warmup, **kw):
        r = run_sharded_search(args, rank, world, dist, text, steps, warmup, **kw)
        t = r["totals"]
        sec = r["seconds_per_search"]
        out[name] = {"seconds_per_search": sec, "nodes_per_s": t["nodes"] / sec, "propagations_per_s": t["props"] / sec,
                     "nodes": t["nodes"], "cuts": t["cuts"], "props": t["props"], "solutions": t["solutions"],
                     "iterations": t["iterations"], "timeout": bool(t.get("timeout")), "steps": steps, "warmup": warmup,
                     "states_moved_between_ranks": r["states_moved_between_ranks"], "nodes_per_rank": r["nodes_per_rank"],
                     "seeded_alike": r["seeded_alike"], "status_page": r["status_page"], "ranks": r["ranks"]}
        if kw.get("time_limit"):
            out[name]["time_limit_s"] = kw["time_limit"]
        if "best" in t and r["objective"] in (2, 3):
            out[name]["best"] = t["best"]
        return out[name]

    q = args.search_queens
    rec(f"queens{q}_all", problems.queens(q, "ALL"), steps=args.search_steps, warmup=1)
    out[f"queens{q}_all"]["workload"] = f"queens-{q} ALL, the whole tree, sharded GPU-per-subtree (BASELINE configs[3] shape); strong scaling"
    # (a pool of 2^25 rows, 32 GB of interval rows: with the default 2^22 the frontiers stay small -- the pool bounds how
    # many parents a depth-first walk in batches may take -- and the same run explores 1.0e9 instead of 1.7e9 nodes/s)
    rec("queens128_all", problems.queens(128, "ALL"), steps=1, warmup=0, time_limit=args.search_time_limit, pool=1 << 25)
    out["queens128_all"]["workload"] = (f"queens-128 ALL stopped after {args.search_time_limit} s on every rank (the reference's -t): "
                                        f"nodes explored per second by the whole job")
    sched = args.search_record_schedule
    if sched > 0:
        # (a child buffer of 2^21: a MIN iteration takes as many parents as it can be sure to hold the children of --
        # parents x widest interval -- and 8,192 parents per iteration, 1.68 s, is where schedule-12 is fastest; 2^19: 2.3 s)
        rec(f"schedule{sched}_min", problems.schedule(sched, 1), steps=1, warmup=0, children=1 << 21)
        out[f"schedule{sched}_min"]["workload"] = (f"schedule-{sched} MIN (examples/schedule.txt style, BASELINE configs[4] shape): the "
                                                   f"incumbent bound travels between the ranks")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100; 3 for --workload search, where a "
                    "step is one complete search)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before (default 10; 1 for --workload search)")
    ap.add_argument("--queens", type=int, default=64)
    ap.add_argument("--schedule", type=int, default=0, help="tasks of a schedule.txt-style optimisation model (tree clauses, "
                    "general kernel) instead of queens (BASELINE configs[4] shape)")
    ap.add_argument("--sudoku", type=int, default=0, help="box size N of an N^2 x N^2 sudoku-shaped != network instead of "
                    "queens (5 = BASELINE configs[2], 25x25)")
    ap.add_argument("--instances", type=int, default=1 << 21, help="node instances per GPU (2^21 queens-64 instances are 2.2 GB per "
                    "launch: eight times the 256 MB Infinity Cache, so a repeated launch cannot be served from it; a launch "
                    "of 2^18 takes 66 us, a quarter of one of 2^20 53 us, an eighth of one of 2^21 51 us: ramp and tail of a "
                    "launch -- 0.607 / 0.639 / 0.644 of the roofline at 2^20 / 2^21 / 2^22)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the timed steps one by one instead of as one hipGraph")
    ap.add_argument("--kernel", type=int, default=0, help="0 best, 1 general, 2 LDS-resident, 3 forbidden sets in LDS, 4 forbidden sets in registers, "
                         "5 = 4 with several nodes per wave (at most 32 variables), 6 clause-resident (small models), 7 interval-only shaving")
    ap.add_argument("--rebuild-sets", action="store_true", help="forbidden-set kernel without resident sets")
    ap.add_argument("--layout", choices=["intervals", "intervals+sets", "sets"], default="intervals",
                    help="which leg is the headline: intervals = the state-only entry csgpu_propagate_batch (default: "
                         "interval states in and out, nothing precomputed); intervals+sets = csgpu_propagate_batch_fb "
                         "with resident forbidden sets; sets = csgpu_propagate_batch_sets (bit-vector states)")
    ap.add_argument("--no-queens128", action="store_true", help="skip the queens-128 sub-record of the default run")
    ap.add_argument("--no-sudoku25", action="store_true", help="skip the sudoku-25 sub-record (BASELINE configs[2]) of the default run")
    ap.add_argument("--workload", choices=["propagate", "search"], default="propagate")
    ap.add_argument("--process-group", action="store_true", help="create the process group even for one rank (the "
                    "collectives of the search workload then run over RCCL / gloo with world size 1)")
    ap.add_argument("--time-limit", type=float, default=0.0, help="search workload: stop every rank after this many seconds "
                    "(the reference's -t; 0 = none); the line then says \"timeout\": true and counts what was explored")
    ap.add_argument("--poll", type=int, default=4, help="search workload: iterations between two looks at the node's "
                    "status page")
    ap.add_argument("--search-queens", type=int, default=17, help="queens-N tree of the search workload (ALL: 17 is "
                    "95,815,104 solutions, 6.2e9 nodes, about a second on one GPU)")
    ap.add_argument("--search-objective", choices=["ALL", "ANY"], default="ALL")
    ap.add_argument("--search-schedule", type=int, default=0, help="search workload on a schedule.txt-style MIN model of this "
                    "many tasks instead of queens (BASELINE configs[4] shape: the incumbent bound travels between the ranks)")
    ap.add_argument("--pool", type=int, default=0, help="search workload: rows of the state pool (default: 8 x --children)")
    ap.add_argument("--children", type=int, default=0, help="search workload: children per iteration at most "
                    "(default: 2^21 for models of at most 32 variables, else 2^19)")
    ap.add_argument("--slice", type=int, default=0, help="search iterations between regular rank exchanges (default: 64; "
                    "a rank that runs dry calls the exchange earlier through the status page)")
    ap.add_argument("--search-slices", type=int, default=0,
                    help="stop a search after this many slices (0 = run to the end): ALL on trees too large to finish")
    ap.add_argument("--no-search", action="store_true", help="skip the `search` sub-record (the sharded search next to the "
                    "propagation headline); profiles of the propagation kernels use it")
    ap.add_argument("--search-steps", type=int, default=2, help="timed searches of the fixed tree in the `search` sub-record")
    ap.add_argument("--search-time-limit", type=float, default=2.0, help="seconds of the queens-128 ALL run of the `search` sub-record")
    ap.add_argument("--search-record-schedule", type=int, default=12, help="tasks of the MIN model of the `search` sub-record (0: none)")
    ap.add_argument("--comm", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (rehearsal on a 1-GPU box, use --comm gloo)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.workload == "search" else 100
    if args.warmup is None:
        args.warmup = 1 if args.workload == "search" else 10

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one process per GPU: measuring one GPU under the label of N would be wrong.  (Without WORLD_SIZE the ranks
        # were started by _launch_ranks_if_needed before anything touched a GPU; a launcher's world must match --gpus.)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: either let bench.py start its ranks (no WORLD_SIZE in "
                         f"the environment) or start them with `python -m torch.distributed.run --nnodes=1 --nproc-per-node "
                         f"{args.gpus} bench.py --gpus {args.gpus} ...`")
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.process_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.comm == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    if args.workload == "search":
        search_workload(args, rank, world, local, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    n_q = args.queens
    text = problems.queens(n_q)
    problem_name = f"queens-{n_q} "
    workload_key = f"queens-{n_q}"
    if args.sudoku:
        text = problems.sudoku(args.sudoku, 0.4, 1)
        problem_name = f"sudoku-{args.sudoku ** 2}x{args.sudoku ** 2} (40 % givens, SURVEY 8d(3)) "
        workload_key = f"sudoku-{args.sudoku ** 2}x{args.sudoku ** 2}"
    if args.schedule:
        text = problems.schedule(args.schedule, 1)
        problem_name = f"schedule-{args.schedule} (examples/schedule.txt style, MIN) "
        workload_key = f"schedule-{args.schedule}"
    legs = run_propagation_legs(args, text, args.instances, seed=12345 + rank, dist=dist, headline_only=(world > 1))
    head = legs["legs"][legs["headline"]]
    model, n, B = legs["model"], legs["n"], legs["B"]
    res_h = head["results"]
    elapsed = torch.tensor([head["wall_s"]], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ok = res_h[:, 0] >= 0
    # value: narrowing events of the consistent nodes only -- there the device count IS the reference's PROPS
    # (every instance is re-checked against the compiled reference below); an inconsistent node's count depends on
    # the revision order and is left out
    totals = torch.tensor([res_h[ok, 1].sum(), res_h[:, 2].sum(), B, int((~ok).sum())], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    props_all, revs_all, nodes_all, fails_all = [float(x) for x in totals.tolist()]
    info = model.device_info()
    steps = head["steps"]
    out = {
        "metric": "constraint propagations/sec + nodes/sec, queens-N, 1/2/4/8 MI355X",
        "value": props_all * steps / elapsed,
        "unit": "propagations/s",
        "nodes_per_s": nodes_all * steps / elapsed,
        "revisions_per_s": revs_all * steps / elapsed,
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": f"{problem_name}propagation-only fixpoint (BASELINE configs[{4 if args.schedule else (2 if args.sudoku else 1)}]), seeded random-walk "
                               f"node instances resident in HBM",
                   "instances_per_gpu": B, "variables": n, "clauses": info["ne_clauses"] + info["tree_clauses"],
                   "entry": head["entry"], "forbidden_sets_precomputed": head["sets_precomputed"],
                   "value_counts": "PROPS of consistent nodes (equal to the reference's, re-checked per instance)",
                   "inconsistent_fraction": fails_all / nodes_all,
                   "props_per_consistent_node": props_all / max(1.0, nodes_all - fails_all),
                   "revisions_per_node": revs_all / nodes_all},
        "roofline": roofline_record(head, workload_key, B),
    }
    if rank == 0:
        out["roofline"]["streaming_ceiling"]["device_copy_gbps_this_run"] = device_copy_gbps()
    if not legs["headline_only"]:
        out["legs"] = {k: leg_summary(v) for k, v in legs["legs"].items()}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, text, model, legs["states_in"], legs["nodes"], head["states_out"], res_h)
        if not (args.sudoku or args.schedule) and n_q == 64 and not args.no_queens128:
            out["queens128"] = queens128_record(args)
            if not args.no_sudoku25:
                out["sudoku25"] = sudoku25_record(args)
            e2e = end_to_end_record()
            if e2e is not None:
                out["end_to_end"] = e2e
    if not args.no_search and not (args.sudoku or args.schedule) and n_q == 64:
        # the propagation buffers are not needed any more: the search engines take their place in HBM
        legs.clear()
        del head, model
        torch.cuda.empty_cache()
        out["search"] = search_record(args, rank, world, dist)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def time_steps(step, steps, warmup, use_graph, dist=None):
    """warmup launches, then `steps` launches captured into one hipGraph (or launched one by one) between HIP
    events on the launching stream and between two barriers.  -> (kernel_ms per launch, wall seconds, launch kind)"""
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    graph = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(steps):
                    step()
            # Untimed replays: the first instantiates and uploads the graph; then the device is kept busy with the same
            # graph for UNTIMED_REPLAY_MS before the timed replay.  Capturing leaves the device idle for tens of ms, and the
            # K launches that follow an idle device run ~11 % slower for the first few ms (measured on one box, tools/
            # bench_cold_start.sh: --steps 20 --warmup 5 gives 0.2387 ms per launch without this, 0.2127 with 20 ms of it,
            # and --steps 100 gives 0.2132 either way) -- which is what a 20-step run is made of.  The W warmup launches
            # above are the caller's; these are part of setting the graph up, and the record says how many there were.
            graph.replay()
            torch.cuda.synchronize()
            replays = 1
            pre = float(os.environ.get("CSOLVE_BENCH_UNTIMED_REPLAY_MS", str(UNTIMED_REPLAY_MS)))
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < pre:
                graph.replay()
                torch.cuda.synchronize()
                replays += 1
            barrier()
        except Exception as exc:  # capture not available: time the plain loop
            print(f"bench: hipGraph capture failed ({exc}); timing the launch loop", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    if graph is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        graph.replay()
        e1.record()
        barrier()
        t1 = time.perf_counter()
        return (e0.elapsed_time(e1) / steps, t1 - t0,
                f"one hipGraph of the timed launches (replayed untimed {replays}x first: instantiation, upload, {pre:g} ms of load "
                f"after the idle gap of the capture)")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    t1 = time.perf_counter()
    return float(np.sum([a.elapsed_time(b) for a, b in ev])) / steps, t1 - t0, "launch loop"


def run_propagation_legs(args, text, instances, seed, dist=None, headline_only=False, steps=None):
    """The batched fixpoint over one resident instance set, through the entries of the C ABI:
      state_only     csgpu_propagate_batch: interval states in, interval states out, nothing precomputed -- what
                     propagate_clauses is defined on and what the CPU baseline gets (the headline where it exists)
      resident_sets  csgpu_propagate_batch_fb: intervals + forbidden sets carried with every state (the search
                     engine's layout); the sets of the inputs are built by an untimed launch
      sets_only      csgpu_propagate_batch_sets: the states carried as bit vectors alone
    Each leg is timed on its own with the same number of steps."""
    model = solve_root(text)
    forced = args.kernel
    model.set_kernel(forced)
    fw = model.forbidden_words()
    n = model.n_vars
    steps = steps or args.steps
    want_sets = fw > 0 and forced in (0, 3, 4, 5) and not args.rebuild_sets
    states_in, nodes, forb_in = make_instances(model, instances, seed=seed, with_sets=want_sets, restore_kernel=forced)
    B = nodes.shape[0]
    legs = {}

    def leg(name, entry, kernel_name, step, states_out, results, layout_bytes_per_node, stores_all, sets_precomputed,
            finish=None):
        kernel_ms, wall, launch = time_steps(step, steps, args.warmup, not args.no_graph, dist)
        res_h = results.cpu().numpy().astype(np.int64)
        so = finish() if finish is not None else states_out
        stored = B if stores_all else int((res_h[:, 0] >= 0).sum())
        legs[name] = {"entry": entry, "kernel": kernel_name, "kernel_ms": kernel_ms, "wall_s": wall, "launch": launch,
                      "steps": steps, "results": res_h, "states_out": so, "n": n, "B": B,
                      "sets_precomputed": sets_precomputed,
                      # what this layout moves: rows in for every node, rows out for the stored ones, record + result
                      "layout_bytes": (layout_bytes_per_node + 32) * B + layout_bytes_per_node * stored,
                      "stored_rows": stored}

    auto = model.kernel()
    names = {1: "cs_propagate_events", 2: "cs_propagate_ne_lds", 3: "cs_propagate_ne_bitset", 4: "cs_propagate_ne_regs",
             5: "cs_propagate_ne_packed", 6: "cs_propagate_clause_rounds", 7: "cs_propagate_ne_shave"}
    # --- state only (every model) ---
    so1 = torch.empty((B, n, 2), dtype=torch.int32, device="cuda")
    r1 = torch.empty((B, 4), dtype=torch.int32, device="cuda")
    k1 = names[auto]
    full_lanes = n in (64, 128, 256)
    leg("state_only", "csgpu_propagate_batch (interval states in, interval states out)", k1,
        lambda: model.propagate(states_in, nodes, so1, r1), so1, r1, 8 * n,
        stores_all=False, sets_precomputed=False)
    headline = "state_only"
    if want_sets and not headline_only:
        so2 = torch.empty((B, n, 2), dtype=torch.int32, device="cuda")
        r2 = torch.empty((B, 4), dtype=torch.int32, device="cuda")
        fo2 = torch.empty((B, n, fw), dtype=torch.int64, device="cuda")
        k2 = "cs_propagate_ne_packed" if (forced in (0, 5) and model.qualifies(5)) else \
            ("cs_propagate_ne_regs" if (forced in (0, 4, 5) and model.qualifies(4)) else "cs_propagate_ne_bitset")
        leg("resident_sets", "csgpu_propagate_batch_fb (intervals + forbidden sets in, the same out)", k2,
            lambda: model.propagate_fb(states_in, nodes, forb_in=forb_in, states_out=so2, forb_out=fo2, results=r2),
            so2, r2, 8 * n + 8 * n * fw, stores_all=(k2 == "cs_propagate_ne_regs" and full_lanes), sets_precomputed=True)
        if model.qualifies(4) and forced in (0, 4):
            sets_in = model.pack_sets(states_in)
            torch.cuda.synchronize()
            fo3 = torch.empty((B, n, fw), dtype=torch.int64, device="cuda")
            r3 = torch.empty((B, 4), dtype=torch.int32, device="cuda")
            leg("sets_only", "csgpu_propagate_batch_sets (bit-vector states in, bit-vector states out)",
                "cs_propagate_ne_regs",
                lambda: model.propagate_sets(sets_in, nodes, sets_out=fo3, results=r3), None, r3, 8 * n * fw,
                stores_all=full_lanes, sets_precomputed=True, finish=lambda: model.unpack_sets(fo3))
    if args.layout == "sets" and "sets_only" in legs:
        headline = "sets_only"
    elif args.layout == "intervals+sets" and "resident_sets" in legs:
        headline = "resident_sets"
    return {"legs": legs, "headline": headline, "model": model, "n": n, "B": B, "states_in": states_in, "nodes": nodes,
            "headline_only": headline_only}


def device_copy_gbps(bytes_each=1 << 30, reps=5):
    """the box's measured copy bandwidth next to the nominal peak -- a device-to-device copy of 1 GiB
    (read + write counted), best of `reps`, outside every timed region."""
    try:
        src = torch.empty(bytes_each, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        best = None
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        del src, dst
        torch.cuda.empty_cache()
        return 2 * bytes_each / (best * 1e-3) / 1e9
    except Exception:  # not enough memory left: the figure is an annotation, not a result
        return None


def roofline_record(leg, workload_key, B):
    """HBM roofline of one leg.  `achieved` / `frac` are on the bytes a node instance NEEDS: the parent state in
    (`struct val_t` per variable: 8 n), the 16-byte node record, the 16-byte result, and the state out (8 n) for the
    CONSISTENT nodes only -- an inconsistent node has no fixpoint to store.  SURVEY 8d's per-node figure (16 n + 32 for
    every node) is kept as `survey_8d_*`; what the layout of the leg moves on top (forbidden sets) as layout_*."""
    n = leg["n"]
    res_h = leg["results"]
    consistent = int((res_h[:, 0] >= 0).sum())
    needed = (8 * n + 32) * B + 8 * n * consistent
    survey_node_bytes = (16 * n + 32) * B
    t = leg["kernel_ms"] * 1e-3
    survey_bytes = 32 * int(res_h[:, 2].sum()) + 8 * int(res_h[:, 1].sum()) + 16 * n * B
    traffic, source = measured_traffic(leg["kernel"], workload_key, B)
    return {"bound": "hbm", "achieved": needed / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": needed / t / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
            "kernel": leg["kernel"], "kernel_ms": leg["kernel_ms"], "launch": leg["launch"],
            "algorithmic_bytes_per_launch": needed, "consistent_nodes": consistent,
            "bytes_counted": "8 n + 32 per node instance (state in, record, result) + 8 n per CONSISTENT node (state out)",
            "bytes_per_node_instance": 16 * n + 32,
            "survey_8d_node_bytes_per_launch": survey_node_bytes,
            "survey_8d_node_frac": survey_node_bytes / t / 1e9 / HBM_PEAK_GBS,
            "layout_bytes_per_launch": leg["layout_bytes"], "layout_gbps": leg["layout_bytes"] / t / 1e9,
            "layout_frac": leg["layout_bytes"] / t / 1e9 / HBM_PEAK_GBS,
            "traffic_over_necessary": None if traffic is None else traffic / needed,
            "output_rows_stored": leg["stored_rows"],
            "survey_8d_formula_gbps": survey_bytes / t / 1e9,
            # what a kernel that only moves rows of this shape reaches on this part (tools/hbm_ceiling.hip: rows in, rows
            # out, no computation; 5.4-6.1 TB/s depending on size, hipMemcpyAsync D2D 5.1-5.3): `peak` stays the nominal
            # 8 TB/s, this says how much of the gap is the memory system's
            "streaming_ceiling": {"gbps": STREAMING_CEILING_GBS, "frac_of_it": needed / t / 1e9 / STREAMING_CEILING_GBS,
                                  "source": "profiles/r01_h_hbm_streaming_ceiling.txt (tools/hbm_ceiling.hip, 2-8 GB per launch)"}}


def leg_summary(leg):
    B, n, t = leg["B"], leg["n"], leg["kernel_ms"] * 1e-3
    ok = leg["results"][:, 0] >= 0
    return {"entry": leg["entry"], "kernel": leg["kernel"], "kernel_ms": leg["kernel_ms"], "nodes_per_s": B / t,
            "propagations_per_s": int(leg["results"][ok, 1].sum()) / t,
            "forbidden_sets_precomputed": leg["sets_precomputed"],
            "frac_of_hbm_peak_on_necessary_bytes": ((8 * n + 32) * B + 8 * n * int(ok.sum())) / t / 1e9 / HBM_PEAK_GBS,
            "frac_of_hbm_peak_on_layout_bytes": leg["layout_bytes"] / t / 1e9 / HBM_PEAK_GBS,
            "layout_bytes_per_node": leg["layout_bytes"] / B}


def end_to_end_record():
    """The north-star scenario as a whole: the reference's own search driver (csolve.c, strategy.c, ... compiled from the
    reference's sources) with its default heuristics (only conflict learning off), once with the reference's CPU
    propagator and once linked against libcsolve_dropin.so (every propagate / propagate_clauses / eval on the GPU),
    on queens-64 and queens-128, same box.  The two searches are both valid but not call for call the same: the
    reference's failure chains (propagate.c:44-54) depend on its revision order."""
    import re
    import subprocess
    import tempfile
    dropin = os.path.join(ROOT, "oracle", "_ref", "csolve_ref_dropin")
    if not (os.path.exists(REF_BIN) and os.path.exists(dropin)):
        return None
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for nq in (64, 128):
            prob = os.path.join(tmp, f"queens{nq}.txt")
            open(prob, "w").write(problems.queens(nq))
            rec = {}
            for name, binary in (("cpu_reference", REF_BIN), ("reference_driver_on_gpu_propagator", dropin)):
                p = subprocess.run([binary, "solve", prob, "-c", "false"], capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    return {"error": p.stderr[-300:]}
                st = json.loads(re.search(r"@STATS (\{.*\})", p.stdout).group(1))
                r = {k: st[k] for k in ("calls", "cuts", "props", "restarts", "solutions", "solve_seconds", "root_seconds")}
                m = re.search(r"@DROPIN (\{.*\})", p.stdout)
                if m:
                    d = json.loads(m.group(1))
                    r["device_call_seconds"] = d["device_call_seconds"]
                    r["shim_host_seconds"] = d["shim_host_seconds"]
                    r["attach_seconds"] = d["attach_seconds"]  # building the device model: contains the HIP runtime's start-up
                    r["root_seconds_without_attach"] = r["root_seconds"] - d["attach_seconds"]
                    r["call_us_median"] = d.get("call_us_median")
                    r["call_us_p90"] = d.get("call_us_p90")
                r["us_per_call"] = 1e6 * st["solve_seconds"] / max(1, st["calls"])
                rec[name] = r
            rec["solve_speedup"] = rec["cpu_reference"]["solve_seconds"] / rec["reference_driver_on_gpu_propagator"]["solve_seconds"]
            rec["per_call_speedup"] = rec["cpu_reference"]["us_per_call"] / rec["reference_driver_on_gpu_propagator"]["us_per_call"]
            out[f"queens{nq}"] = rec
    out["flags"] = "-c false (defaults otherwise: -f true -r 100 -o none)"
    out["note"] = ("solve_seconds = the reference's solve() only; root_seconds (parse + root phase) is listed separately: on the "
                   "GPU build it contains attach_seconds (device model + HIP runtime start-up + start of the resident "
                   "single-node server); root_seconds_without_attach is the rest.  Every propagate_clauses of the driver is "
                   "one request to the resident server (include/csolve_gpu.h), no kernel launch")
    return out


def queens128_record(args):
    """The north-star instance next to the headline: queens-128, 2^19 seeded node instances (1.1 GB per launch), state-only
    entry, a sample of the instances re-checked against the compiled reference (about 6 s of one host core)."""
    import copy
    a = copy.copy(args)
    a.queens, a.sudoku, a.schedule, a.layout = 128, 0, 0, "intervals"
    a.cpu_seconds = min(args.cpu_seconds, 6.0)
    text = problems.queens(128)
    legs = run_propagation_legs(a, text, 1 << 19, seed=777, headline_only=True, steps=min(args.steps, 50))
    leg = legs["legs"]["state_only"]
    rec = roofline_record(leg, "queens-128", legs["B"])
    ok = leg["results"][:, 0] >= 0
    t = leg["kernel_ms"] * 1e-3
    out = {"workload": "queens-128 propagation-only fixpoint, 524288 seeded random-walk node instances resident in HBM",
           "entry": leg["entry"], "forbidden_sets_precomputed": False, "steps": leg["steps"],
           "nodes_per_s": legs["B"] / t, "value": int(leg["results"][ok, 1].sum()) / t, "unit": "propagations/s",
           "roofline": rec}
    out["cpu_baseline"] = cpu_baseline(a, text, legs["model"], legs["states_in"], legs["nodes"], leg["states_out"], leg["results"])
    out["speedup_over_reference_core"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def sudoku25_record(args):
    """BASELINE configs[2] next to the headline: the 25x25 sudoku-shaped != network (625 variables, 72 neighbours each),
    2^18 seeded node instances, state-only entry, a sample re-checked against the compiled reference."""
    import copy
    a = copy.copy(args)
    a.queens, a.sudoku, a.schedule, a.layout = 64, 5, 0, "intervals"
    a.cpu_seconds = min(args.cpu_seconds, 4.0)
    text = problems.sudoku(5, 0.4, 1)  # 40 % of a seeded valid grid revealed
    legs = run_propagation_legs(a, text, 1 << 19, seed=555, headline_only=True, steps=min(args.steps, 30))  # 5.2 GB per launch (2^18: 0.47, 2^19: 0.52, 2^20: 0.53 of the roofline)
    leg = legs["legs"]["state_only"]
    rec = roofline_record(leg, "sudoku-25x25", legs["B"])
    ok = leg["results"][:, 0] >= 0
    t = leg["kernel_ms"] * 1e-3
    out = {"workload": "sudoku-25x25 (40 % givens) propagation-only fixpoint, 524288 seeded random-walk node instances resident in HBM",
           "entry": leg["entry"], "forbidden_sets_precomputed": False, "steps": leg["steps"],
           "nodes_per_s": legs["B"] / t, "value": int(leg["results"][ok, 1].sum()) / t, "unit": "propagations/s",
           "roofline": rec}
    out["cpu_baseline"] = cpu_baseline(a, text, legs["model"], legs["states_in"], legs["nodes"], leg["states_out"], leg["results"])
    out["speedup_over_reference_core"] = out["value"] / out["cpu_baseline"]["value"]
    return out


if __name__ == "__main__":
    main()
