"""CPU tests (gloo, world_size 2, 3 and 5) of the multi-rank search coordinator: seeding by every rank,
the shared status page, work stealing of open states, incumbent exchange, termination.  The engines are the oracle-backed
CPU stand-in (tests/cpu_engine.py); the coordinator code is exactly what runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, golden


def test_transfer_plan_is_deterministic_and_conserving():
    from csolve_amd.parallel import plan_transfers
    plan = plan_transfers([0, 1000, 10, 400], low_water=64)
    assert plan == [(1, 0, 500), (3, 2, 195)]
    pools = [0, 1000, 10, 400]
    for s, d, c in plan:
        pools[s] -= c
        pools[d] += c
    assert sum(pools) == 1410 and min(pools) >= 64
    assert plan_transfers([100, 100], 64) == [] and plan_transfers([0, 1], 64) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, text, out_q, options=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from cpu_engine import OracleEngine
    from csolve_amd.parallel import ShardedSearch
    from oracle.cs_oracle import Model as OModel, Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    om = OModel.parse(text)
    o0 = Oracle(om)
    o0.set_root_phase(True)
    assert o0.propagate(om.root, om.n_vars) >= 0
    om.set_domains(o0.domains())
    om.index()
    options = dict(options or {})
    shuffle = options.pop("engine_shuffle", None)
    eng = OracleEngine(om, shuffle_seed=None if shuffle is None else shuffle + rank)
    sh = ShardedSearch(eng, om.view.objective, om.n_vars, rank, world, dist, engine_device="cpu",
                       **dict(dict(slice_iterations=8, seed_states_per_rank=4, low_water=4), **options))
    root = torch.from_numpy(om.domains()).unsqueeze(0).contiguous()
    local, totals = sh.run(root)
    assert sh.seconds["total"] >= sh.seconds["busy"] > 0 and 0.0 <= sh.idle_fraction() <= 1.0
    out_q.put((rank, local["nodes"], local["solutions"], totals["nodes"], totals["solutions"], totals["best"],
               sh.states_moved, bool(sh.seeded_alike), sh.early_exchanges, bool(totals["timeout"])))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, text, options=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, text, q, options)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=150) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return res


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_all_solutions(world):
    """queens-8 ALL over 2, 3 and 5 ranks: 92 solutions in total, every rank does part of the work, every rank
    computed the same seed frontier by itself, and the total number of explored nodes equals the single-engine
    tree (the common seeding phase counted once)."""
    from csolve_amd import problems
    res = _run(world, problems.queens(8, "ALL"))
    assert all(r[4] == 92 for r in res)
    assert sum(r[2] for r in res) == 92
    assert all(r[1] > 0 for r in res), "a rank stayed idle"
    assert all(r[7] for r in res), "the ranks' seed frontiers differed"
    single = _run(1, problems.queens(8, "ALL"))
    assert res[0][3] == single[0][3]


def test_sharded_search_without_the_page_and_with_rank0_seeding():
    """the fallbacks: no shared status page (ranks on different nodes) and the frontier broadcast from rank 0;
    long slices with the page: a dry rank calls the exchange before the slice is used up"""
    from csolve_amd import problems
    text = problems.queens(8, "ALL")
    single = _run(1, text)
    res = _run(3, text, dict(status_page=False, seed_on_every_rank=False))
    assert all(r[4] == 92 for r in res) and res[0][3] == single[0][3] and not any(r[7] for r in res)
    assert all(r[8] == 0 for r in res)
    res = _run(3, text, dict(slice_iterations=1 << 20, poll_iterations=2))
    assert all(r[4] == 92 for r in res) and res[0][3] == single[0][3]
    assert sum(r[8] for r in res) > 0, "nobody answered a dry rank's call"


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_whose_frontiers_are_ordered_differently_fall_back_to_rank_zeros(world):
    """engines that leave a frontier in a timing-dependent order (the GPU engine's level kernels do): every rank expands
    the root to the same SET of open states in its own order, the checksums differ, rank 0's frontier is broadcast and
    the ranks take their shares of THAT -- no subtree is walked twice or left out: the single-engine tree, 92 solutions"""
    from csolve_amd import problems
    text = problems.queens(8, "ALL")
    single = _run(1, text)
    res = _run(world, text, dict(engine_shuffle=1234))
    assert all(r[4] == 92 for r in res) and sum(r[2] for r in res) == 92
    assert res[0][3] == single[0][3], "a subtree was walked twice or not at all"
    assert not any(r[7] for r in res), "differently ordered frontiers must not pass for the same"
    assert all(r[1] > 0 for r in res)


def test_sharded_minimisation_shares_the_incumbent():
    """schedule-6 MIN over 2 ranks: optimum 22 (reference golden), known to every rank."""
    res = _run(2, open(golden("problems", "schedule6_s1.txt")).read())
    assert all(r[5] == 22 for r in res)


def _oracle_engine(text):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_engine import OracleEngine
    from oracle.cs_oracle import Model as OModel, Oracle
    om = OModel.parse(text)
    o0 = Oracle(om)
    o0.set_root_phase(True)
    assert o0.propagate(om.root, om.n_vars) >= 0
    om.set_domains(o0.domains())
    om.index()
    return om, OracleEngine(om)


def test_lanes_in_one_process():
    """LaneSearch (several engines in one process, a host thread each, the exchange of ShardedSearch without
    collectives) over the oracle-backed CPU engines: queens-7 ALL walks the single-engine tree with every lane
    taking part; schedule-6 MIN reaches 22."""
    from csolve_amd import problems
    from csolve_amd.parallel import LaneSearch
    text = problems.queens(7, "ALL")
    om, single = _oracle_engine(text)
    root = torch.from_numpy(om.domains()).unsqueeze(0).contiguous()
    single.put(root)
    ref = single.run(1 << 30)
    assert ref["solutions"] == 40
    lanes = [_oracle_engine(text)[1] for _ in range(3)]
    tot = LaneSearch(lanes, om.view.objective, slice_iterations=4, seed_states_per_lane=4, low_water=4).run(root)
    assert tot["done"] == 1 and (tot["nodes"], tot["cuts"], tot["solutions"]) == (ref["nodes"], ref["cuts"], 40)
    assert all(x > 0 for x in tot["lanes"])
    text = open(golden("problems", "schedule6_s1.txt")).read()
    om, _ = _oracle_engine(text)
    root = torch.from_numpy(om.domains()).unsqueeze(0).contiguous()
    lanes = [_oracle_engine(text)[1] for _ in range(2)]
    tot = LaneSearch(lanes, om.view.objective, slice_iterations=8, seed_states_per_lane=4, low_water=4).run(root)
    assert tot["done"] == 1 and tot["best"] == 22


def test_time_limit_stops_every_rank_with_partial_results():
    """the reference's -t (shared()->timeout, csolve.c:190-203,408): after the time limit every rank stops in the same
    exchange, the totals say so and hold what was found so far; without a limit the flag stays false"""
    from csolve_amd import problems
    res = _run(3, problems.queens(10, "ALL"), dict(time_limit=0.03, slice_iterations=4, poll_iterations=1))
    assert all(r[9] for r in res), "a rank did not see the timeout"
    assert len({(r[3], r[4]) for r in res}) == 1, "the ranks disagree on the totals"
    assert 0 < res[0][3] and res[0][4] < 724, res[0]  # queens-10 has 724 solutions: the search was cut short
    full = _run(2, problems.queens(6, "ALL"))
    assert not any(r[9] for r in full) and full[0][4] == 4
