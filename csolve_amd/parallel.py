"""Subtree sharding of the search across ranks (one process per GPU, torch.distributed).

Reference mechanism being replaced: `worker_spawn` forks a child that takes the
upper half of the branching variable's interval whenever a worker slot is free
(reference src/csolve.c:105-152), all workers share one page holding the incumbent objective
value, the solution count and the timeout flag (csolve.c:86-97, objective.c:89-93,135).
HIP state does not survive fork(), so ranks exist up front and exchange
  * open states (whole subtrees) taken from the OLDEST end of a rank's pool -- work stealing,
  * the incumbent bound (min / max over ranks) and the found-a-solution / pool-size words.
Over RCCL (backend "nccl") the exchanged tensors stay in device memory and travel over xGMI;
the same code runs over gloo with host tensors (CPU tests, or several ranks sharing one GPU).

Propagation of a single node is never split across ranks: the data path has no collective.

LaneSearch (below) is the same scheme inside one process: several engines on one GPU, a host thread each.
"""
from __future__ import annotations

import torch

OBJ_ANY, OBJ_ALL, OBJ_MIN, OBJ_MAX = 0, 1, 2, 3
INT32_MAX, INT32_MIN = 2**31 - 1, -(2**31)


def plan_transfers(pools, low_water: int):
    """Deterministic rebalancing plan from the gathered pool sizes: pair the richest rank with
    the poorest while the poorest is below `low_water` and the richest can spare states.
    -> list of (src, dst, count).  Every rank computes the same plan."""
    pools = list(pools)
    plan = []
    order = sorted(range(len(pools)), key=lambda r: (pools[r], r))
    lo, hi = 0, len(order) - 1
    while lo < hi:
        poor, rich = order[lo], order[hi]
        if pools[poor] >= low_water:
            break
        give = (pools[rich] - pools[poor]) // 2
        if give <= 0:
            break
        plan.append((rich, poor, give))
        pools[rich] -= give
        pools[poor] += give
        lo += 1
        hi -= 1
    return plan


class ShardedSearch:
    """Runs one search engine per rank and keeps them busy.

    engine: object with put(states), take(k) -> states, run(iterations) -> stats dict,
            set_best(value); states are int32 tensors [k, n_vars, 2] on `engine_device`.
    comm_device: device of the tensors handed to torch.distributed ("cuda" for nccl/RCCL,
            "cpu" for gloo).
    """

    def __init__(self, engine, objective: int, n_vars: int, rank: int, world: int, dist=None,
                 engine_device="cuda", comm_device=None, slice_iterations: int = 64, seed_states_per_rank: int = 64,
                 low_water: int = 64):
        self.engine, self.objective, self.n, self.rank, self.world = engine, objective, n_vars, rank, world
        self.dist = dist
        self.engine_device = engine_device
        self.comm_device = comm_device or engine_device
        self.slice_iterations = slice_iterations
        self.seed_states_per_rank = seed_states_per_rank
        self.low_water = low_water
        self.exchanges = 0
        self.states_moved = 0

    # ---- helpers ---------------------------------------------------------------------------
    def _to_comm(self, t):
        return t.to(self.comm_device).contiguous()

    def _to_engine(self, t):
        return t.to(self.engine_device).contiguous()

    def _gather_words(self, words):
        """all_gather of a few int64 words per rank -> [world, len(words)] (host list)"""
        mine = torch.tensor(words, dtype=torch.int64, device=self.comm_device)
        if self.dist is None or self.world == 1:
            return [list(words)]
        out = torch.empty(self.world * len(words), dtype=torch.int64, device=self.comm_device)
        self.dist.all_gather_into_tensor(out, mine)
        return out.view(self.world, len(words)).cpu().tolist()

    # ---- phases ----------------------------------------------------------------------------
    def seed(self, root_state):
        """Rank 0 expands the root until there are enough open states, then deals them out
        round-robin (the analogue of the reference's repeated interval halving)."""
        stats = None
        if self.world == 1:
            self.engine.put(root_state)
            return
        frontier = None
        if self.rank == 0:
            self.engine.put(root_state)
            want = self.seed_states_per_rank * self.world
            stats = self.engine.run(1)
            while not stats["done"] and stats["pool"] < want:
                stats = self.engine.run(1)
            frontier = self._to_comm(self.engine.take(stats["pool"]))
        count = torch.tensor([0 if frontier is None else frontier.shape[0]], dtype=torch.int64, device=self.comm_device)
        self.dist.broadcast(count, src=0)
        k = int(count.item())
        if k == 0:
            return
        if frontier is None:
            frontier = torch.empty((k, self.n, 2), dtype=torch.int32, device=self.comm_device)
        self.dist.broadcast(frontier, src=0)
        mine = frontier[self.rank::self.world]
        if mine.shape[0] > 0:
            self.engine.put(self._to_engine(mine))

    def _exchange(self, stats):
        """incumbent, termination and work stealing; returns True when the search is over"""
        found = 1 if stats["solutions"] > 0 else 0
        table = self._gather_words([stats["pool"], stats["best"], found])
        pools = [int(r[0]) for r in table]
        if self.objective == OBJ_MIN:
            self.engine.set_best(min(int(r[1]) for r in table))
        elif self.objective == OBJ_MAX:
            self.engine.set_best(max(int(r[1]) for r in table))
        if self.objective == OBJ_ANY and any(int(r[2]) for r in table):
            return True
        if sum(pools) == 0:
            return True
        for src, dst, cnt in plan_transfers(pools, self.low_water):
            self.exchanges += 1
            if self.rank == src:
                states = self._to_comm(self.engine.take(cnt))
                assert states.shape[0] == cnt
                self.dist.send(states, dst=dst)
                self.states_moved += cnt
            elif self.rank == dst:
                buf = torch.empty((cnt, self.n, 2), dtype=torch.int32, device=self.comm_device)
                self.dist.recv(buf, src=src)
                self.engine.put(self._to_engine(buf))
        return False

    def run(self, root_state, max_slices: int = 1 << 40):
        """-> (local stats dict, global totals dict)"""
        self.seed(root_state)
        stats = self.engine.run(0)
        for _ in range(max_slices):
            stats = self.engine.run(self.slice_iterations)
            if self.world == 1:
                if stats["done"]:
                    break
                continue
            if self._exchange(stats):
                break
        totals = dict(stats)
        if self.dist is not None and self.world > 1:
            keys = ("nodes", "cuts", "props", "revisions", "solutions", "iterations")
            t = torch.tensor([stats[k] for k in keys], dtype=torch.int64, device=self.comm_device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            for k, v in zip(keys, t.cpu().tolist()):
                totals[k] = int(v)
            b = torch.tensor([stats["best"]], dtype=torch.int64, device=self.comm_device)
            if self.objective == OBJ_MIN:
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MIN)
            elif self.objective == OBJ_MAX:
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MAX)
            totals["best"] = int(b.item())
        return stats, totals


class LaneSearch:
    """Several engines ("lanes") of one model on ONE GPU, a host thread each.

    An ANY / MIN / MAX iteration is a handful of small dependent launches, so a single engine leaves most of the
    device idle between them (four ranks sharing one GPU finish a schedule.txt-style MIN search 1.8x sooner than one
    rank).  The lanes are what ranks are to ShardedSearch -- own pool, own stream (the engine's device-driven
    bursts), same exchange of incumbent, termination and open states after every slice -- without processes or
    collectives: the engines' calls release the GIL and their bursts overlap on the device.

    Reproducibility: ANY and ALL runs are reproducible whatever the thread timing (the exchange decisions are
    functions of the lanes' statistics at slice boundaries only).  MIN / MAX runs share ONE incumbent word in device
    memory that every lane's accept kernel updates while the others are in the middle of a slice, so which nodes a
    lane cuts depends on when a neighbour's improvement lands: the optimum ("best") is the same in every run, the
    node / cut / iteration counts are not.  The row attaining the optimum is held by the lane that found it
    (best_solution()).
    """

    def __init__(self, engines, objective: int, slice_iterations: int = 32, seed_states_per_lane: int = 64,
                 low_water: int = 64):
        assert len(engines) >= 1
        self.engines, self.objective = list(engines), objective
        self.slice_iterations = slice_iterations
        self.seed_states_per_lane = seed_states_per_lane
        self.low_water = low_water
        self.states_moved = 0

    def run(self, root_state, max_slices: int = 1 << 40):
        """-> totals dict (sums of the lanes' counters, best over the lanes, 'lanes': per-lane node counts)"""
        from concurrent.futures import ThreadPoolExecutor
        lanes = self.engines
        first = lanes[0]
        if self.objective in (OBJ_MIN, OBJ_MAX) and hasattr(first, "share_incumbent"):
            for lane in lanes[1:]:
                lane.share_incumbent(first)  # one word of device memory: a better solution bounds every lane at once
        first.put(root_state)
        stats = [first.run(0)] + [None] * (len(lanes) - 1)
        if len(lanes) > 1:
            want = self.seed_states_per_lane * len(lanes)
            st = first.run(1)
            while not st["done"] and st["pool"] < want:
                st = first.run(1)
            if st["pool"] > 0 and not st["done"]:
                frontier = first.take(st["pool"])
                for i, lane in enumerate(lanes):
                    mine = frontier[i::len(lanes)]
                    if mine.shape[0] > 0:
                        lane.put(mine.contiguous())
        with ThreadPoolExecutor(max_workers=len(lanes)) as pool:
            for _ in range(max_slices):
                stats = list(pool.map(lambda e: e.run(self.slice_iterations), lanes))
                if self.objective == OBJ_MIN:
                    best = min(s["best"] for s in stats)
                elif self.objective == OBJ_MAX:
                    best = max(s["best"] for s in stats)
                if self.objective in (OBJ_MIN, OBJ_MAX):
                    for e in lanes:
                        e.set_best(best)
                if self.objective == OBJ_ANY and any(s["solutions"] > 0 for s in stats):
                    break
                pools = [s["pool"] for s in stats]
                if sum(pools) == 0:
                    break
                for src, dst, cnt in plan_transfers(pools, self.low_water):
                    lanes[dst].put(lanes[src].take(cnt).contiguous())
                    self.states_moved += cnt
        totals = {k: sum(s[k] for s in stats) for k in ("nodes", "cuts", "props", "revisions", "solutions", "iterations")}
        if self.objective == OBJ_MIN:
            totals["best"] = min(s["best"] for s in stats)
        elif self.objective == OBJ_MAX:
            totals["best"] = max(s["best"] for s in stats)
        else:
            totals["best"] = 0
        totals["done"] = int(all(s["done"] for s in stats) or (self.objective == OBJ_ANY and totals["solutions"] > 0))
        totals["lanes"] = [s["nodes"] for s in stats]
        return totals

    def best_solution(self):
        """MIN / MAX: the values of a solution attaining the lanes' common incumbent (the lane that found it holds the
        row; the others report none), or None"""
        for e in self.engines:
            row = e.best_solution() if hasattr(e, "best_solution") else None
            if row is not None:
                return row
        return None
