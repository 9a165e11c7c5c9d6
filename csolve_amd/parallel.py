"""Subtree sharding of the search across ranks (one process per GPU, torch.distributed).

Reference mechanism being replaced: `worker_spawn` forks a child that takes the
upper half of the branching variable's interval whenever a worker slot is free
(reference src/csolve.c:105-152), all workers share one page holding the incumbent objective
value, the solution count and the timeout flag (csolve.c:86-97, objective.c:89-93,135).
HIP state does not survive fork(), so ranks exist up front; every rank expands the root the same way and keeps
its share of the frontier, and the ranks exchange
  * open states (whole subtrees) taken from the OLDEST end of a rank's pool -- work stealing,
  * the incumbent bound (min / max over ranks) and the found-a-solution / pool-size words,
at slice boundaries through collectives and, between them, through a page of shared memory (StatusPage) that a
dry rank uses to call the next exchange early.
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


class StatusPage:
    """One page of memory shared by the ranks of a node (a file in /dev/shm, mapped by every rank): per rank the
    exchange it is waiting at, its incumbent, whether it has found a solution, its pool size.  The analogue of the
    page the reference's forked workers share (reference src/csolve.c:86-97: objective_best, solutions found,
    timeout flag): a rank reads its neighbours' words between bursts of iterations without any collective, so an
    incumbent bounds the other ranks, and a dry rank's request for work is seen, within one burst rather than one
    slice.  Words are aligned int64, written by exactly one rank each."""
    WANT, BEST, FOUND, POOL, TIMEOUT, WORDS = 0, 1, 2, 3, 4, 8

    def __init__(self, path: str, rank: int, world: int, create: bool):
        import mmap
        import os
        import numpy as np
        self.path, self.rank, self.world = path, rank, world
        size = world * self.WORDS * 8
        if create:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, size)
        else:
            fd = os.open(path, os.O_RDWR)
        try:
            self._map = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self.words = np.frombuffer(self._map, dtype=np.int64).reshape(world, self.WORDS)

    def publish(self, want=None, best=None, found=None, pool=None):
        row = self.words[self.rank]
        if best is not None:
            row[self.BEST] = best
        if found is not None:
            row[self.FOUND] = found
        if pool is not None:
            row[self.POOL] = pool
        if want is not None:
            row[self.WANT] = want  # last: the other words are in place when a neighbour sees the request

    def waiting_at(self):
        """the latest exchange any rank is waiting at"""
        return int(self.words[:, self.WANT].max())

    def best(self, objective):
        col = self.words[:, self.BEST]
        return int(col.min()) if objective == OBJ_MIN else int(col.max())

    def any_found(self):
        return bool(self.words[:, self.FOUND].any())

    def set_timeout(self):
        self.words[self.rank, self.TIMEOUT] = 1

    def timed_out(self):
        return bool(self.words[:, self.TIMEOUT].any())

    def close(self):
        self.words = None
        try:
            self._map.close()
        except BufferError:
            pass


class ShardedSearch:
    """Runs one search engine per rank and keeps them busy.

    engine: object with put(states), take(k) -> states, run(iterations) -> stats dict,
            set_best(value); states are int32 tensors [k, n_vars, 2] on `engine_device`.
    comm_device: device of the tensors handed to torch.distributed ("cuda" for nccl/RCCL,
            "cpu" for gloo).
    poll_iterations: a slice is run in bursts of this many iterations; between bursts the rank looks at the
            node's StatusPage (no collective, no device synchronisation beyond the burst's own).
    status_page: None = use one when all ranks can map the same /dev/shm file (one node), False = never.
    time_limit: seconds after which the search stops on every rank with what it has (the reference's -t: SIGALRM sets
            shared()->timeout, which every worker's loop tests, csolve.c:190-203,408); totals["timeout"] tells.

    Time accounting (self.seconds): "seed" (expanding the root, done by every rank alike), "busy" (inside
    engine.run with a non-empty pool), "exchange" (collectives, transfers and the wait for the slowest rank),
    "total"; idle_fraction() = the part of the time after seeding that was not spent in the engine.
    """

    COUNTERS = ("nodes", "cuts", "props", "revisions", "solutions", "iterations")

    def __init__(self, engine, objective: int, n_vars: int, rank: int, world: int, dist=None,
                 engine_device="cuda", comm_device=None, slice_iterations: int = 64, seed_states_per_rank: int = 64,
                 low_water: int = 64, poll_iterations: int = 4, status_page=None, seed_on_every_rank: bool = True,
                 time_limit: float = None):
        self.engine, self.objective, self.n, self.rank, self.world = engine, objective, n_vars, rank, world
        self.time_limit = time_limit  # seconds, like the reference's -t (timeout_init / shared()->timeout, csolve.c:190-203,408)
        self.timed_out = False
        self._deadline = None
        self.dist = dist
        self.engine_device = engine_device
        self.comm_device = comm_device or engine_device
        self.slice_iterations = slice_iterations
        self.seed_states_per_rank = seed_states_per_rank
        self.low_water = low_water
        self.poll_iterations = max(1, poll_iterations)
        self.want_page = status_page
        self.seed_on_every_rank = seed_on_every_rank
        self.page = None
        self.page_used = False  # whether the ranks shared a StatusPage in the last run
        self.exchanges = 0
        self.early_exchanges = 0  # exchanges entered because a neighbour asked, before the slice was used up
        self.states_moved = 0
        self.seed_counters = {k: 0 for k in self.COUNTERS}  # what this rank spent on the common seeding phase
        self.seeded_alike = None
        self.seconds = {"seed": 0.0, "busy": 0.0, "exchange": 0.0, "total": 0.0}

    # ---- helpers ---------------------------------------------------------------------------
    def _to_comm(self, t):
        return t.to(self.comm_device).contiguous()

    def _to_engine(self, t):
        return t.to(self.engine_device).contiguous()

    def _gather_words(self, words):
        """all_gather of a few int64 words per rank -> [world, len(words)] (host list)"""
        if self.dist is None:
            return [list(words)]
        mine = torch.tensor(words, dtype=torch.int64, device=self.comm_device)
        out = torch.empty(self.world * len(words), dtype=torch.int64, device=self.comm_device)
        self.dist.all_gather_into_tensor(out, mine)
        return out.view(self.world, len(words)).cpu().tolist()

    def _open_page(self):
        """rank 0 creates the page, every rank maps it, all agree whether everybody could (one node) or not"""
        import os
        if self.want_page is False or self.dist is None or self.world == 1:
            return
        token = int.from_bytes(os.urandom(7), "little") if self.rank == 0 else 0
        token = int(self._gather_words([token])[0][0])
        path = f"/dev/shm/csolve_amd_page_{token:x}"
        page, ok = None, 1
        if self.rank == 0:
            try:
                page = StatusPage(path, self.rank, self.world, create=True)
            except OSError:
                ok = 0
        self._gather_words([ok])  # the file exists from here on
        if self.rank != 0:
            try:
                page = StatusPage(path, self.rank, self.world, create=False)
            except OSError:
                ok = 0
        everybody = all(int(r[0]) for r in self._gather_words([ok]))
        if self.rank == 0 and page is not None:
            os.unlink(path)  # the mappings keep it alive
        if everybody:
            self.page = page
            self.page_used = True
        elif page is not None:
            page.close()

    # ---- phases ----------------------------------------------------------------------------
    def _expand_root(self, root_state):
        """expand the root until there are enough open states for all ranks -> (frontier, stats)"""
        self.engine.put(root_state)
        want = self.seed_states_per_rank * self.world
        stats = self.engine.run(1)
        while not stats["done"] and stats["pool"] < want and not (self.objective == OBJ_ANY and stats["solutions"] > 0):
            stats = self.engine.run(1)
        frontier = self.engine.take(stats["pool"]) if stats["pool"] > 0 else None
        return frontier, stats

    def seed(self, root_state):
        """Every rank expands the root the same (deterministic) way and keeps every world-th open state: no rank
        waits for rank 0 and nothing is sent.  The ranks compare a checksum of their frontiers; should they ever
        differ, rank 0's frontier is broadcast instead.  (The analogue of the reference's repeated interval
        halving between forked workers, csolve.c:105-152.)
        seed_on_every_rank=False: rank 0 alone expands and broadcasts (the round-1 scheme)."""
        if self.world == 1:
            self.engine.put(root_state)
            self.seeded_alike = True
            return
        frontier, stats = None, None
        if self.seed_on_every_rank or self.rank == 0:
            frontier, stats = self._expand_root(root_state)
        if stats is not None and self.rank != 0:
            self.seed_counters = {k: int(stats[k]) for k in self.COUNTERS}  # rank 0's copy is the one that counts
        k = 0 if frontier is None else int(frontier.shape[0])
        if self.seed_on_every_rank:
            check = 0
            if k > 0:
                flat = frontier.reshape(-1).to(torch.int64)
                weights = torch.arange(flat.numel(), dtype=torch.int64, device=flat.device) % 8191 + 1
                check = int((flat * weights).sum().item())
            table = self._gather_words([k, check])
            self.seeded_alike = all(r == table[0] for r in table)
        else:
            self.seeded_alike = False
        if not self.seeded_alike:
            count = torch.tensor([k if self.rank == 0 else 0], dtype=torch.int64, device=self.comm_device)
            self.dist.broadcast(count, src=0)
            k = int(count.item())
            if k > 0:
                if self.rank == 0:
                    frontier = self._to_comm(frontier)
                else:
                    frontier = torch.empty((k, self.n, 2), dtype=torch.int32, device=self.comm_device)
                self.dist.broadcast(frontier, src=0)
                frontier = self._to_engine(frontier)
        if k == 0:
            return
        mine = frontier[self.rank::self.world]
        if mine.shape[0] > 0:
            self.engine.put(mine.contiguous())

    def _exchange(self, stats):
        """incumbent, termination and work stealing; returns True when the search is over"""
        found = 1 if stats["solutions"] > 0 else 0
        table = self._gather_words([stats["pool"], stats["best"], found, 1 if self._expired() else 0])
        pools = [int(r[0]) for r in table]
        if any(int(r[3]) for r in table):  # one rank's clock is everybody's: all stop in the same exchange
            self.timed_out = True
        if self.objective == OBJ_MIN:
            self.engine.set_best(min(int(r[1]) for r in table))
        elif self.objective == OBJ_MAX:
            self.engine.set_best(max(int(r[1]) for r in table))
        if self.objective == OBJ_ANY and any(int(r[2]) for r in table):
            return True
        if sum(pools) == 0 or self.timed_out:
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

    def _expired(self):
        import time
        return self._deadline is not None and time.perf_counter() >= self._deadline

    def _slice(self, stats, epoch):
        """bursts of iterations until the slice is used up, the pool is dry, or a neighbour waits at the next
        exchange -> stats"""
        import time
        used = 0
        # bursts exist to look at the neighbours' page and at the clock: a rank with neither runs the slice in one call
        # (a MIN / MAX engine enqueues 16 iterations per host round trip; bursts of 4 doubled schedule-12's time)
        burst = self.poll_iterations if (self.page is not None or self._deadline is not None) else self.slice_iterations
        if self.page is None and self._deadline is not None:
            burst = max(self.poll_iterations, 16)  # only the clock to look at
        while used < self.slice_iterations:
            k = min(burst, self.slice_iterations - used)
            had_work = stats["pool"] > 0
            t0 = time.perf_counter()
            stats = self.engine.run(k)
            if had_work:
                self.seconds["busy"] += time.perf_counter() - t0
            used += k
            if stats["done"] or stats["pool"] == 0:
                break
            if self.objective == OBJ_ANY and stats["solutions"] > 0:
                break
            if self._expired():
                if self.page is not None:
                    self.page.set_timeout()
                break
            page = self.page
            if page is not None:
                page.publish(best=stats["best"], found=1 if stats["solutions"] > 0 else 0, pool=stats["pool"])
                if self.objective in (OBJ_MIN, OBJ_MAX):
                    self.engine.set_best(page.best(self.objective))
                if page.waiting_at() > epoch or page.timed_out() or (self.objective == OBJ_ANY and page.any_found()):
                    if used < self.slice_iterations:
                        self.early_exchanges += 1
                    break
        return stats

    def run(self, root_state, max_slices: int = 1 << 40):
        """-> (local stats dict, global totals dict).  Counters of the common seeding phase are counted once."""
        import time
        t_start = time.perf_counter()
        self._deadline = None if self.time_limit is None else t_start + self.time_limit
        self._open_page()
        if self.page is not None:
            self.page.publish(want=0, best=INT32_MAX if self.objective == OBJ_MIN else INT32_MIN, found=0, pool=0)
            self._gather_words([0])  # every row is initialised before anyone reads a neighbour's
        self.seed(root_state)
        stats = self.engine.run(0)
        self.seconds["seed"] = time.perf_counter() - t_start
        epoch = 0
        for _ in range(max_slices):
            stats = self._slice(stats, epoch)
            if self.dist is None:
                if self._expired():
                    self.timed_out = True
                if stats["done"] or self.timed_out:
                    break
                continue
            epoch += 1
            if self.page is not None:
                self.page.publish(want=epoch, best=stats["best"], found=1 if stats["solutions"] > 0 else 0,
                                  pool=stats["pool"])
            t0 = time.perf_counter()
            over = self._exchange(stats)
            self.seconds["exchange"] += time.perf_counter() - t0
            if over:
                break
            stats = self.engine.run(0)
        local = dict(stats)
        for k in self.COUNTERS:
            local[k] = int(local[k]) - self.seed_counters[k]
        totals = dict(local)
        if self.dist is not None:
            t = torch.tensor([local[k] for k in self.COUNTERS], dtype=torch.int64, device=self.comm_device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            for k, v in zip(self.COUNTERS, t.cpu().tolist()):
                totals[k] = int(v)
            b = torch.tensor([stats["best"]], dtype=torch.int64, device=self.comm_device)
            if self.objective == OBJ_MIN:
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MIN)
            elif self.objective == OBJ_MAX:
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MAX)
            totals["best"] = int(b.item())
        totals["timeout"] = local["timeout"] = bool(self.timed_out)
        if self.page is not None:
            self.page.close()
            self.page = None
        self.seconds["total"] = time.perf_counter() - t_start
        return local, totals

    def idle_fraction(self):
        """the part of this rank's time after seeding that it did not spend inside its engine"""
        after = self.seconds["total"] - self.seconds["seed"]
        return max(0.0, 1.0 - self.seconds["busy"] / after) if after > 0 else 0.0


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
