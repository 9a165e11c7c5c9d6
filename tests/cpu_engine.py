"""A CPU stand-in for csolve_amd.solver.Search with the same interface and the same tree
(smallest-interval branching, halving of wide intervals, LIFO pool, incumbent bound), built on
the oracle.  TEST INFRASTRUCTURE: lets the multi-rank coordinator (csolve_amd/parallel.py) run
end to end over gloo without a GPU."""
import numpy as np
import torch

from csolve_amd.parallel import INT32_MAX, INT32_MIN, OBJ_ANY, OBJ_MAX, OBJ_MIN
from oracle.cs_oracle import Oracle

SPLIT_WIDTH = 256


class OracleEngine:
    def __init__(self, omodel, parents_per_iteration=4, shuffle_seed=None):
        """shuffle_seed: permute the pool after every iteration with this seed -- the level kernels of the GPU engine
        (cs_step.hip.h) leave a frontier's survivors in an order that depends on timing, so two ranks that expand the same
        root hold the same SET of open states in different orders"""
        self.rng = None if shuffle_seed is None else np.random.default_rng(shuffle_seed)
        self.m = omodel
        self.orc = Oracle(omodel)
        self.objective = omodel.view.objective
        self.obj_var = omodel.view.obj_var
        self.pool = []
        self.parents = parents_per_iteration
        self.st = dict(nodes=0, cuts=0, props=0, revisions=0, solutions=0, iterations=0, restarts=0, pool=0, pool_peak=0,
                       best=INT32_MAX if self.objective == OBJ_MIN else (INT32_MIN if self.objective == OBJ_MAX else 0),
                       done=0)
        self.found = []

    def put(self, states):
        for s in states.cpu().numpy():
            self.pool.append(s.copy())
        self.st["pool_peak"] = max(self.st["pool_peak"], len(self.pool))

    def take(self, k):
        k = min(k, len(self.pool))
        out, self.pool = self.pool[:k], self.pool[k:]
        n = self.m.n_vars
        return torch.from_numpy(np.stack(out) if out else np.zeros((0, n, 2), np.int32))

    def set_best(self, best):
        if (self.objective == OBJ_MIN and best < self.st["best"]) or (self.objective == OBJ_MAX and best > self.st["best"]):
            self.st["best"] = int(best)

    def _child(self, state, var, lo, hi):
        dom = state.copy()
        tightened = False
        if self.obj_var >= 0:
            o = self.obj_var
            if self.objective == OBJ_MIN and self.st["best"] != INT32_MAX and dom[o, 1] > self.st["best"] - 1:
                dom[o, 1] = self.st["best"] - 1
                tightened = True
            if self.objective == OBJ_MAX and self.st["best"] != INT32_MIN and dom[o, 0] < self.st["best"] + 1:
                dom[o, 0] = self.st["best"] + 1
                tightened = True
            if dom[o, 0] > dom[o, 1]:
                return -1, dom
        status, out = self.orc.instance(dom, var, lo, hi)
        self.st["props"] += self.orc.props()
        if status >= 0 and tightened:
            o = self.obj_var
            status, out = self.orc.instance(out, o, int(out[o, 0]), int(out[o, 1]))
            self.st["props"] += self.orc.props()
        return status, out

    def run(self, iterations):
        for _ in range(iterations):
            if not self.pool or (self.objective == OBJ_ANY and self.st["solutions"] > 0):
                break
            self.st["iterations"] += 1
            take = min(self.parents, len(self.pool))
            parents, self.pool = self.pool[-take:], self.pool[:-take]
            for state in parents:
                width = (state[:, 1].astype(np.int64) - state[:, 0].astype(np.int64))
                width[width == 0] = 1 << 40
                v = int(np.argmin(width))
                lo, hi = int(state[v, 0]), int(state[v, 1])
                if hi - lo + 1 > SPLIT_WIDTH:
                    mid = (lo + hi) >> 1
                    kids = [(lo, mid), (mid + 1, hi)]
                else:
                    kids = [(x, x) for x in range(lo, hi + 1)]
                # LIFO pool: the child pushed last is explored first -> low values last (high when maximising)
                if self.objective != OBJ_MAX:
                    kids.reverse()
                for a, b in kids:
                    self.st["nodes"] += 1
                    status, out = self._child(state, v, a, b)
                    if status < 0:
                        self.st["cuts"] += 1
                    elif (out[:, 0] == out[:, 1]).all():
                        if self.orc_eval_true(out):
                            self.st["solutions"] += 1
                            self.found.append(out[:, 0].copy())
                            if self.objective == OBJ_MIN:
                                self.st["best"] = min(self.st["best"], int(out[self.obj_var, 0]))
                            if self.objective == OBJ_MAX:
                                self.st["best"] = max(self.st["best"], int(out[self.obj_var, 1]))
                    else:
                        self.pool.append(out)
            self.st["pool_peak"] = max(self.st["pool_peak"], len(self.pool))
            if self.rng is not None and len(self.pool) > 1:
                order = self.rng.permutation(len(self.pool))
                self.pool = [self.pool[i] for i in order]
        self.st["pool"] = len(self.pool)
        self.st["done"] = int(not self.pool or (self.objective == OBJ_ANY and self.st["solutions"] > 0))
        return dict(self.st)

    def orc_eval_true(self, state):
        self.orc.set_domains(state)
        lo, hi = self.orc.eval(self.m.root)
        return lo > 0 or hi < 0
