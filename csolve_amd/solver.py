"""Thin Python view of the C ABI: problem models and batched propagation on torch
device buffers.  Mirrors the reference's operator vocabulary: a *model* (constraints +
variables), *states* (one interval per variable), *nodes* (an assignment applied to a
parent state), `propagate` (propagate_clauses for every node of a batch), `eval_root`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import Node, Result, SearchStats, Val, check, load_library

STATUS_FAIL = -1


def _stream_ptr(stream) -> int:
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


class Model:
    """A problem: host model + (after finalize) its device image."""

    def __init__(self, handle):
        self._h = handle
        self.finalized = False

    # ---- construction -----------------------------------------------------------------
    @classmethod
    def from_text(cls, text: str, weights_on: bool = True) -> "Model":
        L = load_library()
        h = C.c_void_p()
        check(L.csgpu_model_from_text(text.encode(), int(weights_on), C.byref(h)))
        return cls(h)

    @classmethod
    def from_file(cls, path: str, weights_on: bool = True) -> "Model":
        L = load_library()
        h = C.c_void_p()
        check(L.csgpu_model_from_file(path.encode(), int(weights_on), C.byref(h)))
        return cls(h)

    @classmethod
    def from_dump(cls, path: str) -> "Model":
        L = load_library()
        h = C.c_void_p()
        check(L.csgpu_model_from_dump(path.encode(), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            load_library().csgpu_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-side queries --------------------------------------------------------------
    @property
    def n_vars(self) -> int:
        return check(load_library().csgpu_model_num_vars(self._h))

    @property
    def n_clauses(self) -> int:
        return check(load_library().csgpu_model_num_clauses(self._h))

    @property
    def objective(self) -> int:
        return check(load_library().csgpu_model_objective(self._h))

    @property
    def objective_var(self) -> int:
        return load_library().csgpu_model_objective_var(self._h)

    def var_names(self):
        L = load_library()
        return [L.csgpu_model_var_name(self._h, i).decode() for i in range(self.n_vars)]

    def domains(self) -> np.ndarray:
        out = np.empty((self.n_vars, 2), dtype=np.int32)
        check(load_library().csgpu_model_get_domains(self._h, out.ctypes.data))
        return out

    def set_domains(self, dom: np.ndarray):
        dom = np.ascontiguousarray(dom, dtype=np.int32)
        assert dom.shape == (self.n_vars, 2)
        check(load_library().csgpu_model_set_domains(self._h, dom.ctypes.data))

    def device_info(self) -> dict:
        info = (C.c_int64 * 8)()
        check(load_library().csgpu_model_device_info(self._h, info))
        keys = ("adjacency_entries", "ne_clauses", "tree_clauses", "tree_nodes", "lds_bytes_per_node",
                "max_list", "skipped_clauses", "max_tree")
        return dict(zip(keys, [int(x) for x in info]))

    # ---- device phases ------------------------------------------------------------------
    def root_propagate(self) -> int:
        """propagate(root, size) on the device; -1 = infeasible, else narrowings."""
        st = C.c_int32()
        check(load_library().csgpu_model_root_propagate(self._h, C.byref(st)))
        return st.value

    def eval_clauses_host(self) -> np.ndarray:
        """eval_<op> of every clause on the current root domains (no finalize needed)."""
        out = np.empty((max(1, self.n_clauses), 2), dtype=np.int32)
        check(load_library().csgpu_model_eval_clauses_host(self._h, out.ctypes.data))
        return out[: self.n_clauses]

    def build_tables(self):
        """Host-only: clause lists + device tables in host memory (no HIP call)."""
        check(load_library().csgpu_model_build_tables(self._h))
        return self

    def normalize(self):
        """normalize(root): host-side rewrite of the trees between the two root propagations"""
        check(load_library().csgpu_model_normalize(self._h))
        return self

    def specialize(self, state, build_only: bool = False) -> "Model":
        """the model rewritten for the subtree below `state` ([n_vars][2] intervals inside the root
        domains): a copy with `state` as root domains, normalised (what the prefix has decided is folded away,
        normalize.c:67-316), propagated and finalized (entailed clauses leave the device tables).  Same results as
        this model for every state inside `state`; shorter clause lists.  build_only: host tables only (no GPU)."""
        state = np.ascontiguousarray(np.asarray(state.cpu() if hasattr(state, "cpu") else state), dtype=np.int32)
        state = state.reshape(-1, self.n_vars, 2)
        assert state.shape[0] == 1, "one state (a common prefix), not a batch"
        state = np.ascontiguousarray(state[0])
        h = C.c_void_p()
        check(load_library().csgpu_model_specialize(self._h, state.ctypes.data, C.byref(h)))
        m = Model(h)
        m.normalize()
        if build_only:
            return m.build_tables()
        if m.root_propagate() < 0:
            raise ValueError("INFEASIBLE PROBLEM")
        return m.finalize()

    def add_conflict(self, elems):
        """a learnt conflict clause "not all of var == value" for elems = [(var, value)] (csgpu_model_add_conflict)"""
        vs = np.ascontiguousarray([e[0] for e in elems], dtype=np.int32)
        cs = np.ascontiguousarray([e[1] for e in elems], dtype=np.int32)
        check(load_library().csgpu_model_add_conflict(self._h, len(elems), vs.ctypes.data, cs.ctypes.data))
        return self

    def finalize(self):
        check(load_library().csgpu_model_finalize(self._h))
        self.finalized = True
        return self

    def set_kernel(self, which: int):
        """0 automatic, 1 general kernel, 2 LDS-resident unit shaving, 3 forbidden sets in LDS,
        4 forbidden sets in registers, 5 the same with two or four nodes per wave, 6 clause-resident (small
        models), 7 interval-only shaving (2-5, 7: pure binary-NE models that fit, see csolve_gpu.h)"""
        check(load_library().csgpu_model_set_kernel(self._h, which))
        return self

    def qualifies(self, which: int) -> bool:
        return bool(load_library().csgpu_model_qualifies(self._h, which))

    def kernel(self) -> int:
        return check(load_library().csgpu_model_get_kernel(self._h))

    def root_state(self, device="cuda") -> torch.Tensor:
        """[1, n_vars, 2] int32 tensor of the root domains."""
        return torch.from_numpy(self.domains()).to(device).unsqueeze(0).contiguous()

    def propagate(self, states_in: torch.Tensor, nodes: torch.Tensor, states_out: torch.Tensor = None,
                  results: torch.Tensor = None, stream=None):
        """Batched propagate_clauses.

        states_in  [P, n_vars, 2] int32 (device)   parent states
        nodes      [B, 4] int32 (device)           rows (var, lo, hi, parent_row)
        returns (states_out [B, n_vars, 2], results [B, 4] = status, props, revisions, rounds)
        """
        n = self.n_vars
        assert states_in.is_cuda and nodes.is_cuda, "device tensors required"
        assert states_in.dtype == torch.int32 and nodes.dtype == torch.int32
        assert states_in.is_contiguous() and nodes.is_contiguous()
        assert states_in.shape[-2:] == (n, 2) and nodes.shape[-1] == 4
        B = nodes.shape[0]
        if states_out is None:
            states_out = torch.empty((B, n, 2), dtype=torch.int32, device=nodes.device)
        if results is None:
            results = torch.empty((B, 4), dtype=torch.int32, device=nodes.device)
        assert states_out.is_contiguous() and results.is_contiguous()
        assert states_out.shape == (B, n, 2) and results.shape == (B, 4)
        check(load_library().csgpu_propagate_batch(self._h, states_in.data_ptr(), nodes.data_ptr(),
                                                   states_out.data_ptr(), results.data_ptr(), B,
                                                   _stream_ptr(stream)))
        return states_out, results

    def forbidden_words(self) -> int:
        """64-bit words of forbidden-set per variable (0: the model does not qualify)"""
        return load_library().csgpu_model_forbidden_words(self._h)

    def propagate_fb(self, states_in, nodes, forb_in=None, states_out=None, forb_out=None, results=None,
                     want_forb=True, stream=None):
        """Batched propagate_clauses with the forbidden sets carried next to the states.
        forb_in/forb_out: int64 tensors [rows, n_vars, FW] (None = rebuild / not wanted).
        returns (states_out, forb_out, results)"""
        n, fw = self.n_vars, self.forbidden_words()
        assert fw > 0, "model does not qualify for the forbidden-set kernel"
        assert states_in.is_cuda and states_in.dtype == torch.int32 and states_in.is_contiguous()
        assert nodes.is_cuda and nodes.dtype == torch.int32 and nodes.is_contiguous()
        B = nodes.shape[0]
        if states_out is None:
            states_out = torch.empty((B, n, 2), dtype=torch.int32, device=nodes.device)
        if results is None:
            results = torch.empty((B, 4), dtype=torch.int32, device=nodes.device)
        if forb_out is None and want_forb:
            forb_out = torch.empty((B, n, fw), dtype=torch.int64, device=nodes.device)
        if forb_in is not None:
            assert forb_in.dtype == torch.int64 and forb_in.is_contiguous() and forb_in.shape[-2:] == (n, fw)
            assert forb_in.shape[0] == states_in.shape[0]
        check(load_library().csgpu_propagate_batch_fb(
            self._h, states_in.data_ptr(), 0 if forb_in is None else forb_in.data_ptr(), nodes.data_ptr(),
            states_out.data_ptr(), 0 if forb_out is None else forb_out.data_ptr(), results.data_ptr(), B,
            _stream_ptr(stream)))
        return states_out, forb_out, results

    # ---- sets-only states (csgpu_sets_*) ---------------------------------------------------------
    def pack_sets(self, states: torch.Tensor, stream=None) -> torch.Tensor:
        """interval states [k, n_vars, 2] -> sets-only states [k, n_vars, FW] (int64)"""
        n, fw = self.n_vars, self.forbidden_words()
        assert states.is_cuda and states.dtype == torch.int32 and states.is_contiguous() and states.shape[-2:] == (n, 2)
        sets = torch.empty((states.shape[0], n, fw), dtype=torch.int64, device=states.device)
        check(load_library().csgpu_sets_pack(self._h, states.data_ptr(), sets.data_ptr(), states.shape[0], _stream_ptr(stream)))
        return sets

    def unpack_sets(self, sets: torch.Tensor, stream=None) -> torch.Tensor:
        n, fw = self.n_vars, self.forbidden_words()
        assert sets.is_cuda and sets.dtype == torch.int64 and sets.is_contiguous() and sets.shape[-2:] == (n, fw)
        states = torch.empty((sets.shape[0], n, 2), dtype=torch.int32, device=sets.device)
        check(load_library().csgpu_sets_unpack(self._h, sets.data_ptr(), states.data_ptr(), sets.shape[0], _stream_ptr(stream)))
        return states

    def propagate_sets(self, sets_in: torch.Tensor, nodes: torch.Tensor, sets_out: torch.Tensor = None,
                       results: torch.Tensor = None, stream=None):
        """Batched propagate_clauses on sets-only states -> (sets_out [B, n_vars, FW], results [B, 4])"""
        n, fw = self.n_vars, self.forbidden_words()
        assert sets_in.is_cuda and sets_in.dtype == torch.int64 and sets_in.is_contiguous() and sets_in.shape[-2:] == (n, fw)
        assert nodes.is_cuda and nodes.dtype == torch.int32 and nodes.is_contiguous() and nodes.shape[-1] == 4
        B = nodes.shape[0]
        if sets_out is None:
            sets_out = torch.empty((B, n, fw), dtype=torch.int64, device=nodes.device)
        if results is None:
            results = torch.empty((B, 4), dtype=torch.int32, device=nodes.device)
        check(load_library().csgpu_propagate_batch_sets(self._h, sets_in.data_ptr(), nodes.data_ptr(), sets_out.data_ptr(),
                                                        results.data_ptr(), B, _stream_ptr(stream)))
        return sets_out, results

    def eval_root(self, states: torch.Tensor, stream=None) -> torch.Tensor:
        """Three-valued value of the root wide-and per state: 1 true, 0 false, 2 undecided."""
        assert states.is_cuda and states.dtype == torch.int32 and states.is_contiguous()
        B = states.shape[0]
        truth = torch.empty((B,), dtype=torch.int32, device=states.device)
        check(load_library().csgpu_eval_batch(self._h, states.data_ptr(), truth.data_ptr(), B, _stream_ptr(stream)))
        return truth

    def eval_clauses(self, state: torch.Tensor, stream=None) -> torch.Tensor:
        """Interval value of every clause for ONE state: [n_clauses, 2]."""
        assert state.is_cuda and state.dtype == torch.int32 and state.is_contiguous()
        out = torch.empty((max(1, self.n_clauses), 2), dtype=torch.int32, device=state.device)
        check(load_library().csgpu_eval_clauses(self._h, state.data_ptr(), out.data_ptr(), _stream_ptr(stream)))
        return out[: self.n_clauses]

    def propagate_one(self, state: np.ndarray, var: int, lo: int, hi: int):
        """Single node through host buffers (the drop-in path)."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        out = np.empty_like(state)
        res = Result()
        check(load_library().csgpu_propagate_one(self._h, state.ctypes.data, Node(var, lo, hi, 0),
                                                 out.ctypes.data, C.byref(res)))
        return res.status, res.props, out


    def propagate_one_causes(self, state: np.ndarray, var: int, lo: int, hi: int, cap: int = 4096):
        """One node of a pure != network with the cause of every bound move (csgpu_propagate_one_causes).
        -> (status, props, fixpoint or None, trace [k, 4] = {variable, 0 lo / 1 hi, new bound, causing variable})"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        out = np.empty_like(state)
        res = Result()
        trace = np.empty((cap, 4), dtype=np.int32)
        cnt = C.c_int32()
        check(load_library().csgpu_propagate_one_causes(self._h, state.ctypes.data, Node(var, lo, hi, 0), out.ctypes.data,
                                                        C.byref(res), trace.ctypes.data, cap, C.byref(cnt)))
        if cnt.value > min(cap, 2048):
            raise OverflowError(f"{cnt.value} trace records, {min(cap, 2048)} kept")
        return res.status, res.props, (out if res.status >= 0 else None), trace[: cnt.value].copy()

    def propagate_one_chain(self, state: np.ndarray, var: int, lo: int, hi: int, cap: int = 4096):
        """The reference's own failure chain of one node (csgpu_propagate_one_chain).
        -> (status -1 / 0, props, bumped variables in the reference's order)"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        st, props, cnt = C.c_int32(), C.c_int32(), C.c_int32()
        bumps = np.empty(cap, dtype=np.int32)
        check(load_library().csgpu_propagate_one_chain(self._h, state.ctypes.data, Node(var, lo, hi, 0), C.byref(st),
                                                        C.byref(props), bumps.ctypes.data, cap, C.byref(cnt)))
        return st.value, props.value, bumps[: min(cap, cnt.value)].copy()

    def propagate_one_traced(self, state: np.ndarray, var: int, lo: int, hi: int, cap: int = 4096):
        """One node with its trail (csgpu_propagate_one_traced).
        -> (status, props, fixpoint or None, trace [k, 4] = {variable, 0 lo / 1 hi / 2 failure, new bound, clause})"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        out = np.empty_like(state)
        res = Result()
        trace = np.empty((cap, 4), dtype=np.int32)
        cnt = C.c_int32()
        check(load_library().csgpu_propagate_one_traced(self._h, state.ctypes.data, Node(var, lo, hi, 0), out.ctypes.data,
                                                        C.byref(res), trace.ctypes.data, cap, C.byref(cnt)))
        if cnt.value > cap:
            raise OverflowError(f"{cnt.value} trace records, room for {cap}")
        return res.status, res.props, (out if res.status >= 0 else None), trace[: cnt.value].copy()

    def propagate_values(self, state: np.ndarray, var: int, values):
        """Several values of `var` on one parent state, host buffers (the drop-in's sibling batch).
        -> (results [count, 4], states_out [count, n_vars, 2]); rows of inconsistent nodes are unspecified"""
        state = np.ascontiguousarray(state, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.int32)
        outs = np.empty((len(values),) + state.shape, dtype=np.int32)
        res = np.empty((len(values), 4), dtype=np.int32)
        check(load_library().csgpu_propagate_values(self._h, state.ctypes.data, var, values.ctypes.data, len(values),
                                                    outs.ctypes.data, res.ctypes.data))
        return res, outs

    def root_propagate_limit(self, limit: int):
        """propagate(root, limit): at most limit + 1 sweeps.  -> (status, rounds)"""
        st, rounds = C.c_int32(), C.c_int32()
        check(load_library().csgpu_model_root_propagate_limit(self._h, limit, C.byref(st), C.byref(rounds)))
        return st.value, rounds.value


def set_linear_fast_paths(on: bool):
    """Process-wide: whether models finalized from now on revise EQ / LT / two-literal OR clauses by direct
    bound propagation (default) or through the expression-tree interpreter (csgpu_set_linear_fast_paths)."""
    load_library().csgpu_set_linear_fast_paths(1 if on else 0)


class Search:
    """Device-resident tree search over a finalized model (csgpu_search_*): a LIFO pool of open
    states in HBM, expanded and propagated in batches.  One instance per GPU/rank."""

    def __init__(self, model: Model, pool_capacity: int = 1 << 20, max_children: int = 1 << 16):
        self.model = model
        self._h = C.c_void_p()
        check(load_library().csgpu_search_create(model._h, pool_capacity, max_children, C.byref(self._h)))

    def close(self):
        if self._h:
            load_library().csgpu_search_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """forget pool, statistics, incumbent and seeds; keep the buffers"""
        check(load_library().csgpu_search_reset(self._h))

    def put(self, states: torch.Tensor):
        """append open states [k, n_vars, 2] (device) to the pool"""
        if states.numel() == 0:
            return
        assert states.is_cuda and states.dtype == torch.int32 and states.is_contiguous()
        assert states.shape[-2:] == (self.model.n_vars, 2)
        check(load_library().csgpu_search_put(self._h, states.data_ptr(), states.shape[0]))

    def take(self, max_states: int) -> torch.Tensor:
        """remove up to max_states of the oldest open states (largest subtrees)"""
        buf = torch.empty((max(1, max_states), self.model.n_vars, 2), dtype=torch.int32, device="cuda")
        cnt = C.c_int64()
        check(load_library().csgpu_search_take(self._h, buf.data_ptr(), max_states, C.byref(cnt)))
        return buf[: cnt.value]

    def set_parents(self, parents_per_iteration: int):
        check(load_library().csgpu_search_set_parents(self._h, int(parents_per_iteration)))

    def set_restart(self, iterations: int):
        check(load_library().csgpu_search_set_restart(self._h, int(iterations)))

    ORDERS = {"none": 0, "smallest-domain": 1, "largest-domain": 2, "smallest-value": 3, "largest-value": 4}

    def set_strategy(self, order="smallest-domain", prefer_failing: bool = False):
        """the reference's -o / -f (csgpu_search_set_strategy): set before the first state is put"""
        o = self.ORDERS[order] if isinstance(order, str) else int(order)
        check(load_library().csgpu_search_set_strategy(self._h, o, int(bool(prefer_failing))))

    def set_restart_on_improvement(self, on: bool = True):
        """MIN / MAX: restart from the seeds on every better solution (csolve.c:418-425)"""
        check(load_library().csgpu_search_set_restart_on_improvement(self._h, int(bool(on))))

    def share_incumbent(self, other: "Search"):
        """keep the incumbent in `other`'s word of device memory (MIN / MAX engines of one model on one device).
        While shared, set_parents beyond the device-driven limit is refused, best_solution() answers only on the
        engine whose own row attains the incumbent, and the library keeps the lender's memory until the last
        borrower is freed."""
        check(load_library().csgpu_search_share_incumbent(self._h, other._h))

    def set_best(self, best: int):
        check(load_library().csgpu_search_set_best(self._h, int(best)))

    def put_cost(self):
        """-> (seconds, states): host time put() has taken since reset() (copy + rebuilding the forbidden sets)"""
        sec, cnt = C.c_double(), C.c_int64()
        check(load_library().csgpu_search_put_cost(self._h, C.byref(sec), C.byref(cnt)))
        return sec.value, cnt.value

    def run(self, max_iterations: int = 1 << 62) -> dict:
        st = SearchStats()
        check(load_library().csgpu_search_run(self._h, max_iterations, C.byref(st)))
        return {k: getattr(st, k) for k, _ in SearchStats._fields_}

    def best_solution(self):
        """MIN/MAX: values of a solution attaining the incumbent, or None"""
        out = np.empty(self.model.n_vars, dtype=np.int32)
        rc = check(load_library().csgpu_search_best_solution(self._h, out.ctypes.data))
        return out if rc == 1 else None

    def solutions(self, max_solutions: int = 1024) -> np.ndarray:
        out = np.empty((max(1, max_solutions), self.model.n_vars), dtype=np.int32)
        k = load_library().csgpu_search_solutions(self._h, out.ctypes.data, max_solutions)
        check(k)
        return out[:k]


def solve_root(text: str, weights_on: bool = True) -> Model:
    """Front end + root phase + finalize: parser.y's Input action up to clauses_init
    (propagate, normalize, propagate, env_generate, clauses_init)."""
    m = Model.from_text(text, weights_on)
    if m.root_propagate() < 0:
        raise ValueError("INFEASIBLE PROBLEM")
    m.normalize()
    if m.root_propagate() < 0:
        raise ValueError("INFEASIBLE PROBLEM")
    return m.finalize()
