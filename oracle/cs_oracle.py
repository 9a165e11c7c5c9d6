"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package csolve_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

CSO_ERROR = -1

OPS = {"VAR": 0, "CONST": 1, "EQ": 2, "LT": 3, "NEG": 4, "ADD": 5, "MUL": 6, "NOT": 7,
       "AND": 8, "OR": 9, "WAND": 10, "CONFL": 11}


class Val(C.Structure):
    _fields_ = [("lo", C.c_int32), ("hi", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("prefer_failing", C.c_int), ("restart_frequency", C.c_uint64), ("order", C.c_int),
                ("max_calls", C.c_uint64), ("max_solutions", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("cuts", C.c_uint64), ("props", C.c_uint64),
                ("restarts", C.c_uint64), ("solutions", C.c_uint64), ("best", C.c_int32),
                ("stopped_early", C.c_int), ("solution_values", C.POINTER(C.c_int32)),
                ("solutions_stored", C.c_uint64)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (a few seconds)."""
    # make knows the dependencies (the host sources under csolve_amd/csrc are part of the library)
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []) + ["liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
        for name in ("cso_neg",):
            getattr(L, name).restype = i32
            getattr(L, name).argtypes = [i32]
        for name in ("cso_add", "cso_mul", "cso_min", "cso_max"):
            getattr(L, name).restype = i32
            getattr(L, name).argtypes = [i32, i32]
        L.cs_model_new.restype = vp
        L.cs_model_free.argtypes = [vp]
        L.cs_model_parse.restype = vp
        L.cs_model_parse.argtypes = [C.c_char_p, C.c_int, C.c_char_p, sz]
        L.cs_model_load.restype = vp
        L.cs_model_load.argtypes = [C.c_char_p, C.c_char_p, sz]
        L.cs_model_save.argtypes = [vp, C.c_char_p]
        L.cs_model_index.argtypes = [vp]
        L.cs_model_normalize.restype = i32
        L.cs_model_normalize.argtypes = [vp]
        L.cs_model_add_var.restype = i32
        L.cs_model_add_var.argtypes = [vp, C.c_char_p, Val]
        L.cs_model_add_node.restype = i32
        L.cs_model_add_node.argtypes = [vp, i32, i32, i32]
        L.cs_model_add_wand.restype = i32
        L.cs_model_add_wand.argtypes = [vp, C.POINTER(i32), i32]
        L.cs_model_add_confl.restype = i32
        L.cs_model_add_confl.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), i32]
        L.cs_model_append_clause.argtypes = [vp, i32]
        L.cs_model_equal.argtypes = [vp, vp, C.c_char_p, sz]
        L.cs_model_first_unbounded.restype = i32
        L.cs_model_first_unbounded.argtypes = [vp]
        L.cso_new.restype = vp
        L.cso_new.argtypes = [vp]
        L.cso_free.argtypes = [vp]
        L.cso_set_root_phase.argtypes = [vp, C.c_int]
        L.cso_set_record_only.argtypes = [vp, C.c_int]
        L.cso_domains.restype = C.POINTER(Val)
        L.cso_domains.argtypes = [vp]
        L.cso_props.restype = u64
        L.cso_props.argtypes = [vp]
        L.cso_reset_stats.argtypes = [vp]
        L.cso_eval.restype = Val
        L.cso_eval.argtypes = [vp, i32]
        L.cso_propagate_node.restype = i32
        L.cso_propagate_node.argtypes = [vp, i32, Val, i32]
        L.cso_propagate.restype = i32
        L.cso_propagate.argtypes = [vp, i32, sz]
        L.cso_propagate_clauses.restype = i32
        L.cso_propagate_clauses.argtypes = [vp, i32]
        L.cso_bind_depth.restype = sz
        L.cso_bind_depth.argtypes = [vp]
        L.cso_bind.argtypes = [vp, i32, Val, i32]
        L.cso_unbind.argtypes = [vp, sz]
        L.cso_log_len.restype = sz
        L.cso_log_len.argtypes = [vp]
        L.cso_log_get.argtypes = [vp, sz, C.POINTER(i32), C.POINTER(Val)]
        L.cso_log_clear.argtypes = [vp]
        L.cso_test_bumps.restype = i32
        L.cso_test_bumps.argtypes = [vp, vp, i32]
        L.cso_instance.restype = i64
        L.cso_instance.argtypes = [vp, vp, i32, Val, vp]
        L.cso_instances.restype = u64
        L.cso_instances.argtypes = [vp, vp, vp, i64, vp, vp]
        L.cso_test_set_strategy.argtypes = [vp, C.c_int, C.c_int]
        L.cso_test_set_prio.argtypes = [vp, i32, i64]
        L.cso_test_var_cmp.argtypes = [vp, i32, i32]
        L.cso_test_heap_load.argtypes = [vp, C.POINTER(i32), i32]
        L.cso_test_heap_op.restype = i32
        L.cso_test_heap_op.argtypes = [vp, C.c_int, i32, i32]
        L.cso_test_heap_get.restype = i32
        L.cso_test_heap_get.argtypes = [vp, C.POINTER(i32)]
        L.cso_test_heap_pos.restype = i32
        L.cso_test_heap_pos.argtypes = [vp, i32]
        L.cso_default_options.argtypes = [C.POINTER(Options)]
        L.cso_solve.argtypes = [vp, C.POINTER(Options), C.POINTER(Result)]
        L.cso_result_free.argtypes = [C.POINTER(Result)]
        _lib = L
    return _lib


class _ModelView(C.Structure):
    """Leading fields of struct cs_model (csolve_amd/csrc/cs_model.h)."""
    _fields_ = [("n_vars", C.c_int32), ("cap_vars", C.c_int32), ("dom", C.POINTER(Val)),
                ("names", C.POINTER(C.c_char_p)), ("prio", C.POINTER(C.c_int64)),
                ("var_node", C.POINTER(C.c_int32)),
                ("n_nodes", C.c_int32), ("cap_nodes", C.c_int32), ("nodes", C.POINTER(C.c_int32)),
                ("n_kids", C.c_int32), ("cap_kids", C.c_int32), ("kids", C.POINTER(C.c_int32)),
                ("root", C.c_int32),
                ("n_top", C.c_int32), ("cap_top", C.c_int32), ("top", C.POINTER(C.c_int32)),
                ("objective", C.c_int32), ("obj_var", C.c_int32), ("weights_on", C.c_int32),
                ("n_clauses", C.c_int32), ("clause_node", C.POINTER(C.c_int32)),
                ("list_off", C.POINTER(C.c_int32)), ("list", C.POINTER(C.c_int32))]


class Model:
    """Owning handle of a cs_model built by the shared host code (parser / loader)."""

    def __init__(self, ptr):
        if not ptr:
            raise ValueError("null model")
        self.ptr = ptr
        self.view = C.cast(ptr, C.POINTER(_ModelView)).contents

    @classmethod
    def parse(cls, text: str, weights_on: bool = True) -> "Model":
        err = C.create_string_buffer(256)
        p = lib().cs_model_parse(text.encode(), int(weights_on), err, 256)
        if not p:
            raise ValueError(err.value.decode())
        return cls(p)

    @classmethod
    def load(cls, path: str) -> "Model":
        err = C.create_string_buffer(256)
        p = lib().cs_model_load(path.encode(), err, 256)
        if not p:
            raise ValueError(err.value.decode())
        return cls(p)

    @classmethod
    def empty(cls) -> "Model":
        return cls(lib().cs_model_new())

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.cs_model_free(self.ptr)
        except Exception:
            pass
        self.ptr = None

    # construction helpers for hand-made models (unit-test vectors)
    def add_var(self, name: str, lo: int, hi: int) -> int:
        return lib().cs_model_add_var(self.ptr, name.encode(), Val(lo, hi))

    def var_node(self, var: int) -> int:
        return self.view.var_node[var]

    def add_node(self, op, a: int, b: int = -1) -> int:
        return lib().cs_model_add_node(self.ptr, OPS[op] if isinstance(op, str) else op, a, b)

    def add_const(self, lo: int, hi: int | None = None) -> int:
        return self.add_node("CONST", lo, lo if hi is None else hi)

    def add_wand(self, elems) -> int:
        arr = (C.c_int32 * max(1, len(elems)))(*elems)
        return lib().cs_model_add_wand(self.ptr, arr, len(elems))

    def add_confl(self, elems) -> int:
        """elems: [(terminal node, conflict value)] -- a learnt conflict clause"""
        n = len(elems)
        nodes = (C.c_int32 * max(1, n))(*[e[0] for e in elems])
        vals = (C.c_int32 * max(1, n))(*[e[1] for e in elems])
        return lib().cs_model_add_confl(self.ptr, nodes, vals, n)

    def append_clause(self, node: int):
        """one more top-level clause (and clause-index entry, if the model is indexed)"""
        if lib().cs_model_append_clause(self.ptr, node) != 0:
            raise ValueError("cs_model_append_clause failed")

    def set_root(self, node: int):
        self.view.root = node

    def index(self):
        if lib().cs_model_index(self.ptr) != 0:
            raise ValueError("cs_model_index failed")

    def normalize(self):
        """host normaliser of the product (csolve_amd/csrc/cs_normalize.c), for model-parity tests"""
        if lib().cs_model_normalize(self.ptr) < 0:
            raise ValueError("cs_model_normalize failed")

    def save(self, path: str):
        if lib().cs_model_save(self.ptr, path.encode()) != 0:
            raise OSError(path)

    def equal(self, other: "Model"):
        why = C.create_string_buffer(256)
        ok = lib().cs_model_equal(self.ptr, other.ptr, why, 256)
        return bool(ok), why.value.decode()

    @property
    def n_vars(self) -> int:
        return self.view.n_vars

    @property
    def n_clauses(self) -> int:
        return self.view.n_clauses

    @property
    def root(self) -> int:
        return self.view.root

    def names(self):
        return [self.view.names[i].decode() for i in range(self.n_vars)]

    def domains(self) -> np.ndarray:
        n = self.n_vars
        out = np.zeros((n, 2), dtype=np.int32)
        for i in range(n):
            out[i, 0] = self.view.dom[i].lo
            out[i, 1] = self.view.dom[i].hi
        return out

    def set_domains(self, dom: np.ndarray):
        for i in range(self.n_vars):
            self.view.dom[i].lo = int(dom[i, 0])
            self.view.dom[i].hi = int(dom[i, 1])

    def list_lengths(self):
        lo = self.view.list_off
        return [lo[i + 1] - lo[i] for i in range(self.n_vars)]


class Oracle:
    def __init__(self, model: Model):
        self.model = model
        self.ptr = lib().cso_new(model.ptr)

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.cso_free(self.ptr)
        except Exception:
            pass
        self.ptr = None

    def set_root_phase(self, on: bool):
        lib().cso_set_root_phase(self.ptr, int(on))

    def set_record_only(self, on: bool):
        lib().cso_set_record_only(self.ptr, int(on))

    def domains(self) -> np.ndarray:
        n = self.model.n_vars
        p = lib().cso_domains(self.ptr)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n, 2)).copy()

    def set_domains(self, dom: np.ndarray):
        p = lib().cso_domains(self.ptr)
        for i in range(self.model.n_vars):
            p[i].lo = int(dom[i, 0])
            p[i].hi = int(dom[i, 1])

    def eval(self, node: int):
        v = lib().cso_eval(self.ptr, node)
        return (v.lo, v.hi)

    def propagate_node(self, node: int, lo: int, hi: int, clause: int = -1) -> int:
        return lib().cso_propagate_node(self.ptr, node, Val(lo, hi), clause)

    def propagate(self, node: int, limit: int) -> int:
        return lib().cso_propagate(self.ptr, node, limit)

    def propagate_clauses(self, var: int) -> int:
        return lib().cso_propagate_clauses(self.ptr, var)

    def props(self) -> int:
        return lib().cso_props(self.ptr)

    def bind_log(self):
        out = []
        var, val = C.c_int32(), Val()
        for i in range(lib().cso_log_len(self.ptr)):
            lib().cso_log_get(self.ptr, i, C.byref(var), C.byref(val))
            out.append((var.value, val.lo, val.hi))
        return out

    def clear_log(self):
        lib().cso_log_clear(self.ptr)

    def instance(self, dom_in: np.ndarray, var: int, lo: int, hi: int):
        """-> (status, dom_out); status = -1 or the reference's PROPS for this node."""
        dom_in = np.ascontiguousarray(dom_in, dtype=np.int32)
        out = np.empty_like(dom_in)
        st = lib().cso_instance(self.ptr, dom_in.ctypes.data, var, Val(lo, hi), out.ctypes.data)
        return int(st), out

    def bumps(self, cap: int = 4096) -> np.ndarray:
        """variables whose priority the last instance() bumped, in the reference's order (failing calls only)"""
        out = np.empty(cap, dtype=np.int32)
        k = lib().cso_test_bumps(self.ptr, out.ctypes.data, cap)
        return out[: min(k, cap)].copy()

    def instances(self, dom_in: np.ndarray, var: np.ndarray, val: np.ndarray):
        """Batch of single-value assignments: dom_in [B,n,2], var [B], val [B]."""
        B = dom_in.shape[0]
        dom_in = np.ascontiguousarray(dom_in, dtype=np.int32)
        out = np.empty_like(dom_in)
        status = np.empty(B, dtype=np.int64)
        L = lib()
        stride = dom_in.strides[0]
        for i in range(B):
            status[i] = L.cso_instance(self.ptr, dom_in.ctypes.data + i * stride, int(var[i]),
                                       Val(int(val[i]), int(val[i])), out.ctypes.data + i * stride)
        return status, out

    def instances_nodes(self, states_in: np.ndarray, nodes: np.ndarray, want_states: bool = True):
        """Same batch interface as the device path: states_in [P,n,2], nodes [B,4] rows
        (var, lo, hi, parent_row).  -> (status[B], states_out[B,n,2] or None, total binds)"""
        states_in = np.ascontiguousarray(states_in, dtype=np.int32)
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        B = nodes.shape[0]
        out = np.empty((B,) + states_in.shape[1:], dtype=np.int32) if want_states else None
        status = np.empty(B, dtype=np.int64)
        binds = lib().cso_instances(self.ptr, states_in.ctypes.data, nodes.ctypes.data, B,
                                    out.ctypes.data if want_states else None, status.ctypes.data)
        return status, out, int(binds)

    def solve(self, prefer_failing=True, restart_frequency=100, order=0, max_calls=0, max_solutions=0):
        opt = Options()
        lib().cso_default_options(C.byref(opt))
        opt.prefer_failing = int(prefer_failing)
        opt.restart_frequency = restart_frequency
        opt.order = order
        opt.max_calls = max_calls
        opt.max_solutions = max_solutions
        res = Result()
        if lib().cso_solve(self.ptr, C.byref(opt), C.byref(res)) != 0:
            raise RuntimeError("cso_solve: model is not indexed")
        n = self.model.n_vars
        sols = [[res.solution_values[s * n + v] for v in range(n)] for s in range(res.solutions_stored)]
        out = dict(calls=res.calls, cuts=res.cuts, props=res.props, restarts=res.restarts,
                   solutions=res.solutions, best=res.best, stopped_early=bool(res.stopped_early),
                   solution_values=sols)
        lib().cso_result_free(C.byref(res))
        return out


def read_walk(path: str):
    """Read a walk file written by oracle/_ref/csolve_ref walk.
    -> dict(n_vars, var[B], value[B], status[B], before[B,n,2], after[B,n,2])"""
    if path.endswith(".gz"):
        import gzip
        with gzip.open(path, "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.int32)
    else:
        raw = np.fromfile(path, dtype=np.int32)
    if raw[0] != 0x4B575343 or raw[1] != 1:
        raise ValueError(path + ": not a walk file")
    n, B = int(raw[2]), int(raw[3])
    rec = raw[4:].reshape(B, 3 + 4 * n)
    return dict(n_vars=n, var=rec[:, 0].copy(), value=rec[:, 1].copy(), status=rec[:, 2].copy(),
                before=rec[:, 3:3 + 2 * n].reshape(B, n, 2).copy(),
                after=rec[:, 3 + 2 * n:].reshape(B, n, 2).copy())
