"""ctypes loader of libcsolve_hip.so (the product's only compute path)."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSOLVE_HIP_LIB") or os.path.join(_HERE, "libcsolve_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "csolve_gpu.h")


class CsolveError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"csolve_gpu error {code}: {message}")
        self.code = code


class Val(C.Structure):
    _fields_ = [("lo", C.c_int32), ("hi", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("var", C.c_int32), ("lo", C.c_int32), ("hi", C.c_int32), ("parent", C.c_int32)]


class SearchStats(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("cuts", C.c_uint64), ("props", C.c_uint64), ("revisions", C.c_uint64),
                ("solutions", C.c_uint64), ("iterations", C.c_uint64), ("restarts", C.c_uint64), ("pool", C.c_int64), ("pool_peak", C.c_int64),
                ("best", C.c_int32), ("done", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("props", C.c_int32), ("revisions", C.c_int32), ("rounds", C.c_int32)]


def declared_symbols(header: str = HEADER_PATH):
    """Names of every function declared in include/csolve_gpu.h."""
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(csgpu_\w+)\s*\(", text)))


_lib = None


def load_library():
    """Load libcsolve_hip.so or raise -- there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C csolve_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch carries its own libamdhip64.so.7; load it first so the process has ONE HIP runtime
    # and device pointers / streams handed over from torch belong to the runtime we launch on.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.csgpu_last_error.restype = C.c_char_p
    L.csgpu_device_count.restype = C.c_int
    L.csgpu_set_device.argtypes = [C.c_int]
    L.csgpu_model_from_text.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.csgpu_model_from_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.csgpu_model_from_dump.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.csgpu_model_free.argtypes = [vp]
    L.csgpu_model_free.restype = None
    for f in ("csgpu_model_num_vars", "csgpu_model_num_clauses", "csgpu_model_objective",
              "csgpu_model_objective_var"):
        getattr(L, f).argtypes = [vp]
    L.csgpu_model_var_name.argtypes = [vp, C.c_int]
    L.csgpu_model_var_name.restype = C.c_char_p
    L.csgpu_model_get_domains.argtypes = [vp, vp]
    L.csgpu_model_set_domains.argtypes = [vp, vp]
    L.csgpu_model_device_info.argtypes = [vp, C.POINTER(i64)]
    L.csgpu_model_root_propagate.argtypes = [vp, C.POINTER(i32)]
    L.csgpu_model_finalize.argtypes = [vp]
    L.csgpu_model_normalize.argtypes = [vp]
    L.csgpu_model_specialize.argtypes = [vp, vp, C.POINTER(vp)]
    L.csgpu_model_build_tables.argtypes = [vp]
    L.csgpu_model_eval_clauses_host.argtypes = [vp, vp]
    L.csgpu_model_set_kernel.argtypes = [vp, C.c_int]
    L.csgpu_model_qualifies.argtypes = [vp, C.c_int]
    L.csgpu_set_linear_fast_paths.argtypes = [C.c_int]
    L.csgpu_sets_pack.argtypes = [vp, vp, vp, i64, vp]
    L.csgpu_sets_unpack.argtypes = [vp, vp, vp, i64, vp]
    L.csgpu_propagate_batch_sets.argtypes = [vp, vp, vp, vp, vp, i64, vp]
    L.csgpu_set_linear_fast_paths.restype = None
    L.csgpu_model_get_kernel.argtypes = [vp]
    L.csgpu_propagate_batch.argtypes = [vp, vp, vp, vp, vp, i64, vp]
    L.csgpu_propagate_batch_obj.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp]
    L.csgpu_model_forbidden_words.argtypes = [vp]
    L.csgpu_propagate_batch_fb.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.csgpu_eval_batch.argtypes = [vp, vp, vp, i64, vp]
    L.csgpu_eval_clauses.argtypes = [vp, vp, vp, vp]
    L.csgpu_search_create.argtypes = [vp, i64, i64, C.POINTER(vp)]
    L.csgpu_search_free.argtypes = [vp]
    L.csgpu_search_free.restype = None
    L.csgpu_search_reset.argtypes = [vp]
    L.csgpu_search_put.argtypes = [vp, vp, i64]
    L.csgpu_search_put_host.argtypes = [vp, vp, i64]
    L.csgpu_search_take.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.csgpu_search_set_best.argtypes = [vp, i32]
    L.csgpu_model_add_conflict.argtypes = [vp, i32, vp, vp]
    L.csgpu_search_put_cost.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64)]
    L.csgpu_objective_better.argtypes = [C.c_int, Val, i32]
    L.csgpu_objective_bound.argtypes = [C.c_int, Val, i32]
    L.csgpu_objective_bound.restype = Val
    L.csgpu_objective_best.argtypes = [C.c_int, Val, i32]
    L.csgpu_objective_best.restype = i32
    L.csgpu_luby_next.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.csgpu_luby_next.restype = None
    L.csgpu_step_check.argtypes = [Val, C.c_uint32]
    L.csgpu_step_val.argtypes = [Val, C.c_uint32, C.c_uint32]
    L.csgpu_step_val.restype = i32
    L.csgpu_search_share_incumbent.argtypes = [vp, vp]
    L.csgpu_search_set_parents.argtypes = [vp, i64]
    L.csgpu_search_set_restart.argtypes = [vp, i64]
    L.csgpu_search_run.argtypes = [vp, i64, C.POINTER(SearchStats)]
    L.csgpu_search_solutions.argtypes = [vp, vp, i64]
    L.csgpu_search_best_solution.argtypes = [vp, vp]
    L.csgpu_search_solutions.restype = i64
    L.csgpu_propagate_one.argtypes = [vp, vp, Node, vp, C.POINTER(Result)]
    L.csgpu_propagate_one_traced.argtypes = [vp, vp, Node, vp, C.POINTER(Result), vp, i32, C.POINTER(i32)]
    L.csgpu_propagate_one_causes.argtypes = [vp, vp, Node, vp, C.POINTER(Result), vp, i32, C.POINTER(i32)]
    L.csgpu_propagate_values.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.csgpu_model_root_propagate_limit.argtypes = [vp, i64, C.POINTER(i32), C.POINTER(i32)]
    L.csgpu_propagate_one_chain.argtypes = [vp, vp, Node, C.POINTER(i32), C.POINTER(i32), vp, i32, C.POINTER(i32)]
    L.csgpu_search_set_strategy.argtypes = [vp, C.c_int, C.c_int]
    L.csgpu_search_set_restart_on_improvement.argtypes = [vp, C.c_int]
    _lib = L
    return L


def check(rc: int):
    if rc < 0:
        raise CsolveError(rc, load_library().csgpu_last_error().decode())
    return rc
