"""CPU tests of the host side: the C ABI library loads and exports what include/csolve_gpu.h
declares, the front end and the clause index replay the reference (checked against the
reference's own post-root dumps), the device tables classify clauses as documented.
No compute call is made here (there is no GPU in this tier)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import golden
from oracle.cs_oracle import OPS, Model as OModel, Oracle, lib as olib

MODELS = ["queens4", "queens8", "queens16", "queens64", "ref_sudoku", "sudoku9_s7", "ref_schedule", "ref_wcet",
          "schedule6_s1"]


def test_library_exports_every_declared_symbol():
    from csolve_amd import _lib
    L = _lib.load_library()
    names = _lib.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_no_device_means_loud_failure():
    """Without a HIP device the compute entry points fail with the library's error; nothing
    falls back to a CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from csolve_amd import CsolveError
    from csolve_amd.solver import Model
    m = Model.from_text(open(golden("problems", "queens8.txt")).read())
    with pytest.raises(CsolveError, match="hip|HIP|device"):
        m.root_propagate()
    d = Model.from_dump(golden("models", "queens8.model"))
    with pytest.raises(CsolveError):
        d.finalize()


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    from csolve_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libcsolve_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load_library()


@pytest.mark.parametrize("text,msg", [
    ("ANY; x = ;", "syntax error"),
    ("x = 1;", "expecting ANY or ALL or MIN or MAX"),
    ("ANY; x = 1 ? 2;", "invalid input `?' in line 1"),
    ("ANY;\n\nx < ;", "in line 3"),
    ("ANY; all_different(a, b;", "expecting ')'"),
])
def test_parse_errors(text, msg):
    from csolve_amd import CsolveError
    from csolve_amd.solver import Model
    with pytest.raises(CsolveError) as e:
        Model.from_text(text)
    assert e.value.code == -2 and msg in str(e.value)


def test_number_and_identifier_tokens():
    """lexer.l:36-102: binary, octal, decimal, hex literals; identifiers with _ @ $; comments."""
    from csolve_amd.solver import Model
    m = Model.from_text("ANY; # comment\n _a@$1 = 0b101; b = 017; c = 0x1F; d = 42; e = 0;\n")
    assert m.var_names() == ["_a@$1", "b", "c", "d", "e"]
    om = OModel.parse("ANY; _a@$1 = 0b101; b = 017; c = 0x1F; d = 42; e = 0;")
    o = Oracle(om)
    o.set_root_phase(True)
    assert o.propagate(om.root, om.n_vars) >= 0
    assert o.domains()[:, 0].tolist() == [5, 15, 31, 42, 0]


@pytest.mark.parametrize("name", MODELS)
def test_front_end_matches_reference_dump(name):
    """Variables (names, order), initial-order weights, root domains: our parser + the oracle's
    root sweeps against what the compiled reference dumped after its own root phase."""
    ref = OModel.load(golden("models", name + ".model"))
    mine = OModel.parse(open(golden("problems", name + ".txt")).read())
    assert mine.names() == ref.names()
    assert [mine.view.prio[i] for i in range(mine.n_vars)] == [ref.view.prio[i] for i in range(ref.n_vars)]
    assert mine.view.objective == ref.view.objective and mine.view.obj_var == ref.view.obj_var
    o = Oracle(mine)
    o.set_root_phase(True)
    assert o.propagate(mine.root, mine.n_vars) >= 0
    assert (o.domains() == ref.domains()).all()


@pytest.mark.parametrize("name", MODELS)
def test_clause_index_replays_clauses_init(name):
    """cs_model_index() on the reference's dumped trees reproduces the reference's own
    per-variable clause lists (parser_support.c:338-396), entry for entry."""
    ref = OModel.load(golden("models", name + ".model"))
    n = ref.n_vars
    want_off = [ref.view.list_off[i] for i in range(n + 1)]
    want = [ref.view.list[i] for i in range(want_off[n])]
    want_nodes = [ref.view.clause_node[i] for i in range(ref.n_clauses)]
    ref.index()
    assert [ref.view.list_off[i] for i in range(n + 1)] == want_off
    assert [ref.view.list[i] for i in range(want_off[n])] == want
    assert [ref.view.clause_node[i] for i in range(ref.n_clauses)] == want_nodes


def test_model_file_round_trip(tmp_path):
    ref = OModel.load(golden("models", "ref_wcet.model"))
    p = str(tmp_path / "x.model")
    ref.save(p)
    again = OModel.load(p)
    ok, why = ref.equal(again)
    assert ok, why


def test_device_tables_classification():
    """Binary != clauses take the 16-byte fast path; everything else is a tree; constant-true
    elements are skipped (queens-64 = 6,048 clauses, list 189, CSR 12,096)."""
    from csolve_amd.solver import Model
    q = Model.from_dump(golden("models", "queens64.model")).build_tables().device_info()
    assert q["ne_clauses"] == 6048 and q["tree_clauses"] == 0 and q["adjacency_entries"] == 12096
    assert q["max_list"] == 189 and q["lds_bytes_per_node"] == 64 * 8 + 16
    s = Model.from_dump(golden("models", "ref_sudoku.model")).build_tables().device_info()
    assert s["ne_clauses"] + s["skipped_clauses"] == 1158 and s["adjacency_entries"] == 1256 and s["max_list"] == 24
    # wcet: 6 of the 15 clauses are `x = y` / `not (x < y)` over two variables (linear fast paths), the sums stay trees
    w = Model.from_dump(golden("models", "ref_wcet.model")).build_tables().device_info()
    assert w["ne_clauses"] == 0 and w["tree_clauses"] == 9 and w["max_tree"] == 53
    # schedule: definitions, precedences and disjunctive pairs all take the linear fast paths
    sch = Model.from_dump(golden("models", "schedule6_s1.model")).build_tables().device_info()
    assert sch["tree_clauses"] == 0 and sch["adjacency_entries"] > 0


def test_generators_are_deterministic_and_reference_shaped():
    from csolve_amd import problems
    assert problems.queens(8) == open(golden("problems", "queens8.txt")).read()
    assert problems.sudoku(3, 0.4, 7) == open(golden("problems", "sudoku9_s7.txt")).read()
    assert problems.schedule(6, 1) == open(golden("problems", "schedule6_s1.txt")).read()
    grid = problems.sudoku_solution(5, 1)
    for r in range(25):
        assert sorted(grid[r]) == list(range(1, 26))
        assert sorted(grid[i][r] for i in range(25)) == list(range(1, 26))
    for br in range(5):
        for bc in range(5):
            assert sorted(grid[br * 5 + i][bc * 5 + j] for i in range(5) for j in range(5)) == list(range(1, 26))


@pytest.mark.parametrize("name", MODELS)
def test_front_end_plus_normaliser_reproduce_the_reference_trees(name):
    """Input action replayed on the host side (root sweeps by the oracle here, by the device in the
    product): parse, propagate, normalize, propagate, clauses_init.  The resulting model -- every
    tree node for node, every per-variable clause list entry for entry, domains, weights -- equals
    what the compiled reference dumped after its own root phase."""
    ref = OModel.load(golden("models", name + ".model"))
    mine = OModel.parse(open(golden("problems", name + ".txt")).read())
    o = Oracle(mine)
    o.set_root_phase(True)
    assert o.propagate(mine.root, mine.n_vars) >= 0
    mine.set_domains(o.domains())
    mine.normalize()
    o = Oracle(mine)
    o.set_root_phase(True)
    assert o.propagate(mine.root, mine.n_vars) >= 0
    mine.set_domains(o.domains())
    mine.index()
    ok, why = mine.equal(ref)
    assert ok, why


# ---- the reference's own unit vectors for the "next" rows: normaliser ------------------------------

def _normalize_cases():
    return json.load(open(golden("ref_unit_normalize.json")))["cases"]


class _Tree:
    """hand-made model for one vector of tests/golden/ref_unit_normalize.json"""

    def __init__(self, terms):
        self.m = OModel.empty()
        self.term = {}
        for name, (lo, hi) in terms.items():
            self.term[name] = self.m.add_const(lo) if lo == hi else self.m.var_node(self.m.add_var(name, lo, hi))

    def build(self, e):
        if isinstance(e, str):
            return self.term[e]
        if e[0] == "CONST":
            return self.m.add_const(e[1])
        kids = [self.build(k) for k in e[1:]]
        return self.m.add_node(e[0], kids[0], kids[1] if len(kids) > 1 else -1)

    def node(self, i):
        v = self.m.view
        return v.nodes[3 * i], v.nodes[3 * i + 1], v.nodes[3 * i + 2]

    def matches(self, i, e):
        """structure of node i == expected expression e (terminals by identity, created constants by value)"""
        op, a, b = self.node(i)
        if isinstance(e, str):
            t = self.term[e]
            return i == t or (op == OPS["CONST"] and self.node(t)[0] == OPS["CONST"] and self.node(t)[1:] == (a, b))
        if e[0] == "CONST":
            return op == OPS["CONST"] and (a, b) == (e[1], e[1])
        if op != OPS[e[0]]:
            return False
        return self.matches(a, e[1]) and (len(e) < 3 or self.matches(b, e[2]))

    def show(self, i):
        op, a, b = self.node(i)
        name = [k for k, v in OPS.items() if v == op][0]
        if name == "VAR":
            return self.m.names()[a]
        if name == "CONST":
            return str(a)
        return "(" + name + " " + self.show(a) + ("" if b < 0 else " " + self.show(b)) + ")"


@pytest.mark.parametrize("case", _normalize_cases(), ids=lambda c: c["ref"].split(" ", 1)[1].replace(" ", "_"))
def test_normaliser_against_the_reference_unit_vectors(case):
    """csolve_amd/csrc/cs_normalize.c on the expressions of the reference's test/test_normalize.c: the result has
    the structure the gtest case expects, and where the gtest expects its argument back no node is created."""
    t = _Tree(case["terms"])
    elems = [t.build(e) for e in (case["wand"] if "wand" in case else [case["in"]])]
    root = t.m.add_wand(elems)
    t.m.set_root(root)
    nodes_before = t.m.view.n_nodes
    t.m.normalize()
    assert t.m.view.root == root, "the wide-and node itself is kept (normalize.c:282-295)"
    op, off, count = t.node(root)
    assert count == len(elems)
    got = [t.m.view.kids[off + i] for i in range(count)]
    want = case["wand_out"] if "wand" in case else [case["out"]]
    for g, w, before in zip(got, want, elems):
        assert t.matches(g, w), f"{case['ref']}: got {t.show(g)}, want {w}"
        if case.get("same") and "wand" not in case:
            assert g == before
    if case.get("same") and "wand" not in case:
        assert t.m.view.n_nodes == nodes_before, "the reference returns its argument: no allocation"
    if "wand" in case:
        created = sum(1 for g, b in zip(got, elems) if g != b)
        assert t.m.view.n_nodes == nodes_before + created


# ---- the reference's unit vectors of the search driver's helpers (objective bound, Luby, value order) ------------

_OBJ = {"ANY": 0, "ALL": 1, "MIN": 2, "MAX": 3}


def test_objective_helpers_against_the_reference_unit_vectors():
    """csgpu_objective_better / _bound / _best -- the functions the kernels and the engine apply the incumbent with
    (cs_arith.h) -- on the vectors of the reference's test/test_objective.c:79-311"""
    from csolve_amd import _lib
    L = _lib.load_library()
    cases = json.load(open(golden("ref_unit_objective.json")))["cases"]
    seen = set()
    for c in cases:
        v = _lib.Val(*(c["val"] or [0, 0])) if "val" in c else None
        if c["fn"] == "better":
            assert bool(L.csgpu_objective_better(_OBJ[c["objective"]], v, c["best"])) == c["expect"], c
        elif c["fn"] == "update_best":
            assert L.csgpu_objective_best(_OBJ[c["objective"]], v, c["best"]) == c["expect_best"], c
        elif c["fn"] == "update_val":
            out = L.csgpu_objective_bound(_OBJ[c["objective"]], v, c["best"])
            assert [out.lo, out.hi] == c["expect_val"], c
        else:
            assert c["fn"] == "best" and c["best"] == c["expect_best"]  # objective_best() is a plain read
        seen.add(c["fn"])
    assert seen == {"better", "update_best", "update_val", "best"} and len(cases) >= 30


def test_luby_and_value_order_against_the_reference_unit_vectors():
    """csgpu_luby_next (the engine's restart schedule) on FailThresholdNext.Basic, csgpu_step_check / _step_val (the
    drop-in's sibling order) on Step.Check / Step.Val of the reference's test/test_csolve.c"""
    from csolve_amd import _lib
    L = _lib.load_library()
    d = json.load(open(golden("ref_unit_search.json")))
    thr, cnt = C.c_uint64(d["luby"]["threshold"]), C.c_uint64(d["luby"]["counter"])
    got = [thr.value]
    for _ in d["luby"]["thresholds"][1:]:
        L.csgpu_luby_next(C.byref(thr), C.byref(cnt))
        got.append(thr.value)
    assert got == d["luby"]["thresholds"] == [1, 1, 2, 1, 1, 2, 4, 1, 1, 2, 1, 1, 2, 4, 8]
    b = _lib.Val(*d["step_check"]["bounds"])
    for row in d["step_check"]["rows"]:
        assert bool(L.csgpu_step_check(b, row["iter"])) == row["expect"], row
    b = _lib.Val(*d["step_val"]["bounds"])
    for seed in (0, 1, 12345):
        vals = [L.csgpu_step_val(b, row["iter"], seed) for row in d["step_val"]["rows"]]
        assert all(b.lo <= v <= b.hi for v in vals) and vals[0] != vals[1]
    # csolve.c:331-338 in full: from the edges inwards, every value of the interval exactly once
    walk = [L.csgpu_step_val(b, i, 0) for i in range(b.hi - b.lo + 1)]
    assert walk[:4] == [3, 17, 4, 16] and sorted(walk) == list(range(3, 18))
    assert [L.csgpu_step_val(b, i, 1) for i in range(4)] == [17, 3, 16, 4]


# ---- the grammar, pinned independently of the front end's self-consistency --------------------------------------

def _sexpr(m, i):
    v = m.view
    op, a, b = v.nodes[3 * i], v.nodes[3 * i + 1], v.nodes[3 * i + 2]
    name = [k for k, x in OPS.items() if x == op][0]
    if name == "VAR":
        return m.names()[a]
    if name == "CONST":
        return str(a) if a == b else f"[{a},{b}]"
    if name == "WAND":
        return "(WAND " + " ".join(_sexpr(m, v.kids[a + k]) for k in range(b)) + ")"
    return "(" + name + " " + _sexpr(m, a) + ("" if b < 0 else " " + _sexpr(m, b)) + ")"


@pytest.mark.parametrize("case", json.load(open(golden("grammar_expectations.json")))["cases"],
                         ids=lambda c: c["ref"].split(" ", 1)[0])
def test_front_end_builds_the_trees_the_reference_grammar_prescribes(case):
    """cs_frontend.c against expectations derived by hand from the ACTIONS of the reference's grammar (src/parser.y, read
    as text): operator precedence and associativity, the desugaring of > <= >= != and binary minus, all_different, the
    objective element and the <obj> variable, one terminal per identifier.  (The compiled reference in oracle/_ref is
    fed by this same front end, so its dumps cannot pin the grammar; these can.)"""
    m = OModel.parse(case["text"], weights_on=False)
    v = m.view
    op, off, count = v.nodes[3 * v.root], v.nodes[3 * v.root + 1], v.nodes[3 * v.root + 2]
    assert op == OPS["WAND"]
    got = [_sexpr(m, v.kids[off + k]) for k in range(count)]
    assert got == case["elems"], (case["ref"], got)
    if "objective" in case:
        assert v.objective == {"ANY": 0, "ALL": 1, "MIN": 2, "MAX": 3}[case["objective"]]
        assert (v.obj_var >= 0) == (case["objective"] in ("MIN", "MAX"))
    if "vars" in case:
        assert m.names() == case["vars"]
    for name, dom in case.get("domains", {}).items():
        assert m.domains()[m.names().index(name)].tolist() == dom


def test_front_end_weights_as_the_reference_grammar_prescribes():
    """the initial ordering weights (env_t.prio) of parser.y's relational / equality actions, derived by hand"""
    w = json.load(open(golden("grammar_expectations.json")))["weights"]
    for case in w["cases"]:
        m = OModel.parse(case["text"], weights_on=True)
        got = dict(zip(m.names(), [m.view.prio[i] for i in range(m.n_vars)]))
        assert got == case["prio"], (case["text"], got)


def test_the_engines_branching_key_orders_variables_like_strategy_var_cmp():
    """csgpu_branch_key (cs_arith.h: what cs_branch and the burst kernels minimise over the open variables) against the
    70 comparisons of test/test_strategy.c VarCmp.*: sign(strategy_var_cmp(a, b)) = sign(key(b) - key(a)) with the
    index bits masked -- the variable the reference's heap would pop is the one with the smallest key"""
    import ctypes as C
    import json
    from csolve_amd import _lib
    L = _lib.load_library()
    L.csgpu_branch_key.argtypes = [C.c_int, C.c_int, _lib.Val, C.c_int64, C.c_int32]
    L.csgpu_branch_key.restype = C.c_uint64
    kinds = {"none": 0, "smallest-domain": 1, "largest-domain": 2, "smallest-value": 3, "largest-value": 4}
    from conftest import golden
    cases = json.load(open(golden("ref_unit_strategy.json")))["cmp"]
    assert len(cases) == 70
    for c in cases:
        ka = L.csgpu_branch_key(kinds[c["order"]], int(c["prefer_failing"]), _lib.Val(*c["a"]["val"]), c["a"]["prio"], 0) >> 16
        kb = L.csgpu_branch_key(kinds[c["order"]], int(c["prefer_failing"]), _lib.Val(*c["b"]["val"]), c["b"]["prio"], 0) >> 16
        sign = (kb > ka) - (kb < ka)
        assert sign == c["sign"], (c, ka, kb)
    # ties go to the lower index
    assert L.csgpu_branch_key(1, 0, _lib.Val(1, 5), 0, 3) < L.csgpu_branch_key(1, 0, _lib.Val(1, 5), 0, 4)


def test_a_specialised_model_drops_what_its_prefix_decides():
    """SURVEY 8f-1 on the host: csgpu_model_specialize copies the model with a subtree's prefix state as root domains;
    the normaliser then folds the valued variables into their clauses (normalize.c:67-75) and the table builder leaves
    constant clauses out of the lists.  (That the search below the prefix gives the same results with either model is
    the GPU suite's test_a_model_specialised_for_a_subtree_gives_the_subtrees_results.)"""
    from csolve_amd._lib import CsolveError
    from csolve_amd.solver import Model
    m = Model.from_dump(golden("models", "schedule6_s1.model"))
    base = m.build_tables().device_info()
    dom = m.domains()
    names = m.var_names()
    prefix = dom.copy()
    starts = [v for v, name in enumerate(names) if name.endswith("_start")]
    for v in starts[:3]:  # three start times decided: every disjunction between two of them is decided with them
        prefix[v, 1] = prefix[v, 0]
    s = m.specialize(prefix, build_only=True)
    info = s.device_info()
    assert s.n_vars == m.n_vars and info["adjacency_entries"] < base["adjacency_entries"]
    assert (s.domains() == prefix).all() and (m.domains() == dom).all()  # the copy is independent of the model
    # a state outside the model's root domains is refused
    bad = dom.copy()
    bad[starts[0], 1] += 1
    with pytest.raises(CsolveError):
        m.specialize(bad, build_only=True)
