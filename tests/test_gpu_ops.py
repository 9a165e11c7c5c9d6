"""GPU tests at operator level: the known-answer vectors of the reference's unit tests
(tests/golden/ref_unit_*.json) on the device.

  * eval vectors: each expression becomes the single clause of a model; the device's clause
    value (cs_eval_clauses, i.e. eval_<op>) must equal the reference's expected interval.
  * propagate vectors: the reference pins them with a mocked bind(), which cannot exist on a
    device that really narrows; so each case is wrapped into a clause (push true -> X, push false
    -> NOT(X), push an interval v -> EQ(X, v)) and the device's root fixpoint of that one clause is
    compared with the oracle's (the oracle itself is pinned on the raw vectors by the CPU tests).
    This drives every operator -- EQ LT NEG ADD MUL NOT AND OR WAND -- through the device with the
    reference's operand values, including the +-infinity sentinels and the saturating cases.
"""
import json
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cases(name):
    return json.load(open(golden(f"ref_unit_{name}.json")))["cases"]


def _build(case, wrap=None):
    """oracle-side model builder; every non-value terminal becomes a variable (the device treats
    constants as immutable), value terminals become constants"""
    from oracle.cs_oracle import Model as OModel
    m = OModel.empty()
    node = {}
    for name, t in case["terms"].items():
        lo, hi = t["dom"]
        node[name] = m.var_node(m.add_var(name, lo, hi)) if lo != hi else m.add_const(lo, hi)
    pending = dict((k, v) for k, v in case["exprs"].items() if v is not None)
    while pending:
        progressed = False
        for name, e in list(pending.items()):
            kids = e[1] if e[0] == "WAND" else [k for k in e[1:] if k is not None]
            if all(k in node for k in kids):
                node[name] = m.add_wand([node[k] for k in kids]) if e[0] == "WAND" else \
                    m.add_node(e[0], node[e[1]], node[e[2]] if e[2] is not None else -1)
                del pending[name]
                progressed = True
        if not progressed:
            break
    top = node[case["target"]]
    if wrap is not None:
        lo, hi = wrap
        if (lo, hi) == (0, 0):
            top = m.add_node("NOT", top)
        elif not (lo > 0 or hi < 0) or lo != hi:
            top = m.add_node("EQ", top, m.add_const(lo, hi))
        elif (lo, hi) != (1, 1):
            top = m.add_node("EQ", top, m.add_const(lo, hi))
    m.set_root(m.add_wand([top]))
    return m


def test_eval_vectors_on_device(tmp_path):
    from csolve_amd.solver import Model
    cases = _cases("eval")
    for i, c in enumerate(cases):
        om = _build(c)
        om.index()
        path = str(tmp_path / f"e{i}.model")
        om.save(path)
        vals = Model.from_dump(path).eval_clauses_host()
        expect = c["expect"]
        if c["fn"] == "eval_wand":
            continue  # a wide-and root is split into its elements as clauses; covered by eval_root tests
        assert vals[0].tolist() == expect, (c["test"], c["fn"], vals[0].tolist(), expect)


def test_propagate_vectors_wrapped_vs_oracle(tmp_path):
    from csolve_amd.solver import Model
    from oracle.cs_oracle import Oracle
    cases = [c for c in _cases("propagate") if c["fn"] not in ("propagate", "propagate_wand", "propagate_term")]
    assert len(cases) > 100
    checked = 0
    for i, c in enumerate(cases):
        om = _build(c, wrap=tuple(c["val"]))
        if om.n_vars == 0:
            continue
        om.index()
        orc = Oracle(om)
        orc.set_root_phase(True)
        want = orc.propagate(om.root, 1 << 20)
        path = str(tmp_path / f"p{i}.model")
        om.save(path)
        gm = Model.from_dump(path)
        got = gm.root_propagate()
        assert (got < 0) == (want < 0), (c["test"], c["fn"], c["val"], got, want)
        if want >= 0:
            assert (gm.domains() == orc.domains()).all(), (c["test"], c["fn"], c["val"])
        checked += 1
    assert checked > 60
