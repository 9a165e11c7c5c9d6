"""GPU edge cases of the batched fixpoint through the C ABI: tiny and degenerate models, ragged
batches, repeated parents, full re-propagation nodes, negative and wide domains (which change the
kernel that is eligible), always against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _oracle_for(text, model):
    from oracle.cs_oracle import Model as OModel, Oracle
    om = OModel.parse(text)
    om.set_domains(model.domains())
    om.index()
    return Oracle(om)


def _check(text, nodes_fn, kernels=(1, 2, 3, 4, 5, 6, 7), batch_sizes=(1, 3, 16, 17, 100)):
    from csolve_amd.solver import solve_root
    model = solve_root(text)
    orc = _oracle_for(text, model)
    n = model.n_vars
    root = model.domains()
    rng = np.random.default_rng(5)
    eligible = [k for k in (1, 2, 3, 4, 5, 6, 7) if model.qualifies(k)]
    ran = []
    for k in kernels:
        if k not in eligible:
            continue
        model.set_kernel(k)
        ran.append(k)
        for B in batch_sizes:
            nodes = nodes_fn(rng, root, B)
            states = torch.from_numpy(root[None].copy()).cuda()
            out, res = model.propagate(states, torch.from_numpy(nodes).cuda())
            torch.cuda.synchronize()
            out, res = out.cpu().numpy(), res.cpu().numpy()
            for i in range(B):
                v, lo, hi, _ = nodes[i]
                st, exp = orc.instance(root, int(v), int(lo), int(hi))
                assert (st < 0) == (res[i, 0] < 0), (k, B, nodes[i].tolist(), st, res[i].tolist())
                if st >= 0:
                    assert (out[i] == exp).all(), (k, B, nodes[i].tolist())
                    assert res[i, 0] == int((exp[:, 0] != exp[:, 1]).sum())
    return model, ran


def _value_nodes(rng, root, B):
    n = root.shape[0]
    nodes = np.zeros((B, 4), dtype=np.int32)
    for i in range(B):
        v = rng.integers(n)
        val = rng.integers(root[v, 0], root[v, 1] + 1)
        nodes[i] = (v, val, val, 0)
    return nodes


def _interval_and_full_nodes(rng, root, B):
    n = root.shape[0]
    nodes = np.zeros((B, 4), dtype=np.int32)
    for i in range(B):
        if i % 3 == 0:
            nodes[i] = (-1, 0, 0, 0)  # full re-propagation of the parent
        else:
            v = rng.integers(n)
            a = rng.integers(root[v, 0], root[v, 1] + 1)
            b = rng.integers(a, root[v, 1] + 1)
            nodes[i] = (v, a, b, 0)  # interval assignment (worker split, csolve.c:121-150)
    return nodes


def test_tiny_models():
    _check("ANY; x != y; 1 <= x; x <= 2; 1 <= y; y <= 2;", _value_nodes)
    _check("ANY; all_different(a, b, c); 0 <= a; a <= 2; 0 <= b; b <= 2; 0 <= c; c <= 2;", _value_nodes)
    _check("ANY; a + 1 != b; a != b; 5 <= a; a <= 6; 5 <= b; b <= 7;", _value_nodes)


def test_interval_assignments_and_full_nodes():
    from csolve_amd import problems
    _check(problems.queens(9), _interval_and_full_nodes)
    _check(problems.sudoku(3, 0.3, 3), _interval_and_full_nodes)


def test_negative_domains_and_offsets():
    text = "ANY; all_different(a-5, b+3, c, d-1); -7 <= a; a <= -1; -9 <= b; b <= -3; -6 <= c; c <= 0; -4 <= d; d <= 2;"
    model, ran = _check(text, _value_nodes)
    assert 3 in ran and model.forbidden_words() == 1


def test_wide_domains_fall_back_from_the_forbidden_set_kernel():
    """root intervals wider than 256 values: no forbidden-set kernel, the unit-shaving kernels and the
    clause-resident kernel handle them (and give the same results)"""
    text = "ANY; all_different(a, b, c); 0 <= a; a <= 1000; 0 <= b; b <= 1000; 500 <= c; c <= 2000; a + 1 != c;"
    model, ran = _check(text, _value_nodes)
    assert model.forbidden_words() == 0 and ran == [1, 2, 6]


def test_128_and_256_value_domains_use_wider_sets():
    text128 = "ANY; all_different(a, b, c); 1 <= a; a <= 128; 1 <= b; b <= 100; 3 <= c; c <= 128;"
    m, ran = _check(text128, _value_nodes)
    assert m.forbidden_words() == 2 and 3 in ran
    text256 = "ANY; all_different(a, b, c); 1 <= a; a <= 256; 1 <= b; b <= 200; 3 <= c; c <= 130;"
    m, ran = _check(text256, _value_nodes)
    assert m.forbidden_words() == 4 and 3 in ran


def test_mixed_model_with_a_tree_clause_uses_the_general_kernels():
    """a clause that stays an expression tree: only the event-driven and the clause-resident kernel take it"""
    text = "ANY; all_different(a, b, c); a + b < 5; 0 <= a; a <= 4; 0 <= b; b <= 4; 0 <= c; c <= 4;"
    model, ran = _check(text, _value_nodes)
    assert ran == [1, 6] and model.device_info()["tree_clauses"] >= 1


def test_empty_batch_and_repeated_parents():
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    model = solve_root(problems.queens(8))
    states = model.root_state().repeat(3, 1, 1).contiguous()
    empty = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    out, res = model.propagate(states, empty)
    assert out.shape[0] == 0 and res.shape[0] == 0
    nodes = torch.tensor([[0, 1, 1, 2], [0, 1, 1, 0], [0, 1, 1, 2], [7, 8, 8, 1]], dtype=torch.int32, device="cuda")
    out, res = model.propagate(states, nodes)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2]) and torch.equal(res[0], res[1])


def test_sets_only_layout_limits_and_empty_batches():
    """csgpu_sets_*: a model that does not qualify for kernel 4 is refused (CSGPU_E_LIMIT), empty batches are
    accepted, a state without allowed values unpacks to the empty interval {1, 0}."""
    from csolve_amd import problems
    from csolve_amd._lib import CsolveError
    from csolve_amd.solver import solve_root
    big = solve_root(problems.sudoku(5, 0.4, 1))  # 625 variables: kernel 3 only
    assert big.forbidden_words() == 1 and not big.qualifies(4)
    with pytest.raises(CsolveError):
        big.pack_sets(big.root_state())
    model = solve_root(problems.queens(12))
    root = model.root_state()
    sets = model.pack_sets(root)
    empty_nodes = torch.empty((0, 4), dtype=torch.int32, device="cuda")
    out, res = model.propagate_sets(sets, empty_nodes)
    assert out.shape[0] == 0 and res.shape[0] == 0
    assert model.pack_sets(root[:0].contiguous()).shape[0] == 0
    assert model.unpack_sets(sets[:0].contiguous()).shape[0] == 0
    dead = sets.clone()
    dead[0, 3, :] = -1  # every value of variable 3 marked
    un = model.unpack_sets(dead)
    assert un[0, 3].tolist() == [1, 0] and torch.equal(un[0, :3], root[0, :3])


def test_root_limit_and_failing_variable_and_value_batches():
    """propagate(root, limit): a chain whose bounds move by one per sweep stops after limit + 1 rounds, wider than
    the fixpoint and containing it; an inconsistent node names a variable whose domain emptied; several values of
    one variable in one launch give what the single-node entry gives for each."""
    from csolve_amd import problems
    from csolve_amd.solver import Model, solve_root
    text = "ANY; a < b; b < c; c < a + 90; 0 <= a; a <= 100; 0 <= b; b <= 100; 0 <= c; c <= 100;"
    free = Model.from_text(text)
    st, rounds = free.root_propagate_limit(-1)
    fix = free.domains()
    cut = Model.from_text(text)
    st2, r2 = cut.root_propagate_limit(1)
    assert st >= 0 and st2 >= 0 and r2 <= 2 <= rounds
    d = cut.domains()
    assert (d[:, 0] <= fix[:, 0]).all() and (d[:, 1] >= fix[:, 1]).all() and (d != fix).any()

    model = solve_root(problems.queens(8))
    root = model.domains()
    for k in (1, 7):
        if not model.qualifies(k):
            continue
        model.set_kernel(k)
        # X1 = 1, then X2 = 2 is on X1's diagonal: X2 is emptied
        st1, _, after = model.propagate_one(root, 0, 1, 1)
        assert st1 >= 0
        nodes = torch.tensor([[1, 2, 2, 0]], dtype=torch.int32, device="cuda")
        out, res = model.propagate(torch.from_numpy(after[None].copy()).cuda(), nodes)
        torch.cuda.synchronize()
        assert int(res[0, 0]) == -1 and 0 <= int(res[0, 3]) < 8, (k, res.tolist())
    model.set_kernel(0)
    vals = [8, 1, 7, 2, 5]
    res, outs = model.propagate_values(root, 3, vals)
    for i, v in enumerate(vals):
        st_one, props_one, out_one = model.propagate_one(root, 3, v, v)
        assert (res[i, 0] < 0) == (st_one < 0)
        if st_one >= 0:
            assert res[i, 0] == st_one and res[i, 1] == props_one and (outs[i] == out_one).all()


def test_root_limit_is_the_references_when_it_binds():
    """propagate(root, limit) on the device against the compiled reference's vectors (tests/golden/root_limit.json):
    verdict and domains after at most limit + 1 sweeps, also where the limit cuts the iteration short of the fixpoint
    (the device then repeats the sweeps in the reference's own order, one clause after the other)"""
    import json
    from conftest import golden
    from csolve_amd.solver import Model
    cases = json.load(open(golden("root_limit.json")))["cases"]
    counts_equal = 0
    for c in cases:
        m = Model.from_text(c["text"])
        st, rounds = m.root_propagate_limit(c["limit"])
        assert (st < 0) == (c["status"] < 0), (c["problem"], c["limit"], st)
        if st < 0:
            continue
        dom = m.domains()
        got = {name: [int(dom[i, 0]), int(dom[i, 1])] for i, name in enumerate(m.var_names())}
        assert got == c["domains"], (c["problem"], c["limit"], got)
        assert rounds <= c["limit"] + 1
        counts_equal += st == c["status"]
    # the count of narrowings (the reference's return value, only ever compared with zero by its callers) agrees
    # wherever the iteration was run in the reference's order
    assert counts_equal >= 1
