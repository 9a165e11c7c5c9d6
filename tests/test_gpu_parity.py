"""GPU parity: the HIP path (through the C ABI) against the reference's golden walks,
against the CPU oracle on the same seeded inputs, and through size-independent
properties at the BASELINE sizes.  Bit-exact: integer interval domains."""
import glob
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

NE_ONLY = {"queens4", "queens8", "queens16", "queens64", "sudoku9_s7", "ref_sudoku"}
WALKS_WITH_MODEL = sorted(os.path.basename(p)[:-8] for p in glob.glob(golden("walks", "*.walk.gz"))
                          if os.path.exists(golden("models", os.path.basename(p)[:-8] + ".model")))


def _text(kind, size):
    from csolve_amd import problems
    if kind == "queens":
        return problems.queens(size)
    if kind == "sudoku":
        return problems.sudoku(size, 0.4, 1)
    if kind.startswith("offsets"):  # offsets48: 48 values per variable
        return problems.offsets(size, int(kind[7:]), 1)
    raise ValueError(kind)


def _kernels(model):
    """every kernel that can run this model"""
    ks = [1]
    if model.qualifies(2):
        ks.append(2)
    if model.forbidden_words() > 0:
        ks.append(3)  # forbidden-set kernel, sets rebuilt from the incoming state
    if model.qualifies(4):
        ks.append(4)  # the same with the sets in registers
    if model.qualifies(5):
        ks.append(5)  # several nodes per wave (at most 32 variables)
    if model.qualifies(6):
        ks.append(6)  # small models: clauses resident in registers, all revised per round
    if model.qualifies(7):
        ks.append(7)  # interval states only: bounds shaved and verified on demand, no forbidden sets
    return ks


def _gpu_walk(model, walk):
    before = torch.from_numpy(walk["before"]).cuda().contiguous()
    B = before.shape[0]
    nodes = np.stack([walk["var"], walk["value"], walk["value"], np.arange(B, dtype=np.int32)], 1).astype(np.int32)
    out, res = model.propagate(before, torch.from_numpy(nodes).cuda())
    torch.cuda.synchronize()
    return out.cpu().numpy(), res.cpu().numpy()


@pytest.mark.parametrize("name", WALKS_WITH_MODEL)
def test_reference_walks_on_reference_model(name):
    """Every golden node instance (recorded from the compiled reference): same verdict, same
    fixpoint domains; on pure != networks also the same PROPS count."""
    from csolve_amd.solver import Model
    from oracle.cs_oracle import read_walk
    walk = read_walk(golden("walks", name + ".walk.gz"))
    model = Model.from_dump(golden("models", name + ".model")).finalize()
    if name in NE_ONLY:
        assert model.kernel() in (3, 4, 5, 7), "pure != networks of this size must take a forbidden-set or the shaving kernel"
    for k in _kernels(model):
        model.set_kernel(k)
        out, res = _gpu_walk(model, walk)
        fail_ref = walk["status"] < 0
        assert ((res[:, 0] < 0) == fail_ref).all()
        ok = ~fail_ref
        assert (out[ok] == walk["after"][ok]).all()
        assert (res[ok, 0] == (out[ok, :, 0] != out[ok, :, 1]).sum(1)).all()  # status = open variables
        if name in NE_ONLY:
            assert (res[ok, 1] == walk["status"][ok]).all()


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-8] for p in glob.glob(golden("walks", "*.walk.gz"))))
def test_reference_walks_through_own_front_end(name):
    """Same instances, but the model is built from the problem TEXT by the product's own
    front end and root phase (device), not loaded from the reference's dump."""
    from csolve_amd.solver import solve_root
    from oracle.cs_oracle import read_walk
    walk = read_walk(golden("walks", name + ".walk.gz"))
    model = solve_root(open(golden("problems", name + ".txt")).read())
    assert model.n_vars == walk["n_vars"]
    for k in _kernels(model):
        model.set_kernel(k)
        out, res = _gpu_walk(model, walk)
        fail_ref = walk["status"] < 0
        assert ((res[:, 0] < 0) == fail_ref).all()
        ok = ~fail_ref
        assert (out[ok] == walk["after"][ok]).all()


@pytest.mark.parametrize("name", ["queens8", "queens64", "ref_sudoku", "ref_schedule", "ref_wcet", "schedule6_s1"])
def test_root_phase_matches_reference_dump(name):
    """parse + device root sweeps reach the root domains the reference reached."""
    from csolve_amd.solver import Model, solve_root
    ref = Model.from_dump(golden("models", name + ".model"))
    mine = solve_root(open(golden("problems", name + ".txt")).read())
    assert mine.var_names() == ref.var_names()
    assert (mine.domains() == ref.domains()).all()


def test_infeasible_root():
    from csolve_amd.solver import Model
    m = Model.from_text("ANY; x = 1; x = 2;")
    assert m.root_propagate() == -1
    m = Model.from_text("ANY; 1 = 2;")
    assert m.root_propagate() == -1
    m = Model.from_text("ANY; x < y; y < x; 0 <= x; x <= 5; 0 <= y; y <= 5;")
    assert m.root_propagate() == -1


def test_unbounded_variable_is_reported():
    from csolve_amd import CsolveError
    from csolve_amd.solver import Model
    m = Model.from_text("ANY; x < y; 0 <= x;")
    assert m.root_propagate() >= 0
    with pytest.raises(CsolveError, match="unbounded variable: "):
        m.finalize()


def _random_nodes(rng, states, count):
    """pick open variables and values inside their current interval"""
    B, n, _ = states.shape
    nodes = np.zeros((count, 4), dtype=np.int32)
    for i in range(count):
        p = rng.integers(B)
        open_vars = np.nonzero(states[p, :, 0] < states[p, :, 1])[0]
        v = open_vars[rng.integers(len(open_vars))] if len(open_vars) else rng.integers(n)
        val = rng.integers(states[p, v, 0], states[p, v, 1] + 1)
        nodes[i] = (v, val, val, p)
    return nodes


@pytest.mark.parametrize("kind,size,kernel", [("queens", 64, 1), ("queens", 64, 2), ("queens", 64, 3),
                                              ("queens", 128, 1), ("queens", 128, 2), ("queens", 128, 3),
                                              ("sudoku", 5, 1), ("sudoku", 5, 2), ("sudoku", 5, 3),
                                              ("queens", 64, 4), ("queens", 128, 4), ("queens", 100, 4),
                                              ("sudoku", 3, 4), ("sudoku", 4, 4),
                                              ("schedule", 16, 1), ("schedule", 16, 6), ("schedule", 10, 6),
                                              ("schedule", 24, 6)])
def test_batch_vs_oracle_and_properties(kind, size, kernel):
    """Seeded multi-level batches at the BASELINE sizes: a sample is checked against the oracle
    bit for bit; the whole batch is checked through properties -- the output is contained in the
    input, is a fixpoint (re-propagating it with every variable marked changed narrows nothing),
    and the verdict does not depend on how the batch is split."""
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    from oracle.cs_oracle import Model as OModel, Oracle
    text = {"queens": lambda: problems.queens(size), "sudoku": lambda: problems.sudoku(size, 0.4, 1),
            "schedule": lambda: problems.schedule(size, 1)}[kind]()
    model = solve_root(text)
    model.set_kernel(kernel)
    assert model.kernel() == kernel
    if 3 <= kernel <= 5:
        assert model.forbidden_words() == {("queens", 64): 1, ("queens", 128): 2, ("queens", 100): 2, ("sudoku", 5): 1,
                                           ("sudoku", 3): 1, ("sudoku", 4): 1}[(kind, size)]
    n = model.n_vars
    omodel = OModel.parse(text)
    omodel.set_domains(model.domains())
    omodel.index()
    orc = Oracle(omodel)

    rng = np.random.default_rng(2024)
    states = model.domains()[None].copy()
    for level in range(4):
        count = 4096 if level == 3 else 512
        nodes = _random_nodes(rng, states, count)
        d_states = torch.from_numpy(states).cuda()
        d_nodes = torch.from_numpy(nodes).cuda()
        out, res = model.propagate(d_states, d_nodes)
        torch.cuda.synchronize()
        out_h, res_h = out.cpu().numpy(), res.cpu().numpy()
        ok = res_h[:, 0] >= 0
        # oracle on a sample
        sample = rng.choice(count, size=min(count, 192), replace=False)
        st, exp = orc.instances(states[nodes[sample, 3]], nodes[sample, 0], nodes[sample, 1])
        assert ((st < 0) == ~ok[sample]).all()
        good = sample[st >= 0]
        assert (out_h[good] == exp[st >= 0]).all()
        if kind != "schedule":
            assert (res_h[good, 1] == st[st >= 0]).all()
        # contained in the parent state with the assignment applied
        parent = states[nodes[:, 3]]
        assert (out_h[ok, :, 0] >= parent[ok, :, 0]).all() and (out_h[ok, :, 1] <= parent[ok, :, 1]).all()
        assert (out_h[ok, :, 0] <= out_h[ok, :, 1]).all()
        assert (res_h[ok, 0] == (out_h[ok, :, 0] != out_h[ok, :, 1]).sum(1)).all()  # status = open variables
        idx = np.nonzero(ok)[0]
        assert (out_h[idx, nodes[idx, 0], 0] == nodes[idx, 1]).all()
        # fixpoint: full re-propagation changes nothing
        again_nodes = np.stack([np.full(len(idx), -1), np.zeros(len(idx)), np.zeros(len(idx)),
                                np.arange(len(idx))], 1).astype(np.int32)
        sub = out[torch.from_numpy(idx).cuda()].contiguous()
        out2, res2 = model.propagate(sub, torch.from_numpy(again_nodes).cuda())
        torch.cuda.synchronize()
        assert (res2[:, 0] >= 0).all() and (res2[:, 1] == 0).all()
        assert torch.equal(out2, sub)
        # splitting the batch does not change anything
        half = count // 2
        o1, r1 = model.propagate(d_states, d_nodes[:half].contiguous())
        o2, r2 = model.propagate(d_states, d_nodes[half:].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([r1, r2])[:, 0], res[:, 0])
        okt = torch.from_numpy(ok).cuda()
        assert torch.equal(torch.cat([o1, o2])[okt], out[okt])
        states = out_h[ok][:256]
        if len(states) == 0:
            break


def test_eval_root_on_solutions():
    """update_solution's check: the root evaluates to true on a solution, false on a
    violated assignment, undecided on an open state."""
    import json
    from csolve_amd.solver import solve_root
    stats = json.load(open(golden("solve_stats.json")))
    rec = next(r for r in stats if r["problem"] == "queens8" and not r["flags"])
    model = solve_root(open(golden("problems", "queens8.txt")).read())
    names = model.var_names()
    sol = np.array([[rec["last_solution"][k]] * 2 for k in names], dtype=np.int32)
    bad = sol.copy()
    bad[0] = bad[1]
    states = np.stack([sol, bad, model.domains()])
    truth = model.eval_root(torch.from_numpy(states).cuda()).cpu().numpy()
    assert truth.tolist() == [1, 0, 2]


def test_propagate_one_host_path():
    from csolve_amd.solver import Model
    from oracle.cs_oracle import read_walk
    walk = read_walk(golden("walks", "queens8.walk.gz"))
    model = Model.from_dump(golden("models", "queens8.model")).finalize()
    for i in range(40):
        st, props, out = model.propagate_one(walk["before"][i], int(walk["var"][i]), int(walk["value"][i]),
                                             int(walk["value"][i]))
        if walk["status"][i] < 0:
            assert st == -1
        else:
            assert st == int((out[:, 0] != out[:, 1]).sum()) and props == walk["status"][i]
            assert (out == walk["after"][i]).all()


@pytest.mark.parametrize("kind,size,kernel", [("queens", 16, 3), ("queens", 64, 3), ("queens", 128, 3), ("sudoku", 3, 3),
                                              ("sudoku", 5, 3), ("queens", 16, 4), ("queens", 64, 4), ("queens", 128, 4),
                                              ("queens", 100, 4), ("sudoku", 3, 4), ("sudoku", 4, 4),
                                              ("queens", 16, 5), ("queens", 13, 5), ("queens", 17, 5), ("queens", 32, 5),
                                              ("offsets48", 20, 5), ("offsets48", 12, 5), ("offsets24", 30, 5),
                                              ("offsets48", 20, 4), ("offsets40", 40, 4), ("offsets48", 20, 3)])
def test_forbidden_sets_inherited_down_a_path(kind, size, kernel):
    """The forbidden-set kernel with the sets carried from parent to child (the search engine's
    mode) against the general kernel and the oracle, five levels deep: same verdicts, same
    domains, same PROPS; and the carried sets equal the sets rebuilt from scratch."""
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    from oracle.cs_oracle import Model as OModel, Oracle
    text = _text(kind, size)
    model = solve_root(text)
    fw = model.forbidden_words()
    assert fw > 0
    model.set_kernel(kernel)  # 3: sets in LDS, 4: sets in registers, 5: in registers, 2 or 4 nodes per wave
    n = model.n_vars
    omodel = OModel.parse(text)
    omodel.set_domains(model.domains())
    omodel.index()
    orc = Oracle(omodel)
    rng = np.random.default_rng(7)
    root = model.root_state()
    full = torch.tensor([[-1, 0, 0, 0]], dtype=torch.int32, device="cuda")
    states, forb, res = model.propagate_fb(root, full)  # root sets
    torch.cuda.synchronize()
    assert int(res[0, 0]) >= 0 and int(res[0, 1]) == 0 and torch.equal(states, root)
    states_h = states.cpu().numpy()
    for level in range(5):
        nodes = _random_nodes(rng, states_h, 768)
        d_nodes = torch.from_numpy(nodes).cuda()
        out3, forb3, res3 = model.propagate_fb(states, d_nodes, forb_in=forb)
        model.set_kernel(1)
        out1, res1 = model.propagate(states, d_nodes)
        model.set_kernel(kernel)
        torch.cuda.synchronize()
        ok = res1[:, 0] >= 0
        assert torch.equal(res3[:, 0] >= 0, ok)
        assert torch.equal(out3[ok], out1[ok])
        assert torch.equal(res3[ok][:, :2], res1[ok][:, :2])  # status (open variables) and PROPS
        # oracle on a sample
        sample = rng.choice(len(nodes), size=96, replace=False)
        st, exp = orc.instances(states_h[nodes[sample, 3]], nodes[sample, 0], nodes[sample, 1])
        r3 = res3.cpu().numpy()
        assert ((st < 0) == (r3[sample, 0] < 0)).all()
        good = sample[st >= 0]
        assert (out3.cpu().numpy()[good] == exp[st >= 0]).all() and (r3[good, 1] == st[st >= 0]).all()
        # the inherited sets are the sets one would rebuild from the child state
        idx = torch.nonzero(ok)[:256, 0]
        sub = out3[idx].contiguous()
        again = torch.stack([torch.full((len(idx),), -1), torch.zeros(len(idx)), torch.zeros(len(idx)),
                             torch.arange(len(idx))], 1).to(torch.int32).cuda()
        s2, f2, r2 = model.propagate_fb(sub, again)
        torch.cuda.synchronize()
        assert torch.equal(s2, sub) and (r2[:, 1] == 0).all()
        assert torch.equal(f2, forb3[idx])
        keep = idx[:128]
        states, forb = out3[keep].contiguous(), forb3[keep].contiguous()
        states_h = states.cpu().numpy()
        if len(keep) == 0:
            break


@pytest.mark.parametrize("kind,size,count", [("queens", 64, 1 << 17), ("queens", 128, 1 << 16), ("queens", 100, 1 << 15),
                                             ("sudoku", 4, 1 << 15), ("sudoku", 3, 1 << 15),
                                             ("queens", 16, (1 << 17) + 3), ("queens", 30, (1 << 16) + 1),
                                             ("offsets48", 20, (1 << 16) + 2), ("offsets24", 14, 1 << 16)])
def test_large_batches_agree_across_kernels(kind, size, count):
    """Batches large enough that every wave walks through several chunks of nodes under full load
    (the small parity batches give each wave at most one chunk): the forbidden-set kernels with
    resident sets (3: LDS, 4: registers, 5: several nodes per wave, where the model qualifies) against the general kernel on
    every node -- verdict, fixpoint, PROPS -- and against each other on the carried sets, over
    repeated launches (a timing-dependent fault shows up as a difference between launches)."""
    import bench
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    text = _text(kind, size)
    model = solve_root(text)
    assert model.forbidden_words() > 0
    states_in, nodes, forb_in = bench.make_instances(model, count, seed=99, walks=4096, with_sets=True, restore_kernel=3)
    model.set_kernel(1)
    o1, r1 = model.propagate(states_in, nodes)
    torch.cuda.synchronize()
    ok = r1[:, 0] >= 0
    assert 0 < int(ok.sum()) < count
    sets = {}
    for k in (3, 4, 5):
        if not model.qualifies(k):
            continue
        model.set_kernel(k)
        for launch in range(3):
            o, f, r = model.propagate_fb(states_in, nodes, forb_in=forb_in)
            torch.cuda.synchronize()
            assert torch.equal(r[:, 0] >= 0, ok), (k, launch)
            assert torch.equal(o[ok], o1[ok]), (k, launch)
            assert torch.equal(r[ok][:, :2], r1[ok][:, :2]), (k, launch)
            if k in sets:
                assert torch.equal(f[ok], sets[k]), (k, launch)
            sets[k] = f[ok]
    # the sets agree on every value of the variables' root domains (bits of values outside a root domain are
    # unspecified: kernel 5 leaves the high word alone when all root domains fit 32 values)
    dom = model.domains()
    fw = model.forbidden_words()
    width = (dom[:, 1].astype(np.int64) - dom[:, 0] + 1)
    mask = np.zeros((model.n_vars, fw), dtype=np.uint64)
    for v in range(model.n_vars):
        for q in range(fw):
            bits = int(min(max(width[v] - 64 * q, 0), 64))
            mask[v, q] = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    d_mask = torch.from_numpy(mask.view(np.int64)).cuda()
    for k in sets:
        assert torch.equal(sets[3] & d_mask, sets[k] & d_mask), k
        if k != 5:
            assert torch.equal(sets[3], sets[k]), k


@pytest.mark.parametrize("kind,size,count", [("queens", 64, 1 << 16), ("queens", 128, 1 << 16), ("sudoku", 5, 1 << 15)])
def test_full_batches_against_the_compiled_reference(kind, size, count):
    """Every instance of a bench-sized batch (BASELINE configs[1], the north-star instance, configs[2]) through every
    kernel the selector can pick for the model -- state-only entry, resident-sets entry, sets-only entry -- against
    the COMPILED REFERENCE replaying the same instances (oracle/_ref/csolve_ref bench: the reference's own bind() +
    propagate_clauses() on one host core): verdict, fixpoint and PROPS of every single node, not a sample and not
    another kernel of this library."""
    import bench
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    if not os.path.exists(bench.REF_BIN):
        pytest.skip("oracle/_ref/csolve_ref not built (needs the reference tree)")
    text = _text(kind, size)
    model = solve_root(text)
    n, fw = model.n_vars, model.forbidden_words()
    states_in, nodes, forb_in = bench.make_instances(model, count, seed=31, walks=4096, with_sets=True, restore_kernel=0)
    si, nd = states_in.cpu().numpy(), nodes.cpu().numpy()
    _, status, after = bench.reference_replay(text, n, si, nd)
    fail = status < 0
    assert 0 < int(fail.sum()) < count

    def check(out, res, what):
        out, res = out.cpu().numpy(), res.cpu().numpy()
        assert ((res[:, 0] < 0) == fail).all(), what
        assert (out[~fail] == after[~fail]).all(), what
        assert (res[~fail, 1] == status[~fail]).all(), what  # PROPS: order-independent on != networks
        assert (res[~fail, 0] == (after[~fail][:, :, 0] != after[~fail][:, :, 1]).sum(1)).all(), what

    ran = []
    for k in (0, 7, 5, 4, 3, 2, 1):
        if k and not model.qualifies(k):
            continue
        model.set_kernel(k)
        out, res = model.propagate(states_in, nodes)
        torch.cuda.synchronize()
        check(out, res, f"state-only entry, kernel {k or model.kernel()}")
        ran.append(k or model.kernel())
        if k in (0, 5, 4, 3) and fw > 0:
            out, _, res = model.propagate_fb(states_in, nodes, forb_in=forb_in)
            torch.cuda.synchronize()
            check(out, res, f"resident sets, kernel {k}")
    model.set_kernel(0)
    if model.qualifies(4):
        sets_out, res = model.propagate_sets(model.pack_sets(states_in), nodes)
        torch.cuda.synchronize()
        check(model.unpack_sets(sets_out), res, "sets-only entry")
    assert (7 in ran) == (size != 5), ran  # queens-64 / -128 take the shaving kernel, sudoku-25 (625 variables) kernel 3


@pytest.mark.parametrize("name", ["ref_schedule", "schedule6_s1", "ref_wcet"])
def test_linear_fast_paths_equal_the_tree_interpreter(name):
    """EQ / LT / two-literal OR clauses on the direct bound-propagation paths against the same clauses
    through the expression-tree interpreter: same verdicts and fixpoints on multi-level random batches;
    the schedule models need no tree at all on the fast paths."""
    from csolve_amd.solver import set_linear_fast_paths, solve_root
    text = open(golden("problems", name + ".txt")).read()
    try:
        set_linear_fast_paths(False)
        slow = solve_root(text)
        set_linear_fast_paths(True)
        fast = solve_root(text)
    finally:
        set_linear_fast_paths(True)
    assert (slow.domains() == fast.domains()).all()
    assert slow.device_info()["tree_clauses"] > fast.device_info()["tree_clauses"]
    if name != "ref_wcet":
        assert fast.device_info()["tree_clauses"] == 0
    rng = np.random.default_rng(11)
    states = fast.domains()[None].copy()
    for level in range(6):
        nodes = _random_nodes(rng, states, 1024)
        d_states, d_nodes = torch.from_numpy(states).cuda(), torch.from_numpy(nodes).cuda()
        of, rf = fast.propagate(d_states, d_nodes)
        os_, rs = slow.propagate(d_states, d_nodes)
        torch.cuda.synchronize()
        ok = rs[:, 0] >= 0
        assert torch.equal(rf[:, 0] >= 0, ok)
        assert torch.equal(of[ok], os_[ok])
        assert torch.equal(rf[ok][:, 0], rs[ok][:, 0])
        # the three-valued root value of every state, open or complete
        assert torch.equal(fast.eval_root(of[ok].contiguous()), slow.eval_root(of[ok].contiguous()))
        states = of[ok][:256].cpu().numpy()
        if len(states) == 0:
            break
    # dives that assign the UPPER bound of every open variable in turn (an unbounded-above objective ends up at
    # 2^31 - 2, where a shifted bound must not turn into a sentinel): same states, same root values
    cur = fast.domains()[None].copy()
    for step in range(fast.n_vars + 1):
        open_vars = np.nonzero(cur[0, :, 0] < cur[0, :, 1])[0]
        d_cur = torch.from_numpy(cur).cuda()
        assert torch.equal(fast.eval_root(d_cur), slow.eval_root(d_cur))
        if len(open_vars) == 0:
            assert int(fast.eval_root(d_cur)[0]) in (0, 1)
            break
        v = int(open_vars[0])
        node = torch.tensor([[v, cur[0, v, 1], cur[0, v, 1], 0]], dtype=torch.int32, device="cuda")
        of, rf = fast.propagate(d_cur, node)
        os_, rs = slow.propagate(d_cur, node)
        torch.cuda.synchronize()
        assert int(rf[0, 0] >= 0) == int(rs[0, 0] >= 0)
        if int(rs[0, 0]) < 0:
            cur[0, v, 1] -= 1  # the upper bound is inconsistent: shave it like the search would
            continue
        assert torch.equal(of, os_)
        cur = of.cpu().numpy()


@pytest.mark.parametrize("kind,size", [("queens", 16), ("queens", 64), ("queens", 128), ("queens", 100), ("sudoku", 3), ("sudoku", 4)])
def test_sets_only_states(kind, size):
    """The sets-only layout (the domains as bit vectors, csgpu_sets_*): pack / unpack round trip, and five
    levels of batches against the interval + sets layout -- same verdicts, PROPS and open-variable counts,
    the unpacked sets are the interval states, and the sets equal the packed interval states bit for bit."""
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    text = problems.queens(size) if kind == "queens" else problems.sudoku(size, 0.4, 1)
    model = solve_root(text)
    assert model.qualifies(4)
    rng = np.random.default_rng(21)
    root = model.root_state()
    sets = model.pack_sets(root)
    assert torch.equal(model.unpack_sets(sets), root)
    full = torch.tensor([[-1, 0, 0, 0]], dtype=torch.int32, device="cuda")
    _, forb, _ = model.propagate_fb(root, full)
    states = root
    for level in range(5):
        nodes = _random_nodes(rng, states.cpu().numpy(), 2048)
        if level == 1:  # some interval assignments and some full re-propagations
            h = states.cpu().numpy()
            for i in range(0, 2048, 7):
                v, p = nodes[i, 0], nodes[i, 3]
                nodes[i, 1], nodes[i, 2] = h[p, v, 0], max(h[p, v, 0], (h[p, v, 0] + h[p, v, 1]) // 2)
            nodes[3::11, 0] = -1
        d_nodes = torch.from_numpy(nodes).cuda()
        out, forb_out, res = model.propagate_fb(states, d_nodes, forb_in=forb)
        sets_out, res_s = model.propagate_sets(sets, d_nodes)
        torch.cuda.synchronize()
        ok = res[:, 0] >= 0
        assert torch.equal(res_s[:, 0] >= 0, ok)
        assert torch.equal(res_s[ok][:, :3], res[ok][:, :3])  # open variables, PROPS, revisions
        assert torch.equal(model.unpack_sets(sets_out[ok].contiguous()), out[ok])
        assert torch.equal(sets_out[ok], model.pack_sets(out[ok].contiguous()))
        keep = torch.nonzero(ok)[:256, 0]
        if len(keep) == 0:
            break
        states, forb, sets = out[keep].contiguous(), forb_out[keep].contiguous(), sets_out[keep].contiguous()


def test_all_kernels_agree_on_irregular_networks():
    """Seeded != networks of irregular shape (3..69 variables, 3..64 values, own lower bound per variable, zero
    to two constraints per pair with arbitrary offsets): every kernel the model qualifies for against the
    general one on multi-level random batches -- verdict, fixpoint, open count and PROPS of every node
    (tools/fuzz_kernels.py runs the same over more models)."""
    import bench
    from csolve_amd import problems
    from csolve_amd.solver import solve_root
    rng = np.random.default_rng(11)
    seen = set()
    for seed in range(16):
        n, values = int(rng.integers(3, 70)), int(rng.integers(3, 65))
        model = solve_root(problems.offsets(n, values, seed + 1))
        with_sets = model.forbidden_words() > 0
        states_in, nodes, forb_in = bench.make_instances(model, 4096, seed=seed, walks=512, with_sets=with_sets, restore_kernel=0)
        model.set_kernel(1)
        o1, r1 = model.propagate(states_in, nodes)
        torch.cuda.synchronize()
        ok = r1[:, 0] >= 0
        for k in (2, 3, 4, 5, 6, 7):
            if not model.qualifies(k):
                continue
            model.set_kernel(k)
            if k in (3, 4, 5) and forb_in is not None:
                o, _, r = model.propagate_fb(states_in, nodes, forb_in=forb_in)
            else:
                o, r = model.propagate(states_in, nodes)
            torch.cuda.synchronize()
            assert torch.equal(r[:, 0] >= 0, ok), (seed, n, values, k)
            assert torch.equal(o[ok], o1[ok]), (seed, n, values, k)
            assert torch.equal(r[ok][:, :2], r1[ok][:, :2]), (seed, n, values, k)
            seen.add(k)
    assert seen == {2, 3, 4, 5, 6, 7}


def test_linear_mixtures_through_both_general_kernels_and_the_oracle():
    """Seeded mixtures of <, <=, =, != and two-literal disjunctions (csolve_amd.problems.linear): the event-driven
    kernel, the clause-resident kernel and the tree interpreter (linear fast paths off) give the same verdicts
    and fixpoints on multi-level random batches, and a sample equals the oracle's."""
    import bench
    from csolve_amd import problems
    from csolve_amd.solver import set_linear_fast_paths, solve_root
    from oracle.cs_oracle import Model as OModel, Oracle
    checked = 0
    for seed in range(1, 13):
        text = problems.linear(6 + 3 * seed, seed)
        try:
            model = solve_root(text)
        except Exception:
            continue  # inconsistent at the root
        assert model.qualifies(6)
        states_in, nodes, _ = bench.make_instances(model, 2048, seed=seed, walks=256)
        outs = {}
        for k in (1, 6):
            model.set_kernel(k)
            o, r = model.propagate(states_in, nodes)
            torch.cuda.synchronize()
            outs[k] = (o, r)
        try:
            set_linear_fast_paths(False)
            slow = solve_root(text)
        finally:
            set_linear_fast_paths(True)
        slow.set_kernel(1)
        outs["tree"] = slow.propagate(states_in, nodes)
        torch.cuda.synchronize()
        ok = outs[1][1][:, 0] >= 0
        for k in (6, "tree"):
            assert torch.equal(outs[k][1][:, 0] >= 0, ok), (seed, k)
            assert torch.equal(outs[k][0][ok], outs[1][0][ok]), (seed, k)
        om = OModel.parse(text)
        om.set_domains(model.domains())
        om.index()
        orc = Oracle(om)
        st_h, nd_h = states_in.cpu().numpy(), nodes.cpu().numpy()
        sample = np.arange(0, 2048, 16)
        st, exp = orc.instances(st_h[nd_h[sample, 3]], nd_h[sample, 0], nd_h[sample, 1])
        assert ((st < 0) == ~ok.cpu().numpy()[sample]).all(), seed
        good = sample[st >= 0]
        assert (outs[6][0].cpu().numpy()[good] == exp[st >= 0]).all(), seed
        checked += 1
    assert checked >= 6
