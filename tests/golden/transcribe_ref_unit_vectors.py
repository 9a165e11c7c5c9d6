#!/usr/bin/env python3
"""Transcribe the known-answer vectors of the reference's own unit tests into JSON.

Reads (as text) /root/reference/test/test_arith.c, test_eval.c and test_propagate.c and
writes VALUES ONLY -- operands, operator names, expected results, expected bind()
arguments -- to tests/golden/ref_unit_{arith,eval,propagate}.json.  No reference code is
copied: each test body is interpreted as a tiny script (declare interval terminals,
build an expression, call one function, expect one result).

Run in the authoring container only (the reference tree does not exist on the GPU box):
    python tests/golden/transcribe_ref_unit_vectors.py
"""
import json
import os
import re
import sys

REF = os.environ.get("CSOLVE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

CONSTS = {"DOMAIN_MIN": -2**31, "DOMAIN_MAX": 2**31 - 1, "PROP_NONE": 0, "PROP_ERROR": -1, "NULL": None}


def num(expr: str) -> int:
    return int(eval(expr, {"__builtins__": {}}, CONSTS))


def split_args(s: str):
    """split on top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def val(expr: str):
    expr = expr.strip()
    m = re.fullmatch(r"VALUE\((.*)\)", expr)
    if m:
        v = num(m.group(1))
        return [v, v]
    m = re.fullmatch(r"INTERVAL\((.*)\)", expr)
    if m:
        a, b = split_args(m.group(1))
        return [num(a), num(b)]
    raise ValueError(expr)


def tests_of(path):
    """yield (suite, name, first_line_no, [statements])"""
    text = open(path).read()
    for m in re.finditer(r"^TEST\((\w+),\s*(\w+)\)\s*\{\n(.*?)^\}", text, re.S | re.M):
        line = text.count("\n", 0, m.start()) + 1
        body = re.sub(r"//.*", "", m.group(3))
        stmts = [s.strip() for s in re.split(r";\s*\n", body) if s.strip()]
        yield m.group(1), m.group(2), line, stmts


def transcribe_arith():
    cases = []
    for suite, name, line, stmts in tests_of(os.path.join(REF, "test", "test_arith.c")):
        for s in stmts:
            m = re.fullmatch(r"EXPECT_EQ\((.*)\)", s.rstrip(";"), re.S)
            if not m:
                continue
            exp, call = split_args(m.group(1))
            c = re.fullmatch(r"(neg|add|mul|min|max)\((.*)\)", call)
            cases.append({"test": f"{suite}.{name}", "line": line, "fn": c.group(1),
                          "args": [num(a) for a in split_args(c.group(2))], "expect": num(exp)})
    return cases


class Env:
    def __init__(self):
        self.terms = {}   # name -> {"dom": [lo,hi], "env": name or None}
        self.exprs = {}   # name -> ["OP", l, r] | ["WAND", [elems]] | None
        self.envs = {}    # env-struct name -> term name
        self.arrays = {}  # wand element arrays: name -> [term/expr names]
        self.skip = False

    def snapshot(self):
        return {"terms": json.loads(json.dumps(self.terms)), "exprs": json.loads(json.dumps(self.exprs))}


def ref(name: str):
    name = name.strip()
    if name == "NULL":
        return None
    assert name.startswith("&"), name
    return name[1:]


def handle_decl(env: Env, s: str) -> bool:
    s1 = " ".join(s.split())
    m = re.fullmatch(r"struct constr_t (\w+) = CONSTRAINT_TERM\((.*)\)", s1)
    if m:
        env.terms[m.group(1)] = {"dom": val(m.group(2)), "env": None}
        return True
    m = re.fullmatch(r"struct env_t (\w+) = \{.*?\.val = &(\w+),.*\}", s1)
    if m:
        env.envs[m.group(1)] = m.group(2)
        return True
    m = re.fullmatch(r"(\w+)\.constr\.term\.env = &(\w+)", s1)
    if m:
        env.terms[m.group(1)]["env"] = m.group(2)
        return True
    m = re.fullmatch(r"struct constr_t (\w+)", s1)
    if m:
        env.exprs[m.group(1)] = None
        return True
    m = re.fullmatch(r"struct wand_expr_t (\w+) ?\[\d+\] = \{(.*)\}", s1)
    if m:
        env.arrays[m.group(1)] = re.findall(r"\.constr = &(\w+)", m.group(2))
        return True
    m = re.fullmatch(r"(?:struct constr_t )?(\w+) = CONSTRAINT_EXPR\((\w+), (.*)\)", s1)
    if m:
        a = split_args(m.group(3))
        env.exprs[m.group(1)] = [m.group(2), ref(a[0]), ref(a[1])]
        return True
    m = re.fullmatch(r"(?:struct constr_t )?(\w+) = CONSTRAINT_WAND\((\d+), (\w+)\)", s1)
    if m:
        env.exprs[m.group(1)] = ["WAND", env.arrays[m.group(3)][: int(m.group(2))]]
        return True
    m = re.fullmatch(r"struct confl_elem_t (\w+) ?\[\d+\] = \{(.*)\}", s1)
    if m:
        env.arrays[m.group(1)] = [[t, val(v)[0]] for v, t in
                                  re.findall(r"\.val = (VALUE\([^)]*\)), \.var = &(\w+)", m.group(2))]
        return True
    m = re.fullmatch(r"(?:struct constr_t )?(\w+) = CONSTRAINT_CONFL\((\d+), (\w+)\)", s1)
    if m:
        env.exprs[m.group(1)] = ["CONFL", env.arrays[m.group(3)][: int(m.group(2))]]
        return True
    return False


def transcribe_eval():
    cases, skipped = [], 0
    for suite, name, line, stmts in tests_of(os.path.join(REF, "test", "test_eval.c")):
        env = Env()
        for s in stmts:
            if handle_decl(env, s):
                continue
            m = re.fullmatch(r"EXPECT_EQ\((.*)\)", " ".join(s.split()), re.S)
            if m:
                exp, call = split_args(m.group(1))
                c = re.fullmatch(r"eval_(\w+)\(&(\w+)\)", call)
                if env.skip or c is None:
                    skipped += 1
                    continue
                case = {"test": f"{suite}.{name}", "line": line, "fn": "eval_" + c.group(1),
                        "target": c.group(2), "expect": val(exp)}
                case.update(env.snapshot())
                cases.append(case)
    return cases, skipped


def transcribe_propagate():
    cases, skipped = [], 0
    for suite, name, line, stmts in tests_of(os.path.join(REF, "test", "test_propagate.c")):
        env = Env()
        binds, other_calls = [], False
        for s in stmts:
            s1 = " ".join(s.split())
            if s1.startswith("MockProxy = new Mock()"):
                binds, other_calls = [], False
                continue
            if s1.startswith("delete(MockProxy)"):
                continue
            if handle_decl(env, s):
                continue
            m = re.match(r"EXPECT_CALL\(\*MockProxy, (\w+)\((.*?)\)\)\s*(.*)", s1)
            if m:
                fn, args, tail = m.group(1), m.group(2), m.group(3)
                if fn == "bind":
                    a = split_args(args)
                    times = re.search(r"\.Times\((\d+)\)", tail)
                    for _ in range(int(times.group(1)) if times else 1):
                        binds.append([env.envs[ref(a[0])], *val(a[1])])
                elif fn in ("strategy_create_conflicts", "strategy_var_order_update", "conflict_reset"):
                    pass  # bookkeeping calls of the failure path; no values to record
                else:
                    other_calls = True
                continue
            m = re.fullmatch(r"EXPECT_EQ\((.*)\)", s1, re.S)
            if m:
                exp, call = split_args(m.group(1))
                c = re.fullmatch(r"(propagate_\w+)\(&(\w+), (.*), NULL\)", call)
                if c is None:
                    c2 = re.fullmatch(r"propagate\(&(\w+), (\d+)\)", call)
                    if c2 is None or env.skip or other_calls:
                        skipped += 1
                        continue
                    case = {"test": f"{suite}.{name}", "line": line, "fn": "propagate", "target": c2.group(1),
                            "limit": int(c2.group(2)), "expect": num(exp), "binds": binds}
                    case.update(env.snapshot())
                    cases.append(case)
                    continue
                if env.skip or other_calls:
                    skipped += 1
                    continue
                case = {"test": f"{suite}.{name}", "line": line, "fn": c.group(1), "target": c.group(2),
                        "val": val(c.group(3)), "expect": num(exp), "binds": binds}
                case.update(env.snapshot())
                cases.append(case)
                binds = []
    return cases, skipped


def main():
    if not os.path.isdir(os.path.join(REF, "test")):
        sys.exit("reference tree not found at " + REF)
    arith = transcribe_arith()
    ev, ev_skip = transcribe_eval()
    pr, pr_skip = transcribe_propagate()
    for name, data in (("arith", arith), ("eval", ev), ("propagate", pr)):
        with open(os.path.join(OUT, f"ref_unit_{name}.json"), "w") as f:
            json.dump({"source": f"reference test/test_{name}.c (values only)", "cases": data}, f, indent=0)
    print(f"arith {len(arith)} cases; eval {len(ev)} cases ({ev_skip} skipped); "
          f"propagate {len(pr)} cases ({pr_skip} skipped)")


if __name__ == "__main__":
    main()
