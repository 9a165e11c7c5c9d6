#!/usr/bin/env python3
"""Transcribe the known-answer vectors of the reference's unit tests of the search driver's helpers into JSON:

  test/test_strategy.c   VarCmp.* (ordering comparisons), VarOrder.* (the priority heap)  -> ref_unit_strategy.json
  test/test_objective.c  ObjectiveBetter / UpdateBest / UpdateVal / Best                  -> ref_unit_objective.json
  test/test_csolve.c     FailThresholdNext.Basic (Luby), Step.Check, Step.Val             -> ref_unit_search.json

VALUES ONLY: every test body is read as a tiny script (declare intervals and priorities, set a mode, call one
function, expect one value).  No reference code is copied.  Authoring container only:
    python tests/golden/transcribe_ref_search_vectors.py
"""
import json
import os
import re

REF = os.environ.get("CSOLVE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
CONSTS = {"DOMAIN_MIN": -2**31, "DOMAIN_MAX": 2**31 - 1}


def num(s):
    return int(eval(s.replace("U", ""), {"__builtins__": {}}, CONSTS))


def val(s):
    m = re.fullmatch(r"VALUE\((.*)\)", s.strip())
    if m:
        return [num(m.group(1))] * 2
    m = re.fullmatch(r"INTERVAL\((.*),(.*)\)", s.strip())
    return [num(m.group(1)), num(m.group(2))]


def tests_of(path):
    """yield (suite, name, first line, [(line no, statement)])"""
    text = open(path).read().split("\n")
    i = 0
    while i < len(text):
        m = re.match(r"TEST\((\w+),\s*(\w+)\)", text[i])
        if not m:
            i += 1
            continue
        first, body, stmt, start = i + 1, [], "", i + 1
        i += 1
        while not text[i].startswith("}"):
            ln = text[i].strip()
            if ln:
                if not stmt:
                    start = i + 1
                stmt += " " + ln
                if ln.endswith(";"):
                    body.append((start, stmt.strip()))
                    stmt = ""
            i += 1
        yield m.group(1), m.group(2), first, body


ORDERS = {"ORDER_NONE": "none", "ORDER_SMALLEST_DOMAIN": "smallest-domain", "ORDER_LARGEST_DOMAIN": "largest-domain",
          "ORDER_SMALLEST_VALUE": "smallest-value", "ORDER_LARGEST_VALUE": "largest-value"}


def strategy():
    path = os.path.join(REF, "test", "test_strategy.c")
    cmp_cases, heap_cases = [], []
    for suite, name, first, body in tests_of(path):
        if suite not in ("VarCmp", "VarOrder") or name in ("Error", "Parent", "Left", "Right"):
            continue
        terms, envs, order, prefer = {}, {}, "none", False
        heap, size, script = {}, None, []
        for ln, st in body:
            m = re.match(r"struct constr_t (\w+) = CONSTRAINT_TERM\((.*)\);", st)
            if m:
                terms[m.group(1)] = val(m.group(2))
                continue
            m = re.match(r"env\[(\d+)\] = \{.*\.val = &(\w+),.*\.order = (\w+), \.prio = (-?\d+),", st)
            if m:
                envs[int(m.group(1))] = {"val": terms[m.group(2)], "prio": int(m.group(4))}
                continue
            m = re.match(r"_order = (\w+);", st)
            if m:
                order = ORDERS[m.group(1)]
                continue
            m = re.match(r"_prefer_failing = (\w+);", st)
            if m:
                prefer = m.group(1) == "true"
                continue
            m = re.match(r"EXPECT_(GT|LT|EQ)\(strategy_var_cmp\(&env\[(\d+)\], &env\[(\d+)\]\), 0\);", st)
            if m:
                cmp_cases.append({"ref": f"test_strategy.c:{ln} {suite}.{name}", "order": order, "prefer_failing": prefer,
                                  "a": envs[int(m.group(2))], "b": envs[int(m.group(3))],
                                  "sign": {"GT": 1, "LT": -1, "EQ": 0}[m.group(1)]})
                continue
            m = re.match(r"_var_order_size = (\d+);", st)
            if m:
                size = int(m.group(1))
                continue
            m = re.match(r"_var_order\[(\d+)\] = &env\[(\d+)\];", st)
            if m:
                heap[int(m.group(1))] = int(m.group(2))
                continue
            m = re.match(r"strategy_var_order_(up|down)\((\d+)\);", st)
            if m:
                script.append({"op": m.group(1), "pos": int(m.group(2)), "line": ln})
                continue
            m = re.match(r"strategy_var_order_swap\((\d+), (\d+)\);", st)
            if m:
                script.append({"op": "swap", "pos": int(m.group(1)), "pos2": int(m.group(2)), "line": ln})
                continue
            m = re.match(r"strategy_var_order_(push|update)\(&env\[(\d+)\]\);", st)
            if m:
                script.append({"op": m.group(1), "var": int(m.group(2)), "line": ln})
                continue
            m = re.match(r"EXPECT_EQ\(strategy_var_order_pop\(\), &env\[(\d+)\]\);", st)
            if m:
                script.append({"op": "pop", "returns": int(m.group(1)), "line": ln})
                continue
            m = re.match(r"EXPECT_EQ\(_var_order\[(\d+)\], &env\[(\d+)\]\);", st)
            if m:
                script[-1].setdefault("heap_after", {})[m.group(1)] = int(m.group(2))
                continue
            m = re.match(r"EXPECT_EQ\(_var_order_size, (\d+)\);", st)
            if m:
                script[-1]["size_after"] = int(m.group(1))
                continue
            m = re.match(r"EXPECT_EQ\((\d+)U, env\[(\d+)\]\.order\);", st)
            if m:
                script[-1].setdefault("position_after", {})[m.group(2)] = int(m.group(1))
                continue
        if suite == "VarOrder":
            if size is None:
                size = len(heap)
            heap_cases.append({"ref": f"test_strategy.c:{first} {suite}.{name}", "order": order, "prefer_failing": prefer,
                               "vars": [envs[i] for i in sorted(envs)], "heap": [heap[i] for i in range(size)],
                               "script": script})
    return {"source": "reference test/test_strategy.c: VarCmp.* and VarOrder.* (strategy.c:79-246)",
            "format": "var = {val: [lo, hi], prio}; cmp: sign of strategy_var_cmp(a, b); heap: initial array of variable "
                      "indices, then operations with the expected array / positions / size / popped variable after each",
            "cmp": cmp_cases, "heap": heap_cases}


def objective():
    path = os.path.join(REF, "test", "test_objective.c")
    cases = []
    for suite, name, first, body in tests_of(path):
        if not suite.startswith("Objective") or suite == "ObjectiveInit" or name == "Errors":
            continue
        obj, best, v = None, None, None
        for ln, st in body:
            m = re.match(r"_objective = OBJ_(\w+);", st)
            if m:
                obj = m.group(1)
                continue
            m = re.match(r"\*_objective_best = (.*);", st)
            if m:
                best = num(m.group(1))
                continue
            m = re.match(r"_objective_val = CONSTRAINT_TERM\((.*)\);", st)
            if m:
                v = val(m.group(1))
                continue
            ref = f"test_objective.c:{ln} {suite}.{name}"
            m = re.match(r"EXPECT_EQ\((true|false), objective_better\(\)\);", st)
            if m:
                cases.append({"ref": ref, "fn": "better", "objective": obj, "best": best, "val": v,
                              "expect": m.group(1) == "true"})
                continue
            if st == "objective_update_best();" or st == "objective_update_val();":
                pending = {"ref": ref, "fn": st[10:-3], "objective": obj, "best": best, "val": v}
                continue
            m = re.match(r"EXPECT_EQ\((-?\d+), \*_objective_best\);", st)
            if m:
                cases.append(dict(pending, expect_best=int(m.group(1))))
                continue
            m = re.match(r"EXPECT_EQ\(_objective_val\.constr\.term\.val, (.*)\);", st)
            if m:
                cases.append(dict(pending, expect_val=val(m.group(1))))
                continue
            m = re.match(r"EXPECT_EQ\((-?\d+), objective_best\(\)\);", st)
            if m:
                cases.append({"ref": ref, "fn": "best", "best": best, "expect_best": int(m.group(1))})
    return {"source": "reference test/test_objective.c:72-311 (objective.c:62-135)",
            "format": "better: objective_better() with the objective value `val` and the incumbent `best`; update_best: the "
                      "incumbent after a solution with objective value `val`; update_val: the objective value after the "
                      "incumbent bound was applied",
            "cases": cases}


def search():
    path = os.path.join(REF, "test", "test_csolve.c")
    out = {"source": "reference test/test_csolve.c:305-337 (csolve.c:76-83), 628-657 (csolve.c:323-338)"}
    for suite, name, first, body in tests_of(path):
        if (suite, name) == ("FailThresholdNext", "Basic"):
            seq = [num(re.match(r"EXPECT_EQ\((\d+)U, _fail_threshold\);", st).group(1)) for ln, st in body
                   if st.startswith("EXPECT_EQ")]
            out["luby"] = {"ref": f"test_csolve.c:{first} FailThresholdNext.Basic", "threshold": 1, "counter": 1,
                           "thresholds": seq, "note": "the first entry is the initial threshold, each further one "
                           "follows a call of fail_threshold_next()"}
        if suite == "Step" and name in ("Check", "Val"):
            bounds, it, rows = None, None, []
            for ln, st in body:
                m = re.match(r"struct val_t v = (.*);", st)
                if m:
                    bounds = val(m.group(1))
                m = re.match(r"s\.iter = (\d+);", st)
                if m:
                    it = int(m.group(1))
                m = re.match(r"EXPECT_EQ\((true|false), step_check\(&s\)\);", st)
                if m:
                    rows.append({"iter": it, "expect": m.group(1) == "true", "line": ln})
                if re.match(r"domain_t v\d = step_val\(&s\);", st):
                    rows.append({"iter": it, "line": ln})
            out["step_" + name.lower()] = {"ref": f"test_csolve.c:{first} Step.{name}", "bounds": bounds, "rows": rows}
    out["step_val"]["expect"] = "every value lies within the bounds; the values of two successive iterations differ"
    return out


def main():
    for name, data in (("strategy", strategy()), ("objective", objective()), ("search", search())):
        p = os.path.join(OUT, f"ref_unit_{name}.json")
        with open(p, "w") as f:
            json.dump(data, f, indent=1)
        n = sum(len(v) for v in data.values() if isinstance(v, list))
        print(p, n or "ok")


if __name__ == "__main__":
    main()
