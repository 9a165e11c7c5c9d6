"""Golden vectors of propagate(root, limit) (reference src/propagate.c:474-485: at most limit + 1 sweeps of
propagate_wand, 379-392) from the compiled reference: `csolve_ref rootlimit <file> <limit>` runs the front end and ONE
call of the reference's propagate() on the raw root and prints the variables' domains.  Only where the limit cuts the
iteration short of the fixpoint do the domains depend on the sweep order, so the cases are chains whose bounds travel
one clause per sweep against the order of the clauses.
    python tests/golden/make_root_limit_golden.py     (authoring container: needs oracle/_ref/csolve_ref)"""
import json
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "csolve_ref")


def chain(n, top):
    # x1 < x2 < ... < xn, all in [0, top]: lower bounds run with the clause order, upper bounds against it
    lines = ["ANY;"] + [f"x{i} < x{i + 1};" for i in range(1, n)] + [f"0 <= x{i}; x{i} <= {top};" for i in range(1, n + 1)]
    return " ".join(lines)


PROBLEMS = {
    "three_chain": "ANY; a < b; b < c; c < a + 90; 0 <= a; a <= 100; 0 <= b; b <= 100; 0 <= c; c <= 100;",
    "creeping_infeasible": "ANY; a < b; b < c; c < a + 2; 0 <= a; a <= 40; 0 <= b; b <= 40; 0 <= c; c <= 40;",
    "chain12": chain(12, 30),
    "sum_chain": "ANY; a + b = c; c + 1 = d; d < a + 7; 0 <= a; a <= 20; 0 <= b; b <= 20; 0 <= c; c <= 50; 0 <= d; d <= 50; b > 2;",
}
LIMITS = {"three_chain": [0, 1, 2, 50], "creeping_infeasible": [0, 3, 10, 25, 200], "chain12": [0, 1, 2, 5, 8, 11, 20],
          "sum_chain": [0, 1, 2, 3, 10]}


def main():
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in PROBLEMS.items():
            path = os.path.join(tmp, name + ".txt")
            open(path, "w").write(text + "\n")
            for limit in LIMITS[name]:
                p = subprocess.run([REF, "rootlimit", path, str(limit)], capture_output=True, text=True, check=True)
                rec = json.loads(re.search(r"@ROOT (\{.*\})", p.stdout).group(1))
                cases.append({"problem": name, "text": text, "limit": limit, "status": rec["status"], "domains": rec["domains"]})
    json.dump({"source": "oracle/_ref/csolve_ref rootlimit (the reference's own propagate(), src/propagate.c:474-485)",
               "cases": cases}, open(os.path.join(HERE, "root_limit.json"), "w"), indent=1)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
