#!/usr/bin/env python3
"""Generate the golden fixtures from the COMPILED REFERENCE (oracle/_ref/csolve_ref).

Authoring container only: needs /root/reference and `make -C oracle ref`.  Writes
  tests/golden/problems/*.txt   problem texts (our generators + the reference's example inputs)
  tests/golden/models/*.model   the reference's post-root trees + clause lists (cs_model files)
  tests/golden/walks/*.walk.gz  node instances from seeded random assignment walks through the
                                reference's propagate_clauses() with conflict learning off
                                (-c false; learnt clauses are outside the hot path and would
                                persist across walks): before, (var, value), status/PROPS, after
  tests/golden/solve_stats.json CALLS/CUTS/PROPS/RESTARTS/solutions of the reference's solve()
The fixtures are data (inputs and expected outputs); no reference source is stored.
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from csolve_amd import problems  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "csolve_ref")
REF_EXAMPLES = "/root/reference/examples"

# name -> (text, walk instances, store model?)
GENERATED = {
    "queens4": (problems.queens(4), 200, True),
    "queens8": (problems.queens(8), 1000, True),
    "queens16": (problems.queens(16), 500, True),
    "queens64": (problems.queens(64), 300, True),
    "queens8_all": (problems.queens(8, "ALL"), 0, False),
    "queens10_all": (problems.queens(10, "ALL"), 0, False),
    "sudoku9_s7": (problems.sudoku(3, 0.4, 7), 300, True),
    "sudoku25_s1": (problems.sudoku(5, 0.4, 1), 150, False),
    "schedule6_s1": (problems.schedule(6, 1), 300, True),
    "schedule12_s1": (problems.schedule(12, 1), 0, False),
}
EXAMPLES = {"ref_sudoku": ("sudoku.txt", 300), "ref_schedule": ("schedule.txt", 200), "ref_wcet": ("wcet.txt", 400)}

DET = ["-c", "false", "-f", "false", "-r", "0"]
SOLVES = [
    ("queens4", DET), ("queens8", []), ("queens8", DET), ("queens16", DET), ("queens16", []),
    ("queens64", []), ("queens8_all", DET), ("queens8_all", []), ("queens10_all", ["-c", "false"]),
    ("sudoku9_s7", []), ("ref_sudoku", []), ("ref_sudoku", DET),
    ("ref_schedule", ["-c", "false"]),
    ("schedule6_s1", ["-c", "false", "-f", "false"]),
    # six minutes of the reference (233,056,571 calls); schedule-14 and -16 of the same generator do not finish
    # within 25 minutes each (with or without conflict learning), so 12 tasks is the largest MIN golden there is
    ("schedule12_s1", ["-c", "false"]),
    ("queens8", ["-c", "false", "-o", "smallest-domain"]), ("queens16", ["-c", "false", "-o", "largest-value", "-r", "0"]),
]


def run(args):
    out = subprocess.run([REF_BIN] + args, capture_output=True, text=True, timeout=1800)
    if out.returncode not in (0, 1):
        raise RuntimeError(f"csolve_ref {args}: rc={out.returncode}\n{out.stderr}")
    return out.stdout


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    for d in ("problems", "models", "walks"):
        os.makedirs(os.path.join(HERE, d), exist_ok=True)
    texts = {}
    for name, (text, walks, model) in GENERATED.items():
        texts[name] = (text, walks, model)
    for name, (fname, walks) in EXAMPLES.items():
        texts[name] = (open(os.path.join(REF_EXAMPLES, fname)).read(), walks, True)

    for name, (text, walks, model) in texts.items():
        path = os.path.join(HERE, "problems", name + ".txt")
        with open(path, "w") as f:
            f.write(text)
        if model:
            print(name, run(["model", path, os.path.join(HERE, "models", name + ".model")]).strip())
        if walks:
            wpath = os.path.join(HERE, "walks", name + ".walk")
            print(name, run(["walk", path, "12345", str(walks), wpath, "-c", "false"]).strip())
            with open(wpath, "rb") as fi, open(wpath + ".gz", "wb") as raw, \
                    gzip.GzipFile(filename="", mode="wb", compresslevel=9, fileobj=raw, mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
            os.remove(wpath)

    stats = []
    for name, flags in SOLVES:
        out = run(["solve", os.path.join(HERE, "problems", name + ".txt")] + flags)
        m = re.search(r"@STATS (\{.*\})", out)
        rec = json.loads(m.group(1))
        sols = re.findall(r"SOLUTION: (.*?)BEST: (-?\d+)", out)
        rec.update(problem=name, flags=flags, n_solution_lines=len(sols))
        if sols:
            rec["last_solution"] = {k.strip(): int(v) for k, v in
                                    (kv.split(" = ") for kv in sols[-1][0].rstrip(", ").split(", "))}
        stats.append(rec)
        print(name, flags, {k: rec[k] for k in ("calls", "cuts", "props", "restarts", "solutions", "best")})
    with open(os.path.join(HERE, "solve_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main()
