"""Synthetic workload generators in the csolve problem-text format.

These produce the shapes BASELINE.json names: N-queens exactly as the reference's
generator writes it (reference scripts/gen_queens.sh:3-37), sudoku-shaped alldiff
networks scaled from 9x9 (reference examples/sudoku.txt:28-59) to 25x25, and
schedule-style optimisation problems after reference examples/schedule.txt.
All generators are deterministic functions of their arguments (own LCG, no `random`).
"""
from __future__ import annotations


class LCG:
    """64-bit LCG (Knuth MMIX constants); identical to the one in oracle/ref_harness.c."""

    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.state = (self.state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self.state >> 33

    def below(self, n: int) -> int:
        return self.next() % n

    def shuffle(self, xs):
        for i in range(len(xs) - 1, 0, -1):
            j = self.below(i + 1)
            xs[i], xs[j] = xs[j], xs[i]
        return xs


def queens(n: int, objective: str = "ANY") -> str:
    """N-queens: three all_different lines plus 1 <= Xi <= N (gen_queens.sh:5-36)."""
    xs = [f"X{i}" for i in range(1, n + 1)]
    lines = [f"# N-queens problem for N={n}", f"{objective};"]
    lines.append("all_different(" + ", ".join(xs) + ");")
    lines.append("all_different(" + ", ".join(f"X{i}+{i}" for i in range(1, n + 1)) + ");")
    lines.append("all_different(" + ", ".join(f"X{i}-{i}" for i in range(1, n + 1)) + ");")
    for i in range(1, n + 1):
        lines.append(f"1 <= X{i}; X{i} <= {n};")
    return "\n".join(lines) + "\n"


def offsets(n: int = 20, values: int = 48, seed: int = 1, objective: str = "ANY") -> str:
    """A small binary != network with irregular shape: n variables of `values` values each, every variable
    with its own lower bound, and for most pairs one or two constraints Vi != Vj + d with seeded offsets
    (neither queens' three-per-pair regularity nor a common window: exercises the general table layout
    of the forbidden-set kernels and windows of more than 32 values on small models)."""
    rng = LCG(seed * 7919 + n * 31 + values)
    lo = [rng.below(9) - 4 for _ in range(n)]
    lines = [f"# offsets network, {n} variables x {values} values, seed {seed}", f"{objective};"]
    for i in range(n):
        for j in range(i + 1, n):
            k = rng.below(4)  # 0: unrelated, 1: one constraint, 2-3: two constraints
            ds = set()
            for _ in range(min(k, 2)):
                ds.add(rng.below(2 * values // 3 + 1) - values // 3 + lo[i] - lo[j])
            for d in sorted(ds):
                lines.append(f"V{i + 1} != V{j + 1} {'+' if d >= 0 else '-'} {abs(d)};")
    for i in range(n):
        lines.append(f"{lo[i]} <= V{i + 1}; V{i + 1} <= {lo[i] + values - 1};")
    return "\n".join(lines) + "\n"


def _cell(r: int, c: int) -> str:
    return f"C{r}_{c}"


def sudoku_solution(box: int, seed: int):
    """A valid box^2 x box^2 grid: the cyclic pattern with seeded row/column/digit shuffles."""
    n = box * box
    rng = LCG(seed)

    def banded():
        bands = rng.shuffle(list(range(box)))
        out = []
        for b in bands:
            out.extend(b * box + r for r in rng.shuffle(list(range(box))))
        return out

    rows, cols = banded(), banded()
    digits = rng.shuffle(list(range(1, n + 1)))
    return [[digits[(box * (r % box) + r // box + c) % n] for c in cols] for r in rows]


def sudoku(box: int = 3, revealed: float = 0.4, seed: int = 1, objective: str = "ANY") -> str:
    """box^2 x box^2 sudoku: givens, then rows / columns / boxes as all_different,
    then 1 <= cell <= n bounds (layout of examples/sudoku.txt, scaled)."""
    n = box * box
    grid = sudoku_solution(box, seed)
    rng = LCG(seed ^ 0x9E3779B97F4A7C15)
    cells = [(r, c) for r in range(n) for c in range(n)]
    rng.shuffle(cells)
    given = sorted(cells[: int(round(revealed * n * n))])
    lines = [f"# sudoku {n}x{n}, {len(given)} givens, seed {seed}", f"{objective};"]
    for r, c in given:
        lines.append(f"{_cell(r, c)} = {grid[r][c]};")
    for r in range(n):
        lines.append("all_different(" + ", ".join(_cell(r, c) for c in range(n)) + ");")
    for c in range(n):
        lines.append("all_different(" + ", ".join(_cell(r, c) for r in range(n)) + ");")
    for br in range(box):
        for bc in range(box):
            lines.append("all_different(" + ", ".join(_cell(br * box + r, bc * box + c)
                                                      for r in range(box) for c in range(box)) + ");")
    for r in range(n):
        lines.append(" ".join(f"1 <= {_cell(r, c)}; {_cell(r, c)} <= {n};" for c in range(n)))
    return "\n".join(lines) + "\n"


def schedule(tasks: int = 16, seed: int = 1, horizon_slack: int = 3) -> str:
    """Single-machine scheduling after examples/schedule.txt: per task release, WCET and
    deadline, pairwise non-overlap disjunctions, MIN end."""
    rng = LCG(seed)
    wcet = [1 + rng.below(4) for _ in range(tasks)]
    total = sum(wcet)
    release = [rng.below(max(1, total // 2)) for _ in range(tasks)]
    lines = [f"# schedule: {tasks} tasks, seed {seed}", "MIN end;"]
    for t in range(tasks):
        dl = total * horizon_slack
        p = f"t{t + 1}"
        lines.append(f"{p}_release = {release[t]};")
        lines.append(f"{p}_release <= {p}_start;")
        lines.append(f"{p}_end = {p}_start + {wcet[t]};")
        lines.append(f"{p}_end <= {p}_release + {dl};")
    for a in range(tasks):
        for b in range(a + 1, tasks):
            pa, pb = f"t{a + 1}", f"t{b + 1}"
            lines.append(f"{pa}_start > {pb}_end | {pb}_start > {pa}_end;")
    for t in range(tasks):
        lines.append(f"end >= t{t + 1}_end;")
    return "\n".join(lines) + "\n"


def linear(n: int = 12, seed: int = 1, objective: str = "ANY") -> str:
    """A seeded mixture of the clause shapes that take the direct bound-propagation paths: `x < y + d`,
    `x <= y + d`, `x = y + d`, two-literal disjunctions of such literals, and a few `!=` (test workload for the
    linear paths of the general kernel and for the clause-resident kernel; no reference example has this mix)."""
    rng = LCG(seed * 104729 + n)
    v = [f"L{i + 1}" for i in range(n)]
    lines = [f"# linear mixture, {n} variables, seed {seed}", f"{objective};"]

    def term(i, d):
        return v[i] if d == 0 else f"{v[i]} {'+' if d > 0 else '-'} {abs(d)}"

    for _ in range(2 * n):
        a, b = rng.below(n), rng.below(n)
        if a == b:
            continue
        d = rng.below(13) - 6
        kind = rng.below(10)
        if kind < 4:
            lines.append(f"{v[a]} < {term(b, d + 8)};")
        elif kind < 6:
            lines.append(f"{v[a]} <= {term(b, d + 6)};")
        elif kind < 7:
            lines.append(f"{v[a]} != {term(b, d)};")
        else:
            c, e = rng.below(n), rng.below(n)
            if c == e:
                continue
            lines.append(f"{v[a]} > {term(b, d)} | {v[c]} > {term(e, rng.below(9) - 4)};")
    # one equality chain so that EQ clauses are present without making the model infeasible
    for i in range(0, n - 1, 5):
        lines.append(f"{v[i + 1]} = {term(i, 1 + rng.below(3))};")
    for i in range(n):
        lo = rng.below(7) - 3
        lines.append(f"{lo} <= {v[i]}; {v[i]} <= {lo + 20 + rng.below(20)};")
    return "\n".join(lines) + "\n"
