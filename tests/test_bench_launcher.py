"""bench.py's own rank launcher (`python bench.py --gpus N` with no WORLD_SIZE): CPU tests with a stand-in rank
script -- the children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0's single JSON line is relayed, a failing
rank fails the launcher.  The real ranks are exercised on the GPU box (tests/test_gpu_bench.py)."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def _run_launcher(tmp_path, body, gpus=3, env_extra=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.argv = ['bench.py', '--gpus', '{gpus}', '--steps', '1']\n"
            f"import importlib.util\n"
            f"spec = importlib.util.spec_from_file_location('bench_launcher', {os.path.join(ROOT, 'bench.py')!r})\n"
            f"src = open({os.path.join(ROOT, 'bench.py')!r}).read().split('if __name__ == \"__main__\":')[0]\n"
            f"ns = {{'__file__': {os.path.join(ROOT, 'bench.py')!r}, '__name__': 'bench_launcher'}}\n"
            f"exec(compile(src, 'bench.py', 'exec'), ns)\n"
            f"ns['_launch_ranks_if_needed'](sys.argv[1:], script={str(script)!r})\n"
            f"print('NOT LAUNCHED')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_the_launcher_starts_the_ranks_and_relays_rank_zero(tmp_path):
    p = _run_launcher(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert "--gpus" in sys.argv
        print("noise from rank", r) if r else print(json.dumps({"n_gpus": w, "argv": sys.argv[1:]}))
    """)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "NOT LAUNCHED" not in p.stdout and "noise" not in p.stdout
    assert json.loads(lines[0]) == {"n_gpus": 3, "argv": ["--gpus", "3", "--steps", "1"]}


def test_a_failing_rank_fails_the_launcher_and_the_others_are_not_left_waiting(tmp_path):
    p = _run_launcher(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)  # a rank waiting for the dead one in a collective
    """, gpus=2, env_extra={"CSOLVE_BENCH_RANK_GRACE": "1"})
    assert p.returncode != 0 and "NOT LAUNCHED" not in p.stdout


def test_no_launch_under_a_launcher_or_for_one_gpu(tmp_path):
    p = _run_launcher(tmp_path, "raise SystemExit(3)", gpus=2, env_extra={"WORLD_SIZE": "2"})
    assert p.returncode == 0 and "NOT LAUNCHED" in p.stdout
    p = _run_launcher(tmp_path, "raise SystemExit(3)", gpus=1)
    assert p.returncode == 0 and "NOT LAUNCHED" in p.stdout
