# GPU box: what a short timed region right after the graph capture costs (bench.py time_steps): --steps 20 --warmup 5, the
# driver's command, with 0 / 20 / 100 / 500 ms of untimed replays of the same graph before the timed one, and --steps 100
for cfg in "20 5 0" "20 5 20" "20 5 100" "20 5 500" "100 10 0" "100 10 200" "20 5 0"; do set -- $cfg; CSOLVE_BENCH_UNTIMED_REPLAY_MS=$3 timeout -k 10 200 python bench.py --gpus 1 --steps $1 --warmup $2 --no-search --no-queens128 --no-sudoku25 --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2 prewarm $3 ms:', round(r['ms_per_step'],4), round(r['roofline']['kernel_ms'],4), round(r['roofline']['frac'],3))"; done
