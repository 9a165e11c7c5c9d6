#!/bin/bash
# GPU box: the GPU test suite, then the propagation bench in its layouts (no CPU leg). OUT=<dir under gpurun_out>
set -e
mkdir -p gpurun_out/${OUT:-matrix}
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${OUT:-matrix}/gpu_tests.txt; cat gpurun_out/${OUT:-matrix}/gpu_tests.txt
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --no-search > gpurun_out/${OUT:-matrix}/bench_default.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --layout sets > gpurun_out/${OUT:-matrix}/bench_sets.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --rebuild-sets > gpurun_out/${OUT:-matrix}/bench_stateonly.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --queens 128 --instances 131072 > gpurun_out/${OUT:-matrix}/bench_q128.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --queens 128 --instances 131072 --layout sets > gpurun_out/${OUT:-matrix}/bench_q128_sets.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --no-cpu --no-search --queens 128 --instances 131072 --rebuild-sets > gpurun_out/${OUT:-matrix}/bench_q128_stateonly.json 2>/dev/null
for f in gpurun_out/${OUT:-matrix}/bench_*.json; do python - $f <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "nodes/s %.3g"%r["nodes_per_s"], "kernel_ms %.4f"%r["roofline"]["kernel_ms"], r["roofline"]["kernel"], "frac %.3f"%r["roofline"]["frac"], "props/node %.2f"%r["config"]["props_per_node"], "fail %.3f"%r["config"]["inconsistent_fraction"])
PY
done
