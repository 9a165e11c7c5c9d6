mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 50 --warmup 5 > gpurun_out/r2b/bench_new.json 2> gpurun_out/r2b/err.txt; tail -2 gpurun_out/r2b/err.txt
python - <<'PY'
import json
r=json.load(open("gpurun_out/r2b/bench_new.json"))
print("value %.3g"%r["value"], "nodes/s %.3g"%r["nodes_per_s"], r["roofline"]["kernel"], "kernel_ms %.4f frac %.3f layout_frac %.3f"%(r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"]["layout_frac"]))
for k,v in r.get("legs",{}).items(): print("  leg", k, v["kernel"], "kernel_ms %.4f nodes/s %.3g frac_nec %.3f frac_layout %.3f"%(v["kernel_ms"], v["nodes_per_s"], v["frac_of_hbm_peak_on_necessary_bytes"], v["frac_of_hbm_peak_on_layout_bytes"]))
q=r.get("queens128")
if q: print("  queens128 kernel_ms %.4f frac %.3f nodes/s %.3g speedup %.0f"%(q["roofline"]["kernel_ms"], q["roofline"]["frac"], q["nodes_per_s"], q["speedup_over_reference_core"]))
print("  cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["sample"][:80])
PY
for a in "--sudoku 4 --instances 65536"; do
  python bench.py --steps 50 --warmup 5 --no-cpu --kernel 7 $a 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('$a', 'kernel_ms %.4f'%r['roofline']['kernel_ms'], 'frac %.3f'%r['roofline']['frac'], 'nodes/s %.3g'%r['nodes_per_s'])"
done
CSGPU_SHAVE_STATIC=1 python bench.py --steps 50 --warmup 5 --no-cpu --kernel 7 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('static shares: kernel_ms %.4f'%r['roofline']['kernel_ms'])"
CSOLVE_HIP_LIB=$PWD/csolve_amd/libcsolve_hip_tl.so python tools/shave_timeline.py 64 262144 2>&1 | grep -v amdgpu.ids
