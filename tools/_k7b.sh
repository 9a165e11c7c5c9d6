for c in 16 8 4 2; do
  echo "csz $c"
  CSGPU_SHAVE_CSZ=$c python bench.py --steps 50 --warmup 5 --no-cpu --kernel 7 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('  q64 kernel_ms %.4f'%r['roofline']['kernel_ms'])"
  CSGPU_SHAVE_CSZ=$c python bench.py --steps 50 --warmup 5 --no-cpu --kernel 7 --queens 128 --instances 131072 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('  q128 kernel_ms %.4f'%r['roofline']['kernel_ms'])"
done
