# GPU box: kernel 2 on the 25x25 sudoku with at most 8 .. 16 waves per workgroup (CSGPU_K2_WAVES): time against occupancy
for w in 8 10 12 14 15 16; do CSGPU_K2_WAVES=$w timeout -k 10 120 python bench.py --sudoku 5 --instances 262144 --no-search --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves<=$w', round(r['roofline']['kernel_ms']*1000,2),'us frac',round(r['roofline']['frac'],3), '8d', round(r['roofline'].get('survey_8d_node_frac',0),3))"; done
