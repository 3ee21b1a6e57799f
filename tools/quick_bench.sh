# usage: bash tools/quick_bench.sh [modes...]   -- prints QPS, recall, scan-kernel ms per step for each screen
for m in "$@"; do python bench.py --steps 10 --warmup 2 --mode $m --no-cpu-baseline --no-paths 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value']), d['recall_at_10'], round(d['roofline']['kernel_ms_avg'],3))"; done
