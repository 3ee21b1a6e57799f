run() { python bench.py --steps 10 --warmup 2 --mode i8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['recall_at_10'], d['roofline']['kernel_ms_avg'])"; }
for dbg in 128 129 256 384 385; do for r in 0 1420; do echo "DEBUG=$dbg ROT=$r"; ZVDB_BF_DEBUG=$dbg ZVDB_BF_ROT=$r run; done; done
