# sweep of two diagnostic switches of the int8 screen (the library takes its switches from its caller, not from the environment:
# bench.py --tuning NAME=VALUE sets them through zvdb_hip_set_tuning and records them in the line)
run() { python bench.py --steps 10 --warmup 2 --mode i8 --no-cpu-baseline --no-paths --tuning ZVDB_BF_DEBUG=$1 --tuning ZVDB_BF_ROT=$2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); assert d['config']['tuning']; print(d['value'], d['recall_at_10'], d['roofline']['kernel_ms_avg'])"; }
for dbg in 128 129 256 384 385; do for r in 0 1420; do echo "DEBUG=$dbg ROT=$r"; run $dbg $r; done; done
