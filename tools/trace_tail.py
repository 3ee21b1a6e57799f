# print the last N dispatches of a rocprofv3 kernel trace (csv) in start order with their gaps: usage trace_tail.py <dir> [N]
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
prev = None
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (a - prev) / 1e3 if prev else 0.0
    nm = r["Kernel_Name"].replace("zvdb::", "").replace("(anonymous namespace)::", "")[:70]
    print(f"{(b - a) / 1e3:10.1f} us  (gap {gap:7.1f})  grid {r['Grid_Size_X']:>8}  {nm}")
    prev = b
