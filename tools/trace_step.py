# print the per-launch timeline of the last step of a rocprofv3 kernel trace (csv): usage trace_step.py <dir>
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].replace("zvdb::", "").replace("(anonymous namespace)::", "")[:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"]) for r in rows]
last = [i for i, k in enumerate(ks) if "k_rerank" in k[0]]
prev_end = None
for k in ks[last[-2] + 1:last[-1] + 1]:
    gap = (k[1] - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(k[2] - k[1]) / 1e3:10.1f} us  (gap {gap:6.1f})  grid {k[3]:>8}  {k[0]}")
    prev_end = k[2]
a, b = ks[last[-2] + 1], ks[last[-1]]
print(f"step span {(b[2] - a[1]) / 1e3:.1f} us")
