# usage (on the GPU box): bash tools/prof_graph.sh  -- rocprofv3 passes of the graph kernels: kernel stats, then FETCH_SIZE (separate pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_graph -o s -- python3 $R/tests/manual/graph_profile.py > $R/gpurun_out/stats_graph.log 2>&1
cp $(find $R/gpurun_out/stats_graph -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_graph_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_graph -o p --output-format csv -- python3 $R/tests/manual/graph_profile.py > $R/gpurun_out/pmc_graph.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_graph/**/*counter_collection.csv", recursive=True)
per = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and any(t in r["Kernel_Name"] for t in ("k_search_ef", "k_search_compat", "k_descend")):
        per.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmc_graph_fetch.csv", "w") as o:
    w = csv.writer(o); w.writerow(["Kernel_Name", "dispatches", "FETCH_SIZE_KB_per_dispatch (x2 = bytes on gfx950 for 16-B/lane reads)"])
    for k, v in per.items():
        w.writerow([k, len(v), " ".join(f"{x:.0f}" for x in v)]); print(k, len(v), [round(x / 1e6, 3) for x in v], "GB (raw, x2 for 16-B/lane)")
PY
grep -E "k_search_ef|k_search_compat|k_descend" $R/gpurun_out/stats_graph_kernel_stats.csv | cut -c1-60,200-330
grep -E "^ef128|^compat" $R/gpurun_out/stats_graph.log
