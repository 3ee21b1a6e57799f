# usage (on the GPU box): bash tools/prof_graph_pmc.sh <leg>  -- FETCH_SIZE pass (own run, --kernel-trace only) of one leg of graph_profile.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-f32}
ZVDB_PROFILE_ROWS=200000 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_graph_$L -o p --output-format csv -- python3 $R/tests/manual/graph_profile.py $L > $R/gpurun_out/pmc_graph_$L.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_graph_$L/**/*counter_collection.csv", recursive=True)
per = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and any(t in r["Kernel_Name"] for t in ("k_search_ef", "k_search_compat", "k_descend")):
        per.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmc_graph_fetch_$L.csv", "w") as o:
    w = csv.writer(o); w.writerow(["Kernel_Name", "dispatches", "FETCH_SIZE_KB_per_dispatch"])
    for k, v in per.items():
        w.writerow([k, len(v), " ".join(f"{x:.0f}" for x in v)]); print(k, len(v), [round(x * 1024 / 1e9, 3) for x in v], "GB raw")
PY
