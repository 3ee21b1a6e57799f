"""Registers / scratch / LDS of every kernel of a .hip file as hipcc allocates them for gfx950:  python tools/kernel_resources.py FILE.hip [extra flags]
(the exact translation units are built with -ffp-contract=off: pass it for exact_kernels / ef_search / rerank / exact_scan / graph_build)"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "zvdb_amd", "csrc"),
       "-S", "--cuda-device-only", src, "-o", "/tmp/kres.s"] + sys.argv[2:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open("/tmp/kres.s").read()
names, rows = [], []
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda key: re.search(r"\." + key + r":\s+(\S+)", blk).group(1)
    names.append(g("name"))
    rows.append((int(g("vgpr_count")), int(g("agpr_count") if False else blk.split()[0]), int(g("vgpr_spill_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size")), int(g("sgpr_count"))))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
print("vgpr agpr spill scratch lds_static sgpr  kernel")
for (v, a, sp, sc, lds, sg), n in sorted(zip(rows, dem), key=lambda x: -x[0][0]):
    print(f"{v:4d} {a:4d} {sp:5d} {sc:7d} {lds:10d} {sg:4d}  {n[:170]}")
