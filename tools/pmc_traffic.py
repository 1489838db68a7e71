#!/usr/bin/env python3
"""tools/pmc_traffic.py <pmc dir> <commit> [kernel-substring] -- HBM bytes per launch of the dominant kernel from the FETCH_SIZE /
WRITE_SIZE passes of tools/pmc.sh, with the corrections of MI355X_MICROARCH.md (HBM section: on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream; both counters are in KB), written to profiles/pmc_traffic.json for bench.py's roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

d, commit = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "k_half8s"
# only the launch the bench line is about: among the dispatches whose name matches, the (kernel name, grid size) pair with the most dispatches -- the bench also launches the
# same kernel on 8 tracks and its blur instantiation (the 8-GPU denominators), which would drag a plain average down
rows = []
for f in glob.glob(d + "/pmc_*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows.append(((r["Kernel_Name"], r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
main = collections.Counter(k for k, c, _ in rows if c == "FETCH_SIZE").most_common(1)[0][0]
agg = collections.defaultdict(list)
for k, c, v in rows:
    if k == main:
        agg[c].append(v)
fetch = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
write = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
out = {"source": "tools/pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes around `python bench.py --steps 10 --warmup 2 --no-cpu`; "
                 "FETCH_SIZE x 2 (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md), both counters in KB",
       "commit": commit, "kernel": main[0].split("(")[0].replace("void ", ""), "grid": int(main[1]), "tracks": 16, "blur": int(os.environ.get("PMC_BLUR", "0")), "dispatches": len(agg["FETCH_SIZE"]),
       "fetch_size_kb": round(fetch, 1), "write_size_kb": round(write, 1),
       "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024))}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "gpurun_out", "pmc_traffic.json"), "w"))
print(json.dumps(out))
