#!/bin/bash
# tools/fetch_calib.sh <outdir> -- FETCH_SIZE (and the TCC request counters, separate pass) of tools/_fetch_calib's known-byte kernels -> a markdown table
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/fetch_calib}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/a -o a -- tools/_fetch_calib 3 > $O/a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/b -o b -- tools/_fetch_calib 3 > $O/b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- tools/_fetch_calib 3 > $O/t.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
known = json.loads(open(O + "/a.log").read().split("\n")[0]) if open(O + "/a.log").read().startswith("{") else None
if known is None:
    for l in open(O + "/a.log"):
        if l.startswith("{"):
            known = json.loads(l); break
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/[ab]/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(O + "/t/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0]] = float(r["AverageNs"])
print("| kernel | bytes read (known) | FETCH_SIZE (KB, raw) | FETCH_SIZE x 1024 / known | TCC_EA0_RDREQ | of them 32 B | RDREQ bytes (64 B, 32 B counted as 32) / known | TCC hit / miss | avg us | GB/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k in ("stream16", "seg640", "seg640x4", "stream4"):
    kb = known["seg_bytes_read"] if k.startswith("seg") else known["buffer_bytes"]
    c = {n: sum(v) / len(v) for n, v in agg.get(k, {}).items()}
    fs, rq, rq32 = c.get("FETCH_SIZE", 0), c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0)
    us = dur.get(k, 0) / 1e3
    print("| %s | %d | %.0f | %.3f | %.0f | %.0f | %.3f | %.0f / %.0f | %.1f | %.0f |" % (k, kb, fs, fs * 1024 / kb, rq, rq32, ((rq - rq32) * 64 + rq32 * 32) / kb, c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0), us, kb / us / 1e3 if us else 0))
PY
rm -rf $O/a/*/*.db $O/b/*/*.db $O/t/*/*.db
