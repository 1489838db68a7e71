cd $GRAFT_REPO_ROOT
for o in 0 1; do echo "LGPU_PB_TILE_ORDER=$o"; for r in 3840x2160:1706x960 1920x1080:1280x720 1280x720:1920x1080 1920x1080:2560x1440; do LGPU_PB_TILE_ORDER=$o python tools/bench_one.py --cold pb:$r:3 pb16:$r:3 2>/dev/null; done; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0 1; do
  LGPU_PB_TILE_ORDER=$o rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pbord$o -o c -- python tools/prof_one.py pb:3840x2160:1706x960:3 > /dev/null 2>&1
  python3 - $o <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/pbord%s/*/*counter_collection.csv" % sys.argv[1])+glob.glob("gpurun_out/pbord%s/*counter_collection.csv" % sys.argv[1]) for r in csv.DictReader(open(f)) if "k_pb_pairs" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("order %s: FETCH_SIZE raw %.0f KB mean over %d dispatches -> x2 = %.1f MB read per frame (source 33.2 MB)" % (sys.argv[1], sum(v)/len(v), len(v), 2*sum(v)/len(v)/1e3*1.024))
PY
done
python -m pytest tests/test_pixbuf_scale.py tests/test_opaque_chain.py -m gpu -q 2>&1 | tail -2
