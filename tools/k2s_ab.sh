#!/bin/bash
# K2 on ONE 1080p frame: the one-copy form k_yuv420p_to_rgb_s (cells of NC chroma columns, workgroup size) against the one-cell-per-lane kernel, alternated in one call
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/k2s; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  rm -rf $O/t
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python tools/prof_k2_single.py > $O/t.log 2>&1
  python - "$label" $O/t <<'P'
import csv,glob,sys
label,d=sys.argv[1],sys.argv[2]
f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'yuv420p' in r['Name']: print("%-28s %-40s calls %s avg %.0f ns min %s max %s"%(label, r['Name'][:40], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
P
}
for rep in 1 2; do
run "classic" LGPU_YUV_S_NC=0
run "s nc1 b256" LGPU_YUV_S_NC=1 LGPU_YUV_S_BLOCK=256
run "s nc1 b512" LGPU_YUV_S_NC=1 LGPU_YUV_S_BLOCK=512
run "s nc1 b1024" LGPU_YUV_S_NC=1 LGPU_YUV_S_BLOCK=1024
run "s nc2 b256" LGPU_YUV_S_NC=2 LGPU_YUV_S_BLOCK=256
run "s nc2 b512" LGPU_YUV_S_NC=2 LGPU_YUV_S_BLOCK=512
run "s nc2 b1024" LGPU_YUV_S_NC=2 LGPU_YUV_S_BLOCK=1024
done
