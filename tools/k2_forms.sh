#!/bin/bash
# K2 launch shapes of k_yuv420p_to_rgb_s against the launch size (cell width, workgroup size, resident groups per CU), rocprofv3 kernel time.
# (profiles/r03/k2_forms.txt was made by an earlier version of this script, when the 16-copy-table and four-column kernels still existed.)
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/k2forms; mkdir -p $O
run() { # label nt w h env...
  local label=$1 nt=$2 w=$3 h=$4; shift 4
  rm -rf $O/t
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python tools/k2_batch.py 200 $nt 1 $w $h > $O/t.log 2>&1
  python - "$label" $O/t $nt $w $h <<'P'
import csv,glob,sys
label,d,nt,w,h=sys.argv[1],sys.argv[2],int(sys.argv[3]),int(sys.argv[4]),int(sys.argv[5])
f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
ab=nt*(w*h*3//2+w*h*4)
for r in csv.DictReader(open(f[0])):
    if 'yuv420p' in r['Name']:
        ns=float(r['AverageNs'])
        print("%d x %dx%d %-10s %-34s avg %8.0f ns min %8s  frac %.3f"%(nt,w,h,label, r['Name'][:34], ns, r['MinNs'], ab/ns/8000))
P
}
for geo in "1 1920 1080" "4 1920 1080" "16 1920 1080" "1 3840 2160"; do
  for wgs in 4 8 16 100000; do
    for blk in 256 512 1024; do
      run "w$wgs b$blk" $geo LGPU_YUV_S_WGS=$wgs LGPU_YUV_S_BLOCK=$blk
    done
  done
  run "nc1 w8 b512" $geo LGPU_YUV_S_WGS=8 LGPU_YUV_S_BLOCK=512 LGPU_YUV_S_NC=1
  run "nc4 w8 b512" $geo LGPU_YUV_S_WGS=8 LGPU_YUV_S_BLOCK=512 LGPU_YUV_S_NC=4
done
