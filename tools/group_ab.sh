#!/bin/bash
# k_pb_half, round-robin order: neighbouring bands per XCD turn (PBH_GROUP) -- time per 16-track launch and L2 -> fabric read bytes (FETCH_SIZE x 2) per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 60 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for g in 5 7 9 13 21 25 26 27 28 29 54; do
  rm -rf /tmp/gp$g
  LGPU_PBH_GROUP=$g rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gp$g -o c -- python bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1
  f=$(python - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob('/tmp/gp$g/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_pb_half<1, 1, 0' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': rows.append((r['Grid_Size'], float(r['Counter_Value'])))
main=collections.Counter(k for k,_ in rows).most_common(1)[0][0]
v=[x for k,x in rows if k==main]
print('%.1f MB' % (2*sum(v)/len(v)*1024/1e6))
PY
)
  echo "group $g: 16 tracks $(LGPU_PBH_GROUP=$g one x) $(LGPU_PBH_GROUP=$g one x) us | 8 tracks $(LGPU_PBH_GROUP=$g one x --tracks 8 --sets 4) | reads $f (algorithmic 663.6 MB)"
done
