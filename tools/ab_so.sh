#!/bin/bash
# interleaved A / B of two builds of the library on the bench launch: tools/ab_so.sh <a.so> <b.so> [bench args]   (a box drifts within a minute: sequential runs mislead)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
one() { LGPU_SO=$1 python bench.py --no-cpu --steps 400 --warmup 100 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  echo "rep $rep: 16 tracks A $(one $A "$@") B $(one $B "$@") A $(one $A "$@") B $(one $B "$@") | 1 track A $(one $A --tracks 1 "$@") B $(one $B --tracks 1 "$@") | 8 tracks A $(one $A --tracks 8 "$@") B $(one $B --tracks 8 "$@")"
done
