#!/bin/bash
# two launch streams at 8 / 16 tracks per step: does the drain of one launch overlap the ramp-up of the next?  interleaved
cd $GRAFT_REPO_ROOT
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
us() { tools/_worker "$@" 2>&1 | grep tool | python -c "import sys,json; print(json.loads(sys.stdin.read())['us_per_step'])"; }
for rep in 1 2 3; do
  for ex in 0 1; do
    echo "rep $rep exchange $ex: 16 tracks one stream $(us --tracks 16 --steps 500 --exchange $ex --overlap 0) two $(us --tracks 16 --steps 500 --exchange $ex --overlap 1) one $(us --tracks 16 --steps 500 --exchange $ex --overlap 0) two $(us --tracks 16 --steps 500 --exchange $ex --overlap 1) | 8 tracks one $(us --tracks 8 --steps 800 --exchange $ex --overlap 0) two $(us --tracks 8 --steps 800 --exchange $ex --overlap 1)"
  done
done
