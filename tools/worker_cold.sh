#!/bin/bash
# the one-frame step of tools/worker.c on two cached frames and on 32 rotating frames (1.6 GB), exchange on / off
cd $GRAFT_REPO_ROOT
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
for rep in 1 2; do
  for a in "--tracks 1 --exchange 1 --sets 2" "--tracks 1 --exchange 1" "--tracks 1 --exchange 0 --sets 2" "--tracks 1 --exchange 0" "--tracks 1 --exchange 1 --overlap 0" "--tracks 8 --exchange 1 --steps 600" "--tracks 16 --exchange 1 --steps 400"; do
    tools/_worker $a 2>&1 | grep tool
  done
done
