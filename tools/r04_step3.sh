#!/bin/bash
cd $GRAFT_REPO_ROOT
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
for rep in 1 2; do
for a in "--exchange 1 --ahead 16 --overlap 1" "--exchange 0 --ahead 16 --overlap 1" "--exchange 1 --ahead 16 --overlap 0" "--exchange 0 --ahead 16 --overlap 0"; do
  echo "$a: $(tools/_worker --tracks 1 --steps 6000 $a 2>&1 | grep tool | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['us_per_step'], 'host', j['host_us_per_step_idle_queue'])")"
done
done
