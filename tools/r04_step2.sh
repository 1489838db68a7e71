#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
timeout 600 python -m pytest tests/test_stepper.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|^E " | head
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
rm -f gpurun_out/r04/worker2.jsonl
for rep in 1 2; do
for a in "--exchange 1 --ahead 16 --overlap 0" "--exchange 1 --ahead 16 --overlap 1" "--exchange 0 --ahead 16 --overlap 1" "--exchange 1 --ahead 1 --overlap 1"; do
  tools/_worker --tracks 1 --steps 6000 $a 2>&1 | grep tool >> gpurun_out/r04/worker2.jsonl
done
done
tools/_worker --tracks 8 --steps 1000 --exchange 1 --ahead 16 --overlap 0 2>&1 | grep tool >> gpurun_out/r04/worker2.jsonl
tools/_worker --tracks 8 --steps 1000 --exchange 1 --ahead 16 --overlap 1 2>&1 | grep tool >> gpurun_out/r04/worker2.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r04/worker2.jsonl'):
    j=json.loads(l); print(j['tracks_per_step'], j['exchange'][:14], 'ahead', j['blocks_per_exchange'], 'streams', j['launch_streams'], 'us/step', j['us_per_step'], 'host', j['host_us_per_step_idle_queue'])
PY
