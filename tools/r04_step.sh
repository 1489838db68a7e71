#!/bin/bash
# round 4: the stepper with batched exchanges (tools/worker.c), the bench line with its new fields, the forced-exchange N > 1 host path on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
timeout 600 python -m pytest tests/test_stepper.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
for ahead in 1 16 1 16; do
  LGPU_WORKER_PROBE=1 tools/_worker --tracks 1 --steps 4000 --exchange 1 --ahead $ahead 2>&1 | tail -7 >> gpurun_out/r04/worker.jsonl
done
tools/_worker --tracks 1 --steps 4000 --exchange 0 --ahead 16 2>&1 | tail -1 >> gpurun_out/r04/worker.jsonl
tools/_worker --tracks 1 --steps 4000 --exchange 0 --ahead 1 2>&1 | tail -1 >> gpurun_out/r04/worker.jsonl
tools/_worker --tracks 16 --steps 500 --exchange 1 --ahead 16 2>&1 | tail -1 >> gpurun_out/r04/worker.jsonl
cat gpurun_out/r04/worker.jsonl
timeout 300 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r04/bench_line.json
LGPU_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r04/bench_forced_exchange.json
cat gpurun_out/r04/bench_line.json gpurun_out/r04/bench_forced_exchange.json
