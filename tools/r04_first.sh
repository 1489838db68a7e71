#!/bin/bash
# round 4, first GPU call: the new at-size parity tests of the benched launch, then the baseline of this box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_pixbuf_scale.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04/first_tests.txt
for i in 1 2; do
  timeout 200 python bench.py --no-cpu --steps 600 --warmup 100 2>/dev/null | tail -1 >> gpurun_out/r04/first_bench.jsonl
  timeout 200 python bench.py --no-cpu --tracks 1 --steps 2000 --warmup 200 2>/dev/null | tail -1 >> gpurun_out/r04/first_bench.jsonl
  timeout 200 python bench.py --no-cpu --tracks 8 --steps 1000 --warmup 200 2>/dev/null | tail -1 >> gpurun_out/r04/first_bench.jsonl
done
cat gpurun_out/r04/first_tests.txt
python - <<'PY'
import json
for l in open('gpurun_out/r04/first_bench.jsonl'):
    j = json.loads(l); print(j['config']['tracks_per_gpu'], j['value'], j['roofline']['launch_us'], j['roofline']['frac'])
PY
