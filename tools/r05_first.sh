#!/bin/bash
# round 5, first GPU call: the baseline of this box (bench lines of the headline and of config 5 with the gaussian), the counters of the blur instantiation of k_pb_half
# (not on record until now), the counters of config 4 in one launch, and the single-frame op timings the review lists
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/first; mkdir -p $O
python bench.py > $O/bench_default.json 2>$O/bench_default.err
for a in "--blur 1" "--blur 1 --tracks 8" "--blur 1 --tracks 1" "--tracks 8" "--tracks 1"; do
  timeout 300 python bench.py --no-cpu $a 2>/dev/null | grep "^{" >> $O/bench_shapes.jsonl
done
tools/pmc.sh gpurun_out/r05/pmc_blur16 --blur 1 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r05/pmc_blur16 k_pb_half > $O/pmc_pb_half_blur.md
tools/pmc.sh gpurun_out/r05/pmc_blur8 --blur 1 --tracks 8 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r05/pmc_blur8 k_pb_half > $O/pmc_pb_half_blur_8tracks.md
mkdir -p gpurun_out/r04/pmc
bash tools/pmc_ops.sh c4rgb24 pb3 pb1 composite
bash tools/pmc_case.sh gpurun_out/r05/pmc_c4rgba k_gauss5_colorkey python tools/prof_one.py c4rgba > $O/pmc_c4rgba.md 2>&1
cp gpurun_out/r04/pmc_c4rgb24.md gpurun_out/r04/pmc_pb_4k_to_1706x960.md gpurun_out/r04/pmc_pb_1080p_to_720p.md gpurun_out/r04/pmc_composite.md $O/ 2>/dev/null
bash tools/cold_ops.sh > $O/op_timings.txt 2>/dev/null
rm -rf gpurun_out/r05/pmc_*/*/*.db gpurun_out/r05/pmc_*/*/*/*.db
head -c 1800 $O/bench_default.json; echo
python - <<'PY'
import json
for l in open('gpurun_out/r05/first/bench_shapes.jsonl'):
    j = json.loads(l); print(j['config'].get('tracks_per_gpu'), j['config'].get('blur'), j['value'], j['roofline']['launch_us'], j['roofline']['frac'])
PY
head -30 $O/op_timings.txt
