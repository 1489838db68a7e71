#!/bin/bash
# round 5 evidence run on ONE box: rocprofv3 kernel stats of the default bench (profiled run first, on the fresh box), the plain bench line, PMC passes + traffic json of the
# current build (headline kernel and the blur instantiation), bench lines of the other shapes, the forced-exchange N > 1 host path, the C worker, the op timings with the batch rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r05; mkdir -p $O
C=$(cat tools/_commit 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_first -o t -- python bench.py --no-cpu > $O/bench_line_inside_the_rocprofv3_run.log 2>&1
cp $O/trace_first/t_kernel_stats.csv $O/final_kernel_stats.csv
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep "^{" > $O/bench_driver_shape.json
for a in "--tracks 1" "--tracks 8" "--blur 1" "--blur 1 --tracks 8" "--blur 1 --tracks 1" "--resize-backend polyphase" "--launch-streams 2"; do
  timeout 300 python bench.py --no-cpu $a 2>/dev/null | grep "^{" >> $O/final_bench.jsonl
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_blur -o t -- python bench.py --no-cpu --blur 1 > /dev/null 2>&1
cp $O/trace_blur/t_kernel_stats.csv $O/blur_kernel_stats.csv
LGPU_BENCH_FORCE_EXCHANGE=1 python bench.py --no-cpu 2>/dev/null | grep "^{" > $O/bench_forced_exchange.json
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
for a in "--tracks 1 --exchange 1 --ahead 16" "--tracks 1 --exchange 0 --ahead 16" "--tracks 8 --exchange 1 --steps 600" "--tracks 16 --exchange 1 --ahead 16 --steps 500"; do LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH timeout 300 tools/_worker $a 2>&1 | grep tool >> $O/worker.jsonl; done
timeout 900 tools/pmc.sh gpurun_out/pmc_final_r05 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final_r05 k_pb_half > $O/final_pmc_pixbuf_chain.md
python tools/pmc_traffic.py gpurun_out/pmc_final_r05 $C k_pb_half > $O/pmc_traffic_pixbuf.json
timeout 900 tools/pmc.sh gpurun_out/pmc_final_r05_blur --blur 1 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final_r05_blur k_pb_half > $O/final_pmc_pixbuf_chain_blur.md
bash tools/cold_ops.sh > $O/op_timings.txt 2>/dev/null
mkdir -p gpurun_out/r04/pmc
timeout 900 bash tools/pmc_ops.sh c4rgb24 pb3 > /dev/null 2>&1
cp gpurun_out/r04/pmc_c4rgb24.md gpurun_out/r04/pmc_pb_4k_to_1706x960.md $O/ 2>/dev/null
bash tools/pmc_case.sh gpurun_out/r05/pmc_c4rgba k_gauss5_colorkey python tools/prof_one.py c4rgba > $O/pmc_c4rgba.md 2>&1
rm -rf $O/trace_first $O/trace_blur gpurun_out/pmc_final_r05*/*/*.db gpurun_out/r05/pmc_*/*/*.db gpurun_out/r05/pmc_*/*.db
head -c 2200 $O/bench_default.json; echo; cat $O/pmc_traffic_pixbuf.json; grep "k_pb_half" $O/final_kernel_stats.csv $O/blur_kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
for l in open('gpurun_out/final_r05/final_bench.jsonl'):
    j = json.loads(l); print(j['config'].get('tracks_per_gpu'), j['config']['workload'][-40:], j['config'].get('launch_streams'), j['value'], j['roofline']['launch_us'], j['roofline']['frac'])
PY
