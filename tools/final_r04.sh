#!/bin/bash
# round 4 evidence run on ONE box: rocprofv3 kernel stats of the default bench (profiled run first, on the fresh box), the plain bench line, power / clock samples under
# the launch, PMC passes + traffic json of the current build, bench lines of the other shapes, the forced-exchange N > 1 host path, the C worker, the op timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
C=$(cat tools/_commit 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_first -o t -- python bench.py --no-cpu > $O/bench_line_inside_the_rocprofv3_run.log 2>&1
cp $O/trace_first/t_kernel_stats.csv $O/final_kernel_stats.csv
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --no-cpu --launch-streams 2 2>/dev/null | grep "^{" > $O/bench_two_streams.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep "^{" > $O/bench_driver_shape.json
# power / clocks beside 30 s of back-to-back launches
python bench.py --no-cpu --steps 150000 --warmup 100 > $O/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|power\|Temperature (Sensor junction)\|Temperature (Sensor memory)" | tr '\n' ';' | cut -c1-600 >> $O/power_clock_samples.txt; echo >> $O/power_clock_samples.txt
  sleep 3
done
wait $BP
python -c "import json; j=json.loads(open('$O/clock_probe_bench.json').readlines()[-1]); print('launch_us under the samples', j['roofline']['launch_us'], 'stream probe', j['roofline']['box_class'])" >> $O/power_clock_samples.txt
for a in "--tracks 1" "--tracks 8" "--blur 1" "--blur 1 --tracks 1" "--resize-backend polyphase" "--l2-translucent 0"; do
  python bench.py --no-cpu $a 2>/dev/null | grep "^{" >> $O/final_bench.jsonl
done
LGPU_BENCH_FORCE_EXCHANGE=1 python bench.py --no-cpu 2>/dev/null | grep "^{" > $O/bench_forced_exchange.json
gcc -O2 -o tools/_worker tools/worker.c -Iinclude -Llives_amd -llivesgpu -Wl,-rpath,$PWD/lives_amd
for a in "--tracks 1 --exchange 1 --ahead 16" "--tracks 1 --exchange 1 --ahead 16 --sets 2" "--tracks 1 --exchange 1 --ahead 1" "--tracks 1 --exchange 0 --ahead 16" "--tracks 8 --exchange 1 --steps 600" "--tracks 16 --exchange 1 --ahead 16 --steps 500" "--tracks 16 --exchange 1 --overlap 0 --steps 500"; do LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH tools/_worker $a 2>&1 | grep tool >> $O/worker.jsonl; done
tools/pmc.sh gpurun_out/pmc_final_r04 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final_r04 k_pb_half > $O/final_pmc_pixbuf_chain.md
python tools/pmc_traffic.py gpurun_out/pmc_final_r04 $C k_pb_half > $O/pmc_traffic_pixbuf.json
cp gpurun_out/pmc_final_r04/trace/t_kernel_stats.csv $O/kernel_stats_late_in_the_call.csv
bash tools/cold_ops.sh > $O/op_timings.txt 2>/dev/null
bash tools/edge_ab.sh 2>/dev/null | head -6 > $O/edge_timings.txt
rm -rf $O/trace_first gpurun_out/pmc_final_r04/*/*.db
cat $O/bench_default.json | head -c 1500; echo; cat $O/power_clock_samples.txt | tail -3; cat $O/pmc_traffic_pixbuf.json; head -22 $O/op_timings.txt
