#!/bin/bash
# the round's closing evidence run on ONE box: bench lines of every shape, rocprofv3 kernel stats + PMC of the default bench, traffic json, kernel traces of the blur
# chain / the polyphase chain / tools/bench_ops.py, the op table, the resize tables, the HBM streams, the C worker
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
C=$(cat tools/_commit 2>/dev/null || echo unknown)
# first, on the fresh box: the profiled run (its own bench line sits in trace.log) and right behind it the plain default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_first -o t -- python bench.py --no-cpu > $O/bench_line_inside_the_rocprofv3_run.log 2>&1
cp $O/trace_first/t_kernel_stats.csv $O/final_kernel_stats.csv
python bench.py > $O/bench_default.json 2>$O/bench_default.err
tools/_hbm_stream > $O/hbm_calibration.txt 2>&1
for a in "--tracks 1" "--blur 1" "--blur 1 --tracks 1" "--resize-backend polyphase" "--resize-backend polyphase --tracks 1" "--resize-backend polyphase --blur 1" "--resize-backend polyphase --blur 1 --tracks 1" "--l2-translucent 0"; do
  python bench.py --no-cpu $a 2>/dev/null | grep "^{" >> $O/final_bench.jsonl
done
LGPU_BENCH_FORCE_EXCHANGE=1 python bench.py --no-cpu 2>/dev/null | grep "^{" > $O/bench_forced_exchange.json
for a in "--tracks 1 --exchange 1 --pixbuf 1" "--tracks 1 --exchange 0 --pixbuf 1" "--tracks 16 --exchange 1 --pixbuf 1 --steps 500" "--tracks 1 --exchange 1 --pixbuf 0"; do tools/_worker $a 2>&1 | grep tool >> $O/worker.jsonl; done
tools/pmc.sh gpurun_out/pmc_final > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final k_pb_half > $O/final_pmc_pixbuf_chain.md
python tools/pmc_traffic.py gpurun_out/pmc_final $C k_pb_half > $O/pmc_traffic_pixbuf.json
cp gpurun_out/pmc_final/trace/t_kernel_stats.csv $O/kernel_stats_late_in_the_call.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_blur -o t -- python bench.py --no-cpu --blur 1 > /dev/null 2>&1
cp gpurun_out/trace_blur/t_kernel_stats.csv $O/kernel_stats_blur_chain.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_poly -o t -- python bench.py --no-cpu --resize-backend polyphase > /dev/null 2>&1
cp gpurun_out/trace_poly/t_kernel_stats.csv $O/kernel_stats_polyphase_chain.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_ops -o t -- python tools/bench_ops.py --no-cpu > /dev/null 2>&1
cp gpurun_out/trace_ops/t_kernel_stats.csv $O/kernel_stats_bench_ops.csv
python tools/bench_ops.py > $O/ops_roofline.md 2>/dev/null
python tools/bench_resize.py --pixbuf 2>/dev/null > $O/pixbuf_ratios.jsonl
python tools/bench_resize.py 2>/dev/null > $O/polyphase_ratios.jsonl
rm -rf $O/trace_first gpurun_out/trace_blur gpurun_out/trace_poly gpurun_out/trace_ops gpurun_out/pmc_final/*/*.db
