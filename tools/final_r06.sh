#!/bin/bash
# round 6 evidence run on ONE box: rocprofv3 kernel stats of the default bench (first, on the fresh box), the plain bench line (with the seam_chain leg), the per-op table
# of the build that ships (every batch form as x1 / x8 / x16 rows, cold; the single-frame entry points; tools/bench_ops.py's full list with the oracle beside it), the
# host-side profile of the seam leg, PMC passes of the headline kernel and the blur instantiation
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r06; mkdir -p $O
C=$(cat tools/_commit 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_first -o t -- python bench.py --no-cpu --no-seam > $O/bench_line_inside_the_rocprofv3_run.log 2>&1
cp $O/trace_first/t_kernel_stats.csv $O/final_kernel_stats.csv
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep "^{" > $O/bench_driver_shape.json
for a in "--tracks 1" "--tracks 8" "--blur 1" "--blur 1 --tracks 8" "--blur 1 --tracks 1"; do
  timeout 300 python bench.py --no-cpu $a 2>/dev/null | grep "^{" >> $O/final_bench.jsonl
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_blur -o t -- python bench.py --no-cpu --no-seam --blur 1 > /dev/null 2>&1
cp $O/trace_blur/t_kernel_stats.csv $O/blur_kernel_stats.csv
python tools/seam_profile.py 2>&1 | grep -v amdgpu.ids > $O/seam_host_profile.txt
bash tools/ops_table_r06.sh > $O/ops_roofline.md 2>$O/ops_roofline.err
timeout 600 python tools/bench_ops.py --json $O/ops_single_with_oracle.json > $O/ops_single_with_oracle.md 2>/dev/null
timeout 900 tools/pmc.sh gpurun_out/pmc_final_r06 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final_r06 k_pb_half > $O/final_pmc_pixbuf_chain.md
python tools/pmc_traffic.py gpurun_out/pmc_final_r06 $C k_pb_half > $O/pmc_traffic_pixbuf.json
timeout 900 tools/pmc.sh gpurun_out/pmc_final_r06_blur --blur 1 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_final_r06_blur k_pb_half > $O/final_pmc_pixbuf_chain_blur.md
rm -rf $O/trace_first $O/trace_blur gpurun_out/pmc_final_r06*/*/*.db
head -c 3000 $O/bench_default.json; echo; cat $O/pmc_traffic_pixbuf.json; grep "k_pb_half" $O/final_kernel_stats.csv $O/blur_kernel_stats.csv | cut -c1-160
