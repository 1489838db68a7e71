#!/bin/bash
# round-3 evidence run: kernel trace + PMC of the default bench (pinned pixbuf chain), of the blur chain, of the polyphase chain, and a kernel trace of tools/bench_ops.py
cd $GRAFT_REPO_ROOT
C=$(cat gpurun_out/.commit 2>/dev/null || echo unknown)
tools/pmc.sh gpurun_out/pmc_pb
python tools/pmc_summary.py gpurun_out/pmc_pb k_pb_half > gpurun_out/pmc_pb.md
python tools/pmc_traffic.py gpurun_out/pmc_pb $C k_pb_half > gpurun_out/pmc_traffic_pixbuf.json
cp gpurun_out/pmc_pb/trace/t_kernel_stats.csv gpurun_out/pb_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_blur -o t -- python bench.py --no-cpu --blur 1 > gpurun_out/trace_blur.log 2>&1
cp gpurun_out/trace_blur/t_kernel_stats.csv gpurun_out/pb_blur_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_poly -o t -- python bench.py --no-cpu --resize-backend polyphase > gpurun_out/trace_poly.log 2>&1
cp gpurun_out/trace_poly/t_kernel_stats.csv gpurun_out/poly_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_ops -o t -- python tools/bench_ops.py > gpurun_out/bench_ops.log 2>&1
cp gpurun_out/trace_ops/t_kernel_stats.csv gpurun_out/ops_kernel_stats.csv
rm -rf gpurun_out/trace_blur gpurun_out/trace_poly gpurun_out/trace_ops gpurun_out/pmc_pb/*/*.db
