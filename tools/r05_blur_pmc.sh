#!/bin/bash
# round 5: SQ counters of the blur instantiation of k_pb_half at 48 / 24 / 16 bands per track (same launch otherwise)
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/blur_bands_pmc; mkdir -p $O
for b in 48 24 16; do
  export LGPU_PBH_TH=$((100000+b))
  B="python bench.py --steps 10 --warmup 2 --no-cpu --blur 1"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/b$b/pmc_a -o a -- $B > $O/a.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/b$b/pmc_c -o c -- $B > $O/c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/b$b/pmc_d -o d -- $B > $O/d.log 2>&1
  mkdir -p $O/b$b/trace; echo "Name,Calls,AverageNs,MinNs,MaxNs,Percentage" > $O/b$b/trace/t_kernel_stats.csv
  python tools/pmc_summary.py $O/b$b k_pb_half > $O/bands_$b.md
  rm -rf $O/b$b
done
unset LGPU_PBH_TH
tail -25 $O/bands_48.md; tail -25 $O/bands_24.md; tail -25 $O/bands_16.md
