#!/bin/bash
# tools/pmc.sh <outdir> -- rocprofv3 kernel trace + PMC passes (separate runs) of the default bench workload
O=${1:-gpurun_out/pmc}; shift
B="python bench.py --steps 10 --warmup 2 --no-cpu --no-seam $*"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
# the kernel trace runs the bench at its default length (1000 timed steps): the average duration is then the steady-state one the bench's HIP events report
# (with 10 steps the 300 wake-up launches, the first ~100 of them at ramping clocks, dominate the average)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --no-cpu --no-seam $* > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_a -o a -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/pmc_b -o b -- $B > $O/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c -o c -- $B > $O/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d -o d -- $B > $O/d.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_e -o e -- $B > $O/e.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_f -o f -- $B > $O/f.log 2>&1
rm -f $O/*/*.db
tail -2 $O/e.log
