#!/bin/bash
# tools/pmc_k2.sh <outdir> -- rocprofv3 kernel trace + PMC passes (separate runs) of the K2 batch workload (tools/k2_batch.py)
O=${1:-gpurun_out/pmc_k2}
B="python tools/k2_batch.py 20"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_a -o a -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b -o b -- $B > $O/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c -o c -- $B > $O/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d -o d -- $B > $O/d.log 2>&1
rm -f $O/*/*.db
tail -2 $O/d.log
