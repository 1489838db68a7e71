#!/bin/bash
# tools/pmc_case.sh <outdir> <kernel substring> <command...> -- kernel trace + the SQ / LDS / memory counter passes (separate runs) of one command
O=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- "$@" > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_a -o a -- "$@" > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b -o b -- "$@" > $O/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c -o c -- "$@" > $O/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d -o d -- "$@" > $O/d.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_f -o f -- "$@" > $O/f.log 2>&1
rm -f $O/*/*.db
python tools/pmc_summary.py $O "$K"
