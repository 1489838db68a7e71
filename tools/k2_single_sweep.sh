#!/bin/bash
# K2 on ONE 1080p frame (BASELINE config 2): back-to-back launch interval of the one-column kernel and of the default form, then
# rocprofv3 kernel stats + counters of the single-frame kernel
cd $GRAFT_REPO_ROOT
echo "one-column kernel (LGPU_YUV_S_NC=0): $(LGPU_YUV_S_NC=0 python tools/prof_k2_single.py 2>/dev/null | tail -1)"
echo "default: $(python tools/prof_k2_single.py 2>/dev/null | tail -1)"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_k2s; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/prof_k2_single.py > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_a -o a -- python tools/prof_k2_single.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b -o b -- python tools/prof_k2_single.py > $O/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c -o c -- python tools/prof_k2_single.py > $O/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d -o d -- python tools/prof_k2_single.py > $O/d.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_f -o f -- python tools/prof_k2_single.py > $O/f.log 2>&1
rm -f $O/*/*.db
python tools/pmc_summary.py $O yuv420p
