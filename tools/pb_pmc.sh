#!/bin/bash
# rocprofv3 PMC passes (separate runs, no trace domains) of one scaler case: tools/pb_pmc.sh <outdir> <case>   e.g. pb16:3840x2160:1706x960:3
O=${1:-gpurun_out/pb_pmc}; C=${2:-pb16:3840x2160:1706x960:3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
B="python tools/bench_one.py --cold $C"
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o $n -- $B > $O/$n.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run c SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS
run d SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH_LEVEL SQ_INSTS_WAVE32_LDS SQ_CYCLES SQ_WAVES_EQ_64 SQ_INST_CYCLES_VMEM SQ_INSTS_SENDMSG
rm -f $O/*/*.db
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + '/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, d in acc.items():
        if 'k_pb' not in k: continue
        print(k)
        for c, v in sorted(d.items()): print('   %-28s %16.0f' % (c, v))
PY
