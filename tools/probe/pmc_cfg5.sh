#!/bin/bash
# SQ counters of the config-5 edge kernel (one dynamics + JVP evaluation at B = 4096).  usage (through gpurun):
#   tools/probe/pmc_cfg5.sh OUT.txt [LIB]
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/$1
[ -n "$2" ] && export TFEP_HIP_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
C5="$R/tools/measure_cfg5.py --evals-only 1 --batch 4096"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/c5a -- python3 $C5 > /tmp/c5a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/c5b -- python3 $C5 > /tmp/c5b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d /tmp/c5c -- python3 $C5 > /tmp/c5c.log 2>&1
python3 - > $OUT <<PY
import csv, glob, collections
for tag in ('c5a', 'c5b', 'c5c'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob('/tmp/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'egnn_edge_kernel' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in sorted(agg.items()):
        print(tag, k, 'launches', n, 'per launch', round(v / max(n, 1), 1))
PY
tail -2 /tmp/c5a.log >> $OUT; tail -3 /tmp/c5c.log | head -5 >> $OUT
cat $OUT
