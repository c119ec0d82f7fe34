#!/bin/bash
# SQ counters of the block kernel, rows_per_wave 16 vs 64 (usage through gpurun: tools/probe/pmc_inv.sh <cfg2|cfg4i> <batch>)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export INV_CFG=$1 INV_BATCH=$2 TFEP_INV_LOOKAHEAD=0
for rows in 16 64; do
  export TFEP_INV_ROWS_PER_WAVE=$rows
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pa$rows -- python3 $R/tools/probe/inv_cfg.py > /tmp/pa$rows.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pb$rows -- python3 $R/tools/probe/inv_cfg.py > /tmp/pb$rows.log 2>&1
  python3 - <<PY
import csv, glob, collections
for tag in ('pa$rows', 'pb$rows'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob('/tmp/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'inverse_block' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in sorted(agg.items()):
        print('rows=$rows', k, 'launches', n, 'per launch', round(v / max(n, 1), 1))
PY
  tail -1 /tmp/pa$rows.log
done
