#!/bin/bash
# kernel timeline of the last training steps of one cfg2 layer (which copies / helper kernels sit between the GEMMs)
# usage (through gpurun, from the repo root): tools/probe/timeline_train.sh <out file under gpurun_out> [measure_configs arm]
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
ARM=${2:-train}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python3 $GRAFT_REPO_ROOT/tools/measure_configs.py $ARM > /tmp/p_tl.out 2> /tmp/p_tl.err < /dev/null
python3 $GRAFT_REPO_ROOT/tools/timeline_prof.py /tmp/p_tl $OUT ${3:-4000} < /dev/null
wc -l $OUT
