#!/bin/bash
# rocprofv3 kernel trace of the blocked inverse (cfg2 one layer at B=8192, cfg4 at B=16384) for profiles/.
# usage (through gpurun, from the repo root): tools/profile_inverse.sh <out dir under gpurun_out>
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_inv -- python3 $GRAFT_REPO_ROOT/tools/measure_configs.py cfg2inv cfg4 > $OUT/inverse_under_rocprof.jsonl 2> /tmp/p_inv.err < /dev/null
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_inv $OUT/inverse < /dev/null
ls $OUT
