#!/bin/bash
# rocprofv3 kernel trace of one cfg2 layer's training step (forward + backward of all parameters, B=16384) for profiles/.
# usage (through gpurun, from the repo root): tools/profile_train.sh <out dir under gpurun_out>
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python3 $GRAFT_REPO_ROOT/tools/measure_configs.py train > $OUT/train_under_rocprof.jsonl 2> /tmp/p_tr.err < /dev/null
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_tr $OUT/train < /dev/null
ls $OUT
