#!/bin/bash
# kernel timeline of the headline forward (bench.py, cfg2 arm only)
# usage (through gpurun, from the repo root): tools/probe/timeline_bench.sh <out file under gpurun_out> [last_n_events]
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tl
export TFEP_BENCH_ARMS=none
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-arms > /tmp/p_tl.out 2> /tmp/p_tl.err < /dev/null
python3 $GRAFT_REPO_ROOT/tools/timeline_prof.py /tmp/p_tl $OUT ${2:-400} < /dev/null
wc -l $OUT
