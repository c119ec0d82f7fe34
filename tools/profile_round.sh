#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on a 1-GPU box (run through gpurun from the repo root).
# usage: tools/profile_round.sh <out dir under gpurun_out>
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
python3 $B > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/p_kt.err
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_kt $OUT/bench
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/p_f.err
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_f $OUT/bench_fetch
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/p_w.err
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_w $OUT/bench_write
echo "write done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_s -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/p_s.err
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_s $OUT/bench_sq
echo "sq done"
cd $GRAFT_REPO_ROOT
python3 tools/split_check.py --speed > $OUT/split_check.txt 2>&1
TFEP_DIAG=16 python3 tools/split_phases.py > $OUT/split_phases.txt 2>&1
ls -la $OUT
