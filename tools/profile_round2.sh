#!/bin/bash
# Round-2 rocprofv3 evidence for profiles/ on a 1-GPU box (run through gpurun from the repo root).
# usage: tools/profile_round2.sh <out dir under gpurun_out>
# Counters are collected in their own passes, with no trace domains beside them (gpurun refuses the combination).
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
S="$GRAFT_REPO_ROOT/tools/summarize_prof.py"
C5="$GRAFT_REPO_ROOT/tools/measure_cfg5.py"
python3 $B --steps 5 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extra-arms > $OUT/bench_under_rocprof.json 2> /tmp/p_kt.err
python3 $S /tmp/p_kt $OUT/bench
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_f.err
python3 $S /tmp/p_f $OUT/bench_fetch
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_w.err
python3 $S /tmp/p_w $OUT/bench_write
echo "write done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_s -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_s.err
python3 $S /tmp/p_s $OUT/bench_sq
echo "sq done"
# config 5: two dynamics + JVP evaluations at the full batch
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $C5 --evals-only 2 > $OUT/cfg5_under_rocprof.json 2> /tmp/p_c5.err
python3 $S /tmp/p_c5 $OUT/cfg5
echo "cfg5 trace done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/p_c5s -- python3 $C5 --evals-only 1 --batch 4096 > /dev/null 2> /tmp/p_c5s.err
python3 $S /tmp/p_c5s $OUT/cfg5_sq
echo "cfg5 sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_c5f -- python3 $C5 --evals-only 1 --batch 4096 > /dev/null 2> /tmp/p_c5f.err
python3 $S /tmp/p_c5f $OUT/cfg5_fetch
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_c5w -- python3 $C5 --evals-only 1 --batch 4096 > /dev/null 2> /tmp/p_c5w.err
python3 $S /tmp/p_c5w $OUT/cfg5_write
echo "cfg5 traffic done"
ls -la $OUT
