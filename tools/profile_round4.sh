#!/bin/bash
# Round-4 rocprofv3 evidence for profiles/ on a 1-GPU box (run through gpurun from the repo root).
# usage: tools/profile_round4.sh <out dir under gpurun_out> [part ...]     parts: bench trace pmc inverse others (default: all)
# Counters are collected in their own passes, with no trace domains beside them (gpurun refuses the combination).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
PARTS=${@:-bench trace pmc inverse others}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
S="$GRAFT_REPO_ROOT/tools/summarize_prof.py"
MC="$GRAFT_REPO_ROOT/tools/measure_configs.py"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python3 $B --steps 5 --warmup 2 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench done rc=$?"
fi
if has trace; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-extra-arms > $OUT/bench_under_rocprof.json 2> /tmp/p_kt.err
  python3 $S /tmp/p_kt $OUT/bench; echo "kernel trace done"
fi
if has pmc; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_f.err
  python3 $S /tmp/p_f $OUT/bench_fetch
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_w.err
  python3 $S /tmp/p_w $OUT/bench_write
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_s -- python3 $B --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_s.err
  python3 $S /tmp/p_s $OUT/bench_sq; echo "pmc done"
fi
if has inverse; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_inv -- python3 $GRAFT_REPO_ROOT/tools/probe/inv_one.py > $OUT/inverse_under_rocprof.txt 2> /tmp/p_inv.err
  python3 $S /tmp/p_inv $OUT/inverse; echo "inverse trace done"
fi
if has others; then
  python3 $MC cfg1 cfg2inv cfg3shard train cfg4 > $OUT/other_configs.jsonl 2> $OUT/other_configs.err; echo "others done rc=$?"
fi
ls -la $OUT
