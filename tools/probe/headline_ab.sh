#!/bin/bash
# same-box A/B of the headline (cfg2 forward only) between library builds: usage headline_ab.sh <out file> <lib1> <lib2> ...
# (each library twice, alternating)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
: > $OUT
for round in 1 2; do
for lib in "$@"; do
    echo "== $lib (round $round)" >> $OUT
    TFEP_HIP_LIB=$GRAFT_REPO_ROOT/$lib TFEP_BENCH_ARMS=none timeout -k 10 200 python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'fused_avg_ms': d['roofline'].get('avg_launch_ms'), 'frac': d['roofline']['frac']}))" >> $OUT
done
done
cat $OUT
