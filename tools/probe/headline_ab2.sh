#!/bin/bash
# headline A/B of a probe library against the shipped one, alternating on one box: tools/probe/headline_ab2.sh <probe .so> [reps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-arms 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print('shipped ', round(r['ms_per_step'],1), 'ms/step  fused', round(r['roofline']['avg_launch_ms'],2))"
  TFEP_HIP_LIB=$1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-arms 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print('probe   ', round(r['ms_per_step'],1), 'ms/step  fused', round(r['roofline']['avg_launch_ms'],2))"
done
