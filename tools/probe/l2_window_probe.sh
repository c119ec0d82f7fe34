#!/bin/bash
# what would the fused split kernel gain if its operand stream came from L2 instead of beyond it?
# A/B on ONE box: the shipped library against a timing build whose k-loop re-reads a 128-column window of the operand
# panels (L2-resident; results wrong).  Wall time, in-kernel clock (TFEP_DIAG=16) and FETCH_SIZE of both.
# usage (through gpurun): bash tools/probe/l2_window_probe.sh <out dir under gpurun_out>
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m tfep_amd.build --probe /tmp/libprobe128.so -DTFEP_PROBE_KWINDOW=128 > $OUT/build.log 2>&1
echo "probe built"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extra-arms"
for rep in 1 2 3; do
  python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped  rep $rep: ms_per_step %.1f fused launch %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" | tee -a $OUT/ab.txt
  TFEP_HIP_LIB=/tmp/libprobe128.so python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('L2-window rep $rep: ms_per_step %.1f fused launch %.2f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" | tee -a $OUT/ab.txt
done
echo "--- in-kernel counters, shipped" | tee -a $OUT/ab.txt
TFEP_DIAG=16 python3 tools/split_phases.py 2>/dev/null | tee -a $OUT/ab.txt
echo "--- in-kernel counters, L2-window probe" | tee -a $OUT/ab.txt
TFEP_HIP_LIB=/tmp/libprobe128.so TFEP_DIAG=16 python3 tools/split_phases.py 2>/dev/null | tee -a $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
TFEP_HIP_LIB=/tmp/libprobe128.so rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_pf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-arms > /dev/null 2> /tmp/p_pf.err
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py /tmp/p_pf $OUT/probe_fetch
head -3 $OUT/probe_fetch_pmc.csv | cut -c 1-200 | tee -a $OUT/ab.txt
