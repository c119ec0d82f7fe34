#!/bin/bash
# same-box A/B of the config-5 edge kernel: alternate libraries, N single dynamics + JVP evaluations each
# usage: tools/probe/cfg5_ab.sh OUT LIB_A LIB_B [...]
out=$1; shift
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)" >> "$out"
    TFEP_HIP_LIB=$lib python tools/measure_cfg5.py --evals-only 7 >> "$out" 2>&1 || exit 1
  done
done
