#!/bin/bash
# same-box A/B of the workgroup -> tile walk of the fused kernel (TFEP_BLOCK_MAP=1: row super-tiles fastest, the shipped
# default; 2: column super-tiles fastest, the activation panel Infinity-Cache resident).  usage: map_mode_ab.sh OUT
out=$1; : > "$out"
for round in 1 2; do
  for mode in 1 2; do
    echo "== TFEP_BLOCK_MAP=$mode (round $round)" >> "$out"
    TFEP_BLOCK_MAP=$mode python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-arms >> "$out" 2>&1 || exit 1
  done
done
