#!/bin/bash
# Compile every kernel file to gfx950 assembly and scan it for inline-asm MFMA hazards the compiler cannot see
# (tools/probe/mfma_hazard_scan.py: VALU write -> MFMA operand; tools/probe/mfma_result_hazard_scan.py: MFMA result ->
# early read by a spill store / copy / epilogue instruction).  Run after any change to a kernel with inline-asm MFMAs
# (split_gemm_kernel.h, egnn.hip, inverse_block.hip).  usage: tools/scan_hazards.sh   (from the repo root; ~4 min)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/tfep_scan
mkdir -p $OUT
rc=0
for f in split_gemm split_gemm_layouts egnn inverse_block; do
    extra=""
    [ $f = egnn ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-result $extra --cuda-device-only -S \
        $ROOT/tfep_amd/csrc/$f.hip -o $OUT/$f.s 2>/dev/null || { echo "compile failed: $f"; rc=1; continue; }
    echo "== $f"
    python3 $ROOT/tools/probe/mfma_hazard_scan.py $OUT/$f.s | tail -3 || rc=1
    python3 $ROOT/tools/probe/mfma_result_hazard_scan.py $OUT/$f.s | tail -4 || rc=1
done
exit $rc
