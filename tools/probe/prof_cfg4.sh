#!/bin/bash
# kernel trace of the cfg4 measurements (tools/measure_configs.py cfg4).  usage through gpurun: tools/probe/prof_cfg4.sh OUTPREFIX
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $R/tools/measure_configs.py cfg4 > $R/$1.jsonl 2> /tmp/p_c4.err
python3 $R/tools/summarize_prof.py /tmp/p_c4 $R/$1
head -30 $R/$1_kernel_stats.csv | cut -c1-170
