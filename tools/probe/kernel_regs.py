#!/usr/bin/env python
"""Register / spill / scratch figures of the kernels in a hipcc -save-temps .s file:  kernel_regs.py <file.s> [name substring]"""
import re
import sys

s = open(sys.argv[1]).read()
meta = s[s.index('amdhsa.kernels:'):]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for entry in re.split(r'\n  - ', meta)[1:]:
    name = re.search(r'\.name:\s+(\S+)', entry).group(1)
    if sub not in name:
        continue
    f = {k: (re.search(r'\.%s:\s+(\d+)' % k, entry) or [None, '?'])[1] for k in
         ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size')}
    print(f"{name[:90]:90s} vgpr {f['vgpr_count']:>3} agpr {f['agpr_count']:>3} spill v{f['vgpr_spill_count']} s{f['sgpr_spill_count']} scratch {f['private_segment_fixed_size']}")
