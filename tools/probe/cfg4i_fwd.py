#!/usr/bin/env python
"""cfg4-i forward (4-layer MAF + circular RQ-8 + periodic embedding, 512 torsions, B = 131 072): ms per forward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

r = bench.cfg4_i_arm(torch.device('cuda'))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('ms', 'samples_per_s')}, round(r['roofline']['frac'], 4),
      'inverse', round(r['inverse']['ms'], 2), round(r['inverse']['at_8192_rows']['ms'], 2))
