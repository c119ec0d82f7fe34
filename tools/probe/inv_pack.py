"""Blocked inverse of one cfg2 layer: the 16-row block kernel packed onto fewer CUs (TFEP_INV_WPW waves per workgroup) with
and without look-ahead GEMMs on a side stream (TFEP_INV_LOOKAHEAD).  INV_BATCH (default 8192)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == 'one':
    import time
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tfep_amd import _lib
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    B, D = int(os.environ.get('INV_BATCH', 8192)), 3000
    with torch.device(dev):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    x = torch.randn(B, D, device=dev).clamp_(-4.9, 4.9)
    with torch.no_grad():
        y, _ = flow(x)
        xi, _ = flow.inverse(y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            xi, _ = flow.inverse(y)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    bp = flow[0]._blocked_plan(dev)
    fused = bp['fused']
    lds16 = _lib.load().tfep_inverse_block_lds_bytes_rows(2, fused['cache_len'], fused['max_feats'], 16)
    print(f"WPW={os.environ.get('TFEP_INV_WPW', '-')} LOOK={os.environ.get('TFEP_INV_LOOKAHEAD', '-')} ROWS={os.environ.get('TFEP_INV_ROWS_PER_WAVE', '-')}: "
          f"{ms:.1f} ms, round trip {float((xi - x).abs().max()):.1e}, LDS per 16-row wave {lds16} B", flush=True)
    sys.exit(0)

for rnd in range(2):
    for wpw, look in (('1', '0'), ('2', '0'), ('4', '0'), ('2', '1'), ('4', '1'), ('8', '1'), ('1', '1')):
        env = dict(os.environ, TFEP_INV_WPW=wpw, TFEP_INV_LOOKAHEAD=look, TFEP_INV_ROWS_PER_WAVE='16')
        subprocess.run([sys.executable, os.path.abspath(__file__), 'one'], env=env, check=False)
