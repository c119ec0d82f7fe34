#!/usr/bin/env python
"""Time the plain split-f16 product y = a w^T for a list of shapes "M,N,K[,tile_n[,k_split]]" (dense, no masks): us per launch,
us per 32-wide k-tile of a 256-row tile, TFLOP/s (fp32-equivalent)."""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tfep_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda')
for spec in sys.argv[1:]:
    v = [int(t) for t in spec.split(',')]
    M, N, K = v[:3]
    tile_n = v[3] if len(v) > 3 else 0
    ks = v[4] if len(v) > 4 else 1
    ld = v[5] if len(v) > 5 and v[5] else K          # operands of ld columns, the product over the first K of them (k-ranges)
    a = torch.randn(M, ld, device=dev)
    w = torch.randn(N + 400, ld, device=dev) * 0.01
    a_s, a_inv = ops.split_rows(a, ld)
    w_s, w_inv0 = ops.split_rows(w, ld, per_tensor=True)
    w_inv = torch.zeros(4, device=dev)
    w_inv[:2] = w_inv0
    del a, w
    out = torch.empty((ks, M, N) if ks > 1 else (M, N), device=dev)
    d = _lib.GemmDesc()
    d.x, d.ldx, d.w, d.ldw = a_s.data_ptr(), ld, w_s.data_ptr(), ld
    kr = torch.tensor([0, K] * 64, dtype=torch.int32, device=dev)
    if ld != K:
        d.k_ranges = kr.data_ptr()
    d.y, d.ldy = out.data_ptr(), N
    d.B, d.N, d.n_rows_w, d.k_padded, d.act = M, N, N + 400, ld, 0
    d.split, d.x_inv_scale, d.w_inv_scale = 1, a_inv.data_ptr(), w_inv.data_ptr()
    d.tile_n = tile_n
    if ks > 1:
        d.k_split, d.slab_stride = ks, M * N
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    gap = int(os.environ.get('GAP_CYCLES', 0))      # a one-thread spin kernel before every launch (the GPU "busy" but idle)
    if gap:
        tot = 0.0
        for _ in range(n):
            torch.cuda._sleep(gap)
            e0.record()
            _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        us = tot * 1e3 / n
    else:
        e0.record()
        for _ in range(n):
            _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
    tn = tile_n or 256
    tiles = -(-M // 256) * -(-N // tn) * ks
    rounds = -(-tiles // 256)
    print(f'{spec:>28}: {us:9.1f} us  {tiles} tiles = {tiles / 256:.2f} rounds; per k-tile of a round {us / rounds / (K / ks / 32):.2f} us;'
          f' {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s')
    if int(os.environ.get('TFEP_DIAG', 0)) & 16:
        buf = (ctypes.c_ulonglong * 4)()
        _lib.call('tfep_diag_split_cycles', buf)
        _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), st)
        torch.cuda.synchronize()
        _lib.call('tfep_diag_split_cycles', buf)
        nwg = max(1, buf[2])
        print(f'      per workgroup: k-loop {buf[0] / nwg:.0f} cycles ({buf[0] / nwg / (K / ks / 32):.0f} per k-tile), epilogue {buf[1] / nwg:.0f} cycles,'
              f' lifetime {buf[3] / nwg / 100:.1f} us -> clock {(buf[0] + buf[1]) / max(1, buf[3]) * 100:.0f} MHz')
    del a_s, w_s, out
