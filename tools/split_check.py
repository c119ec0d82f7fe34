#!/usr/bin/env python
"""GPU check of the split-f16 GEMM: conversion round trip, accuracy against float64, and speed against the
exact-fp32 MFMA kernel and the device's matrix-pipe ceiling for the same instruction mix."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from tfep_amd import _lib, ops  # noqa: E402


def unsplit(s, inv, cols):
    """Rebuild fp32 values from split rows (host side, for the check only)."""
    R = s.shape[0]
    raw = s.view(torch.float16).reshape(R, -1, 2, 8).float()          # (R, groups, hi/lo, 8)
    v = (raw[:, :, 0] + raw[:, :, 1]).reshape(R, -1)[:, :cols]
    return v * (inv.reshape(-1, 1) if inv.numel() == R else inv[0])


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    torch.manual_seed(0)
    dev = 'cuda'
    tm, tn, tk = ops.tile_sizes()
    # ---- conversion round trip
    x = torch.randn(300, 1000, device=dev) * torch.logspace(-3, 3, 300, device=dev)[:, None]
    xs, inv = ops.split_rows(x, ops.round_up(1000, tk))
    err = ((unsplit(xs, inv, 1000).double() - x.double()).abs() / x.abs().amax(1, keepdim=True).double()).max().item()
    print(f'split_rows per-row: max |err| / rowmax = {err:.2e}')
    ws, winv = ops.split_rows(x, ops.round_up(1000, tk), per_tensor=True)
    err = ((unsplit(ws, winv, 1000).double() - x.double()).abs().max() / x.abs().max()).item()
    print(f'split_rows per-tensor: max |err| / max = {err:.2e}')

    # ---- accuracy
    for (B, K, N) in [(1000, 1000, 700), (513, 4097, 300)]:
        kp, npad = ops.round_up(K, tk), ops.round_up(N, tk)
        a = torch.randn(B, K, device=dev)
        a = torch.where(a > 0, a, torch.expm1(a))                       # ELU-like activations
        w = torch.randn(N, K, device=dev) / K ** 0.5
        bias = torch.randn(npad, device=dev)
        ap = ops.pad_columns(a, kp)
        wp = ops.masked_weight_prepare(w, None, None, n_rows_padded=npad, k_padded=kp)
        ref = a.double() @ w.double().T + bias[:N].double()
        y32 = ops.masked_linear_packed(ap, wp, bias, N)
        as_, ainv = ops.split_rows(ap, kp)
        ws_, winv = ops.split_rows(wp, kp, per_tensor=True)
        ysp = ops.masked_linear_split(as_, ainv, ws_, winv, bias, N)
        scale = (a.double().abs() @ w.double().abs().T)
        e32 = ((y32.double() - ref).abs() / scale).max().item()
        esp = ((ysp.double() - ref).abs() / scale).max().item()
        r32 = ((y32.double() - ref).norm() / ref.norm()).item()
        rsp = ((ysp.double() - ref).norm() / ref.norm()).item()
        print(f'B={B} K={K} N={N}: max err/sum|a w|: fp32-mfma {e32:.2e} split {esp:.2e}; rel L2: {r32:.2e} / {rsp:.2e}')
        yelu = ops.masked_linear_split(as_, ainv, ws_, winv, bias, N, act=1)
        refe = torch.where(ref > 0, ref, torch.expm1(ref))
        print(f'   ELU epilogue rel L2 {((yelu.double() - refe).norm() / refe.norm()).item():.2e}')

    # ---- speed (dense, hidden-layer sized)
    if '--speed' in sys.argv:
        B, K, N = 65536, 15008, 15104
        kp, npad = ops.round_up(K, tk), ops.round_up(N, tk)
        a = torch.randn(B, kp, device=dev)
        wp = torch.randn(npad, kp, device=dev) / K ** 0.5
        bias = torch.zeros(npad, device=dev)
        out = torch.empty(B, npad, device=dev)
        flops = 2.0 * B * kp * npad
        t32 = timeit(lambda: ops.masked_linear_packed(a, wp, bias, npad, act=1, out=out), 3)
        as_, ainv = ops.split_rows(a, kp)
        ws_, winv = ops.split_rows(wp, kp, per_tensor=True)
        tsp = timeit(lambda: ops.masked_linear_split(as_, ainv, ws_, winv, bias, npad, act=1, out=out), 3)
        tcv = timeit(lambda: ops.split_rows(a, kp, out=as_, inv_scale=ainv), 3)
        print(f'dense {B}x{kp}x{npad}: fp32-mfma {t32 * 1e3:.1f} ms = {flops / t32 / 1e12:.1f} TF; '
              f'split {tsp * 1e3:.1f} ms = {flops / tsp / 1e12:.1f} TF-equiv; split_rows(A) {tcv * 1e3:.2f} ms')
        # ceiling
        blocks, iters = 256 * 4, 400
        scratch = torch.empty(blocks * 256, device=dev)
        f = lambda: _lib.call('tfep_diag_split_mfma_peak', _lib.ptr(scratch), blocks, iters, _lib.stream_of(scratch))
        tp = timeit(f, 3)
        fl = blocks * 4 * iters * 25 * 4 * 3 * (16 * 16 * 32 * 2)
        print(f'split MFMA ceiling: {fl / tp / 1e12:.0f} TF f16 = {fl / tp / 3e12:.0f} TF fp32-equivalent')


if __name__ == '__main__':
    main()
