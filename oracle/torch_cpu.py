"""Oracle (torch, CPU): one MAF layer forward (MADE + plain RQ neural spline) on all host cores.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  This is the timed ``cpu_baseline`` leg of ``bench.py``
(all host cores, fp32, no autograd, one MAF layer at a time on a 1024-sample chunk) and is held
to the goldens by ``tests/test_oracle_golden.py``.  It restates the same algorithm as ``oracle/made.py`` +
``oracle/transformers.py`` with torch CPU operators (MKL GEMMs, intra-op thread pool) instead of numpy, the way the
reference itself runs on a CPU:

  * effective weight ``mask * g * v / ||v||_row`` recomputed per call (reference masked.py:369-371, 433-439; the second
    ``weight * mask`` of masked.py:270 is the identity on it), ``F.linear`` + ELU (made.py:320-326);
  * spline parameters (spline.py:351-415), knots by cumulative sums, bin = #{knots < x} - 1 over the knots extended by
    the two far sentinels of the linear tails (spline.py:567-650), the RQ map and log-derivative (spline.py:485-494,
    546-564).
Plain (non-circular, fixed-bound, learnable boundary slopes) splines only: the configuration BASELINE.json times.
"""
import math

import torch
import torch.nn.functional as F


def effective_weight(layer):
    """``mask o g v / ||v||`` with 0 (not NaN) on masked entries.  Ref: masked.py:369-371, 433-439.  In-place where that
    saves a pass over the matrix (5.6 GB per cfg2 layer: the timed baseline is bound by these passes, not by the GEMM)."""
    mask = layer['mask']
    if 'weight_v' in layer:
        v = layer['weight_v']
        w = v * (layer['weight_g'] / torch.linalg.vector_norm(v, dim=1, keepdim=True))
        w.masked_fill_(mask == 0, 0.0)
    else:
        w = layer['weight'] * mask
    return w


def made_forward(x, layers):
    """MADE.forward.  Ref: conditioners/made.py:320-329, 355; masked.py:265-277.  (The reference multiplies the effective
    weight by the mask a second time inside the masked linear, masked.py:270: the identity on a weight that is already
    exactly zero where the mask is -- bit-identical, not repeated here.)"""
    h = x
    for i, layer in enumerate(layers):
        h = F.linear(h, effective_weight(layer), layer['bias'])
        if i + 1 < len(layers):
            h = F.elu(h)
    return h


def spline_forward(x, parameters, x0, xf, n_bins, min_bin_size=1e-4, min_slope=1e-4):
    """Plain RQ spline, domain == codomain == [x0, xf].  Returns ``(y, log_det_J)``."""
    B, D = x.shape
    K = n_bins
    p = parameters.reshape(B, 3 * K + 1, D)                      # parameter-major (spline.py:351-352)
    span = (xf - x0) - K * min_bin_size
    widths = torch.softmax(p[:, :K], dim=1) * span + min_bin_size
    heights = torch.softmax(p[:, K:2 * K], dim=1) * span + min_bin_size
    slopes = F.softplus(p[:, 2 * K:] + math.log(math.expm1(1.0 - min_slope))) + min_slope

    zero = torch.zeros(B, 1, D, dtype=x.dtype)
    kx = x0 + torch.cat([zero, torch.cumsum(widths, dim=1)], dim=1)      # (B, K+1, D)
    ky = x0 + torch.cat([zero, torch.cumsum(heights, dim=1)], dim=1)
    # linear tails through far sentinel knots on the boundary-slope lines (spline.py:589-607)
    far = 1000.0 * (kx[:, -1:] - kx[:, :1])
    kx = torch.cat([kx[:, :1] - far, kx, kx[:, -1:] + far], dim=1)       # (B, K+3, D)
    ky = torch.cat([ky[:, :1] - far * slopes[:, :1], ky, ky[:, -1:] + far * slopes[:, -1:]], dim=1)
    sl = torch.cat([slopes[:, :1], slopes, slopes[:, -1:]], dim=1)       # (B, K+3, D)

    idx = ((x[:, None, :] > kx).sum(dim=1, keepdim=True) - 1).clamp_(0, K + 1)      # strict >: on-knot -> lower bin
    xk, yk = kx.gather(1, idx)[:, 0], ky.gather(1, idx)[:, 0]
    w = kx.gather(1, idx + 1)[:, 0] - xk
    h = ky.gather(1, idx + 1)[:, 0] - yk
    d0, d1 = sl.gather(1, idx)[:, 0], sl.gather(1, idx + 1)[:, 0]
    s = h / w
    e = (x - xk) / w
    e1 = e * (1.0 - e)
    den = s + (d1 + d0 - 2.0 * s) * e1
    y = yk + h * (s * e * e + d0 * e1) / den
    dydx = s * s * (d1 * e * e + 2.0 * s * e1 + d0 * (1.0 - e) ** 2) / (den * den)
    return y, torch.log(dydx).sum(dim=1)


@torch.no_grad()
def maf_forward(x, made_layers, x0, xf, n_bins):
    """One MAF layer without conditioning features: ``y, log_det_J``.  Ref: flows/autoregressive.py:144-177."""
    return spline_forward(x, made_forward(x, made_layers), x0, xf, n_bins)
