"""TFEP loss (reference ``tfep/loss.py:26-140``) on the fused HIP reduction.

``BoltzmannKLDivLoss`` keeps the reference constructor / call signature.  The whole-batch
``mean`` / ``softmax``-weighted sum is computed from the sufficient statistics of
``tfep_tfep_reduce``; when ``torch.distributed`` is initialised (one process per GPU, RCCL),
``process_group`` all-reduces those few scalars so every rank returns the loss of the GLOBAL
batch.
"""
from typing import Optional

import torch

from . import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.tfep.*)
from .distributed import allreduce_stats


class BoltzmannKLDivLoss(torch.nn.Module):
    r"""KL divergence between Boltzmann distributions A and mapped B'.

    ``loss = mean_i(u_B(x_i) - log|det J(x_i)| [- u_A(x_i)])`` or, with ``log_weights``,
    ``sum_i softmax(log_w)_i (...)`` (reference loss.py:125-140).
    """

    def __init__(self, ignore_nan: bool = False, process_group=None, distributed: bool = False):
        super().__init__()
        #: Whether to ignore NaNs when computing the loss or not.
        self.ignore_nan = ignore_nan
        self.process_group = process_group
        self.distributed = distributed or process_group is not None

    def forward(
            self,
            target_potentials: torch.Tensor,
            log_det_J: Optional[torch.Tensor] = None,
            log_weights: Optional[torch.Tensor] = None,
            ref_potentials: Optional[torch.Tensor] = None,
    ) -> torch.Tensor:
        args = (target_potentials, log_det_J, log_weights, ref_potentials)
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args):
            return _LossFunction.apply(self, *args)
        return self._value(*args)[0]

    def _value(self, target_potentials, log_det_J, log_weights, ref_potentials):
        stats = reduce_stats(target_potentials.detach(), _det(log_det_J), _det(ref_potentials), _det(log_weights),
                             None, kT=1.0, ignore_nan=self.ignore_nan)
        if self.distributed:
            stats = allreduce_stats(stats, self.process_group)
        if log_weights is not None:
            loss = stats[4] / stats[3]                 # sum softmax(log_w) * r
        else:
            loss = stats[1] / stats[0]                 # (nan)mean(r)
        return loss.to(target_potentials.dtype), stats


def reduce_stats(target_potentials, log_det_J=None, ref_potentials=None, log_weights=None, bias=None, kT=1.0,
                 ignore_nan=False):
    """The 9 float64 sufficient statistics of the batch: ``torch.ops.tfep.tfep_reduce`` (``tfep_tfep_reduce``)."""
    ops.check_device_tensor(target_potentials, 'target_potentials')
    return torch.ops.tfep.tfep_reduce(target_potentials, log_det_J, ref_potentials, log_weights, bias, float(kT),
                                      bool(ignore_nan))


def _det(t):
    return None if t is None else t.detach()


class _LossFunction(torch.autograd.Function):
    """d loss / d u_B[i] = w_i, d/d log_det_J[i] = d/d u_A[i] = -w_i, d/d log_w[i] = w_i (r_i - loss), with
    w_i = 1/N or softmax(log_w)_i over the GLOBAL batch (NaN samples get zero weight with ignore_nan)."""

    @staticmethod
    def forward(ctx, module, uB, ldj, lw, uA):
        loss, stats = module._value(uB, ldj, lw, uA)
        ctx.module = module
        ctx.stats = stats
        ctx.loss = loss.detach()          # (not the output tensor itself: output -> grad_fn -> ctx -> output is a reference cycle)
        ctx.save_for_backward(uB, ldj, lw, uA)
        return loss

    @staticmethod
    def backward(ctx, g):
        uB, ldj, lw, uA = ctx.saved_tensors
        stats = ctx.stats
        r = uB
        if ldj is not None:
            r = r - ldj
        if uA is not None:
            r = r - uA
        if lw is not None:
            w = torch.exp(lw.double() - stats[2]) / stats[3]
        else:
            w = torch.full_like(r, 1.0, dtype=torch.float64) / stats[0]
        if ctx.module.ignore_nan:
            nan = torch.isnan(r)
            w_r = torch.where(nan, torch.zeros_like(w), w)
        else:
            w_r = w
        gw = (g.double() * w_r).to(uB.dtype)
        g_lw = None
        if lw is not None and ctx.needs_input_grad[3]:
            rr = torch.where(torch.isnan(r), torch.zeros_like(r), r) if ctx.module.ignore_nan else r
            g_lw = (g.double() * w * (rr.double() - ctx.loss.double())).to(lw.dtype)
        return (None, gw, None if ldj is None else -gw, g_lw, None if uA is None else -gw)
