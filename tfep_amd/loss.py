"""TFEP loss (reference ``tfep/loss.py:26-140``) on the fused HIP reduction.

``BoltzmannKLDivLoss`` keeps the reference constructor / call signature.  The whole-batch
``mean`` / ``softmax``-weighted sum is computed from the sufficient statistics of
``tfep_tfep_reduce``; when ``torch.distributed`` is initialised (one process per GPU, RCCL),
``process_group`` all-reduces those few scalars so every rank returns the loss of the GLOBAL
batch.
"""
from typing import Optional

import torch

from . import ops
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
        stats = ops.tfep_reduce(target_potentials, log_det_J, ref_potentials, log_weights, None,
                                kT=1.0, ignore_nan=self.ignore_nan)
        if self.distributed:
            stats = allreduce_stats(stats, self.process_group)
        if log_weights is not None:
            loss = stats[4] / stats[3]                 # sum softmax(log_w) * r
        else:
            loss = stats[1] / stats[0]                 # (nan)mean(r)
        return loss.to(target_potentials.dtype)
