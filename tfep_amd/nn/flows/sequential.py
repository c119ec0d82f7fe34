"""Sequence of flows (reference ``tfep/nn/flows/sequential.py:24-68``)."""
import torch


class SequentialFlow(torch.nn.Sequential):
    """Chain normalizing flows; returns the mapped coordinates and the cumulative log|det J|."""

    def n_parameters(self):
        """int: The total number of parameters that can be optimized."""
        return sum(flow.n_parameters() for flow in self)

    def forward(self, x):
        return self._pass(x, inverse=False)

    def inverse(self, y):
        return self._pass(y, inverse=True)

    def _pass(self, x, inverse):
        cumulative_log_det_J = torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        flows = reversed(self) if inverse else self
        name = 'inverse' if inverse else 'forward'
        for flow in flows:
            x, log_det_J = getattr(flow, name)(x)
            cumulative_log_det_J += log_det_J
        return x, cumulative_log_det_J
