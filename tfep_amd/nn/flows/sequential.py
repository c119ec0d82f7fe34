"""Sequence of flows (reference ``tfep/nn/flows/sequential.py:24-68``)."""
import torch


def _run_chain(layers, method, x):
    """Apply ``layer.<method>`` along ``layers``; the per-layer log|det J| are summed on the device."""
    total = None
    for layer in layers:
        x, log_det_J = getattr(layer, method)(x)
        total = log_det_J if total is None else total + log_det_J
    if total is None:                                   # no layers: the identity map
        total = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
    return x, total


class SequentialFlow(torch.nn.Sequential):
    """Chain of normalizing flows: ``forward`` runs them in order, ``inverse`` in reverse order with each layer's
    ``inverse``; both return the mapped coordinates and the cumulative log|det J|."""

    def n_parameters(self):
        """int: The total number of parameters that can be optimized."""
        return sum(layer.n_parameters() for layer in self)

    def forward(self, x):
        return _run_chain(list(self), 'forward', x)

    def inverse(self, y):
        return _run_chain(list(self)[::-1], 'inverse', y)
