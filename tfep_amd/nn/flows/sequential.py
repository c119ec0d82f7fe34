"""Sequence of flows (reference ``tfep/nn/flows/sequential.py:24-68``)."""
import os

import torch


_side_streams = {}


def _side_stream(device):
    """One extra HIP stream per device for work that overlaps the main stream (weight packing of the next layer)."""
    key = str(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device)
    return _side_streams[key]


def _run_chain(layers, method, x):
    """Apply ``layer.<method>`` along ``layers``; the per-layer log|det J| are summed on the device.

    Forward pass on a HIP device: while layer i computes (matrix-core bound), the masked weight-norm re-pack of layer
    i + 1 (HBM bound, independent of x) runs on a side stream.
    """
    total = None
    overlap = (method == 'forward' and x.is_cuda and len(layers) > 1
               and os.environ.get('TFEP_OVERLAP_PACK', '1') != '0')
    for i, layer in enumerate(layers):
        if overlap and i + 1 < len(layers) and hasattr(layers[i + 1], 'prepack_async'):
            layers[i + 1].prepack_async(x.device, _side_stream(x.device), x.shape[0])
        x, log_det_J = getattr(layer, method)(x)
        total = log_det_J if total is None else total + log_det_J
    if total is None:                                   # no layers: the identity map
        total = torch.full((x.shape[0],), 0.0, dtype=x.dtype, device=x.device)   # (a kernel, not a memset: ops.zeros)
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
