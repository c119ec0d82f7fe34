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
    overlap = (method == 'forward' and x.is_cuda and len(layers) > 1
               and os.environ.get('TFEP_OVERLAP_PACK', '1') != '0')
    # The range guard of the split-f16 default (AutoregressiveFlow.split_guard) costs a host synchronisation per layer call; along a
    # sequence the layers run on the split kernels at once and leave their flags on the device, read together at the end: one
    # synchronisation per flow call, and the host keeps launching ahead between the layers.  A flagged layer (rare: features
    # more than 2^19 apart in scale) and everything after it are then repeated with the guard deciding layer by layer.
    from . import autoregressive
    defer = (method == 'forward' and x.is_cuda and len(layers) > 1 and autoregressive.deferred_flags is None
             and not torch.cuda.is_current_stream_capturing() and os.environ.get('TFEP_DEFER_GUARD', '1') != '0')

    def run(first, x, flags):
        terms, inputs = [], []
        for i in range(first, len(layers)):
            layer = layers[i]
            if overlap and i + 1 < len(layers) and hasattr(layers[i + 1], 'prepack_async'):
                layers[i + 1].prepack_async(x.device, _side_stream(x.device), x.shape[0])
            inputs.append(x)
            n_before = len(flags) if flags is not None else 0
            x, log_det_J = getattr(layer, method)(x)
            if flags is not None:
                for k in range(n_before, len(flags)):
                    flags[k] = flags[k] + (i,)
            terms.append(log_det_J)
        return x, terms, inputs

    if defer:
        autoregressive.deferred_flags = flags = []
        try:
            y, terms, inputs = run(0, x, flags)
        finally:
            autoregressive.deferred_flags = None
        if flags:
            counts = torch.cat([c for _, c, _ in flags]).tolist()          # the one synchronisation
            first = None
            for (layer, _, i), c in zip(flags, counts):
                layer.last_split_guard = dict(feature_scales_out_of_range=bool(c), exact=False)
                if c and (first is None or i < first):
                    first = i
            if first is not None:
                y, redo, _ = run(first, inputs[first], None)
                terms = terms[:first] + redo
    else:
        y, terms, _ = run(0, x, None)
    total = None
    for log_det_J in terms:
        total = log_det_J if total is None else total + log_det_J
    if total is None:                                   # no layers: the identity map
        total = torch.full((y.shape[0],), 0.0, dtype=y.dtype, device=y.device)   # (a kernel, not a memset: ops.zeros)
    return y, total


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
