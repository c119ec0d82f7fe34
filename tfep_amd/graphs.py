"""HIP-graph replay of a flow for fixed shapes.

Small flows (cfg1: 2 MAF layers on 66 features) and the blocked inverse (thousands of tiny launches per
layer) are launch-bound: the GPU waits for the host between kernels.  ``GraphedFlow`` captures one call of
``flow.forward`` / ``flow.inverse`` on the current device into a HIP graph (``torch.cuda.CUDAGraph`` is
hipGraph on ROCm) and replays it with one host call.  Everything inside a call -- the masked weight re-pack,
the GEMMs, the transformer kernels -- is launched on the capturing stream through the C ABI, so it is all in
the graph and reads the CURRENT parameter values at every replay (optimiser updates are in place).
"""
import torch


class GraphedFlow:
    """``g = GraphedFlow(flow, batch_size, n_features); y, log_det_J = g(x)`` (no autograd)."""

    def __init__(self, flow, batch_size, n_features, inverse=False, device=None, warmup=2):
        self.flow = flow
        self.inverse = inverse
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.static_in = torch.zeros(batch_size, n_features, dtype=torch.float32, device=device)
        fn = flow.inverse if inverse else flow.forward
        # Warm-up off the default stream: builds the execution plans (host-side index work, device->host
        # reads of masks) and sets kernel attributes -- none of which may happen during capture.
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                fn(self.static_in)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out, self.static_ldj = fn(self.static_in)

    def __call__(self, x):
        if x.shape != self.static_in.shape:
            raise ValueError(f'GraphedFlow was captured for shape {tuple(self.static_in.shape)}, got {tuple(x.shape)}')
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out.clone(), self.static_ldj.clone()
