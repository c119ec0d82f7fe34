"""HIP-graph replay of a flow for fixed shapes.

Small flows (cfg1: 2 MAF layers on 66 features) and the blocked inverse (thousands of tiny launches per
layer) are launch-bound: the GPU waits for the host between kernels.  ``GraphedFlow`` captures one call of
``flow.forward`` / ``flow.inverse`` on the current device into a HIP graph (``torch.cuda.CUDAGraph`` is
hipGraph on ROCm) and replays it with one host call.  Everything inside a call -- the masked weight re-pack,
the GEMMs, the transformer kernels -- is launched on the capturing stream through the C ABI, so it is all in
the graph and reads the CURRENT parameter values at every replay (optimiser updates are in place).
"""
import torch


def _capturing_flags():
    """Context manager: the range guards of the MAF layers captured inside add their flags to the device counter ``['count']`` of
    the returned dict."""
    import contextlib
    from .nn.flows import autoregressive

    @contextlib.contextmanager
    def ctx():
        flags = {}
        before = autoregressive.capture_flags
        autoregressive.capture_flags = flags
        try:
            yield flags
        finally:
            autoregressive.capture_flags = before
    return ctx()


class GraphedFlow:
    """``g = GraphedFlow(flow, batch_size, n_features); y, log_det_J = g(x)`` (no autograd).

    The range guard of the split-f16 default (``AutoregressiveFlow.split_guard``) cannot choose the arithmetic inside a graph:
    the captured call runs on the split kernels and computes every guarded layer's flag on the device; with ``check_range``
    (default) ``g(x)`` reads the flags after the replay -- one host synchronisation -- and, if one is set, repeats the call
    eagerly, where the guard sends the flagged layers to the exact-fp32 kernels (``last_call_guarded``)."""

    def __init__(self, flow, batch_size, n_features, inverse=False, device=None, warmup=2, check_range=True):
        self.flow = flow
        self.inverse = inverse
        self.check_range = check_range
        self.last_call_guarded = False
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
        with torch.no_grad(), _capturing_flags() as flags, torch.cuda.graph(self.graph):
            self.static_out, self.static_ldj = fn(self.static_in)
        self._flag_total = flags.get('count')
        self.n_guarded_calls = flags.get('calls', 0)

    def __call__(self, x):
        if x.shape != self.static_in.shape:
            raise ValueError(f'GraphedFlow was captured for shape {tuple(self.static_in.shape)}, got {tuple(x.shape)}')
        self.static_in.copy_(x)
        self.graph.replay()
        self.last_call_guarded = False
        if self.check_range and self._flag_total is not None and int(self._flag_total.item()):
            self.last_call_guarded = True
            with torch.no_grad():
                return (self.flow.inverse if self.inverse else self.flow.forward)(self.static_in)
        return self.static_out.clone(), self.static_ldj.clone()


def _held_by_a_live_graph(params):
    """True when the gradient accumulator node of one of ``params`` is owned by somebody else -- a live autograd graph.
    The node exists only while a graph refers to it: mark the one we are handed, drop it, ask again."""
    import gc
    import uuid
    token = uuid.uuid4().hex
    for p in params:
        node = torch.autograd.graph.get_gradient_edge(p).node
        node.metadata['tfep_probe'] = token
        del node
    gc.collect()
    held = False
    for p in params:
        node = torch.autograd.graph.get_gradient_edge(p).node
        if node.metadata.pop('tfep_probe', None) == token:
            held = True
        del node
    return held


class GraphedTrainingStep:
    """One training step -- ``flow(x)``, ``loss_fn(y, log_det_J)``, ``backward()``, ``optimizer.step()`` -- captured into
    a HIP graph for a fixed input shape and replayed with one host call.

    A cfg1-sized step (2 MAF layers on 66 features, batch 1024) is ~100 kernel launches whose host side costs 4 ms while
    the kernels themselves take a fraction of that; replayed as a graph the step runs at kernel speed.  Everything inside
    is launched on the capturing stream through the C ABI or by torch ops, reads the CURRENT parameters (the optimiser
    updates them in place) and writes the gradients into the same ``.grad`` tensors at every replay.

    ``step(x)`` returns the loss of that step (a tensor that the next call overwrites).  ``loss_fn`` must be built from
    capturable ops (no ``.item()`` / host round trips); the optimiser must be capturable (SGD; Adam with
    ``capturable=True``).  No tensor computed from the parameters under grad mode may be alive when the step is constructed
    (checked: RuntimeError).  Activations are not kept across the forward inside a capture (the layer recomputes them), the
    weights are packed on every replay.

    Construction leaves the model and the optimiser as it found them.  The capture needs ``warmup`` real steps first
    (plans, kernel attributes, the optimiser's lazily created state) on ``sample_input`` (default: zeros); the parameters
    and every optimiser state tensor that existed before are then restored IN PLACE (the graph holds their addresses), and
    state the warm-up created (momentum buffers, Adam moments and step counters) is zeroed in place -- the value a fresh
    optimiser starts from.  The first replay is therefore the first step of training.  ``.grad`` of the parameters is
    owned by the graph afterwards (overwritten by every replay)."""

    def __init__(self, flow, loss_fn, optimizer, batch_size, n_features, device=None, warmup=3, sample_input=None):
        self.flow, self.loss_fn, self.optimizer = flow, loss_fn, optimizer
        self._flag_total = None
        for group in optimizer.param_groups:
            # State that the warm-up steps create is zeroed again before the capture; for SGD's momentum buffer that equals a
            # fresh optimiser only when the first step's buffer is the plain gradient: with dampening != 0 a fresh optimiser
            # starts from buf = grad where a zeroed buffer gives (1 - dampening) grad.
            if isinstance(optimizer, torch.optim.SGD) and group.get('momentum', 0) != 0 and group.get('dampening', 0) != 0:
                raise ValueError('GraphedTrainingStep: SGD with momentum and dampening != 0 is not supported (the first replayed '
                                 'step would differ from a fresh optimiser\'s); use dampening = 0')
        self.params = [p for group in optimizer.param_groups for p in group['params'] if p.requires_grad]
        if _held_by_a_live_graph(self.params):
            # The gradient accumulator of a parameter belongs to the stream on which the first live graph over it was built.
            # With such a graph still alive (the output of an earlier flow(x) / flow.inverse(y) under grad mode, kept by the
            # caller) the backward inside the capture synchronises with that stream, and ending such a capture takes the
            # process down inside the HIP runtime.
            raise RuntimeError('GraphedTrainingStep: an autograd graph over the parameters is still alive (a tensor computed '
                               'from the flow outside torch.no_grad()?): delete it before capturing the step')
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.static_in = torch.zeros(batch_size, n_features, dtype=torch.float32, device=device)
        if sample_input is not None:
            self.static_in.copy_(sample_input)
        # what the warm-up steps must not leave behind: parameter values and optimiser state
        saved_params = [p.detach().clone() for p in self.params]
        saved_state = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(p, {}).items()}
                       for p in self.params}
        # Warm-up off the default stream (plans, kernel attributes, optimiser state), as torch's whole-network capture asks
        from .nn.flows import _backward
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        was = _backward.FORCE_RECOMPUTE
        _backward.FORCE_RECOMPUTE = True               # warm up the path the capture takes (no activations kept inside one)
        try:
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._step(self.static_in)
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)      # the capture allocates the gradients in the graph's own pool
            with _capturing_flags() as flags, torch.cuda.graph(self.graph):
                self.static_loss = self._step(self.static_in)
            self._flag_total = flags.get('count')
        finally:
            _backward.FORCE_RECOMPUTE = was
            # (also on failure: a warm-up step that produced NaN must not stay in the weights)
            with torch.no_grad():
                for p, v in zip(self.params, saved_params):
                    p.copy_(v)
                for p in self.params:
                    before = saved_state[id(p)]
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            if k in before:
                                v.copy_(before[k])
                            else:
                                v.zero_()
                        elif k in before:
                            optimizer.state[p][k] = before[k]
        if sample_input is None:
            self.static_in.zero_()

    def _step(self, x):
        y, ldj = self.flow(x)
        loss = self.loss_fn(y, ldj)
        # torch.autograd.grad, not loss.backward(): the parameters' gradient accumulators are then never executed.  They
        # belong to whichever stream first built a graph over the parameters; if such a graph is still alive (the output of
        # an earlier flow(x) under grad mode, kept by the caller), backward() would synchronise the capture with that
        # stream -- and ending such a capture takes the process down inside the HIP runtime.
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        for p, g in zip(self.params, grads):
            p.grad = g
        self.optimizer.step()
        return loss.detach()

    def range_guard_tripped(self):
        """True when, in the LAST replayed step, the input of a layer spanned more feature scales than the split-f16 kernels the
        captured step runs on carry at fp32 accuracy (``AutoregressiveFlow.split_guard``; the eager path sends such a call to the
        exact-fp32 kernels -- a replayed step cannot, its update is already applied).  One host synchronisation; pin
        ``layer.split_gemm = False`` before capturing a step for such data."""
        return self._flag_total is not None and bool(int(self._flag_total.item()))

    def __call__(self, x):
        if x.shape != self.static_in.shape:
            raise ValueError(f'GraphedTrainingStep was captured for shape {tuple(self.static_in.shape)}, got {tuple(x.shape)}')
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_loss
