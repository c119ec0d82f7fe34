"""E(n)-equivariant graph network dynamics with the ``tfep.nn.dynamics.egnn`` API on gfx950 kernels.

Mirrors reference ``tfep/nn/dynamics/egnn.py``: ``EGNNDynamics`` (:28-219) with the same constructor, submodule /
parameter names (``time_embedding._log_gammas``, ``h_embedding.{weight,bias}``, ``graph_layer_{i}.{distance_embedding.
_log_gammas, message_mlp.{0,2}, attention_mlp.0, update_x_mlp.{0,2}, update_h_mlp.{0,2}}``), the ``_node_types_one_hot``
buffer and the identity initialisation (:136-138).  The torch submodules only HOLD the parameters (same construction
order, so the same seed gives the reference's initial values); the arithmetic is ``csrc/egnn.hip``:

  ``forward(t, x)``   velocity ``(batch, 3 n_nodes)``                                   (egnn.py:143-194)
  ``jvp(t, x, v)``    velocity and the directional derivative ``J v`` in the same pass -- what the trace estimators of
                      the continuous flow are built from (``e . (J e)`` is the Hutchinson estimate, continuous.py:307-324)

Per call: the parameters are re-packed (``tfep_egnn_pack_layer``), one embedding kernel, one edge kernel per layer, one
node kernel between layers, one finishing kernel.  No edge list, no scatter_add.  There is no autograd through the
kernels: under grad mode ``forward`` is ``torch_forward``, the same map as differentiable torch operators (what makes a
``ContinuousFlow`` over these dynamics trainable: its loss differentiates a trace of the Jacobian).
"""
import ctypes
import math
import os

import torch

from ... import _lib
from ..embeddings.radial import BehlerParrinelloRadialExpansion, GaussianBasisExpansion
from ..graph import FixedGraph


class _EGLayer(torch.nn.Module):
    """Parameters of one equivariant layer under the reference's names (egnn.py:225-270)."""

    def __init__(self, r_cutoff, node_feat_dim, distance_feat_dim, speed_factor):
        super().__init__()
        self.speed_factor = speed_factor
        self.distance_embedding = BehlerParrinelloRadialExpansion.from_range(
            r_cutoff=r_cutoff, n_gaussians=distance_feat_dim, max_mean=r_cutoff, trainable_stds=True,
            force_zero_after_cutoff=False)
        F = node_feat_dim
        self.message_mlp = torch.nn.Sequential(
            torch.nn.Linear(2 * F + distance_feat_dim, F), torch.nn.SiLU(), torch.nn.Linear(F, F), torch.nn.SiLU())
        self.attention_mlp = torch.nn.Sequential(torch.nn.Linear(F, 1), torch.nn.Sigmoid())
        self.update_x_mlp = torch.nn.Sequential(
            torch.nn.Linear(F, F), torch.nn.SiLU(), torch.nn.Linear(F, 1, bias=False), torch.nn.Tanh())
        self.update_h_mlp = torch.nn.Sequential(torch.nn.Linear(2 * F, F), torch.nn.SiLU(), torch.nn.Linear(F, F))

    def forward(self, *args, **kwargs):
        raise RuntimeError('_EGLayer holds parameters only; EGNNDynamics runs the layers on the HIP kernels')


class EGNNDynamics(FixedGraph):
    """Velocity field of a continuous normalizing flow (Satorras et al. 2021, as varied by tfep).  Arguments as
    reference egnn.py:73-128."""

    def __init__(
        self,
        node_types,
        r_cutoff,
        time_feat_dim=16,
        node_feat_dim=64,
        distance_feat_dim=64,
        n_layers=4,
        speed_factor=1.0,
        initialize_identity=True,
    ):
        super().__init__(node_types=node_types)
        self.time_embedding = GaussianBasisExpansion.from_range(n_gaussians=time_feat_dim, max_mean=1.0,
                                                                trainable_stds=True)
        self.h_embedding = torch.nn.Linear(in_features=len(self._node_types_one_hot[1]) + time_feat_dim,
                                           out_features=node_feat_dim)
        self._n_layers = n_layers
        for layer_idx in range(n_layers):
            eg_layer = _EGLayer(r_cutoff=r_cutoff, node_feat_dim=node_feat_dim, distance_feat_dim=distance_feat_dim,
                                speed_factor=speed_factor)
            if initialize_identity:
                eg_layer.update_x_mlp[-2].weight.data.fill_(0.0)
            self.add_module('graph_layer_' + str(layer_idx), eg_layer)
        self._dims = (int(time_feat_dim), int(node_feat_dim), int(distance_feat_dim))
        #: The three per-edge F x F products on split-f16 operands (fp32-equivalent, ~3x the fp32-MFMA rate); False: exact
        #: fp32 MFMA; None: the ``TFEP_EGNN_SPLIT`` environment switch (default on).
        self.split_gemm = None

    # ------------------------------------------------------------------ reference API
    def forward(self, t, x):
        """Velocity ``(batch_size, n_nodes*3)`` at time ``t`` and positions ``x``."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.torch_forward(t, x)                    # differentiable (to any order): see there
        if not self.kernels_supported():
            return self.torch_forward(t, x)                    # widths beyond the kernels' 64: the same map, slower
        return self._run(t, x)[0]

    def dense_route_bytes(self, batch, dtype=torch.float32):
        """Lower bound on what ``torch_forward`` keeps alive for a backward at this batch size: per layer the pair tensors
        ``[B, n, n, .]`` of the message MLP's two hidden activations (2 F), the radial basis (G) and a few scalars per pair."""
        _, F, G = self._dims
        n = self.n_nodes
        return int(batch) * n * n * (2 * F + G + 8) * torch.empty((), dtype=dtype).element_size() * self._n_layers

    def kernels_supported(self):
        """The HIP kernels cover ``node_feat_dim, distance_feat_dim <= 64`` (one wave holds a feature column in
        registers); wider dynamics run on ``torch_forward`` (and a ``ContinuousFlow`` over them on autograd, like any
        user-supplied torch dynamics).  ``jvp`` / ``vjp`` exist for the kernels only."""
        return _lib.load().tfep_egnn_tile(self._dims[1], self._dims[2]) != 0

    def torch_forward(self, t, x):
        """The same velocity as a composite of torch operators on the device -- the DIFFERENTIABLE route.

        The kernels have forward-mode and reverse-mode products with respect to the positions but no parameter gradients,
        and the loss of a continuous flow needs the gradient of a trace of the Jacobian: second derivatives.  Under grad
        mode the dynamics therefore run the way the reference itself does -- plain torch ops that autograd differentiates,
        ``create_graph`` included (egnn.py:143-194, 272-369; continuous.py:231-278) -- in a DENSE formulation: every
        ordered pair ``[b, source i, destination j]`` of a sample with a "kept" mask (``i != j``, distance <= cutoff)
        instead of an edge list and ``scatter_add``.  Same numbers as the kernels to float32 rounding
        (tests/test_gpu_continuous.py), the reference's memory appetite (B n^2 (2F + G) floats per layer): meant for the
        sizes at which the reference can train; inference goes through the kernels (``torch.no_grad()``)."""
        import torch.nn.functional as Fn
        _lib.check_device_tensor(x, 'x', dtype=x.dtype)        # (no CPU path here either)
        B = x.shape[0]
        n = self.n_nodes
        if x.dim() != 2 or x.shape[1] != 3 * n:
            raise ValueError(f'x must have shape (batch_size, {3 * n}), got {tuple(x.shape)}')
        dt, dev = x.dtype, x.device
        # The dense formulation keeps B n^2 (2F + G) values per layer alive for the backward (more with ``create_graph``):
        # 800 TB at BASELINE cfg5's size (B = 16 384, n = 256).  Fail with the reason, not with an allocator error
        # somewhere inside the composite.
        if torch.is_grad_enabled():
            need = self.dense_route_bytes(B, dt)
            free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            if need > free:
                raise _lib.TfepHipError(
                    f'EGNNDynamics: the differentiable (autograd) route needs about {need / 2 ** 30:.0f} GiB for batch {B} x {n} nodes '
                    f'({self._n_layers} layers x B n^2 (2F + G) values kept for the backward) and {free / 2 ** 30:.0f} GiB are free.  '
                    'It is taken because grad mode is on and the input or a parameter requires a gradient: for inference call under '
                    'torch.no_grad() (or freeze the parameters) -- that runs on the HIP kernels at any size; for training reduce the '
                    'batch (the kernels have no parameter-gradient pass yet: DESIGN.md section 7).')
        t = torch.as_tensor(t, dtype=dt, device=dev).reshape(1)
        te = self.time_embedding
        t_emb = torch.exp(-torch.exp(te._log_gammas.to(dt)) * (t[:, None] - te._means.to(dt)) ** 2)[0]      # radial.py:110-130
        one_hot = self._node_types_one_hot.to(dt)
        h = self.h_embedding(torch.cat([one_hot, t_emb[None].expand(n, -1)], dim=-1))[None].expand(B, -1, -1)  # egnn.py:196-219
        pos0 = x.reshape(B, n, 3)
        pos = pos0
        off = ~torch.eye(n, dtype=torch.bool, device=dev)
        zero, one = torch.zeros((), dtype=dt, device=dev), torch.ones((), dtype=dt, device=dev)
        for layer in self._layers():
            emb = layer.distance_embedding
            rc = float(emb.r_cutoff)
            diff = pos[:, None, :, :] - pos[:, :, None, :]                  # x[dest j] - x[src i]   (graph.py:254)
            # (the i == j diagonal is not an edge: distance 1 there keeps non-finite values out of the autograd graph)
            d = torch.sqrt(torch.where(off[None], (diff * diff).sum(-1), one))
            keep = off[None] & (d <= rc)                                    # graph.py:297
            direction = diff / d[..., None]                                 # graph.py:260
            sw = 0.5 * torch.cos(math.pi / rc * d) + 0.5                    # radial.py:161-176
            if emb.force_zero_after_cutoff:
                sw = torch.where(d > rc, zero, sw)
            rbf = torch.exp(-torch.exp(emb._log_gammas.to(dt)) * (d[..., None] - emb._means.to(dt)) ** 2) * sw[..., None]
            F = h.shape[-1]
            w0 = layer.message_mlp[0]
            # first linear of the message MLP by input block: [h_src, h_dest, rbf] (egnn.py:246-251)
            z1 = (Fn.linear(h, w0.weight[:, :F])[:, :, None, :] + Fn.linear(h, w0.weight[:, F:2 * F], w0.bias)[:, None, :, :]
                  + Fn.linear(rbf, w0.weight[:, 2 * F:]))
            m = Fn.silu(layer.message_mlp[2](Fn.silu(z1)))
            m = m * torch.sigmoid(layer.attention_mlp[0](m))                # egnn.py:323-325
            node_msg = torch.where(keep[..., None], m, zero).sum(dim=1)     # messages arriving at dest j (egnn.py:331)
            upd = layer.update_h_mlp[2](Fn.silu(layer.update_h_mlp[0](torch.cat([h, node_msg], dim=-1))))
            mag = torch.tanh(layer.update_x_mlp[2](Fn.silu(layer.update_x_mlp[0](m))))
            disp = torch.where(keep[..., None], layer.speed_factor * direction * mag, zero).sum(dim=1)      # egnn.py:355-361
            h, pos = h + upd, pos + disp
        vel = pos - pos0                                                    # translation invariance (egnn.py:185)
        vel = vel - vel.mean(dim=1, keepdim=True)                           # centre of geometry preserved (:189-191)
        return vel.reshape(B, 3 * n)

    def jvp(self, t, x, v, trace=None, frobenius=None, scale=1.0, velocity_squared_norm=None, need_jvp=True):
        """``(velocity, J v)`` with ``J = d velocity / d x``; optionally ``trace += scale v . (J v)``,
        ``frobenius += scale |J v|^2`` and ``velocity_squared_norm = |velocity|^2`` (all ``(batch,)``, in place)."""
        return self._run(t, x, v, trace, frobenius, scale, velocity_squared_norm, need_jvp)

    def vjp(self, t, x, g, trace=None, frobenius=None, scale=1.0, velocity_squared_norm=None):
        """``(velocity, g^T J)`` with ``J = d velocity / d x``: a reverse pass through the kernels, the number
        ``torch.autograd.grad(velocity, x, g)`` gives in the reference (continuous.py:307-361).  Optionally
        ``trace += scale (g^T J) . g``, ``frobenius += scale |g^T J|^2``, ``velocity_squared_norm = |velocity|^2``.
        About 2.5 x the cost of ``jvp`` (the layer chain is recomputed once per reduction side; see csrc/egnn.hip)."""
        return self._run_vjp(t, x, g, trace, frobenius, scale, velocity_squared_norm)

    # ------------------------------------------------------------------ execution
    def _layers(self):
        return [self._modules['graph_layer_' + str(i)] for i in range(self._n_layers)]

    def _tile(self):
        nt = _lib.load().tfep_egnn_tile(self._dims[1], self._dims[2])
        if nt == 0:
            raise NotImplementedError(f'EGNNDynamics kernels support node_feat_dim, distance_feat_dim <= 64 '
                                      f'(got {self._dims[1]}, {self._dims[2]})')
        return nt

    def _layer_params(self, layer, device):
        """ctypes view of one layer's parameter tensors (kept alive by the returned list)."""
        def f32(p):
            p = p.detach()
            if p.device != device or p.dtype != torch.float32 or not p.is_contiguous():
                p = p.to(device=device, dtype=torch.float32).contiguous()
            return p
        tensors = dict(
            dist_means=f32(layer.distance_embedding._means), dist_log_gammas=f32(layer.distance_embedding._log_gammas),
            msg0_w=f32(layer.message_mlp[0].weight), msg0_b=f32(layer.message_mlp[0].bias),
            msg2_w=f32(layer.message_mlp[2].weight), msg2_b=f32(layer.message_mlp[2].bias),
            att_w=f32(layer.attention_mlp[0].weight), att_b=f32(layer.attention_mlp[0].bias),
            ux0_w=f32(layer.update_x_mlp[0].weight), ux0_b=f32(layer.update_x_mlp[0].bias),
            ux2_w=f32(layer.update_x_mlp[2].weight),
            uh0_w=f32(layer.update_h_mlp[0].weight), uh0_b=f32(layer.update_h_mlp[0].bias),
            uh2_w=f32(layer.update_h_mlp[2].weight), uh2_b=f32(layer.update_h_mlp[2].bias))
        p = _lib.EgnnLayerParams()
        p.F, p.G = self._dims[1], self._dims[2]
        for k, v in tensors.items():
            setattr(p, k, v.data_ptr())
        return p, tensors

    def _run(self, t, x, v=None, trace=None, frob=None, scale=1.0, vel_sq=None, need_jvp=True):
        _lib.check_device_tensor(x, 'x')
        x, _ = _lib.rows(x, 'x')
        x = x.contiguous()
        B, D = x.shape
        n = self.n_nodes
        if D != 3 * n:
            raise ValueError(f'x must have shape (batch_size, {3 * n}), got {tuple(x.shape)}')
        if B == 0:                             # an empty batch: empty velocity (and tangent); the accumulators stay as they are
            return torch.empty_like(x), (torch.empty_like(x) if v is not None and need_jvp else None)
        tan = v is not None
        if tan:
            _lib.check_device_tensor(v, 'v')
            v = v.contiguous()
            if v.shape != x.shape:
                raise ValueError('v must have the shape of x')
        for name, acc in (('trace', trace), ('frobenius', frob), ('velocity_squared_norm', vel_sq)):
            if acc is not None:
                _lib.check_device_tensor(acc, name)
                if acc.shape != (B,) or not acc.is_contiguous():
                    raise ValueError(f'{name} must be a contiguous (batch,) tensor')
        dev, stream = x.device, _lib.stream_of(x)
        nt = self._tile()
        fp = 16 * nt
        split = (os.environ.get('TFEP_EGNN_SPLIT', '1') != '0') if self.split_gemm is None else bool(self.split_gemm)
        f32 = dict(dtype=torch.float32, device=dev)
        layers = self._layers()
        L = len(layers)
        t = float(t)

        # ---- re-pack the parameters (every call, like the masked weights of the MAF path)
        n_packed = _lib.load().tfep_egnn_packed_floats(nt)
        packed = torch.empty(L, n_packed, **f32)
        keep = []
        params0 = None
        for li, layer in enumerate(layers):
            p, tensors = self._layer_params(layer, dev)
            keep.append(tensors)
            params0 = p if li == 0 else params0
            _lib.call('tfep_egnn_pack_layer', ctypes.byref(p), nt, _lib.ptr(packed[li]), stream)

        # ---- node embedding and layer-0 source / destination terms (the same for every sample)
        emb = torch.empty(3, n, fp, **f32)
        one_hot = self._node_types_one_hot.detach().to(**f32).contiguous()
        tm = self.time_embedding._means.detach().to(**f32).contiguous()
        tg = self.time_embedding._log_gammas.detach().to(**f32).contiguous()
        we = self.h_embedding.weight.detach().to(**f32).contiguous()
        be = self.h_embedding.bias.detach().to(**f32).contiguous()
        _lib.call('tfep_egnn_embed', _lib.ptr(one_hot), n, one_hot.shape[1], t, _lib.ptr(tm), _lib.ptr(tg), len(tm),
                  _lib.ptr(we), _lib.ptr(be), ctypes.byref(params0), nt, _lib.ptr(emb[0]), _lib.ptr(emb[1]),
                  _lib.ptr(emb[2]), stream)

        big = None
        if L > 1:       # h, nm, P, Q (+ tangents): (B, n, fp) each, reused by every layer
            big = torch.empty(8 if tan else 4, B, n, fp, **f32)
        pos, dpos = x, v
        h, h_bstride, dh = emb[0], 0, None
        P, Q, dP, dQ, pq_bstride = emb[1], emb[2], None, None, 0
        r_cutoff = float(layers[0].distance_embedding.r_cutoff)
        for li, layer in enumerate(layers):
            last = li == L - 1
            a = _lib.EgnnEdgeArgs()
            a.B, a.n_nodes, a.nt = B, n, nt
            a.split = int(split)
            a.r_cutoff, a.speed_factor = float(layer.distance_embedding.r_cutoff), float(layer.speed_factor)
            a.packed = packed[li].data_ptr()
            a.pos, a.dpos = pos.data_ptr(), (dpos.data_ptr() if tan else None)
            a.P, a.Q, a.pq_bstride = P.data_ptr(), Q.data_ptr(), pq_bstride
            a.dP, a.dQ = (dP.data_ptr() if dP is not None else None), (dQ.data_ptr() if dQ is not None else None)
            pos_out = torch.empty(B, D, **f32)
            dpos_out = torch.empty(B, D, **f32) if tan else None
            a.pos_out, a.dpos_out = pos_out.data_ptr(), (dpos_out.data_ptr() if tan else None)
            if not last:
                a.nm, a.dnm = big[1].data_ptr(), (big[5].data_ptr() if tan else None)
            _lib.call('tfep_egnn_edge', ctypes.byref(a), stream)
            if not last:
                # (each wave of the node kernel reads a node's h completely before writing it: in place from layer 1 on)
                na = _lib.EgnnNodeArgs()
                na.B, na.n_nodes, na.nt = B, n, nt
                na.packed, na.packed_next = packed[li].data_ptr(), packed[li + 1].data_ptr()
                na.h, na.h_bstride, na.dh = h.data_ptr(), h_bstride, (dh.data_ptr() if dh is not None else None)
                na.nm, na.dnm = big[1].data_ptr(), (big[5].data_ptr() if tan else None)
                na.h_out, na.P_out, na.Q_out = big[0].data_ptr(), big[2].data_ptr(), big[3].data_ptr()
                if tan:
                    na.dh_out, na.dP_out, na.dQ_out = big[4].data_ptr(), big[6].data_ptr(), big[7].data_ptr()
                _lib.call('tfep_egnn_node', ctypes.byref(na), stream)
                h, h_bstride, P, Q, pq_bstride = big[0], n, big[2], big[3], n
                if tan:
                    dh, dP, dQ = big[4], big[6], big[7]
            pos, dpos = pos_out, dpos_out
        vel = torch.empty(B, D, **f32)
        jv = torch.empty(B, D, **f32) if (tan and need_jvp) else None
        _lib.call('tfep_egnn_finish', _lib.ptr(pos), _lib.ptr(x), _lib.ptr(dpos), _lib.ptr(v) if tan else None, B, n,
                  _lib.ptr(vel), _lib.ptr(jv), float(scale), _lib.ptr(trace), _lib.ptr(frob), _lib.ptr(vel_sq), stream)
        del keep, r_cutoff
        return vel, jv


def _run_vjp(self, t, x, g, trace=None, frob=None, scale=1.0, vel_sq=None):
    """Forward pass keeping every layer's inputs, then the reverse pass (see ``EGNNDynamics.vjp``)."""
    _lib.check_device_tensor(x, 'x')
    _lib.check_device_tensor(g, 'g')
    x, g = x.contiguous(), g.contiguous()
    B, D = x.shape
    n = self.n_nodes
    if D != 3 * n or g.shape != x.shape:
        raise ValueError(f'x and g must have shape (batch_size, {3 * n})')
    if B == 0:
        return torch.empty_like(x), torch.empty_like(x)
    for name, acc in (('trace', trace), ('frobenius', frob), ('velocity_squared_norm', vel_sq)):
        if acc is not None:
            _lib.check_device_tensor(acc, name)
            if acc.shape != (B,) or not acc.is_contiguous():
                raise ValueError(f'{name} must be a contiguous (batch,) tensor')
    dev, stream = x.device, _lib.stream_of(x)
    nt = self._tile()
    fp = 16 * nt
    split = (os.environ.get('TFEP_EGNN_SPLIT', '1') != '0') if self.split_gemm is None else bool(self.split_gemm)
    f32 = dict(dtype=torch.float32, device=dev)
    layers = self._layers()
    L = len(layers)
    t = float(t)
    n_packed = _lib.load().tfep_egnn_packed_floats(nt)
    packed = torch.empty(L, n_packed, **f32)
    keep = []
    params0 = None
    for li, layer in enumerate(layers):
        p, tensors = self._layer_params(layer, dev)
        keep.append(tensors)
        params0 = p if li == 0 else params0
        _lib.call('tfep_egnn_pack_layer', ctypes.byref(p), nt, _lib.ptr(packed[li]), stream)
    emb = torch.empty(3, n, fp, **f32)
    one_hot = self._node_types_one_hot.detach().to(**f32).contiguous()
    tm = self.time_embedding._means.detach().to(**f32).contiguous()
    tg = self.time_embedding._log_gammas.detach().to(**f32).contiguous()
    we = self.h_embedding.weight.detach().to(**f32).contiguous()
    be = self.h_embedding.bias.detach().to(**f32).contiguous()
    _lib.call('tfep_egnn_embed', _lib.ptr(one_hot), n, one_hot.shape[1], t, _lib.ptr(tm), _lib.ptr(tg), len(tm),
              _lib.ptr(we), _lib.ptr(be), ctypes.byref(params0), nt, _lib.ptr(emb[0]), _lib.ptr(emb[1]),
              _lib.ptr(emb[2]), stream)

    # ---- forward, every layer's inputs kept: saved[l] = (h_l, nm_l, P_l, Q_l) (B, n, fp) each; pos[l]
    saved = torch.empty(max(L - 1, 1), 4, B, n, fp, **f32)          # layers 1 .. L-1 (layer 0: the shared embedding) + nm
    nm0 = torch.empty(B, n, fp, **f32) if L > 1 else None
    pos = [x]
    H, P, Q, bstride = [emb[0]], [emb[1]], [emb[2]], [0]
    NM = []
    for li, layer in enumerate(layers):
        last = li == L - 1
        a = _lib.EgnnEdgeArgs()
        a.B, a.n_nodes, a.nt, a.split = B, n, nt, int(split)
        a.r_cutoff, a.speed_factor = float(layer.distance_embedding.r_cutoff), float(layer.speed_factor)
        a.packed = packed[li].data_ptr()
        a.pos, a.P, a.Q, a.pq_bstride = pos[li].data_ptr(), P[li].data_ptr(), Q[li].data_ptr(), bstride[li]
        pos_out = torch.empty(B, D, **f32)
        a.pos_out = pos_out.data_ptr()
        if not last:
            nm = nm0 if li == 0 else saved[li - 1][1]
            a.nm = nm.data_ptr()
            NM.append(nm)
        _lib.call('tfep_egnn_edge', ctypes.byref(a), stream)
        pos.append(pos_out)
        if not last:
            na = _lib.EgnnNodeArgs()
            na.B, na.n_nodes, na.nt = B, n, nt
            na.packed, na.packed_next = packed[li].data_ptr(), packed[li + 1].data_ptr()
            na.h, na.h_bstride, na.nm = H[li].data_ptr(), bstride[li], NM[li].data_ptr()
            sl = saved[li]                                        # inputs of layer li + 1
            na.h_out, na.P_out, na.Q_out = sl[0].data_ptr(), sl[2].data_ptr(), sl[3].data_ptr()
            _lib.call('tfep_egnn_node', ctypes.byref(na), stream)
            H.append(sl[0]); P.append(sl[2]); Q.append(sl[3]); bstride.append(n)
    vel = torch.empty(B, D, **f32)
    _lib.call('tfep_egnn_finish', _lib.ptr(pos[L]), _lib.ptr(x), None, None, B, n, _lib.ptr(vel), None, 1.0, None, None,
              _lib.ptr(vel_sq), stream)

    # ---- reverse: velocity = C (pos_L - x) with the centring projector C
    gc = torch.empty(B, D, **f32)
    _lib.call('tfep_egnn_center', _lib.ptr(g), B, n, 1.0, _lib.ptr(gc), stream)
    g_pos_out = gc
    g_h = None
    g_nm = None
    work = torch.empty(4, B, n, fp, **f32)                        # g_P, g_Q, g_h, g_nm
    for li in range(L - 1, -1, -1):
        layer = layers[li]
        g_pos = torch.empty(B, D, **f32)
        for src_owned in (0, 1):
            a = _lib.EgnnEdgeBwdArgs()
            a.B, a.n_nodes, a.nt, a.split, a.src_owned = B, n, nt, int(split), src_owned
            a.r_cutoff, a.speed_factor = float(layer.distance_embedding.r_cutoff), float(layer.speed_factor)
            a.packed = packed[li].data_ptr()
            a.pos, a.P, a.Q, a.pq_bstride = pos[li].data_ptr(), P[li].data_ptr(), Q[li].data_ptr(), bstride[li]
            a.g_pos_out = g_pos_out.data_ptr()
            a.g_nm = g_nm.data_ptr() if g_nm is not None else None
            a.g_lane = work[0 if src_owned else 1].data_ptr()     # sources: g_P; destinations: g_Q
            a.g_pos = g_pos.data_ptr()
            _lib.call('tfep_egnn_edge_backward', ctypes.byref(a), stream)
        g_pos_out = g_pos
        if li > 0:
            nb = _lib.EgnnNodeBwdArgs()
            nb.B, nb.n_nodes, nb.nt = B, n, nt
            nb.packed, nb.packed_next = packed[li - 1].data_ptr(), packed[li].data_ptr()
            nb.h, nb.h_bstride, nb.nm = H[li - 1].data_ptr(), bstride[li - 1], NM[li - 1].data_ptr()
            nb.g_h_next = g_h.data_ptr() if g_h is not None else None
            nb.g_P, nb.g_Q = work[0].data_ptr(), work[1].data_ptr()
            # (each wave reads a node's cotangents completely before it writes them: g_h in place from the second time on)
            nb.g_h, nb.g_nm = work[2].data_ptr(), work[3].data_ptr()
            _lib.call('tfep_egnn_node_backward', ctypes.byref(nb), stream)
            g_h, g_nm = work[2], work[3]
    # g^T J = (d pos_L / d x)^T C g - C g
    out = torch.empty(B, D, **f32)
    ptrs = (ctypes.c_void_p * 4)(gc.data_ptr(), None, None, None)
    coef = (ctypes.c_float * 4)(-1.0, 0.0, 0.0, 0.0)
    _lib.call('tfep_ode_axpy', _lib.ptr(g_pos_out), ptrs, coef, 1, g_pos_out.numel(), _lib.ptr(out), stream)
    if trace is not None or frob is not None:
        _lib.call('tfep_row_dots', _lib.ptr(out), _lib.ptr(g), B, D, float(scale), _lib.ptr(trace), _lib.ptr(frob), stream)
    del keep
    return vel, out


EGNNDynamics._run_vjp = _run_vjp
