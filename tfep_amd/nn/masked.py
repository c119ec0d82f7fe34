"""Masked linear layers with the ``tfep.nn.masked`` API, computing on gfx950 kernels.

Mirrors reference ``tfep/nn/masked.py``: ``create_autoregressive_mask`` (:36-108),
``MaskedLinear`` (:115-213), ``masked_linear`` (:305), ``masked_weight_norm`` (:312-330),
``remove_masked_weight_norm`` (:333-348).  Same constructor arguments, parameter / buffer
names (``weight`` | ``weight_g`` + ``weight_v``, ``bias``, ``mask``) and error behaviour.

The arithmetic runs in ``libtfep_hip.so``: the effective weight ``mask o g v/||v||`` is
rebuilt by ``tfep_masked_weight_prepare`` on every forward (the reference does the same in
a forward pre-hook, masked.py:397-398) and the product by the fp32-MFMA GEMM.  There is no
CPU path: calling ``forward`` on CPU tensors raises.
"""
import numpy as np
import torch
from torch.nn.parameter import Parameter

from .. import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.tfep.*)


# =============================================================================
# CREATE AUTOREGRESSIVE MASKS
# =============================================================================

def create_autoregressive_mask(degrees_in, degrees_out, strictly_less=True, transpose=False, dtype=None):
    """0/1 mask connecting inputs to outputs of (strictly) greater degree.

    Same semantics as reference masked.py:36-108: with ``transpose=True`` the mask has
    shape ``(n_out, n_in)`` and ``mask[o, i] = deg_out[o] > deg_in[i]`` (``>=`` if
    ``strictly_less`` is False).
    """
    degrees_in = torch.as_tensor(np.asarray(degrees_in) if not torch.is_tensor(degrees_in) else degrees_in)
    degrees_out = torch.as_tensor(np.asarray(degrees_out) if not torch.is_tensor(degrees_out) else degrees_out)
    if transpose:
        a, b = degrees_out[:, None], degrees_in[None, :]
    else:
        a, b = degrees_out[None, :], degrees_in[:, None]
    mask = (a > b) if strictly_less else (a >= b)
    if dtype is None:
        dtype = torch.get_default_dtype()
    return mask.to(dtype)


# =============================================================================
# FUNCTIONAL API
# =============================================================================

class MaskedLinearFunc(torch.autograd.Function):
    r"""``y = x (M o A)^T + b`` with the analytic backward of the reference (masked.py:220-302):
    ``grad_input = g (M o A)``, ``grad_weight = (g^T x) o M``, ``grad_bias = sum_b g`` -- every product on
    the fp32-MFMA GEMM of ``libtfep_hip.so``.  With ``weight_g`` the weight is the masked weight-norm
    parametrisation ``M o g v/||v||`` (masked.py:351-404) and the gradients are those of ``g`` and ``v``
    (masked entries of ``grad_v`` and fully-masked rows of ``grad_g`` are zero, masked.py:401-402, :429).
    """

    @staticmethod
    def forward(ctx, input, weight, bias=None, mask=None, weight_g=None):
        x2 = input.detach().reshape(-1, input.shape[-1])
        n_out, k = weight.shape
        tm, tn, tk = ops.tile_sizes()
        k_pad, n_pad = ops.round_up(k, tk), ops.round_up(n_out, tk)
        w = ops.masked_weight_prepare(weight.detach(), None if weight_g is None else weight_g.detach(), mask,
                                      n_rows_padded=n_pad, k_padded=k_pad)
        xp = ops.pad_columns(x2, k_pad)
        y = ops.masked_linear_packed(xp, w, None if bias is None else bias.detach(), n_out)
        ctx.save_for_backward(xp, w, weight, mask, weight_g)
        ctx.has_bias = bias is not None
        ctx.in_shape = input.shape
        return y.reshape(*input.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, grad_output):
        from .flows._backward import _gemm, _transpose
        from .. import _lib
        xp, w, weight, mask, weight_g = ctx.saved_tensors
        n_out, k = weight.shape
        n_pad, k_pad = w.shape
        tm, tn, tk = ops.tile_sizes()
        f32 = dict(dtype=torch.float32, device=xp.device)
        g2 = grad_output.reshape(-1, n_out).float()
        B = g2.shape[0]
        gp = ops.pad_columns(g2, n_pad)
        grad_input = grad_weight = grad_bias = grad_g = None
        if ctx.needs_input_grad[0]:
            wt = _transpose(w, n_pad, k_pad, torch.zeros(k_pad, n_pad, **f32))
            gx = _gemm(gp, wt, torch.empty(B, k_pad, **f32), B, k_pad, k_pad)
            grad_input = gx[:, :k].reshape(ctx.in_shape)
        if ctx.needs_input_grad[1] or (weight_g is not None and ctx.needs_input_grad[4]):
            Bp = ops.round_up(B, tk)
            gT = _transpose(gp, B, n_pad, torch.zeros(n_pad, Bp, **f32))
            xT = _transpose(xp, B, k_pad, torch.zeros(k_pad, Bp, **f32))
            gw = _gemm(gT, xT, torch.zeros(n_pad, k_pad, **f32), n_pad, k_pad, k_pad, accumulate=1)
            grad_weight = torch.empty_like(weight)
            if weight_g is not None:
                grad_g = torch.empty_like(weight_g)
            w_c = weight.detach().contiguous()
            g_c = None if weight_g is None else weight_g.detach().contiguous()
            m_c = None if mask is None else mask.contiguous()
            _lib.call('tfep_weight_norm_backward', _lib.ptr(gw), k_pad, _lib.ptr(w_c), _lib.ptr(g_c), _lib.ptr(m_c), n_out, k, None, None,
                      _lib.ptr(grad_weight), _lib.ptr(grad_g), _lib.stream_of(xp))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = torch.empty(n_out, **f32)
            _lib.call('tfep_column_sums', _lib.ptr(gp), n_pad, B, n_out, _lib.ptr(grad_bias), 0, _lib.stream_of(xp))
        return grad_input, grad_weight, grad_bias, None, grad_g


def masked_linear(input, weight, bias=None, mask=None):
    r"""``y = x (M o A)^T + b`` (reference masked.py:265-277, :305), differentiable.

    ``input`` may have extra leading dimensions ``(batch, *, in_features)``.
    """
    ops.check_device_tensor(input, 'input')
    return torch.ops.tfep.masked_linear(input, weight, bias, mask, None)


# =============================================================================
# MODULE API
# =============================================================================

class MaskedLinear(torch.nn.Linear):
    r"""Masked linear transformation :math:`y = x \cdot (M \circ A)^T + b`.

    Constructor, attributes and ``state_dict`` keys as reference masked.py:115-213.
    """

    def __init__(self, in_features, out_features, bias=True, mask=None):
        super().__init__(in_features, out_features, bias=bias)
        self.register_buffer('mask', mask)
        # Masked weights start at exactly 0 (masked.py:172-176).
        if self.mask is not None:
            self.weight.data = self.weight.data * self.mask

    def n_parameters(self):
        """int: The total number of (unmasked) parameters (masked.py:178-186)."""
        if self.mask is None:
            n = self._weight_numel()
        else:
            n = (self.mask != 0).sum()
        if self.bias is not None:
            n = n + self.bias.numel()
        return n

    def _weight_numel(self):
        return self.out_features * self.in_features

    @property
    def has_weight_norm(self):
        return 'weight_g' in self._parameters

    def effective_weight(self):
        """The masked (and weight-normalised) weight as a dense ``(out, in)`` HIP tensor."""
        if self.has_weight_norm:
            w = ops.masked_weight_prepare(self.weight_v.detach(), self.weight_g.detach(), self.mask,
                                          k_padded=self.in_features)
        else:
            w = ops.masked_weight_prepare(self._parameters['weight'].detach(), None, self.mask,
                                          k_padded=self.in_features)
        return w

    def __getattr__(self, name):
        # With weight norm the reference exposes ``module.weight`` as the recomputed tensor
        # (masked.py:395); here it is computed on demand.
        if name == 'weight' and 'weight_g' in self.__dict__.get('_parameters', {}):
            return self.effective_weight()
        return super().__getattr__(name)

    def forward(self, input):
        ops.check_device_tensor(input, 'input')
        if self.has_weight_norm:
            return torch.ops.tfep.masked_linear(input, self.weight_v, self.bias, self.mask, self.weight_g)
        return torch.ops.tfep.masked_linear(input, self._parameters['weight'], self.bias, self.mask, None)

    def extra_repr(self):
        return 'in_features={}, out_features={}, bias={}, weight_norm={}'.format(
            self.in_features, self.out_features, self.bias is not None, self.has_weight_norm)


# =============================================================================
# WEIGHT NORMALIZATION
# =============================================================================

#: attribute set on a parameter by a backward that returns its gradient already masked: the ``data_ptr()`` of that
#: gradient (consumed by the hooks below, which pass exactly that tensor through and mask anything else)
GRAD_IS_MASKED = '_tfep_grad_is_masked'


def masked_weight_norm(module, name='weight', dim=0):
    """NaN-free weight normalisation of a (masked) linear module (reference masked.py:312-404).

    Replaces parameter ``name`` with ``name_g`` (per-row norm, shape ``(out, 1)``) and
    ``name_v`` (direction).  Gradient hooks keep masked entries of ``v`` and fully-masked
    rows of ``g`` at zero gradient (masked.py:401-402), for optimisers run by the caller.
    """
    if name + '_g' in module._parameters:
        raise RuntimeError("Cannot register two weight_norm hooks on the same parameter {}".format(name))
    if dim != 0:
        raise ValueError('masked_weight_norm supports dim=0 (one norm per output row) only.')
    mask = getattr(module, 'mask', None)
    weight = module._parameters[name]
    del module._parameters[name]
    g = Parameter(torch.linalg.vector_norm(weight.data, ord=2, dim=1, keepdim=True))
    v = Parameter(weight.data)
    module.register_parameter(name + '_g', g)
    module.register_parameter(name + '_v', v)
    if mask is not None:
        # The hooks read the module's mask buffer when they run (it follows the module across devices; nothing is copied
        # per step and nothing but torch ops on the gradient's device runs, so they can be captured in a HIP graph).  The
        # MAF layer's own backward already returns masked gradients (tfep_weight_norm_backward applies these very rules)
        # and says so on the parameter (GRAD_IS_MASKED): the hook then passes the gradient through instead of re-masking
        # 4.5 GB per cfg2 output layer and step.
        def _g_hook(grad, module=module, g=g):
            if g.__dict__.pop(GRAD_IS_MASKED, None) == grad.data_ptr():
                return grad
            m = module.mask
            key = (m._version, m.data_ptr())
            cached = module.__dict__.get('_tfep_live_rows')
            if cached is None or cached[0] != key:
                cached = module.__dict__['_tfep_live_rows'] = (key, (m != 0).any(dim=1, keepdim=True))
            return grad * cached[1].to(dtype=grad.dtype, device=grad.device)

        def _v_hook(grad, module=module, v=v):
            if v.__dict__.pop(GRAD_IS_MASKED, None) == grad.data_ptr():
                return grad
            return grad.masked_fill(module.mask.to(grad.device) == 0, 0.0)

        module._tfep_wn_hooks = (g.register_hook(_g_hook), v.register_hook(_v_hook))
    return module


def remove_masked_weight_norm(module, name='weight'):
    """Fold ``g``/``v`` back into a plain ``weight`` parameter (reference masked.py:333-348)."""
    if name + '_g' not in module._parameters:
        raise ValueError("weight_norm of '{}' not found in {}".format(name, module))
    g = module._parameters[name + '_g']
    v = module._parameters[name + '_v']
    norm = torch.linalg.vector_norm(v.data, ord=2, dim=1, keepdim=True)
    w = v.data * (g.data / norm)
    mask = getattr(module, 'mask', None)
    if mask is not None:
        w[mask == 0.0] = 0.0
    for h in getattr(module, '_tfep_wn_hooks', ()):
        h.remove()
    del module._parameters[name + '_g']
    del module._parameters[name + '_v']
    module.register_parameter(name, Parameter(w))
    return module
