"""Rational-quadratic neural spline transformer (reference ``tfep/nn/transformers/spline.py``)."""
from typing import Optional

import torch

from ... import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.tfep.*)
from .transformer import MAFTransformer


class NeuralSplineTransformer(MAFTransformer):
    r"""Neural spline transformer, incl. circular splines and learnable domain bounds.

    Constructor, registered buffers (``x0, xf, n_bins, _y0, _yf, _circular,
    _identity_boundary_slopes, _learn_lower_bound, _learn_upper_bound, _min_bin_size,
    _min_slope``) and checks as reference spline.py:71-163.  The parameter layout of
    ``forward`` / ``inverse`` is the reference's (spline.py:187-226): reshaped to
    ``(batch, n_parameters_per_feature, n_features)`` the rows are widths, heights, slopes
    and the optional shift / domain parameters.
    """

    def __init__(
            self,
            x0: torch.Tensor,
            xf: torch.Tensor,
            n_bins: int,
            y0: Optional[torch.Tensor] = None,
            yf: Optional[torch.Tensor] = None,
            circular: bool = False,
            identity_boundary_slopes: bool = False,
            learn_lower_bound: bool = False,
            learn_upper_bound: bool = False,
            min_bin_size: float = 1e-4,
            min_slope: float = 1e-4,
    ):
        super().__init__()
        if y0 is None:
            y0 = x0.detach()
        if yf is None:
            yf = xf.detach()
        if circular and (learn_lower_bound or learn_upper_bound):
            raise ValueError('Cannot instantiate a circular spline with learnable limits.')
        if circular and not (torch.allclose(x0, y0) and torch.allclose(xf, yf)):
            raise ValueError('x0==y0 and xf==yf must hold for all periodic degrees of freedom.')
        if min_bin_size <= 0.:
            raise ValueError('The minimum bin size should be positive.')
        if (min_slope <= 0.) or (min_slope >= 1.):
            raise ValueError('The minimum slope should be between 0 and 1.')

        self.register_buffer('x0', x0)
        self.register_buffer('xf', xf)
        self.register_buffer('n_bins', torch.as_tensor(n_bins))
        self.register_buffer('_y0', y0)
        self.register_buffer('_yf', yf)
        self.register_buffer('_circular', torch.as_tensor(circular))
        self.register_buffer('_identity_boundary_slopes', torch.as_tensor(identity_boundary_slopes))
        self.register_buffer('_learn_lower_bound', torch.as_tensor(learn_lower_bound))
        self.register_buffer('_learn_upper_bound', torch.as_tensor(learn_upper_bound))
        self.register_buffer('_min_bin_size', torch.as_tensor(min_bin_size))
        self.register_buffer('_min_slope', torch.as_tensor(min_slope))
        self._cfg = None

    def host(self):
        """Host-side copy of the scalar buffers (``n_bins``, flags, minimum sizes).  They are registered
        buffers like in the reference, so after ``.to('cuda')`` reading them costs a device->host sync:
        do it once, not on every forward (and never inside a HIP-graph capture)."""
        h = self.__dict__.get('_host_cache')
        if h is None:
            h = dict(n_bins=int(self.n_bins), circular=bool(self._circular),
                     identity=bool(self._identity_boundary_slopes), learn_lower=bool(self._learn_lower_bound),
                     learn_upper=bool(self._learn_upper_bound), min_bin=float(self._min_bin_size),
                     min_slope=float(self._min_slope))
            self.__dict__['_host_cache'] = h
        return h

    @property
    def n_parameters_per_feature(self) -> int:
        """Parameters per feature (reference spline.py:165-182)."""
        h = self.host()
        n = 3 * h['n_bins'] + 1
        if h['learn_lower']:
            n += 1
        if h['learn_upper']:
            n += 1
        if h['identity']:
            n -= 1 if h['circular'] else 2
        return n

    def _apply(self, fn, *args, **kwargs):
        self._cfg = None
        self.__dict__.pop('_host_cache', None)
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._cfg = None
        self.__dict__.pop('_host_cache', None)
        return super()._load_from_state_dict(*args, **kwargs)

    def config(self, device):
        """Device descriptor handed to the kernels (rebuilt after .to() / load_state_dict)."""
        if self._cfg is None or self._cfg.x0.device != device:
            h = self.host()
            f32 = dict(device=device, dtype=torch.float32)
            self._cfg = ops.SplineConfig(
                self.x0.to(**f32), self.xf.to(**f32), self._y0.to(**f32), self._yf.to(**f32),
                h['n_bins'], h['circular'], h['identity'], h['learn_lower'], h['learn_upper'],
                h['min_bin'], h['min_slope'])
        return self._cfg

    def _op_args(self, device):
        cfg, h = self.config(device), self.host()
        return (cfg.x0, cfg.xf, cfg.y0, cfg.yf, h['n_bins'], h['circular'], h['identity'], h['learn_lower'],
                h['learn_upper'], h['min_bin'], h['min_slope'])

    def forward(self, x, parameters):
        ops.check_device_tensor(x, 'x')
        return tuple(torch.ops.tfep.spline_forward(x, parameters, *self._op_args(x.device)))   # differentiable

    def inverse(self, y, parameters):
        ops.check_device_tensor(y, 'y')
        return tuple(torch.ops.tfep.spline_inverse(y, parameters, *self._op_args(y.device)))

    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        """Zeros: equal bins, unit slopes, zero shift, unit domain scale (reference spline.py:263-297)."""
        if not (torch.allclose(self.x0, self._y0) and torch.allclose(self.xf, self._yf)):
            raise ValueError('The identity neural spline transformer can be '
                             'implemented only if x0=y0 and xf=yf.')
        return torch.zeros(size=(self.n_parameters_per_feature, n_features)).to(self.x0).reshape(-1)

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return degrees_in.tile((self.n_parameters_per_feature,))
