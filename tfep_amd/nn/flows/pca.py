"""Flow in PCA-whitened coordinates (reference ``tfep/nn/flows/pca.py:26-125``).

Same constructor, buffers (``mean``, ``whitening_matrix``, ``blackening_matrix``, ``whitening_log_det_J``: the
reference's state_dict schema) and pass logic.  The whitening statistics are estimated once, at construction, with torch
(mean / covariance / ``torch.linalg.eigh``, like the reference's ``utils.math.cov``, pca.py:52-76); the two dense
``(B, D) x (D, D)`` products of every pass run on the HIP GEMM kernel through ``torch.ops.tfep.masked_linear`` (no mask,
no weight norm), which also gives them their gradient."""
import torch

from ... import ops
from ... import torch_ops  # noqa: F401  (registers torch.ops.tfep.*)


class PCAWhitenedFlow(torch.nn.Module):
    """Wraps ``flow``: whiten with the PCA of the data ``x`` given at construction, run the flow, and -- ``blacken`` --
    map the result back (reference pca.py:26-125)."""

    def __init__(self, flow, x, blacken=True):
        super().__init__()
        self.flow = flow
        self.blacken = blacken
        x = x.detach()
        if x.dim() != 2:
            raise ValueError('The function supports only 2D matrices')
        # mean and covariance (ddof = 1), reference utils/math.py:67-134
        mean = torch.mean(x, 0)
        xc = x - mean
        cov = torch.matmul(xc.t(), xc) / (x.shape[0] - 1)
        eigvalues, eigvectors = torch.linalg.eigh(cov)
        if torch.any(eigvalues < 0.0):
            raise ValueError(
                'Cannot determine the PCA whitening matrix since some of the '
                'eigenvalues of the covariance matrix estimate are negative. '
                'Likely, this is due to an insufficient number of samples.')
        singular_values = torch.sqrt(eigvalues)
        self.register_buffer('mean', mean)
        self.register_buffer('whitening_matrix', torch.matmul(eigvectors, torch.diag(1. / singular_values)))
        self.register_buffer('blackening_matrix', torch.matmul(torch.diag(singular_values), eigvectors.t()))
        self.register_buffer('whitening_log_det_J', -torch.sum(torch.log(singular_values)))
        self._dev = {}

    def _apply(self, fn, *args, **kwargs):
        self._dev = {}
        return super()._apply(fn, *args, **kwargs)

    def n_parameters(self):
        """int: The total number of parameters that can be optimized."""
        return self.flow.n_parameters()

    def forward(self, x):
        return self._pass(x, inverse=False)

    def inverse(self, y):
        return self._pass(y, inverse=True)

    def _operands(self, device):
        """float32 device copies for the kernels: ``(mean, W^T, B^T)`` (the GEMM computes ``x weight^T``), re-made when a
        buffer is replaced or written (load_state_dict)."""
        bufs = (self.mean, self.whitening_matrix, self.blackening_matrix)
        key = tuple((b._version, b.data_ptr()) for b in bufs)
        cached = self._dev.get(str(device))
        if cached is None or cached[0] != key:
            f32 = dict(device=device, dtype=torch.float32)
            cached = (key, self.mean.to(**f32).contiguous(), self.whitening_matrix.to(**f32).t().contiguous(),
                      self.blackening_matrix.to(**f32).t().contiguous())
            self._dev[str(device)] = cached
        return cached[1:]

    def _whiten(self, x):
        mean, w_t, _ = self._operands(x.device)
        return torch.ops.tfep.masked_linear(x - mean, w_t, None, None, None)

    def _blacken(self, x):
        mean, _, b_t = self._operands(x.device)
        return torch.ops.tfep.masked_linear(x, b_t, None, None, None) + mean

    def _pass(self, x, inverse):
        ops.check_device_tensor(x, 'x')
        if x.dim() != 2 or x.shape[1] != self.mean.shape[0]:
            raise RuntimeError(f'PCAWhitenedFlow: x has shape {tuple(x.shape)}, the whitening matrix was estimated for '
                               f'{self.mean.shape[0]} features')
        # whiten on the way in unless this is an inverse pass whose forward left the output whitened; blacken on the
        # way out likewise (reference pca.py:96-124)
        whiten = not inverse or self.blacken
        blacken = inverse or self.blacken
        if whiten:
            x = self._whiten(x)
        y, log_det_J = self.flow.inverse(x) if inverse else self.flow(x)
        if blacken:
            y = self._blacken(y)
        if not (whiten and blacken):            # only one of the two: their Jacobians do not cancel
            ldj_w = self.whitening_log_det_J.to(device=log_det_J.device, dtype=log_det_J.dtype)
            log_det_J = log_det_J + ldj_w if whiten else log_det_J - ldj_w
        return y, log_det_J
