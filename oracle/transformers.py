"""Oracle (numpy): MAF transformers and the periodic embedding.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
Reference files restated here: ``tfep/nn/transformers/{affine,spline,moebius,
mixed}.py`` and ``tfep/nn/embeddings/mafembed.py``.

Conventions: ``x`` is ``(B, D)``; ``parameters`` is ``(B, P*D)`` in the
reference's parameter-major layout (column ``p*D + f`` is parameter ``p`` of
feature ``f``; spline.py:351-352, affine.py:138-141).  Every transformer
returns ``(y, log_det_J)`` with ``log_det_J`` of shape ``(B,)``.
"""
import numpy as np


# -----------------------------------------------------------------------------
# affine / volume preserving  (transformers/affine.py)
# -----------------------------------------------------------------------------

def affine_split(parameters):
    """(B, 2D) -> shift (B, D), log_scale (B, D).  Ref: affine.py:136-141."""
    b = parameters.shape[0]
    p = parameters.reshape(b, 2, -1)
    return p[:, 0], p[:, 1]


def affine_forward(x, parameters):
    """y = x*exp(a) + b; ldj = sum a.  Ref: affine.py:321-323."""
    shift, log_scale = affine_split(parameters)
    y = x * np.exp(log_scale) + shift
    return y, np.sum(log_scale, axis=1)


def affine_inverse(y, parameters):
    """x = (y-b)*exp(-a); ldj = -sum a.  Ref: affine.py:361-363."""
    shift, log_scale = affine_split(parameters)
    x = (y - shift) * np.exp(-log_scale)
    return x, -np.sum(log_scale, axis=1)


def volume_preserving_forward(x, parameters, periodic_indices=None, periodic_limits=None):
    """y = x + b (periodic wrap on selected columns), ldj = 0.  Ref: affine.py:366-411."""
    y = x + parameters
    if periodic_indices is not None:
        lo, hi = periodic_limits
        y[:, periodic_indices] = np.mod(y[:, periodic_indices], hi - lo) + lo
    return y, np.zeros(x.shape[0], dtype=x.dtype)


def volume_preserving_inverse(y, parameters, periodic_indices=None, periodic_limits=None):
    """x = y - b (periodic wrap), ldj = 0.  Ref: affine.py:414-456."""
    x = y - parameters
    if periodic_indices is not None:
        lo, hi = periodic_limits
        x[:, periodic_indices] = np.mod(x[:, periodic_indices], hi - lo) + lo
    return x, np.zeros(y.shape[0], dtype=y.dtype)


# -----------------------------------------------------------------------------
# rational-quadratic neural spline  (transformers/spline.py)
# -----------------------------------------------------------------------------

def spline_n_parameters_per_feature(n_bins, circular=False, identity_boundary_slopes=False,
                                    learn_lower_bound=False, learn_upper_bound=False):
    """Ref: spline.py:165-182."""
    n = 3 * n_bins + 1
    if learn_lower_bound:
        n += 1
    if learn_upper_bound:
        n += 1
    if identity_boundary_slopes:
        n -= 1 if circular else 2
    return n


def _softmax(a, axis):
    m = np.max(a, axis=axis, keepdims=True)
    e = np.exp(a - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def _softplus(a):
    """torch.nn.functional.softplus, beta=1, threshold=20."""
    with np.errstate(over='ignore'):
        return np.where(a > 20.0, a, np.log1p(np.exp(np.minimum(a, 20.0))))


def spline_get_parameters(parameters, x0, xf, n_bins, y0=None, yf=None, circular=False,
                          identity_boundary_slopes=False, learn_lower_bound=False,
                          learn_upper_bound=False, min_bin_size=1e-4, min_slope=1e-4):
    """Split + normalise the conditioner output.  Ref: spline.py:319-417.

    Returns ``x0, y0, widths (B,K,D), heights (B,K,D), slopes (B,K+1,D), shifts|None``.
    """
    dt = parameters.dtype
    x0 = np.asarray(x0, dtype=dt)
    xf = np.asarray(xf, dtype=dt)
    y0 = x0 if y0 is None else np.asarray(y0, dtype=dt)
    yf = xf if yf is None else np.asarray(yf, dtype=dt)
    min_bin_size = np.asarray(min_bin_size, dtype=dt)
    min_slope = np.asarray(min_slope, dtype=dt)
    K = n_bins
    P = spline_n_parameters_per_feature(K, circular, identity_boundary_slopes,
                                        learn_lower_bound, learn_upper_bound)
    b = parameters.shape[0]
    p = parameters.reshape(b, P, -1)                          # spline.py:351-352

    widths = p[:, :K]
    heights = p[:, K:2 * K]
    if identity_boundary_slopes:                              # spline.py:359-365
        n_slopes = K - 1
    elif circular:
        n_slopes = K
    else:
        n_slopes = K + 1
    slopes = p[:, 2 * K:2 * K + n_slopes]

    if circular:                                              # spline.py:368-375
        shifts = p[:, -1]
        if not identity_boundary_slopes:
            slopes = np.concatenate([slopes, slopes[:, :1]], axis=1)
    else:
        shifts = None

    if identity_boundary_slopes:                              # spline.py:378-380
        zeros = np.zeros_like(widths[:, :1])
        slopes = np.concatenate([zeros, slopes, zeros], axis=1)

    min_interval = K * min_bin_size                           # spline.py:384-391
    rescaled_width = xf - x0 - min_interval
    rescaled_height = yf - y0 - min_interval
    if learn_lower_bound or learn_upper_bound:
        domain_scale = np.exp(p[:, -1:])
        rescaled_width = rescaled_width * domain_scale
        rescaled_height = rescaled_height * domain_scale

    widths = _softmax(widths, axis=1) * rescaled_width + min_bin_size    # spline.py:394-395
    heights = _softmax(heights, axis=1) * rescaled_height + min_bin_size

    x0_out, y0_out = x0, y0                                   # spline.py:399-410
    if learn_lower_bound and learn_upper_bound:
        domain_shift = p[:, -2]
        x0_out = x0 + domain_shift
        y0_out = y0 + domain_shift
    elif learn_lower_bound:
        x0_out = xf - rescaled_width.squeeze(1) - min_interval
        y0_out = yf - rescaled_height.squeeze(1) - min_interval

    offset = np.log(np.exp(1. - min_slope) - 1.)              # spline.py:414-415
    slopes = _softplus(slopes + offset) + min_slope
    return x0_out, y0_out, widths.astype(dt), heights.astype(dt), slopes.astype(dt), shifts


def _assign_bins(x, x0, y0, widths, heights, slopes, inverse):
    """Bin search with two sentinel knots.  Ref: spline.py:567-650."""
    b, K, d = widths.shape
    dt = widths.dtype
    cum_w = np.cumsum(widths, axis=1, dtype=dt)               # spline.py:572-573
    cum_h = np.cumsum(heights, axis=1, dtype=dt)
    x0 = np.atleast_1d(np.asarray(x0, dtype=dt))
    y0 = np.atleast_1d(np.asarray(y0, dtype=dt))

    n_knots = K + 3
    knots_x = np.empty((b, n_knots, d), dtype=dt)             # spline.py:589-594
    knots_x[:, 1] = x0
    knots_x[:, 2:-1] = np.expand_dims(x0, -2) + cum_w
    knots_y = np.empty((b, n_knots, d), dtype=dt)
    knots_y[:, 1] = y0
    knots_y[:, 2:-1] = np.expand_dims(y0, -2) + cum_h

    dx = cum_w[:, -1] * dt.type(1000.)                        # spline.py:599-607
    knots_x[:, 0] = x0 - dx
    knots_x[:, -1] = knots_x[:, -2] + dx
    dy0 = slopes[:, 0] * dx
    knots_y[:, 0] = y0 - dy0
    dyf = slopes[:, -1] * dx
    knots_y[:, -1] = knots_y[:, -2] + dyf

    slopes_p = np.concatenate([slopes[:, 0:1], slopes, slopes[:, -1:]], axis=1)   # spline.py:611-614
    widths_p = np.concatenate([dx[:, None], widths, dx[:, None]], axis=1)
    heights_p = np.concatenate([dy0[:, None], heights, dyf[:, None]], axis=1)

    knots = knots_y if inverse else knots_x                   # spline.py:622-625 (strict >)
    bin_idx = np.sum(x[:, None, :] > knots, axis=1) - 1

    bi = np.arange(b)[:, None]
    fi = np.arange(d)[None, :]
    w = widths_p[bi, bin_idx, fi]                             # spline.py:629-639
    h = heights_p[bi, bin_idx, fi]
    xk = knots_x[bi, bin_idx, fi]
    yk = knots_y[bi, bin_idx, fi]
    dk = slopes_p[bi, bin_idx, fi]
    dk1 = slopes_p[bi, bin_idx + 1, fi]
    s = h / w                                                 # spline.py:643
    return w, h, xk, yk, dk, dk1, s, bin_idx


def _log_det_J(dk, dk1, s, eps, eps_1m, eps2, inverse):
    """Ref: spline.py:546-564."""
    num = s ** 2 * (dk1 * eps2 + 2 * s * eps_1m + dk * (1 - eps) ** 2)
    den = (s + (dk1 + dk - 2 * s) * eps_1m) ** 2
    ldj = np.sum(np.log(num / den), axis=1)
    return -ldj if inverse else ldj


def neural_spline(x, x0, y0, widths, heights, slopes, return_bins=False):
    """Functional RQ-spline forward.  Ref: spline.py:424-501."""
    w, h, xk, yk, dk, dk1, s, bins = _assign_bins(x, x0, y0, widths, heights, slopes, inverse=False)
    eps = (x - xk) / w
    eps_1m = eps * (1 - eps)
    eps2 = eps ** 2
    num = h * (s * eps2 + dk * eps_1m)
    den = s + (dk1 + dk - 2 * s) * eps_1m
    y = yk + num / den
    ldj = _log_det_J(dk, dk1, s, eps, eps_1m, eps2, inverse=False)
    if return_bins:
        return y, ldj, bins
    return y, ldj


def neural_spline_inverse(y, x0, y0, widths, heights, slopes):
    """Functional RQ-spline inverse.  Ref: spline.py:504-543."""
    w, h, xk, yk, dk, dk1, s, _ = _assign_bins(y, x0, y0, widths, heights, slopes, inverse=True)
    y_myk = y - yk
    t = dk1 + dk - 2 * s
    a = h * (s - dk) + y_myk * t
    bq = h * dk - y_myk * t
    c = -s * y_myk
    eps = 2 * c / (-bq - np.sqrt(bq ** 2 - 4 * a * c))
    x = eps * w + xk
    ldj = _log_det_J(dk, dk1, s, eps, eps * (1 - eps), eps ** 2, inverse=True)
    return x, ldj


def spline_forward(x, parameters, x0, xf, n_bins, **cfg):
    """NeuralSplineTransformer.forward.  Ref: spline.py:184-241."""
    nx0, ny0, widths, heights, slopes, shifts = spline_get_parameters(parameters, x0, xf, n_bins, **cfg)
    if cfg.get('circular', False):                            # spline.py:236-238
        xf_ = np.asarray(xf, dtype=x.dtype)
        x = np.mod(x - nx0 + shifts, xf_ - nx0) + nx0
    return neural_spline(x, nx0, ny0, widths, heights, slopes)


def spline_inverse(y, parameters, x0, xf, n_bins, **cfg):
    """NeuralSplineTransformer.inverse.  Ref: spline.py:243-261."""
    nx0, ny0, widths, heights, slopes, shifts = spline_get_parameters(parameters, x0, xf, n_bins, **cfg)
    x, ldj = neural_spline_inverse(y, nx0, ny0, widths, heights, slopes)
    if shifts is not None:                                    # spline.py:257-259
        xf_ = np.asarray(xf, dtype=x.dtype)
        x = np.mod(x - nx0 - shifts, xf_ - nx0) + nx0
    return x, ldj


# -----------------------------------------------------------------------------
# Moebius  (transformers/moebius.py)
# -----------------------------------------------------------------------------

def moebius_forward(x, parameters, dimension, max_radius=0.99, unit_sphere=False):
    """Moebius transformer on d-vectors.  Ref: moebius.py:374-478 (functional), :104-111 (reshape).

    The log-det is ``log|det|`` of the explicit d x d Jacobian (moebius.py:460-476).
    """
    b, n_feat = x.shape
    dt = x.dtype
    xv = x.reshape(b, -1, dimension)
    w = parameters.reshape(b, -1, dimension)
    w_norm = np.linalg.norm(w, axis=-1, keepdims=True)
    rescaling = dt.type(max_radius) / (1 + w_norm)            # moebius.py:437-441
    if not unit_sphere:
        x_norm = np.linalg.norm(xv, axis=-1, keepdims=True)
        rescaling = x_norm * rescaling
    w = rescaling * w
    w_norm = rescaling * w_norm
    if unit_sphere:                                           # moebius.py:446-452
        numerator = 1 - w_norm ** 2
    else:
        numerator = x_norm ** 2 - w_norm ** 2
    diff = xv - w
    diff_norm = np.linalg.norm(diff, axis=-1, keepdims=True)
    y = numerator / diff_norm ** 2 * diff - w

    numerator = numerator[..., None]                          # moebius.py:458-472
    diff_norm = diff_norm[..., None]
    dd_outer = diff[..., :, None] * diff[..., None, :]
    eye = np.eye(dimension, dtype=dt)
    jac = numerator * (eye / diff_norm ** 2 - 2 / diff_norm ** 4 * dd_outer)
    if not unit_sphere:
        xn = x_norm[..., None]
        jac2 = eye - xv[..., :, None] * xv[..., None, :] / xn ** 2
        jac = np.einsum('...ij,...jk->...ik', jac, jac2)
        jac = y[..., :, None] * xv[..., None, :] / xn ** 2 + jac
    ldj = np.linalg.slogdet(jac)[1].sum(axis=-1)              # moebius.py:475-476
    return y.reshape(b, n_feat), ldj.astype(dt)


def moebius_inverse(y, parameters, dimension, max_radius=0.99, unit_sphere=False):
    """Inverse = forward with -w.  Ref: moebius.py:142-147."""
    return moebius_forward(y, -parameters, dimension, max_radius, unit_sphere)


# -----------------------------------------------------------------------------
# mixed transformer  (transformers/mixed.py)
# -----------------------------------------------------------------------------

def mixed_run(x, parameters, transformers, indices, par_lengths, inverse=False):
    """Dispatch index groups to sub-transformers.  Ref: mixed.py:165-186.

    ``transformers[i]`` is a pair ``(forward_fn, inverse_fn)`` of callables taking
    ``(x_group, parameters_group)``; ``parameters`` is grouped BY TRANSFORMER
    (tensor_split at the cumulative parameter lengths, mixed.py:64-68, 175).
    """
    y = np.empty_like(x)
    ldj = np.zeros(x.shape[0], dtype=x.dtype)
    splits = np.cumsum(par_lengths[:-1])
    pars = np.split(parameters, splits, axis=1)
    for (fwd, inv), ind, par in zip(transformers, indices, pars):
        fn = inv if inverse else fwd
        y[:, ind], l = fn(x[:, ind], par)
        ldj = ldj + l
    return y, ldj


# -----------------------------------------------------------------------------
# periodic embedding  (embeddings/mafembed.py)
# -----------------------------------------------------------------------------

def periodic_embedding(x, limits, periodic_indices, nonperiodic_indices):
    """[x_nonperiodic..., cos t0, sin t0, cos t1, sin t1, ...].  Ref: mafembed.py:112-145."""
    b = x.shape[0]
    lo, hi = limits
    period_scale = 2 * np.pi / (hi - lo)
    xp = (x[:, periodic_indices] - lo) * x.dtype.type(period_scale)
    emb = np.stack([np.cos(xp), np.sin(xp)], axis=2).reshape(b, -1)
    return np.concatenate([x[:, nonperiodic_indices], emb], axis=1).astype(x.dtype)


def periodic_embedding_degrees_out(degrees_in, periodic_indices, nonperiodic_indices):
    """Ref: mafembed.py:147-167."""
    degrees_in = np.asarray(degrees_in)
    return np.concatenate([degrees_in[nonperiodic_indices],
                           np.repeat(degrees_in[periodic_indices], 2)])


# -----------------------------------------------------------------------------
# Flip-invariant and mixed embeddings (mafembed.py:174-446)
# -----------------------------------------------------------------------------

def _elu(v):
    return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))


def _mlp2(v, w0, b0, w1, b1):
    """Linear -> ELU -> Linear with torch.nn.Linear weights (out, in).  Ref: mafembed.py:219-228."""
    return _elu(v @ w0.T + b0) @ w1.T + b1


def flip_invariant_embedding(x, embedded_indices, nonembedded_indices, vector_dimension, emb_net, weight_net):
    """[x_nonembedded..., per vector: softmax-weighted sum of emb_net(v) and emb_net(-v)].
    ``emb_net`` / ``weight_net``: (w0, b0, w1, b1).  Ref: mafembed.py:257-306."""
    b = x.shape[0]
    v = x[:, embedded_indices].reshape(-1, vector_dimension)
    e = np.stack([_mlp2(v, *emb_net), _mlp2(-v, *emb_net)], axis=1)           # (b * n_vec, 2, E)
    a = np.stack([_mlp2(v, *weight_net), _mlp2(-v, *weight_net)], axis=1)     # (b * n_vec, 2, 1)
    a = a - a.max(axis=1, keepdims=True)
    wgt = np.exp(a)
    wgt = wgt / wgt.sum(axis=1, keepdims=True)
    emb = (wgt * e).sum(axis=1).reshape(b, -1)
    return np.concatenate([x[:, nonembedded_indices], emb], axis=1).astype(x.dtype)


def flip_invariant_embedding_degrees_out(degrees_in, embedded_indices, nonembedded_indices, vector_dimension,
                                         embedding_dimension):
    """Ref: mafembed.py:308-348."""
    degrees_in = np.asarray(degrees_in)
    vec = degrees_in[embedded_indices].reshape(-1, vector_dimension)
    if not np.all(vec == vec[:, [0]]):
        raise ValueError('The same degree must be assigned to all components of each embedded vectors.')
    return np.concatenate([degrees_in[nonembedded_indices], np.repeat(vec[:, 0], embedding_dimension)])


def mixed_embedding(x, parts, nonembedded_indices):
    """``parts``: list of (indices, callable) -- each callable embeds ``x[:, indices]``.  Ref: mafembed.py:411-427."""
    return np.concatenate([x[:, nonembedded_indices]] + [f(x[:, idx]) for idx, f in parts], axis=1).astype(x.dtype)
