"""Oracle (numpy): MADE degrees, autoregressive masks, masked linear, weight norm.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
Reference files restated here (paths relative to the reference root):
``tfep/nn/conditioners/made.py`` and ``tfep/nn/masked.py``.
"""
import math

import numpy as np


# -----------------------------------------------------------------------------
# degrees (host-side integer logic)
# -----------------------------------------------------------------------------

def round_robin(x, length, err_msg=None):
    """Tile ``x`` until ``length`` elements.  Ref: conditioners/made.py:441-461."""
    x = np.asarray(x)
    n_round_robin, n_remaining = divmod(length, len(x))
    if n_round_robin == 0:
        if err_msg is None:
            err_msg = f'Length {length} is smaller than the array (len={len(x)}).'
        raise ValueError(err_msg)
    out = np.tile(x, n_round_robin)
    if n_remaining != 0:
        out = np.concatenate([out, x[:n_remaining]])
    return out


def generate_degrees(n_features, order='ascending', max_value=None,
                     conditioning_indices=None, repeats=1, rng_perm=None):
    """Degrees of the MADE input nodes.  Ref: conditioners/made.py:32-145.

    ``order='random'`` uses ``torch.randperm`` in the reference; the oracle
    takes the permutation explicitly through ``rng_perm``.
    """
    n_noncond = n_features
    if conditioning_indices is not None:
        n_noncond -= len(conditioning_indices)

    if max_value is None:
        try:
            max_value = len(repeats) - 1                       # made.py:102-105
        except TypeError:
            max_value = int(math.ceil(n_noncond / repeats)) - 1  # made.py:106-108

    if order == 'ascending':
        degrees = np.arange(max_value + 1)
    elif order == 'descending':
        degrees = np.arange(max_value, -1, -1)
    elif order == 'random':
        degrees = np.asarray(rng_perm)
    else:
        raise ValueError("Accepted string values for 'order' "
                         "are 'ascending', 'descending', and 'random'.")

    # made.py:121-125  repeat_interleave, truncate, tile.
    degrees = np.repeat(degrees, repeats)[:n_noncond]
    degrees = round_robin(degrees, n_noncond)

    if conditioning_indices is not None:                       # made.py:128-143
        cond = [int(i) for i in np.asarray(conditioning_indices).tolist()]
        cond_set = set(cond)
        noncond = [i for i in range(n_features) if i not in cond_set]
        out = np.empty(n_features, dtype=degrees.dtype)
        out[cond] = -1
        out[noncond] = degrees
        degrees = out
    return degrees.astype(np.int64)


def degrees_hidden(degrees_in, degrees_out, hidden_layers=2):
    """Degrees of the hidden nodes.  Ref: conditioners/made.py:366-434."""
    degrees_in = np.asarray(degrees_in)
    degrees_out = np.asarray(degrees_out)
    try:
        hidden_layers = hidden_layers.tolist()
    except AttributeError:
        pass
    max_degree_out = degrees_out.max()
    relevant = degrees_in < max_degree_out                     # made.py:390-391

    if isinstance(hidden_layers, int):                         # made.py:395-403
        n_rel = int(relevant.sum())
        n_out = len(degrees_out)
        width = int(np.ceil((n_rel * n_out) ** 0.5))
        width = max(width, n_rel)
        hidden_layers = [width for _ in range(hidden_layers)]

    if isinstance(hidden_layers[0], int):                      # made.py:407-424
        out = []
        for layer_idx, width in enumerate(hidden_layers):
            motif = degrees_in[relevant]
            out.append(round_robin(
                motif, width,
                err_msg=(f'Hidden layer {layer_idx} is too small for the number'
                         ' of input features. Increase the size of the layer or'
                         ' explicitly pass the degrees for the hidden layers.')))
        return out

    out = [np.asarray(x) for x in hidden_layers]               # made.py:425-432
    for layer_idx, deg in enumerate(out):
        if np.any(deg >= max_degree_out):
            raise ValueError(f'The {layer_idx}-th hidden layer contain '
                             'nodes with degrees that will be ignored '
                             'by the output layer.')
    return out


def create_autoregressive_mask(degrees_in, degrees_out, strictly_less=True,
                               transpose=False, dtype=np.float32):
    """0/1 connectivity mask.  Ref: masked.py:36-108."""
    degrees_in = np.asarray(degrees_in)
    degrees_out = np.asarray(degrees_out)
    if transpose:
        if strictly_less:
            mask = degrees_out[:, None] > degrees_in[None, :]
        else:
            mask = degrees_out[:, None] >= degrees_in[None, :]
    else:
        if strictly_less:
            mask = degrees_out[None, :] > degrees_in[:, None]
        else:
            mask = degrees_out[None, :] >= degrees_in[:, None]
    return mask.astype(dtype)


def made_masks(degrees_in, degrees_out, hidden_layers=2, dtype=np.float32):
    """All (out, in) masks of a MADE network.  Ref: conditioners/made.py:286-329.

    Hidden layers use ``>=``, the output layer uses strict ``>`` (made.py:308-309).
    """
    hidden = degrees_hidden(degrees_in, degrees_out, hidden_layers)
    masks = []
    prev = np.asarray(degrees_in)
    for layer_idx in range(len(hidden) + 1):
        is_out = layer_idx == len(hidden)
        cur = np.asarray(degrees_out) if is_out else hidden[layer_idx]
        masks.append(create_autoregressive_mask(prev, cur, strictly_less=is_out,
                                                transpose=True, dtype=dtype))
        prev = cur
    return masks


# -----------------------------------------------------------------------------
# masked linear + weight norm (floating point)
# -----------------------------------------------------------------------------

def weight_norm_effective(weight_g, weight_v, mask=None):
    """``W = v * (g / ||v||_row)`` with masked entries forced to 0 (NaN-safe).

    Ref: masked.py:369-371 (compute_weight = torch _weight_norm, dim=0, then
    _ApplyMask) and masked.py:433-439 (entries with mask == 0 are set to 0.0,
    which also removes the NaNs of fully-masked rows).
    """
    v = np.asarray(weight_v)
    g = np.asarray(weight_g).reshape(-1, 1)
    with np.errstate(divide='ignore', invalid='ignore'):
        norm = np.sqrt(np.sum(v * v, axis=1, keepdims=True, dtype=v.dtype))
        w = v * (g / norm)
    if mask is not None:
        w = np.where(np.asarray(mask) == 0.0, np.zeros((), dtype=w.dtype), w)
    return w.astype(v.dtype)


def masked_linear(x, weight, bias=None, mask=None):
    """``y = x (M o W)^T + b``.  Ref: masked.py:265-277."""
    w = np.asarray(weight)
    if mask is not None:
        w = w * np.asarray(mask).astype(w.dtype)
    y = np.asarray(x) @ w.T
    if bias is not None:
        y = y + np.asarray(bias)
    return y


def elu(x):
    """ELU, alpha = 1 (torch.nn.ELU default).  Ref: conditioners/made.py:320."""
    x = np.asarray(x)
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def made_layers_from_state(state, prefix=''):
    """Collect the MaskedLinear layers of one MADE from a state_dict-like mapping.

    Keys follow the reference schema ``{prefix}layers.{0,2,4,...}.{bias,weight_g,
    weight_v,weight,mask}`` (made.py:320-329, masked.py:170,391-394).
    """
    idx = sorted({int(k[len(prefix):].split('.')[1]) for k in state
                  if k.startswith(prefix + 'layers.')})
    layers = []
    for i in idx:
        p = f'{prefix}layers.{i}.'
        layer = {'bias': np.asarray(state[p + 'bias']), 'mask': np.asarray(state[p + 'mask'])}
        if p + 'weight_g' in state:
            layer['weight_g'] = np.asarray(state[p + 'weight_g'])
            layer['weight_v'] = np.asarray(state[p + 'weight_v'])
        else:
            layer['weight'] = np.asarray(state[p + 'weight'])
        layers.append(layer)
    return layers


def made_forward(x, layers, return_hidden=False):
    """MADE forward: [MaskedLinear, ELU]* MaskedLinear.  Ref: conditioners/made.py:355-356, 320-326.

    With weight norm the effective weight is recomputed from (g, v) on every
    call, exactly like the reference forward pre-hook (masked.py:397-398).
    """
    h = np.asarray(x)
    hidden = []
    for li, layer in enumerate(layers):
        if 'weight_g' in layer:
            w = weight_norm_effective(layer['weight_g'], layer['weight_v'], layer['mask'])
        else:
            w = layer['weight']
        h = masked_linear(h, w.astype(h.dtype), layer['bias'].astype(h.dtype), layer['mask'])
        if li != len(layers) - 1:
            h = elu(h)
            hidden.append(h)
    if return_hidden:
        return h, hidden
    return h
