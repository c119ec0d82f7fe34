"""Fixed-topology graph utilities with the ``tfep.nn.graph`` API (reference ``tfep/nn/graph.py``).

``FixedGraph`` (:29-116), ``get_all_edges`` (:119-163), ``fix_node_indices_batch_size`` (:166-218),
``compute_edge_distances`` (:222-263), ``prune_long_edges`` (:266-301) and ``unsorted_segment_sum`` (:304-316), kept for
callers that build their own graph networks.  The edge lists are host-side integer work (plain index arithmetic, no
floating point); the segment sum is ``tfep_segment_sum``.  ``tfep_amd.nn.dynamics.EGNNDynamics`` itself never builds an
edge list: its kernels walk all ordered pairs on the fly (``csrc/egnn.hip``).
"""
from typing import Optional, Sequence

import torch

from .. import _lib
from ..utils.misc import ensure_tensor_sequence


class FixedGraph(torch.nn.Module):
    """Graph with a fixed topology: one-hot node types (buffer ``_node_types_one_hot``) and cached edges."""

    def __init__(self, node_types: Sequence[int], mask: Optional[torch.Tensor] = None):
        super().__init__()
        node_types = ensure_tensor_sequence(node_types)
        one_hot = torch.nn.functional.one_hot(node_types).to(torch.get_default_dtype())
        self.register_buffer('_node_types_one_hot', one_hot)
        self._last_batch_edges = get_all_edges(batch_size=1, n_nodes=self.n_nodes, mask=mask)
        self._n_edges = int(self._last_batch_edges.shape[1])

    @property
    def n_nodes(self):
        """int: Number of nodes in the graph."""
        return len(self._node_types_one_hot)

    @property
    def n_edges(self):
        """int: Number of (directional) edges of one sample."""
        return self._n_edges

    def get_edges(self, batch_size):
        """``(2, batch_size*n_edges)`` node indices in ``[0, batch_size*n_nodes)``; cached for the last batch size."""
        self._last_batch_edges = fix_node_indices_batch_size(
            node_indices=self._last_batch_edges, new_batch_size=batch_size, n_indices=self._n_edges, n_nodes=self.n_nodes)
        return self._last_batch_edges


def get_all_edges(batch_size, n_nodes, mask=None):
    """All ordered pairs ``src != dest`` (or the non-zeros of ``mask[src, dest]``), sample after sample."""
    if mask is None:
        if n_nodes == 1:
            return torch.empty(2, 0)
        src = torch.arange(n_nodes).repeat_interleave(n_nodes)
        dest = torch.arange(n_nodes).repeat(n_nodes)
        off = src != dest
        edges = torch.stack([src[off], dest[off]])
    else:
        if mask.shape != (n_nodes, n_nodes):
            raise ValueError('mask must have shape (n_nodes, n_nodes)')
        edges = mask.nonzero().t()
    return fix_node_indices_batch_size(node_indices=edges, new_batch_size=batch_size, n_indices=edges.shape[-1],
                                       n_nodes=n_nodes)


def fix_node_indices_batch_size(node_indices: torch.Tensor, new_batch_size: int, n_indices: int, n_nodes: int):
    """Grow / shrink ``(*, old_batch*n_indices)`` node indices to ``new_batch_size`` samples."""
    wanted = new_batch_size * n_indices
    have = node_indices.shape[-1]
    if have == wanted:
        return node_indices
    if have > wanted:
        return node_indices[..., :wanted]
    one = node_indices[..., :n_indices]
    shift = torch.arange(0, new_batch_size * n_nodes, n_nodes, device=one.device).unsqueeze(-1)
    return (one.unsqueeze(-2) + shift).flatten(start_dim=-2)


def compute_edge_distances(x, edges, normalize_directions=False, inverse_directions=False):
    """Distances ``(n_edges,)`` and direction vectors ``x[dest] - x[src]`` ``(n_edges, 3)`` across the edges."""
    a, b = (edges[0], edges[1]) if inverse_directions else (edges[1], edges[0])
    directions = x[a] - x[b]
    distances = torch.sqrt(torch.sum(directions ** 2, dim=-1))
    if normalize_directions:
        directions = directions / distances.unsqueeze(-1)
    return distances, directions


def prune_long_edges(r_cutoff, edges, distances, *args):
    """Drop the edges longer than ``r_cutoff`` (and the matching rows of ``args``)."""
    keep = distances <= r_cutoff
    return (edges[:, keep], distances[keep], *[a[keep] for a in args])


def unsorted_segment_sum(data, segment_ids, n_segments):
    """``out[s] = sum of the rows of data with segment_ids == s`` on the HIP device (``tfep_segment_sum``)."""
    _lib.check_device_tensor(data, 'data')
    if data.dim() != 2:
        raise ValueError('data must be (n_rows, n_columns)')
    data = data.contiguous()
    ids = segment_ids.to(device=data.device, dtype=torch.int64).contiguous()
    if ids.shape != (data.shape[0],):
        raise ValueError('segment_ids must have one entry per row of data')
    out = torch.empty(n_segments, data.shape[1], dtype=data.dtype, device=data.device)
    _lib.call('tfep_segment_sum', _lib.ptr(data), _lib.ptr(ids), data.shape[0], data.shape[1], n_segments,
              _lib.ptr(out), _lib.stream_of(data))
    return out
