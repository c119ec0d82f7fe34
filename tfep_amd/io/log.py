"""Per-sample log of potentials, log|det J| and CVs, in the on-disk format of the reference's ``TFEPLogger``
(``tfep/io/log.py:34-73`` format, ``:75-131`` constructor, ``:157-485`` read/save API, ``:537-643`` files / indices).

Layout under ``save_dir_path`` (files written by either implementation are readable by the other)::

    metadata.json                {"batch_size", "n_samples_per_epoch", "version"}
    train/epoch-<E>.npz          one 1D array of length n_samples_per_epoch per tensor name + the bool array "__mask"
                                 (True where a batch has been saved); entry i = sample i % batch_size of batch i // batch_size
    eval/step-<S>.npz            one 1D array per name, appended batch by batch (any length)

The hot path produces these values sharded over ranks (one process per GPU); :func:`gather_to_rank0` brings the shards of a
batch to rank 0, which is the only rank that should own a logger (the class is not multi-process safe, like the
reference's).
"""
import json
import os
import warnings

import numpy as np
import torch


class _Slot:
    """The one npz archive of a kind ('train' or 'eval') that is held in memory."""

    def __init__(self, directory, prefix):
        self.directory = directory
        self.prefix = prefix
        self.index = None
        self.arrays = None

    @property
    def path(self):
        return os.path.join(self.directory, f'{self.prefix}-{self.index}.npz')

    def dump(self):
        np.savez_compressed(self.path, **self.arrays)


def _as_numpy(value):
    return value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value)


class TFEPLogger:
    """Store and retrieve per-sample quantities by training epoch / batch or by optimisation step.

    Same constructor, properties, methods, warnings and errors as the reference class; not multi-process or thread safe.
    """

    VERSION = '0.1'
    METADATA_FILE_NAME = 'metadata.json'
    INDEX_NAMES = ['trajectory_sample_index', 'dataset_sample_index']
    MASK_NAME = '__mask'

    def __init__(self, save_dir_path='tfep_logs', data_loader=None, train_subdir_name='train', eval_subdir_name='eval'):
        self._save_dir_path = os.path.realpath(save_dir_path)
        self._slots = {'train': _Slot(os.path.join(save_dir_path, train_subdir_name), 'epoch'),
                       'eval': _Slot(os.path.join(save_dir_path, eval_subdir_name), 'step')}
        metadata_path = os.path.join(save_dir_path, self.METADATA_FILE_NAME)
        resuming = os.path.isfile(metadata_path)          # the metadata file is written last by a new logger
        if resuming:
            with open(metadata_path) as f:
                meta = json.load(f)
            self._batch_size, self._n_samples_per_epoch = meta['batch_size'], meta['n_samples_per_epoch']
        elif data_loader is None:
            raise ValueError("When creating a new logger, 'data_loader' must be passed.")
        else:
            self._read_loader(data_loader)
        os.makedirs(save_dir_path, exist_ok=True)
        for slot in self._slots.values():
            os.makedirs(slot.directory, exist_ok=True)
        if not resuming:
            with open(metadata_path, 'w') as f:
                json.dump({'batch_size': self.batch_size, 'n_samples_per_epoch': self.n_samples_per_epoch,
                           'version': self.VERSION}, f)

    # ------------------------------------------------------------------ sizes
    @property
    def batch_size(self):
        """The batch size of the training dataset."""
        return self._batch_size

    @property
    def n_samples_per_epoch(self):
        """Samples per training epoch (the dataset size minus the dropped remainder under ``drop_last``)."""
        return self._n_samples_per_epoch

    @property
    def n_batches_per_epoch(self):
        return -(-self._n_samples_per_epoch // self._batch_size)

    @property
    def save_dir_path(self):
        return self._save_dir_path

    def _read_loader(self, loader):
        batch_size, drop_last = loader.batch_size, getattr(loader, 'drop_last', False)
        if batch_size is None:                            # a custom batch sampler carries both
            batch_size, drop_last = loader.batch_sampler.batch_size, loader.batch_sampler.drop_last
        n = len(loader.dataset)
        self._batch_size = batch_size
        self._n_samples_per_epoch = n - n % batch_size if drop_last else n

    # ------------------------------------------------------------------ indices and archives
    def _resolve(self, step_idx, epoch_idx, batch_idx, need_batch):
        """(step, epoch, batch) from whichever of them was given."""
        per_epoch = self.n_batches_per_epoch
        if step_idx is not None:
            epoch_idx, batch_idx = divmod(step_idx, per_epoch)
        elif epoch_idx is None:
            raise ValueError("Either step_idx or epoch_idx must be passed.")
        elif batch_idx is not None:
            step_idx = epoch_idx * per_epoch + batch_idx
        elif need_batch:
            raise ValueError("To save tensors either 'step_idx' or both 'epoch_idx' and 'batch_idx' must be passed.")
        return step_idx, epoch_idx, batch_idx

    def _open(self, kind, index):
        slot = self._slots[kind]
        if slot.index != index:
            slot.index = index
            if os.path.isfile(slot.path):
                with np.load(slot.path) as archive:
                    slot.arrays = {name: archive[name] for name in archive.files}
            elif kind == 'train':
                slot.arrays = {self.MASK_NAME: np.zeros(self.n_samples_per_epoch, dtype=bool)}
            else:
                slot.arrays = {}
        return slot

    def _selection(self, slot, kind, remove_nans):
        """Boolean selection of entries, or None for "everything" (evaluation data without NaN filtering)."""
        arrays = slot.arrays
        if remove_nans is False:
            return arrays[self.MASK_NAME] if kind == 'train' else None
        if remove_nans is True:
            columns = [v for k, v in arrays.items() if k != self.MASK_NAME]
        else:
            columns = [arrays[remove_nans]]
        keep = None
        for column in columns:
            finite = ~np.isnan(column)
            keep = finite if keep is None else keep & finite
        if kind == 'train':
            keep = keep & arrays[self.MASK_NAME]
        return keep

    @classmethod
    def _warn_if_no_indices(cls, tensors):
        if not any(name in tensors for name in cls.INDEX_NAMES):
            warnings.warn(f"tensors does not contain any sample indices among the following attributes: {cls.INDEX_NAMES}. "
                          "Without it, it might be difficult to match training and evaluation configurations to their "
                          "reference potential.")

    @staticmethod
    def _export(data, as_numpy):
        return data if as_numpy else {k: torch.tensor(v) for k, v in data.items()}

    # ------------------------------------------------------------------ training data
    def save_train_tensors(self, tensors, step_idx=None, epoch_idx=None, batch_idx=None):
        """Save ``(batch_size,)`` tensors of a training batch, or ``(n_samples_per_epoch,)`` tensors of a whole epoch
        (only ``epoch_idx`` given)."""
        self._warn_if_no_indices(tensors)
        _, epoch_idx, batch_idx = self._resolve(step_idx, epoch_idx, batch_idx, need_batch=False)
        slot = self._open('train', epoch_idx)
        saved = slot.arrays[self.MASK_NAME]
        for name, value in tensors.items():
            value = _as_numpy(value)
            if batch_idx is None:
                slot.arrays[name] = value
                saved[:] = True
                continue
            if name not in slot.arrays:
                slot.arrays[name] = np.empty(self.n_samples_per_epoch, dtype=value.dtype)
            begin = self.batch_size * batch_idx
            slot.arrays[name][begin:begin + len(value)] = value
            saved[begin:begin + len(value)] = True
        slot.dump()

    def read_train_tensors(self, names=None, step_idx=None, epoch_idx=None, batch_idx=None, remove_nans=False, as_numpy=False):
        """The saved entries of an epoch or of one of its batches (unsaved batches are left out)."""
        _, epoch_idx, batch_idx = self._resolve(step_idx, epoch_idx, batch_idx, need_batch=False)
        slot = self._open('train', epoch_idx)
        if names is None:
            names = [k for k in slot.arrays if k != self.MASK_NAME]
        keep = self._selection(slot, 'train', remove_nans)
        window = slice(None) if batch_idx is None else slice(self.batch_size * batch_idx, self.batch_size * (batch_idx + 1))
        return self._export({name: slot.arrays[name][window][keep[window]] for name in names}, as_numpy)

    # ------------------------------------------------------------------ evaluation data
    def save_eval_tensors(self, tensors, step_idx=None, epoch_idx=None, batch_idx=None, update=False):
        """Append the tensors evaluated with the network trained for ``step_idx`` steps; with ``update`` entries whose
        trajectory / dataset sample index is already stored are overwritten instead."""
        self._warn_if_no_indices(tensors)
        step_idx, _, _ = self._resolve(step_idx, epoch_idx, batch_idx, need_batch=True)
        slot = self._open('eval', step_idx)
        names = list(slot.arrays) if slot.arrays else list(tensors)
        try:
            new = {name: _as_numpy(tensors[name]) for name in names}
        except KeyError:
            raise KeyError("'tensors' must include all the following Tensors: " + str(names))
        if update:
            key = next((k for k in self.INDEX_NAMES if k in new), None)
            if key is not None and key in slot.arrays:
                _, incoming, stored = np.intersect1d(new[key], slot.arrays[key], assume_unique=True, return_indices=True)
                if len(incoming):
                    for name in names:
                        slot.arrays[name][stored] = new[name][incoming]
                        new[name] = np.delete(new[name], incoming)
        for name in names:
            slot.arrays[name] = np.concatenate((slot.arrays[name], new[name])) if name in slot.arrays else new[name]
        slot.dump()

    def read_eval_tensors(self, names=None, step_idx=None, epoch_idx=None, batch_idx=None, remove_nans=False, sort_by=None,
                          as_numpy=False):
        """The tensors saved for a step; ``sort_by`` reorders every array by that one and rewrites the file in that order."""
        step_idx, _, _ = self._resolve(step_idx, epoch_idx, batch_idx, need_batch=True)
        slot = self._open('eval', step_idx)
        if sort_by is not None:
            order = np.argsort(slot.arrays[sort_by])
            slot.arrays = {k: v[order] for k, v in slot.arrays.items()}
            slot.dump()
        data = slot.arrays if names is None else {name: slot.arrays[name] for name in names}
        keep = self._selection(slot, 'eval', remove_nans)
        if keep is not None:
            data = {k: v[keep] for k, v in data.items()}
        return self._export(data, as_numpy)


def gather_to_rank0(tensors, group=None):
    """Concatenate the per-rank shards of a batch on rank 0 (rank order = row order of :func:`tfep_amd.distributed.shard_rows`).

    Returns the gathered dict on rank 0 and ``None`` elsewhere; without an initialised process group the input is returned.
    Shards may have different lengths (the remainder rows go to the first ranks).  One ``all_gather`` of the lengths and one
    padded ``all_gather`` per tensor: a few kB per step, nothing on the data path.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = {}
    lengths = None
    for name in sorted(tensors):
        t = tensors[name].detach()
        if lengths is None:
            mine = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
            all_len = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(all_len, mine, group=group)
            lengths = [int(v) for v in all_len]
        longest = max(lengths)
        padded = t.new_zeros((longest,) + tuple(t.shape[1:]))
        padded[:t.shape[0]] = t
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        if rank == 0:
            out[name] = torch.cat([p[:n] for p, n in zip(parts, lengths)])
    return out if rank == 0 else None
