"""Host-side data helpers the feature descriptors need (reference: `utils/data.py`).

The pandas ETL of the reference is outside the hot path.  Provided: `get_auto_embedding_dim` (used by
`SparseFeature(embed_dim=None)`), the reference's dict-of-columns dataset / DataLoader factory, and -- SURVEY.md 8 row
f3 -- `DeviceDataLoader`: the same batches from columns that live in HBM, without the per-row dict building of
`TorchDataset.__getitem__` (`utils/data.py:18-19`) and the per-column `.to(device)` of the training loop
(`trainers/ctr_trainer.py:67`), which at GPU step times (0.7 ms) are what an epoch would otherwise spend its time on.
"""
import ctypes as C

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, random_split


def get_auto_embedding_dim(num_classes):
    """emb_dim = floor(6 * num_classes^(1/4)) (`utils/data.py:65-75`, DCN rule of thumb)."""
    return int(np.floor(6 * np.power(num_classes, 0.25)))


class TorchDataset(Dataset):
    """dict-of-columns dataset: item i = ({name: column[i]}, y[i]) (`utils/data.py:11-22`)."""

    def __init__(self, x, y):
        super().__init__()
        self.x, self.y = x, y

    def __getitem__(self, index):
        return {k: v[index] for k, v in self.x.items()}, self.y[index]

    def __len__(self):
        return len(self.y)


class DataGenerator(object):
    """Train / val / test DataLoaders from column dicts (`utils/data.py:37-62`)."""

    def __init__(self, x, y):
        super().__init__()
        self.dataset = TorchDataset(x, y)
        self.length = len(self.dataset)

    def generate_dataloader(self, x_val=None, y_val=None, x_test=None, y_test=None, split_ratio=None, batch_size=16,
                            num_workers=8):
        if split_ratio is not None:
            n_train = int(self.length * split_ratio[0])
            n_val = int(self.length * split_ratio[1])
            n_test = self.length - n_train - n_val
            print("the samples of train : val : test are  %d : %d : %d" % (n_train, n_val, n_test))
            train, val, test = random_split(self.dataset, (n_train, n_val, n_test))
        else:
            train, val, test = self.dataset, TorchDataset(x_val, y_val), TorchDataset(x_test, y_test)
        mk = lambda ds, sh: DataLoader(ds, batch_size=batch_size, shuffle=sh, num_workers=num_workers)
        return mk(train, True), mk(val, False), mk(test, False)


class DeviceDataLoader(object):
    """Batches of a dict-of-columns dataset served from HBM.

    `x`: {name: array-like [n]} (numpy arrays or tensors, any of the dtypes the lookup accepts), `y`: [n].  Columns are
    uploaded once.  A batch is a dict of contiguous row VIEWS (no copy, no launch) plus the label view -- exactly what
    `DataLoader(TorchDataset(x, y), batch_size)` yields after the trainer's `.to(device)`.  `shuffle=True` draws a new
    row permutation per epoch (torch.randperm on the device, seeded by `generator`) and applies it to ALL columns with
    ONE launch (`swr_take_rows`, csrc/take.hip).  `drop_last` as in torch.  Iterating yields `(x_dict, y)`; `len()` is the
    number of batches."""

    def __init__(self, x, y, batch_size, device="cuda", shuffle=False, drop_last=False, generator=None):
        from .. import _hip as H
        self._H = H
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise H.SwrError("DeviceDataLoader keeps its columns in HBM: device must be a GPU")
        as_dev = lambda v: (v if torch.is_tensor(v) else torch.as_tensor(np.ascontiguousarray(v))).to(self.device).contiguous()
        self.names = list(x.keys())
        self.cols = [as_dev(x[k]).reshape(-1) for k in self.names] + [as_dev(y).reshape(-1)]
        self.n = int(self.cols[-1].numel())
        for k, c in zip(self.names, self.cols):
            if c.numel() != self.n:
                raise ValueError(f"column {k!r} has {c.numel()} rows, the labels {self.n}")
            if c.element_size() not in (1, 2, 4, 8):
                raise ValueError(f"column {k!r}: unsupported dtype {c.dtype}")
        if len(self.cols) > 96:
            raise ValueError("DeviceDataLoader: at most 95 feature columns")
        self.batch_size, self.shuffle, self.drop_last, self.generator = int(batch_size), bool(shuffle), bool(drop_last), generator
        self._shuffled = [torch.empty_like(c) for c in self.cols] if self.shuffle else None

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def _permute(self):
        H = self._H
        perm = torch.randperm(self.n, device=self.device, generator=self.generator)
        tab = (H.TakeColumn * len(self.cols))()
        for j, (src, dst) in enumerate(zip(self.cols, self._shuffled)):
            tab[j] = H.TakeColumn(src.data_ptr(), dst.data_ptr(), src.element_size(), 0)
        H.check(H.lib.swr_take_rows(tab, len(self.cols), H.ptr(perm), self.n, self.n, H.ptr(H.err_flag(self.device)),
                                    H.stream()), "swr_take_rows")
        return self._shuffled

    def __iter__(self):
        cols = self._permute() if self.shuffle and self.n else self.cols
        for b in range(len(self)):
            lo, hi = b * self.batch_size, min((b + 1) * self.batch_size, self.n)
            yield {k: c[lo:hi] for k, c in zip(self.names, cols)}, cols[-1][lo:hi]
