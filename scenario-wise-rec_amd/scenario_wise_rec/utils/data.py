"""Host-side data helpers the feature descriptors need (reference: `utils/data.py`).

The pandas / DataLoader input pipeline of the reference is outside the hot path;
only `get_auto_embedding_dim` (used by `SparseFeature(embed_dim=None)`) and a minimal tensor dataset
are provided.
"""
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
