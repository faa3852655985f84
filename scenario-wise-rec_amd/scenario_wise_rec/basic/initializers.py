"""Embedding-table initialisers (reference: `basic/initializers.py`): callables
`(vocab_size, embed_dim) -> torch.nn.Embedding`.  The nn.Embedding object is only the parameter
holder (state_dict key `<...>.embed_dict.<name>.weight`); lookups run in the fused gather kernel."""
import torch


class _Init(object):
    def _fill(self, weight):
        raise NotImplementedError

    def __call__(self, vocab_size, embed_dim):
        embed = torch.nn.Embedding(vocab_size, embed_dim)
        self._fill(embed.weight)
        return embed


class RandomNormal(_Init):
    """N(mean, std); the features' default is RandomNormal(0, 1e-4) (`basic/features.py:62`)."""

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = mean, std

    def _fill(self, weight):
        torch.nn.init.normal_(weight, self.mean, self.std)


class RandomUniform(_Init):
    def __init__(self, minval=0.0, maxval=1.0):
        self.minval, self.maxval = minval, maxval

    def _fill(self, weight):
        torch.nn.init.uniform_(weight, self.minval, self.maxval)


class XavierNormal(_Init):
    def __init__(self, gain=1.0):
        self.gain = gain

    def _fill(self, weight):
        torch.nn.init.xavier_normal_(weight, self.gain)


class XavierUniform(_Init):
    def __init__(self, gain=1.0):
        self.gain = gain

    def _fill(self, weight):
        torch.nn.init.xavier_uniform_(weight, self.gain)


class Pretrained(object):
    """Table from a given 2-D weight, optionally frozen (`basic/initializers.py:76-92`)."""

    def __init__(self, embedding_weight, freeze=True):
        self.embedding_weight = torch.as_tensor(embedding_weight, dtype=torch.float32)
        self.freeze = freeze

    def __call__(self, vocab_size, embed_dim):
        assert vocab_size == self.embedding_weight.shape[0] and embed_dim == self.embedding_weight.shape[1]
        return torch.nn.Embedding.from_pretrained(self.embedding_weight, freeze=self.freeze)
