"""Feature descriptors (reference: `basic/features.py:5-95`).

A descriptor is the schema of one input column AND the owner of its embedding table: the table is
created lazily by `get_embedding_layer()` and cached on the descriptor, so two models built from the
same descriptor objects share one table.  `hash_seed` is a build-side addition for
hashed vocabularies (BASELINE config 5): raw ids are mapped to rows by the gather kernel itself.
"""
from ..utils.data import get_auto_embedding_dim
from .initializers import RandomNormal


class SparseFeature(object):
    """One categorical column looked up in a [vocab_size, embed_dim] table."""

    def __init__(self, name, vocab_size, embed_dim=None, shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001), hash_seed=0):
        self.name = name
        self.vocab_size = vocab_size
        self.embed_dim = get_auto_embedding_dim(vocab_size) if embed_dim is None else embed_dim
        self.shared_with = shared_with
        self.padding_idx = padding_idx
        self.initializer = initializer
        self.hash_seed = int(hash_seed)

    def __repr__(self):
        return f'<SparseFeature {self.name} with Embedding shape ({self.vocab_size}, {self.embed_dim})>'

    def get_embedding_layer(self):
        if not hasattr(self, 'embed'):
            self.embed = self.initializer(self.vocab_size, self.embed_dim)
        return self.embed


class SequenceFeature(SparseFeature):
    """Multi-hot / behaviour-sequence column (reference `basic/features.py:5-46`): ids of shape (batch, seq_len), padded;
    `pooling` in "sum" / "mean" / "concat".  Positions equal to `padding_idx` (or to -1 when no padding_idx is given) are
    masked out of the pooled embedding (`basic/layers.py:137-140`).  The lookup is one pooled-gather launch
    (csrc/embed_bag.hip)."""

    def __init__(self, name, vocab_size, embed_dim=None, pooling="mean", shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001), hash_seed=0):
        super().__init__(name, vocab_size, embed_dim, shared_with, padding_idx, initializer, hash_seed)
        self.pooling = pooling

    def __repr__(self):
        return f'<SequenceFeature {self.name} with Embedding shape ({self.vocab_size}, {self.embed_dim})>'


class DenseFeature(object):
    """One numeric column, passed through as a single float (embed_dim is fixed to 1)."""

    def __init__(self, name):
        self.name = name
        self.embed_dim = 1

    def __repr__(self):
        return f'<DenseFeature {self.name}>'
