"""In-scope multi-domain CTR models (reference: `models/multi_domain/__init__.py:1-12`).
AdaSparse, Sarnet, M2M, AdaptDHM and M3oE are outside the hot path."""
from .star import Star
from .mmoe import MMOE
from .ple import PLE
from .sharebottom import SharedBottom
from .epnet import EPNet
from .ppnet import PPNet
from .hamur import HamurLarge, HamurSmall
