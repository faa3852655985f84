"""In-scope multi-domain CTR models (reference: `models/multi_domain/__init__.py:1-12`).
AdaSparse, Sarnet, M2M and AdaptDHM are outside the hot path; M3oE is row f4 of SURVEY.md 8."""
from .star import Star
from .mmoe import MMOE
from .ple import PLE
from .sharebottom import SharedBottom
from .epnet import EPNet
from .ppnet import PPNet
from .hamur import HamurLarge, HamurSmall
from .m3oe import M3oE
