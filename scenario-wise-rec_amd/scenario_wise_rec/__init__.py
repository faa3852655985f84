"""scenario_wise_rec -- MI355X-native build of the multi-domain CTR training hot path.

Same import paths, class names, constructor signatures, `forward(x_dict)` contract and `state_dict()`
layout as the reference package (Xiaopengli1/Scenario-Wise-Rec), so it drops in for that path; the
compute runs in hand-written HIP kernels (libswr.so, include/swr.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
