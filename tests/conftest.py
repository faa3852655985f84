import os
import sys

import pytest

# The product decides by batch size whether a step forks side streams (ops.SIDE_MIN_BATCH: below 32 768 rows one stream is faster).
# The tests run mostly small batches and exist to exercise the multi-stream step -- forks, joins, captured branches, the skew
# harness -- so they force the forks on; tests/test_graph_gpu.py::test_side_streams_by_batch_size covers the automatic choice and
# the single-stream step.
os.environ.setdefault("SWR_SIDE_STREAM", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "scenario-wise-rec_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Family-level golden parity first, component tests next, multi-process (data-parallel) tests last: under `-x` a
# failure in a later group cannot hide the results of an earlier one.
_ORDER = ("test_parity_gpu", "test_baseline_shapes_gpu", "test_ops_gpu", "test_lazy_adam_gpu", "test_graph_gpu")


def _group(item):
    name = item.fspath.basename[:-3]
    if name == "test_parallel":
        return len(_ORDER) + 1
    return _ORDER.index(name) if name in _ORDER else len(_ORDER)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_group)          # stable: the order inside a file is kept
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
