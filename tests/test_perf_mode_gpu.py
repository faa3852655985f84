"""The bf16 perf mode (SWR_GEMM=bf16: one bf16 MFMA product per k-group instead of the six that give fp32-class accuracy):
not the parity path -- SURVEY.md fact 5 -- so what is pinned here is that its error is MEASURED and of the expected size:
well outside the 1e-4 logit tolerance, well inside what bf16 operands allow (K ~ 500 products of ~2^-9 relative error).
The mode is read once per process (csrc/gemm.hip gemm_mode), hence the subprocess."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import json, sys
sys.path[:0] = [%(root)r, %(root)r + "/scenario-wise-rec_amd", %(root)r + "/tests"]
import numpy as np, torch
import bench
from _golden import logit, perturb_product
from test_baseline_shapes_gpu import oracle_for, small_config
from scenario_wise_rec import _hip as H
cfg = small_config(2, 20000, 4096)
model, feats = bench.build_model(cfg, seed=11)
perturb_product(model, 23)
state0 = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
x, y = bench.synth_batch(cfg, cfg["batch"], seed=5)
model.cuda().train()
p = model({k: torch.from_numpy(v).cuda() for k, v in x.items()}).detach().cpu().numpy()
torch.cuda.synchronize(); H.check_errors()
op, _l, _g = oracle_for(cfg, state0).loss_and_grads(x, y)
print(json.dumps({"mode": H.lib.swr_gemm_precision_mode(), "max_logit_err": float(np.abs(logit(p) - logit(op)).max())}))
'''


def _run(mode):
    env = dict(os.environ)
    env.pop("SWR_GEMM", None)
    if mode:
        env["SWR_GEMM"] = mode
    out = subprocess.run([sys.executable, "-c", PROBE % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bf16_perf_mode_error_is_what_bf16_operands_give():
    exact, fast = _run(None), _run("bf16")
    assert exact["mode"] == 1 and fast["mode"] == 2
    assert exact["max_logit_err"] < 1e-4                       # the parity path (six products)
    assert 2e-4 < fast["max_logit_err"] < 5e-2, fast          # KuaiRand MMoE widths, K0 = 516: ~1e-3 .. 1e-2
    print(f"bf16 perf mode: max logit error {fast['max_logit_err']:.3e} (parity path {exact['max_logit_err']:.3e})")
