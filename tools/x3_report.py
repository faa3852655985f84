#!/usr/bin/env python3
"""Error report of a libswr build over the parity suite: for every golden case (tests/test_parity_gpu.py::test_train_three_steps), the
six BASELINE shapes (tests/test_baseline_shapes_gpu.py) and, with --full, config 2 at full size (test_cfg2_full), the LARGEST logit
error against the reference / oracle probabilities and the largest gradient error as a fraction of the tensor's largest entry --
the numbers behind "adopt the three-product split only where logit error <= 5e-5 and gradient error <= 1e-4" (VERDICT round 5, item 6).
The comparison functions are wrapped, the tests run unchanged; a test that fails its own tolerance is reported as FAIL with the
numbers seen until then.   usage (GPU box): [SWR_LIB=.../libswr_x3.so] python tools/x3_report.py [--full]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "scenario-wise-rec_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import _golden

REC = {"logit": 0.0, "grad": 0.0, "grad_name": ""}
_real_probs = _golden.assert_probs_close
_real_allclose = np.testing.assert_allclose


def probs(got, want, tol=1e-4):
    want = np.asarray(want)
    nz = want != 0.0
    if nz.any():
        e = np.abs(_golden.logit(np.asarray(got)[nz]) - _golden.logit(want[nz].astype(np.float64)))
        w = want[nz].astype(np.float64)
        e = np.maximum(0.0, e - 2 * 6e-8 / np.minimum(w, 1 - w))         # (minus the fp32 quantisation of a saturated probability)
        REC["logit"] = max(REC["logit"], float(e.max()))
    return _real_probs(got, want, tol)


def allclose(actual, desired, rtol=1e-7, atol=0, err_msg="", **kw):
    a, d = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    if rtol == 0 and a.shape == d.shape and a.size and ("weight" in err_msg or "bias" in err_msg or err_msg.startswith("grad") or "." in err_msg):
        top = max(1e-6, float(np.abs(d).max()))
        r = float(np.abs(a - d).max()) / top
        if r > REC["grad"]:
            REC["grad"], REC["grad_name"] = r, err_msg
    return _real_allclose(actual, desired, rtol=rtol, atol=atol, err_msg=err_msg, **kw)


_golden.assert_probs_close = probs
np.testing.assert_allclose = allclose
import test_baseline_shapes_gpu as TB      # noqa: E402
import test_parity_gpu as TP               # noqa: E402
TP.assert_probs_close = probs
TB.assert_probs_close = probs


def run(label, fn):
    REC.update(logit=0.0, grad=0.0, grad_name="")
    status = "ok"
    try:
        fn()
    except Exception as e:                 # noqa: BLE001
        status = "FAIL " + (str(e).strip().splitlines() or [type(e).__name__])[0][:90]
        if os.environ.get("X3_TRACE"):
            traceback.print_exc()
    print(f"{label:34s} logit {REC['logit']:.2e}  grad/max {REC['grad']:.2e} ({REC['grad_name'][:40]})  {status}", flush=True)


print("library:", os.environ.get("SWR_LIB", "(product build)"))
for name in TP.SINGLE:
    run("golden " + name, lambda: TP.test_train_three_steps(name))
for n, cap, batch in ((1, 8000, 4096), (2, 20000, 4096), (3, 20000, 4096), (4, 12000, 4096), (5, 6000, 1024), (6, 6000, 2048)):
    run(f"baseline shape cfg{n} b{batch}", lambda: TB.run_config(TB.small_config(n, cap, batch)))
if "--full" in sys.argv:
    import test_full_size_gpu as TF
    TF.assert_probs_close = probs
    run("cfg2 full size (B 65536)", TF.test_cfg2_full)
    run("cfg5 shard step (B 32768)", TF.test_cfg5_shard_step_gradients)
    run("cfg6 shard step (B 32768)", TF.test_cfg6_shard_step_gradients)
