"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/swr.h
declares, its structs have the layout the Python binding assumes, and the host-side mirror keeps the
reference's module API (state_dict keys / shapes) -- no device compute."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from _golden import Case, case_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "swr.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from scenario_wise_rec import _hip as H
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(H.lib, n), f"libswr.so does not export {n}"
        assert n in H.EXPORTS, f"{n} is declared in swr.h but not bound in _hip.py"
    assert set(H.EXPORTS) == set(names)


def test_abi_version_and_status_strings():
    from scenario_wise_rec import _hip as H
    assert H.lib.swr_abi_version() == H.ABI_VERSION == 8
    assert H.lib.swr_status_str(0) == b"ok"
    assert b"workspace" in H.lib.swr_status_str(-6)
    assert H.lib.swr_device_available() in (0, 1)


def test_struct_layouts_match_the_header():
    """Compile a C program against include/swr.h and compare sizeof / offsetof with the ctypes mirrors."""
    from scenario_wise_rec import _hip as H
    structs = {"swr_sparse_slot": H.SparseSlot, "swr_dense_slot": H.DenseSlot, "swr_embed_grad_slot": H.EmbedGradSlot,
               "swr_tower_args": H.TowerArgs, "swr_bnmix_args": H.BnMixArgs,
               "swr_gemm_args": H.GemmArgs, "swr_gemm_tn_args": H.GemmTnArgs, "swr_act_range": H.ActRange,
               "swr_mix_desc": H.MixDesc, "swr_adam_hyper": H.AdamHyper, "swr_adam_table": H.AdamTable,
               "swr_dp_table": H.DpTable, "swr_star_layer_args": H.StarLayerArgs,
               "swr_take_column": H.TakeColumn, "swr_layernorm_args": H.LayerNormArgs,
               "swr_onehot_table": H.OnehotTable, "swr_fl_piece": H.FlPiece, "swr_fl_plan": H.FlPlan,
               "swr_fl_offsets": H.FlOffsets}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "swr.h"', 'int main(void){']
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(l.split() for l in out.strip().splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


@pytest.mark.parametrize("name", [n for n in case_names() if "_dp" not in n and "mmoe_" not in n])
def test_state_dict_layout_is_the_reference_one(name):
    """Same keys, shapes and dtypes as the reference's state_dict (checkpoints interchange)."""
    from _golden import build_product_model
    c = Case(name)
    model = build_product_model(c, device="cpu")
    want = c.group("state0")
    got = model.state_dict()
    assert list(got) == list(want)                      # same ORDER too
    for k, v in want.items():
        assert tuple(got[k].shape) == tuple(v.shape), k
        assert np.array_equal(got[k].numpy(), v), k


def test_arena_lays_fused_layers_back_to_back():
    from _golden import build_product_model
    from scenario_wise_rec import ops
    model = build_product_model(Case("mmoe"), device="cpu")
    model.build_arena()
    a = model.arena()
    assert a is not None and a["p"].numel() >= sum(p.numel() for p in model.parameters())
    ws = [e.block(0)[0].weight for e in model.experts] + [g.block(0)[0].weight for g in model.gates]
    cat = ops._cat_params(ws)
    assert cat.data_ptr() == ws[0].data_ptr() and cat.shape[0] == sum(w.shape[0] for w in ws)   # zero-copy
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.build_arena()                                  # idempotent
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    for p in model.parameters():                         # gradient views shadow the parameters
        assert p.grad is not None and p.grad.shape == p.shape
    sd = {k: torch.randn_like(v) if v.dtype.is_floating_point else v for k, v in before.items()}
    model.load_state_dict(sd)                            # in-place: the views survive
    assert model.arena() is not None


def test_cpu_tensors_are_rejected_loudly():
    from _golden import build_product_model, to_device
    from scenario_wise_rec._hip import SwrError
    c = Case("mmoe")
    model = build_product_model(c, device="cpu")
    with pytest.raises(SwrError, match="no CPU fallback"):
        model(to_device(c.batch(0)[0], "cpu"))


def test_hamur_mutates_hyper_dims_like_the_reference():
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.models.multi_domain import HamurSmall
    hd = [16]
    HamurSmall([SparseFeature("a", 5, 8)], 2, [24, 12], hd, 3)
    assert hd == [16, 9]                                 # hamur.py:288 appends k*k in place


def test_epnet_mlp_collapses_to_one_linear():
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.models.multi_domain import EPNet
    m = EPNet([SparseFeature("d", 3, 8)], [SparseFeature("a", 5, 8)], [32, 16])
    assert [k for k in m.state_dict() if k.startswith("mlp.")] == ["mlp.mlp.0.weight", "mlp.mlp.0.bias"]   # epnet.py:21


def test_evaluate_multi_domain_loss_host_logic():
    """Per-domain log-loss / AUC bookkeeping of `evaluate_multi_domain_loss` (ctr_trainer.py:113-152) against
    sklearn called directly; an empty domain gives None."""
    from sklearn.metrics import log_loss, roc_auc_score
    from scenario_wise_rec.trainers import CTRTrainer
    rng = np.random.default_rng(0)
    n, D = 400, 4
    dom = rng.integers(0, 3, size=n)                     # domain 3 is empty
    p = rng.random(n).astype(np.float32) * 0.98 + 0.01
    y = (rng.random(n) < p).astype(np.float32)

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return x["p"]

    tr = CTRTrainer(Stub(), "t", optimizer_fn=torch.optim.SGD, optimizer_params={"lr": 0.1}, device="cpu")
    batches = [({"p": torch.from_numpy(p[i:i + 100]), "domain_indicator": torch.from_numpy(dom[i:i + 100])},
                torch.from_numpy(y[i:i + 100])) for i in range(0, n, 100)]
    dl, da, tl, ta = tr.evaluate_multi_domain_loss(tr.model, batches, D)
    assert dl[3] is None and da[3] is None
    for d in range(3):
        assert dl[d] == pytest.approx(log_loss(y[dom == d], p[dom == d]))
        assert da[d] == pytest.approx(roc_auc_score(y[dom == d], p[dom == d]))
    assert tl == pytest.approx(log_loss(y, p)) and ta == pytest.approx(roc_auc_score(y, p))
    auc, ll = tr.evaluate(tr.model, batches)
    assert auc == pytest.approx(ta) and ll == pytest.approx(tl)
    assert tr.predict(tr.model, batches) == pytest.approx(p.tolist())


def test_bench_configs_have_the_baseline_shapes():
    """bench.py's configurations are the BASELINE.json shapes: K0 = F_s * E + F_d, domain counts,
    per-GPU batches; synthetic batches are reproducible and within the vocabularies."""
    sys.path.insert(0, ROOT)
    import bench
    want = {1: (49, 3, 4096), 2: (516, 5, 65536), 3: (376, 3, 131072 // 8), 4: (96, 4, 65536 // 8), 5: (388, 8, 262144 // 8),
            6: (452, 8, 262144 // 8)}
    for n, (k0, D, B) in want.items():
        cfg = bench.CONFIGS[n]
        assert len(cfg["vocabs"]) * cfg["embed_dim"] + cfg["n_dense"] == k0, n
        assert len(cfg["domain_shares"]) == D and cfg["batch"] == B, n
    assert bench.gather_bytes_per_sample(bench.CONFIGS[2]) == 4384
    cfg = dict(bench.CONFIGS[2], vocabs=[1000, 50000, 8, 2])
    x1, y1 = bench.synth_batch(cfg, 512, seed=3)
    x2, y2 = bench.synth_batch(cfg, 512, seed=3)
    assert all(np.array_equal(x1[k], x2[k]) for k in x1) and np.array_equal(y1, y2)
    for i, v in enumerate(cfg["vocabs"]):
        assert x1[f"s{i}"].min() >= 0 and x1[f"s{i}"].max() < v
    assert set(np.unique(x1["domain_indicator"])) <= set(range(5))


def test_ctrtrainer_gpus_without_a_launcher_says_how_to_launch(monkeypatch):
    """`CTRTrainer(gpus=[0, 1])` (reference: nn.DataParallel, ctr_trainer.py:45-47) outside `torch.distributed.run` raises
    with the command line to use instead of silently training on one GPU; the row chunk of a rank is torch.chunk's."""
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.models.multi_domain import MMOE
    from scenario_wise_rec.trainers import CTRTrainer
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    feats = [SparseFeature("s0", 5, 4), DenseFeature("d0")]
    model = MMOE(feats, 2, 2, {"dims": [4]}, {"dims": [4]})
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        CTRTrainer(model, "x", gpus=[0, 1])
    tr = CTRTrainer.__new__(CTRTrainer)
    tr._world, tr._rank = 4, 2
    xs, ys = tr._my_rows({"a": torch.arange(12)}, torch.arange(12))
    assert xs["a"].tolist() == [6, 7, 8] and ys.tolist() == torch.arange(12).chunk(4)[2].tolist()
    with pytest.raises(ValueError, match="split evenly"):
        tr._my_rows({"a": torch.arange(10)}, torch.arange(10))


def test_no_kernel_spills_registers():
    """tools/check_scratch.py: hipcc -Rpass-analysis=kernel-resource-usage over every csrc/*.hip -- no kernel of the library may
    have ScratchSize > 0 (a spilled kernel moves its registers through HBM: bnmix_bwd lost 40 % to 108 bytes per lane in round 3;
    gemm_rows_x6<8>, layernorm_bwd<4> and bnmix_bwd<8, 5> did until round 5).  Cross-compiles without a GPU (~3 min, 4 jobs)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_scratch.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
