#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (the reference lives at /root/reference and
never travels):

    python tests/golden/make_golden.py

For every in-scope model family it builds the reference model on a small
schema, re-initialises it so that outputs are input-sensitive (the default
N(0, 1e-4) embeddings make every model almost constant),
and records, per case `<name>.npz`:

    meta                      json: family, hyper-parameters, feature schema
    x<s>/<col>, y<s>          inputs of training step s = 0, 1, 2
    state0/<key>              full state_dict before training
    eval_probs                model.eval() forward of x0 on state0
    train_probs, loss0        first training forward (batch statistics)
    grad/<key>                parameter gradients of the first step
                              (absent key = grad None, e.g. PPNet agn tables)
    state1/<key>, state3/<key>  state_dict after 1 and 3 Adam(1e-3, wd 1e-5) steps
    losses                    the three training losses
    eval3_probs               model.eval() forward of x0 after the third step

Only DATA is written: inputs and the reference's outputs.  No reference source
is copied.  The files are consumed by tests/test_oracle_golden.py (oracle vs
reference) and tests/test_parity_gpu.py (HIP path vs the same vectors).
"""
import copy
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from scenario_wise_rec.basic.features import DenseFeature, SequenceFeature, SparseFeature  # noqa: E402  (reference)
from scenario_wise_rec.models.multi_domain import (  # noqa: E402  (reference)
    MMOE, PLE, EPNet, HamurLarge, HamurSmall, M3oE, PPNet, SharedBottom, Star)

OUT = os.path.dirname(os.path.abspath(__file__))
LR, WD = 1e-3, 1e-5


def make_schema(rng, n_sparse, embed_dim, n_dense, vocab_hi=200, prefix="s"):
    vocabs = [2, 3, vocab_hi] + [int(v) for v in rng.integers(4, vocab_hi, size=max(0, n_sparse - 3))]
    feats = [{"kind": "sparse", "name": f"{prefix}{i}", "vocab_size": vocabs[i], "embed_dim": embed_dim}
             for i in range(n_sparse)]
    feats += [{"kind": "dense", "name": f"{prefix}d{i}"} for i in range(n_dense)]
    return feats


def build_features(schema):
    out = []
    for f in schema:
        if f["kind"] == "sparse":
            out.append(SparseFeature(f["name"], vocab_size=f["vocab_size"], embed_dim=f["embed_dim"], shared_with=f.get("shared_with")))
        elif f["kind"] == "sequence":
            out.append(SequenceFeature(f["name"], vocab_size=f["vocab_size"], embed_dim=f["embed_dim"], pooling=f["pooling"],
                                       shared_with=f.get("shared_with"), padding_idx=f.get("padding_idx")))
        else:
            out.append(DenseFeature(f["name"]))
    return out


def make_batch(rng, schemas, B, domain_num, idx_dtype=np.int64, dense_dtype=np.float32,
               dom_dtype=np.int64, empty_domain=None, out_of_range=False):
    x = {}
    for schema in schemas:
        for f in schema:
            if f["kind"] == "sparse":
                x[f["name"]] = rng.integers(0, f["vocab_size"], size=B).astype(
                    idx_dtype if f["vocab_size"] <= np.iinfo(idx_dtype).max else np.int64)
            elif f["kind"] == "sequence":
                # padded id sequences [B, L]: random lengths 0..L (some rows empty), the tail filled with the padding id
                L, pad = f["seq_len"], f.get("padding_idx")
                ids = rng.integers(1 if pad == 0 else 0, f["vocab_size"], size=(B, L))
                if pad is not None:
                    lens = rng.integers(0, L + 1, size=B)
                    ids[np.arange(L)[None, :] >= lens[:, None]] = pad
                x[f["name"]] = ids.astype(idx_dtype if f["vocab_size"] <= np.iinfo(idx_dtype).max else np.int64)
            else:
                x[f["name"]] = rng.random(B).astype(dense_dtype)
    dom = rng.integers(0, domain_num, size=B)
    if empty_domain is not None:
        dom[dom == empty_domain] = (empty_domain + 1) % domain_num
    if out_of_range:
        dom[::7] = domain_num          # id == D  -> probability 0.0 (mmoe.py:53-55)
        dom[3::11] = -1
    x["domain_indicator"] = dom.astype(dom_dtype)
    y = (rng.random(B) < 0.3).astype(np.float32)
    return x, y


def perturb(model, seed):
    """Input-sensitive re-initialisation (deterministic)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "embed_dict" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
            elif name.startswith(("u.", "v.")):                     # HAMUR U,V (all-ones by default)
                p.copy_(torch.rand(p.shape, generator=g) * 0.3 + 0.1)
            elif name.startswith(("share_parm_b", "domain_specific_b.")):  # STAR biases, U(0,1) by default
                p.copy_(torch.rand(p.shape, generator=g) * 0.98 + 0.02)
            elif name.startswith("b_list") or name in ("bias1", "bias2", "dn_share_bias") \
                    or "dn_bias" in name or (name.endswith(".bias") and p.dim() == 1):
                # Biases that feed a BatchNorm (or a domain norm) have a mathematically ZERO gradient:
                # the reference's value is summation-order noise (~1e-9) and Adam's update is then
                # driven by weight_decay * p.  Keep |p| >= 0.02 so that term (>= 2e-7) dominates the
                # noise and the trajectory is well defined; BN betas are re-drawn just below.
                sign = (torch.rand(p.shape, generator=g) < 0.5).float() * 2 - 1
                p.copy_(sign * (torch.rand(p.shape, generator=g) * 0.13 + 0.02))
            elif name in ("gamma1", "gamma2", "dn_share_gamma") or "dn_gamma" in name:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif name.startswith(("slot_bias", "shared_bias")):                 # M3oE STAR-front biases (zeros by default)
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for mod_name, mod in model.named_modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.2)
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def to_torch(x):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in x.items()}


def run_case(name, family, ctor, hyper, schemas, domain_num, B=250, seed=0, **batch_kw):
    rng = np.random.default_rng(1000 + seed)
    torch.manual_seed(seed)
    model = ctor()
    perturb(model, seed + 17)
    rec = {"meta": json.dumps({"family": family, "hyper": hyper, "schemas": schemas,
                               "domain_num": domain_num, "B": B, "lr": LR, "weight_decay": WD})}
    batches = [make_batch(rng, schemas, B, domain_num, **batch_kw) for _ in range(3)]
    for s, (x, y) in enumerate(batches):
        for k, v in x.items():
            rec[f"x{s}/{k}"] = v
        rec[f"y{s}"] = y
    for k, v in sd_numpy(model).items():
        rec[f"state0/{k}"] = v

    x0, y0 = batches[0]
    model.eval()
    with torch.no_grad():
        rec["eval_probs"] = model(to_torch(x0)).numpy().copy()

    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
    crit = torch.nn.BCELoss()
    losses = []
    model.train()
    for s, (x, y) in enumerate(batches):
        p = model(to_torch(x))
        loss = crit(p, torch.from_numpy(y).float())
        model.zero_grad()
        loss.backward()
        if s == 0:
            rec["train_probs"] = p.detach().numpy().copy()
            rec["loss0"] = np.float32(loss.item())
            for k, prm in model.named_parameters():
                if prm.grad is not None:
                    rec[f"grad/{k}"] = prm.grad.numpy().copy()
        opt.step()
        losses.append(loss.item())
        if s == 0:
            for k, v in sd_numpy(model).items():
                rec[f"state1/{k}"] = v
    for k, v in sd_numpy(model).items():
        rec[f"state3/{k}"] = v
    rec["losses"] = np.asarray(losses, dtype=np.float32)
    model.eval()
    with torch.no_grad():
        rec["eval3_probs"] = model(to_torch(x0)).numpy().copy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  losses={losses}")
    return model, batches


def run_dp_case(name, ctor, hyper, schemas, domain_num, n_shards, B=256, seed=0):
    """nn.DataParallel semantics (`ctr_trainer.py:45-47`) emulated on CPU: each
    replica runs its row shard with ITS OWN batch statistics, the loss is the
    mean over the global batch, gradients are summed, replica 0's BN buffers
    persist.  Records the summed gradients and the state after one Adam step."""
    rng = np.random.default_rng(2000 + seed)
    torch.manual_seed(seed)
    model = ctor()
    perturb(model, seed + 17)
    x, y = make_batch(rng, schemas, B, domain_num)
    rec = {"meta": json.dumps({"family": "MMOE", "hyper": hyper, "schemas": schemas, "domain_num": domain_num,
                               "B": B, "lr": LR, "weight_decay": WD, "n_shards": n_shards})}
    for k, v in x.items():
        rec[f"x0/{k}"] = v
    rec["y0"] = y
    for k, v in sd_numpy(model).items():
        rec[f"state0/{k}"] = v
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
    model.train()
    model.zero_grad()
    sh = B // n_shards
    replicas = [model] + [copy.deepcopy(model) for _ in range(n_shards - 1)]
    probs = []
    for r, rep in enumerate(replicas):
        xs = {k: v[r * sh:(r + 1) * sh] for k, v in x.items()}
        p = rep(to_torch(xs))
        loss = torch.nn.functional.binary_cross_entropy(p, torch.from_numpy(y[r * sh:(r + 1) * sh]), reduction="sum") / B
        loss.backward()
        probs.append(p.detach().numpy())
    for rep in replicas[1:]:
        for (k, p0), (_, pr) in zip(model.named_parameters(), rep.named_parameters()):
            if pr.grad is not None:
                p0.grad = pr.grad.clone() if p0.grad is None else p0.grad + pr.grad
    rec["train_probs"] = np.concatenate(probs)
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            rec[f"grad/{k}"] = prm.grad.numpy().copy()
    opt.step()
    for k, v in sd_numpy(model).items():
        rec[f"state1/{k}"] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    rng = np.random.default_rng(7)

    # ---- MMoE, KuaiRand-like: 5 domains, 4 experts, E=16 (BASELINE config 2 in miniature)
    sch = make_schema(rng, 6, 16, 3)
    hyper = {"domain_num": 5, "n_expert": 4, "expert_params": {"dims": [32]}, "tower_params": {"dims": [16]}}
    ctor = lambda: MMOE(build_features(sch), **hyper)
    run_case("mmoe", "MMOE", ctor, hyper, [sch], 5, seed=1)
    run_case("mmoe_empty_domain", "MMOE", ctor, hyper, [sch], 5, seed=2, empty_domain=3)
    run_case("mmoe_out_of_range_domain", "MMOE", ctor, hyper, [sch], 5, seed=3, out_of_range=True)
    run_case("mmoe_narrow_dtypes", "MMOE", ctor, hyper, [sch], 5, seed=4,
             idx_dtype=np.int16, dense_dtype=np.float16, dom_dtype=np.int8)
    run_case("mmoe_b1000", "MMOE", ctor, hyper, [sch], 5, B=1000, seed=5)
    run_dp_case("mmoe_dp2", ctor, hyper, [sch], 5, n_shards=2, seed=6)
    run_dp_case("mmoe_dp8", ctor, hyper, [sch], 5, n_shards=8, seed=7)
    # SequenceFeature pooled lookups (`basic/layers.py:73-87`): mean with padding id 0 sharing the table of sparse s2,
    # sum with its own table and padding id 3, mean without padding_idx (every position counts), next to plain features
    # (pooling="concat" yields [B, L*E] per feature, which no in-scope model's input_dim = sum(embed_dim) accepts)
    sch_seq = sch + [
        {"kind": "sequence", "name": "h_mean", "vocab_size": sch[2]["vocab_size"], "embed_dim": 16, "pooling": "mean",
         "shared_with": "s2", "padding_idx": 0, "seq_len": 7},
        {"kind": "sequence", "name": "h_sum", "vocab_size": 50, "embed_dim": 16, "pooling": "sum", "padding_idx": 3, "seq_len": 5},
        {"kind": "sequence", "name": "h_all", "vocab_size": 30, "embed_dim": 16, "pooling": "mean", "seq_len": 4}]
    run_case("mmoe_seq", "MMOE", lambda: MMOE(build_features(sch_seq), **hyper), hyper, [sch_seq], 5, seed=9)
    # two-layer experts and towers (stacked BN blocks)
    hyper2 = {"domain_num": 3, "n_expert": 3, "expert_params": {"dims": [24, 12]}, "tower_params": {"dims": [8, 4]}}
    run_case("mmoe_deep", "MMOE", lambda: MMOE(build_features(sch), **hyper2), hyper2, [sch], 3, seed=8)

    # ---- SharedBottom, MovieLens-like: 3 domains, E=8 (config 1 in miniature)
    sch1 = make_schema(rng, 6, 8, 1)
    hyper = {"domain_num": 3, "bottom_params": {"dims": [128]}, "tower_params": {"dims": [8]}}
    run_case("sharedbottom", "SharedBottom", lambda: SharedBottom(build_features(sch1), **hyper), hyper, [sch1], 3, seed=11)

    # ---- PLE, Mind-like: 4 domains, E=32, no dense (config 4 in miniature) + a 2-level variant
    sch4 = make_schema(rng, 3, 32, 0)
    hyper = {"domain_num": 4, "n_level": 1, "n_expert_specific": 2, "n_expert_shared": 1,
             "expert_params": {"dims": [64, 32]}, "tower_params": {"dims": [16]}}
    run_case("ple", "PLE", lambda: PLE(build_features(sch4), **hyper), hyper, [sch4], 4, seed=21)
    hyper_l2 = {"domain_num": 3, "n_level": 2, "n_expert_specific": 2, "n_expert_shared": 2,
                "expert_params": {"dims": [16]}, "tower_params": {"dims": [8]}}
    run_case("ple_2level", "PLE", lambda: PLE(build_features(sch4), **hyper_l2), hyper_l2, [sch4], 3, seed=22)

    # ---- STAR, Ali-CCP-like: 3 domains, E=16 (config 3 in miniature)
    sch3 = make_schema(rng, 5, 16, 4)
    hyper = {"num_domains": 3, "fcn_dims": [64, 32, 16, 8], "aux_dims": [16]}
    run_case("star", "Star", lambda: Star(build_features(sch3), **hyper), hyper, [sch3], 3, seed=31)

    # ---- PPNet / EPNet: id + agnostic feature groups (config 5 in miniature)
    sch_id = make_schema(rng, 2, 16, 0, prefix="id")
    sch_agn = make_schema(rng, 4, 16, 2, prefix="ag")
    hyper = {"domain_num": 4, "fcn_dims": [32, 16, 8]}
    run_case("ppnet", "PPNet",
             lambda: PPNet(build_features(sch_id), build_features(sch_agn), **hyper), hyper, [sch_id, sch_agn], 4, seed=41)
    sch_sce = [{"kind": "sparse", "name": "domain_indicator_f", "vocab_size": 4, "embed_dim": 16}]
    hyper = {"fcn_dims": [32, 16]}
    run_case("epnet", "EPNet",
             lambda: EPNet(build_features(sch_sce), build_features(sch_agn), **hyper), hyper, [sch_sce, sch_agn], 4, seed=42)

    # ---- HAMUR small / large
    sch5 = make_schema(rng, 4, 16, 2)
    hyper = {"domain_num": 3, "fcn_dims": [48, 24], "hyper_dims": [16], "k": 5}
    run_case("hamur_small", "HamurSmall",
             lambda: HamurSmall(build_features(sch5), 3, [48, 24], [16], 5), hyper, [sch5], 3, seed=51)
    hyper = {"domain_num": 2, "fcn_dims": [64, 48, 40, 32, 24, 16, 12], "hyper_dims": [16], "k": 4}
    run_case("hamur_large", "HamurLarge",
             lambda: HamurLarge(build_features(sch5), 2, [64, 48, 40, 32, 24, 16, 12], [16], 4), hyper, [sch5], 2, seed=52)

    # ---- M3oE: STAR front + MMoE body + domain experts, LayerNorm blocks
    sch6 = make_schema(rng, 5, 16, 2)
    hyper = {"domain_num": 4, "fcn_dims": [64, 32, 32, 16], "expert_num": 3, "exp_d": 1, "exp_t": 1, "bal_d": 1, "bal_t": 1}
    run_case("m3oe", "M3oE", lambda: M3oE(build_features(sch6), **hyper), hyper, [sch6], 4, seed=61)
    hyper = {"domain_num": 3, "fcn_dims": [48, 24, 24, 20, 12], "expert_num": 2, "exp_d": 0.5, "exp_t": 1, "bal_d": -0.3, "bal_t": 1}
    run_case("m3oe_deep_out_of_range", "M3oE", lambda: M3oE(build_features(sch6), **hyper), hyper, [sch6], 3, seed=62,
             out_of_range=True)


if __name__ == "__main__":
    main()
