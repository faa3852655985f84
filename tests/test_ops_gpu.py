"""Op-level parity of the HIP kernels (through the C ABI via scenario_wise_rec.ops) against numpy /
the oracle, on seeded inputs.  Integer / copy work is bit-exact; fp32 products are compared with an
fp64 reference at fp32-roundoff tolerances written next to each check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda") if dtype is None else \
        torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


@pytest.mark.parametrize("B,dims,n_dense,idx_dtype", [
    (250, [16, 16, 16], 2, np.int64), (1000, [8, 8], 1, np.int16), (64, [32], 0, np.int32),
    (333, [6, 10, 3], 3, np.int64),          # dims not multiples of 4: scalar path
    (1, [16], 1, np.int64), (4097, [64, 64], 4, np.int8)])
def test_gather_bit_exact(B, dims, n_dense, idx_dtype):
    """K1 is a pure copy: bit-exact against numpy fancy indexing, sparse block first, dense last."""
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(B)
    vocabs = [int(min(rng.integers(2, 300), np.iinfo(idx_dtype).max)) for _ in dims]
    feats = [DenseFeature("d0")] if n_dense else []      # dense first in the list: output still puts it last
    feats += [SparseFeature(f"s{i}", v, d) for i, (v, d) in enumerate(zip(vocabs, dims))]
    feats += [DenseFeature(f"d{i}") for i in range(1, n_dense)]
    layer = EmbeddingLayer(feats)
    tables = {f"s{i}": rng.standard_normal((v, d)).astype(np.float32) for i, (v, d) in enumerate(zip(vocabs, dims))}
    with torch.no_grad():
        for k, t in tables.items():
            layer.embed_dict[k].weight.copy_(torch.from_numpy(t))
    layer.to("cuda")
    x = {f"s{i}": rng.integers(0, v, size=B).astype(idx_dtype) for i, v in enumerate(vocabs)}
    x.update({f"d{i}": rng.random(B).astype(np.float16 if i % 2 else np.float32) for i in range(n_dense)})
    out = layer({k: _dev(v) for k, v in x.items()}, feats, squeeze_dim=True).detach().cpu().numpy()
    want = np.concatenate([tables[f"s{i}"][x[f"s{i}"].astype(np.int64)] for i in range(len(dims))] +
                          [x[f"d{i}"].astype(np.float32)[:, None] for i in range(n_dense)], axis=1)
    assert out.shape == want.shape
    assert np.array_equal(out, want)


def test_gather_out_of_range_index_raises():
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    f = [SparseFeature("s0", 10, 8)]
    layer = EmbeddingLayer(f).to("cuda")
    layer({"s0": torch.tensor([1, 2, 10], device="cuda")}, f, squeeze_dim=True)
    with pytest.raises(IndexError):
        H.check_errors()


@pytest.mark.parametrize("B,dim,vocabs,limit", [(250, 16, [2, 7, 200], 1 << 30), (1000, 8, [3, 500], 1 << 30),
                                                (777, 16, [5, 4000, 9000], 64 * 1024),     # two sparse-mode tables
                                                (512, 12, [40], 1 << 30), (300, 64, [11, 13], 1 << 30),
                                                (2000, 16, [1000, 3000, 70000], 1 << 30),  # row-range parts; sorted dense
                                                (600, 6, [40, 3000], 1 << 30)])            # columns not 16-byte aligned
@pytest.mark.parametrize("scale", [1e-3, 5e-6])     # 5e-6: every |g| < 2^-13, the direct sums take their one-limb form
def test_embedding_backward(B, dim, vocabs, limit, scale):
    """K3 against np.add.at in fp64.  The fixed-point accumulation is exact to 2^-60, so the only error
    is the final fp32 rounding: rtol 2e-7 of the largest entry."""
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(B + dim)
    feats = [SparseFeature(f"s{i}", v, dim) for i, v in enumerate(vocabs)]
    feats.append(SparseFeature("shared", vocabs[0], dim, shared_with="s0"))       # two slots, one table
    layer = EmbeddingLayer(feats)
    layer.dense_table_limit_bytes = limit
    layer.to("cuda")
    x = {f.name: rng.integers(0, f.vocab_size, size=B) for f in feats}
    x["s0"][: B // 2] = 1                                                          # a long run of one row
    out = layer({k: _dev(v) for k, v in x.items()}, feats, squeeze_dim=True)
    g = rng.standard_normal(out.shape).astype(np.float32) * np.float32(scale)
    out.backward(_dev(g))
    torch.cuda.synchronize()
    for i, f in enumerate(feats[:-1]):
        want = np.zeros((f.vocab_size, dim), np.float64)
        np.add.at(want, x[f.name], g[:, i * dim:(i + 1) * dim].astype(np.float64))
        if i == 0:
            np.add.at(want, x["shared"], g[:, len(vocabs) * dim:(len(vocabs) + 1) * dim].astype(np.float64))
        w = layer.embed_dict[f.name].weight
        if f.vocab_size * dim * 4 > limit:
            urow, ugrad = w._swr_sparse_grad
            urow, ugrad = urow.cpu().numpy(), ugrad.cpu().numpy()
            got = np.zeros_like(want)
            rows = urow[urow >= 0]
            assert len(np.unique(rows)) == len(rows)                               # each row listed once
            got[rows] = ugrad[urow >= 0]
            assert np.all(ugrad[urow < 0] == 0)
            assert w.grad is None or not torch.count_nonzero(w.grad)
        else:
            got = w.grad.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * np.abs(want).max())


@pytest.mark.parametrize("limit", [1 << 30, 1024])
def test_embedding_backward_wide_dynamic_range(limit, monkeypatch):
    """Gradients from 1e-15 to 5e5, signed zeros, a subnormal: the general (three-case) fixed-point conversion and
    the short mid-range one must agree with an fp64 sum; dense (direct / sorted) and sparse-mode tables.  (The exact
    fixed-point paths: SWR_K3_MFMA=0 keeps the 3 000-row table off the fp32-accumulating MFMA segment sums, which are
    tested in test_embedding_backward_mfma_segment_sums.)"""
    monkeypatch.setenv("SWR_K3_MFMA", "0")             # (the default)
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(11)
    B, dim = 3000, 16
    feats = [SparseFeature("a", 5, dim), SparseFeature("b", 3000, dim), SparseFeature("c", 40000, dim)]
    layer = EmbeddingLayer(feats)
    layer.dense_table_limit_bytes = limit
    layer.to("cuda")
    x = {f.name: rng.integers(0, f.vocab_size, size=B) for f in feats}
    mag = 10.0 ** rng.integers(-15, 6, size=(B, 3 * dim))
    g = (rng.standard_normal((B, 3 * dim)) * mag).astype(np.float32)
    g[::7, ::3] = 0.0
    g[1::7, 1::3] = -0.0
    g[5, 5] = 1e-41                                   # subnormal: below the accumulator resolution, counts as 0
    g[6, :] = 1e-3                                    # one all-mid-range row next to the wild ones
    layer(({k: _dev(v) for k, v in x.items()}), feats, squeeze_dim=True).backward(_dev(g))
    torch.cuda.synchronize()
    from scenario_wise_rec import _hip as H
    H.check_errors()
    for i, f in enumerate(feats):
        want = np.zeros((f.vocab_size, dim), np.float64)
        np.add.at(want, x[f.name], g[:, i * dim:(i + 1) * dim].astype(np.float64))
        w = layer.embed_dict[f.name].weight
        if f.vocab_size * dim * 4 > limit:
            urow, ugrad = (t.cpu().numpy() for t in w._swr_sparse_grad)
            got = np.zeros_like(want)
            got[urow[urow >= 0]] = ugrad[urow >= 0]
        else:
            got = w.grad.cpu().numpy()
        # exact integer sums: the result is the fp32 rounding of the true sum (+ B * 2^-60 of truncation)
        np.testing.assert_allclose(got, want, rtol=1.2e-7, atol=B * 2.0 ** -60)


@pytest.mark.parametrize("B", [4096, 40000])
def test_embedding_backward_one_limb_form_is_exact(B):
    """Direct sums, gradients around the 2^-13 switch of the one-limb LDS form (csrc/embed_bwd.hip to_fixed_wide_n): waves
    whose values are all below it add one limb, the others two, into the same accumulators; negative sums carry into the
    high limb at the flush.  Exact integer sums: the result is the fp32 rounding of the fp64 sum."""
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(B)
    dim = 16
    feats = [SparseFeature("a", 7, dim), SparseFeature("b", 300, dim), SparseFeature("c", 1472, dim)]
    layer = EmbeddingLayer(feats).to("cuda")
    x = {f.name: rng.integers(0, f.vocab_size, size=B) for f in feats}
    g = (rng.standard_normal((B, 3 * dim)) * 2e-5).astype(np.float32)
    g[B // 2:] *= np.float32(16.0)                     # second half: some values above 2^-13 in most waves
    g[:, 5] = -np.abs(g[:, 5])                          # one all-negative column (carry of a negative low limb)
    g[::3, 7] = 0.0
    layer(({k: _dev(v) for k, v in x.items()}), feats, squeeze_dim=True).backward(_dev(g))
    torch.cuda.synchronize()
    from scenario_wise_rec import _hip as H
    H.check_errors()
    for i, f in enumerate(feats):
        want = np.zeros((f.vocab_size, dim), np.float64)
        np.add.at(want, x[f.name], g[:, i * dim:(i + 1) * dim].astype(np.float64))
        got = layer.embed_dict[f.name].weight.grad.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1.2e-7, atol=B * 2.0 ** -60)


def test_embedding_backward_flags_out_of_range_gradient():
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    feats = [SparseFeature("a", 5, 8)]
    layer = EmbeddingLayer(feats).to("cuda")
    g = torch.zeros(64, 8, device="cuda")
    g[3, 2] = 3e6                                      # >= 2^20
    layer({"a": torch.arange(64, device="cuda") % 5}, feats, squeeze_dim=True).backward(g)
    with pytest.raises(H.SwrError):
        H.check_errors()


@pytest.mark.parametrize("B,dim,vocabs,limit", [(8192, 16, [4371900, 35], 1 << 20), (8192, 32, [748000, 20000, 300], 1 << 20),
                                                (5000, 16, [70000, 3], 1 << 30),       # sorted DENSE table (stripes), ragged rows
                                                (257, 8, [100000], 1024), (16384, 16, [1 << 22], 1 << 20), (31, 64, [5000], 1024),
                                                (4096, 16, [90000], 1024)])            # (+ the shared slot: a segment of 2 B keys)
def test_embedding_backward_counting_sort_is_bitwise_the_radix_sort(B, dim, vocabs, limit, monkeypatch):
    """Short segments (<= 16 384 keys per table: a strong-scaling shard, config 4's per-GPU batch) are sorted by ONE counting
    launch (csrc/radix_sort.h rank_sort_kernel) instead of two launches per radix pass.  Both sorts are stable, so the row lists,
    the dense gradients and the order of every sum are the same: bitwise, including the run markers of repeated rows."""
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(B + dim + len(vocabs))
    feats = [SparseFeature(f"s{i}", v, dim) for i, v in enumerate(vocabs)]
    feats.append(SparseFeature("shared", vocabs[0], dim, shared_with="s0"))
    layer = EmbeddingLayer(feats)
    layer.dense_table_limit_bytes = limit
    layer.to("cuda")
    x = {f.name: rng.integers(0, f.vocab_size, size=B) for f in feats}
    x["s0"][: B // 3] = x["s0"][0]                                                 # a long run of one row
    x["s0"][B // 3: B // 2] = rng.integers(0, 50, size=B // 2 - B // 3)            # many short runs
    x["shared"][: B // 4] = x["s0"][0]
    xd = {k: _dev(v) for k, v in x.items()}
    g = _dev(rng.standard_normal((B, dim * len(feats))).astype(np.float32) * np.float32(1e-3))
    res = []
    # radix passes | the counting sort in a launch of its own | the counting sort sharing ONE launch with the small tables' direct sums
    # (rank_direct_kernel: the default at these sizes when the lookup has both kinds of tables)
    for mode, fuse in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("SWR_RANK_SORT", mode)
        monkeypatch.setenv("SWR_K3_FUSE", fuse)
        layer.zero_grad()
        for p in layer.parameters():
            p._swr_sparse_grad = None
        layer(xd, feats, squeeze_dim=True).backward(g)
        torch.cuda.synchronize()
        out = []
        for p in layer.parameters():
            sg = getattr(p, "_swr_sparse_grad", None)
            out.append((sg[0].clone(), sg[1].clone()) if sg is not None else (p.grad.clone(),))
        res.append(out)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            for u, v in zip(a, b):
                assert torch.equal(u, v)
    # and against fp64 sums (the sort decides nothing about the values)
    want = np.zeros((vocabs[0], dim), np.float64)
    gh = g.cpu().numpy().astype(np.float64)
    np.add.at(want, x["s0"], gh[:, :dim])
    np.add.at(want, x["shared"], gh[:, len(vocabs) * dim:])
    first = res[2][0]
    if len(first) == 2:
        urow, ugrad = first[0].cpu().numpy(), first[1].cpu().numpy()
        got = np.zeros_like(want)
        got[urow[urow >= 0]] = ugrad[urow >= 0]
    else:
        got = first[0].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * np.abs(want).max())


def test_embedding_backward_is_deterministic():
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(5)
    feats = [SparseFeature("a", 3, 16), SparseFeature("b", 1000, 16)]
    layer = EmbeddingLayer(feats).to("cuda")
    x = {k: _dev(rng.integers(0, v, size=20000)) for k, v in (("a", 3), ("b", 1000))}
    g = _dev(rng.standard_normal((20000, 32)).astype(np.float32))
    res = []
    for _ in range(3):
        layer.zero_grad()
        layer(x, feats, squeeze_dim=True).backward(g)
        res.append([p.grad.clone() for p in layer.parameters()])
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,dim,vocabs", [(5000, 16, [40, 1000, 17, 4096]), (2048, 8, [100, 3]), (4100, 32, [300, 2000]),
                                          (3000, 64, [1000, 33]), (2500, 24, [50]), (70000, 16, [1472, 455])])
def test_embedding_backward_mfma_segment_sums(B, dim, vocabs, monkeypatch):
    """Mid-size dense-gradient tables at batches >= 2048: dEmb = OneHot^T dE on the matrix pipes (csrc/embed_bwd.hip,
    segsum_mfma_kernel).  fp32 accumulation in a fixed order: against np.add.at in fp64 within 1e-6 of the sum of the
    addends' magnitudes, bitwise equal from run to run, and equal to the exact fixed-point path (SWR_K3_MFMA=0) to the
    same bound; a shared table (two lookups) and a run of one row included."""
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    rng = np.random.default_rng(B + dim)
    feats = [SparseFeature(f"s{i}", v, dim) for i, v in enumerate(vocabs)]
    feats.append(SparseFeature("shared", vocabs[0], dim, shared_with="s0"))
    layer = EmbeddingLayer(feats).to("cuda")
    x = {f.name: rng.integers(0, f.vocab_size, size=B) for f in feats}
    x["s0"][: B // 3] = vocabs[0] - 1
    xd = {k: _dev(v) for k, v in x.items()}
    g = (rng.standard_normal((B, (len(vocabs) + 1) * dim)) * 10.0 ** rng.integers(-4, 1, size=(B, 1))).astype(np.float32)
    gd = _dev(g)

    def run():
        layer.zero_grad()
        layer(xd, feats, squeeze_dim=True).backward(gd)
        torch.cuda.synchronize()
        return [layer.embed_dict[f.name].weight.grad.clone() for f in feats[:-1]]
    monkeypatch.setenv("SWR_K3_MFMA", "1")             # (opt-in path: see csrc/embed_bwd.hip seg_enabled)
    got = run()
    again = run()
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    monkeypatch.setenv("SWR_K3_MFMA", "0")
    exact = run()
    for i, f in enumerate(feats[:-1]):
        want = np.zeros((f.vocab_size, dim), np.float64)
        mag = np.zeros((f.vocab_size, dim), np.float64)
        cols = [(x[f.name], slice(i * dim, (i + 1) * dim))]
        if i == 0:
            cols.append((x["shared"], slice(len(vocabs) * dim, (len(vocabs) + 1) * dim)))
        for idx, sl in cols:
            np.add.at(want, idx, g[:, sl].astype(np.float64))
            np.add.at(mag, idx, np.abs(g[:, sl]).astype(np.float64))
        bound = 1e-6 * mag + 1e-30
        assert np.all(np.abs(got[i].cpu().numpy() - want) <= bound), f.name
        assert np.all(np.abs(exact[i].cpu().numpy() - want) <= 1.5e-7 * np.abs(want) + 1e-12), f.name


def test_embedding_backward_mfma_flags_non_finite_gradients_without_spreading_them(monkeypatch):
    """An Inf in one sample's gradient must not reach the other rows of its 16-row tile (0 * Inf inside the one-hot product):
    it contributes nothing and raises the sticky error word, like the fixed-point paths."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.basic.features import SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    monkeypatch.setenv("SWR_K3_MFMA", "1")
    feats = [SparseFeature("a", 64, 16)]
    layer = EmbeddingLayer(feats).to("cuda")
    B = 4096
    idx = torch.arange(B, device="cuda") % 64
    g = torch.ones(B, 16, device="cuda")
    g[5, 3] = float("inf")
    g[70, 1] = float("nan")
    layer({"a": idx}, feats, squeeze_dim=True).backward(g)
    torch.cuda.synchronize()
    got = layer.embed_dict["a"].weight.grad.cpu().numpy()
    want = np.full((64, 16), B / 64.0)
    want[5, 3] -= 1.0
    want[70 % 64, 1] -= 1.0
    assert np.array_equal(got, want)
    with pytest.raises(H.SwrError):
        H.check_errors()


@pytest.mark.parametrize("M,N,K", [(250, 148, 516), (64, 1, 16), (1000, 33, 7), (31, 300, 52), (4096, 256, 376),
                                   (5003, 148, 516), (4100, 64, 36), (8192, 20, 132), (4500, 96, 288), (5000, 300, 132), (4200, 512, 64), (4100, 2048, 36),
                                   (6000, 160, 300),
                                   # tn through the WIDE kernel (5 x 9 tiles of 32): whole, ragged rows + columns, a K not a multiple of 4
                                   # (the 16-byte reduce does not apply), more rows than one stage ring per split
                                   (4096, 160, 288), (5003, 148, 288), (70000, 160, 264), (4133, 130, 258)])
# (tn on the bf16-split kernel -- 516, 132, 288: ragged last tile folded into the full blocks)
def test_gemm_forms(M, N, K):
    """nt / nn / tn on the f32 MFMA pipe against fp64 matmul: exact-fp32 fma chains, so the error is
    fp32 summation roundoff, bounded here by 2e-6 * sum|a||b| per output."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)          # asymmetric operands (transpose-detecting)
    b = rng.standard_normal(N).astype(np.float32)
    dA, dW = _dev(A), _dev(W)
    C = torch.empty((M, N), device="cuda")
    ops.gemm("nt", dA, dW, C, M, N, K, bias=_dev(b))
    bound = 2e-6 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1)
    assert np.all(np.abs(C.cpu().numpy() - (A.astype(np.float64) @ W.astype(np.float64).T + b)) <= bound)
    Wio = np.ascontiguousarray(W.T)
    C2 = torch.empty((M, N), device="cuda")
    ops.gemm("nn", dA, _dev(Wio), C2, M, N, K)
    assert np.all(np.abs(C2.cpu().numpy() - A.astype(np.float64) @ Wio.astype(np.float64)) <= bound)
    G = rng.standard_normal((M, N)).astype(np.float32)
    dWg = torch.empty((N, K), device="cuda")
    cs = torch.empty(N, device="cuda")
    ops.gemm_tn(_dev(G), dA, dWg, M, N, K, colsum=cs)
    want = G.astype(np.float64).T @ A.astype(np.float64)
    bound_t = 2e-6 * (np.abs(G).astype(np.float64).T @ np.abs(A).astype(np.float64) + 1)
    assert np.all(np.abs(dWg.cpu().numpy() - want) <= bound_t)
    np.testing.assert_allclose(cs.cpu().numpy(), G.astype(np.float64).sum(0), rtol=0, atol=2e-6 * np.abs(G).sum(0).max())


def test_gemm_identity_detects_transposes():
    """A = I with an asymmetric B: a swapped row/column in the fragment or C mapping cannot pass."""
    from scenario_wise_rec import ops
    n = 96
    B = np.arange(n * n, dtype=np.float32).reshape(n, n) / 7.0
    I = np.eye(n, dtype=np.float32)
    C = torch.empty((n, n), device="cuda")
    ops.gemm("nt", _dev(I), _dev(B), C, n, n, n)
    assert np.array_equal(C.cpu().numpy(), B.T)
    ops.gemm("nn", _dev(I), _dev(B), C, n, n, n)
    assert np.array_equal(C.cpu().numpy(), B)
    ops.gemm_tn(_dev(I), _dev(B), C, n, n, n)
    assert np.array_equal(C.cpu().numpy(), B)


def test_grouped_gemm_and_prologue():
    from scenario_wise_rec import ops
    rng = np.random.default_rng(3)
    M, G, N, K = 300, 5, 16, 32
    X = rng.standard_normal((M, G * K)).astype(np.float32)
    W = rng.standard_normal((G * N, K)).astype(np.float32)
    sc = rng.standard_normal(G * K).astype(np.float32)
    sh = rng.standard_normal(G * K).astype(np.float32)
    C = torch.empty((M, G * N), device="cuda")
    ops.gemm("nt", _dev(X), _dev(W), C, M, N, K, a_scale=_dev(sc), a_shift=_dev(sh), a_relu=True, groups=G,
             gsA=K, gsB=N * K, gsC=N, gsScale=K)
    Xp = np.maximum(X.astype(np.float64) * sc + sh, 0)
    want = np.concatenate([Xp[:, g * K:(g + 1) * K] @ W[g * N:(g + 1) * N].astype(np.float64).T for g in range(G)], 1)
    np.testing.assert_allclose(C.cpu().numpy(), want, rtol=0, atol=2e-5)


# last two of the first five: rows wider than 1024 floats.  (M, N, group, n_groups): trailing softmax groups that are not one float4
# (PLE's gates: groups of n_expert_specific + n_expert_shared columns, ple.py:89-94) -- the tail workgroups of the float4 kernels
@pytest.mark.parametrize("M,N,group,n_groups", [(250, 148, 4, 1), (33, 5, 4, 1), (4099, 64, 4, 1), (300, 1540, 4, 1), (130, 1792, 4, 1),
                                                 (1027, 76, 3, 4), (515, 588, 3, 4), (261, 84, 9, 1), (70, 72, 5, 8), (3, 75, 3, 1)])
def test_linear_bn_act_block_vs_oracle(M, N, group, n_groups):
    """One [Linear -> BatchNorm1d(train) -> ReLU | softmax] block, forward, running stats and every
    gradient, against the oracle tape in fp64.  Tolerance 2e-5 absolute on O(1) values."""
    from oracle import tape as T
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M * N)
    K = 52
    n_sm = group * n_groups if N >= 8 else 0         # trailing softmax groups
    X = rng.standard_normal((M, K)); W = rng.standard_normal((N, K)) * 0.3; b = rng.standard_normal(N) * 0.1
    g = rng.random(N) + 0.5; be = rng.standard_normal(N) * 0.2
    dY = rng.standard_normal((M, N))
    # oracle
    x_, W_, b_, g_, be_ = (T.param(a.copy()) for a in (X, W, b, g, be))
    z = T.linear(x_, W_, b_)
    y, mu, var = T.batchnorm_train(z, g_, be_, 1e-5)
    if n_sm:
        y = T.cat1([T.relu(T.slice1(y, 0, N - n_sm))] +
                   [T.softmax_rows(T.slice1(y, N - n_sm + i * group, N - n_sm + (i + 1) * group)) for i in range(n_groups)])
    else:
        y = T.relu(y)
    T.backward(y, seed=dY)
    # product
    t = {k: _dev(v, torch.float32).requires_grad_(True) for k, v in dict(X=X, W=W, b=b, g=g, be=be).items()}
    rm, rv, nbt = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    acts = [(0, N - n_sm, "relu", 1), (N - n_sm, N, "softmax", group)] if n_sm else "relu"
    bn = {"gamma": [t["g"]], "beta": [t["be"]], "running_mean": [rm], "running_var": [rv], "nbt": [nbt],
          "eps": 1e-5, "momentum": 0.1}
    Y = ops.linear_bn_act(t["X"], [t["W"]], [t["b"]], bn=bn, acts=acts, training=True)
    Y.backward(_dev(dY, torch.float32))
    np.testing.assert_allclose(Y.detach().cpu().numpy(), y.v, rtol=0, atol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mu, rtol=0, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * var * M / (M - 1), rtol=1e-5)
    assert int(nbt) == 1
    for name, ref in (("X", x_), ("W", W_), ("g", g_), ("be", be_)):
        scale = np.abs(ref.g).max()
        np.testing.assert_allclose(t[name].grad.cpu().numpy(), ref.g, rtol=0, atol=3e-5 * scale, err_msg=name)
    assert np.abs(t["b"].grad.cpu().numpy()).max() < 1e-3 * np.abs(W_.g).max()      # mathematically zero


def test_fused_adam_matches_oracle():
    from oracle.optim import Adam
    from scenario_wise_rec.optim import FusedAdam
    rng = np.random.default_rng(0)
    shapes = [(37, 5), (11,), (128, 16)]
    ps = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    params = [torch.nn.Parameter(_dev(p)) for p in ps]
    opt = FusedAdam(params, lr=1e-3, weight_decay=1e-5)
    ref = Adam(lr=1e-3, weight_decay=1e-5)
    state = {i: p.astype(np.float64) for i, p in enumerate(ps)}
    for step in range(5):
        gs = [rng.standard_normal(s).astype(np.float32) * 10 ** rng.uniform(-6, 0) for s in shapes]
        for p, g in zip(params, gs):
            p.grad = _dev(g)
        opt.step()
        ref.step(state, {i: g.astype(np.float64) for i, g in enumerate(gs)})
    for i, p in enumerate(params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), state[i], rtol=0, atol=2e-6)


@pytest.mark.parametrize("M,H_,n_expert,sel,n_before", [
    (515, 32, 9, [[0, 1, 8], [2, 3, 8], [4, 5, 8], [6, 7, 8]], 576),     # PLE, config 4: 4 gates of 3 behind 576 other columns
    (1000, 16, 4, [[0, 1, 2, 3]] * 3, 0),                                 # identity selection (the butterfly backward)
    (77, 8, 5, [[0, 4], [1, 4], [2, 3]], 5)])                             # gate tensor at an odd column offset
def test_moe_mix_with_the_gates_in_their_own_tensor(M, H_, n_expert, sel, n_before):
    """MoeMix(X, desc, width, G): experts in X, gate probabilities in a column block of ANOTHER tensor (a split_cols view, as in
    PLE: ple.py:107-125) -- same pooled output and gradients, bit for bit, as the concatenated form; the gate gradient lands in the
    split_cols gradient tensor when that is 16-byte friendly."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M + H_)
    n_out, n_sel = len(sel), len(sel[0])
    ng = n_out * n_sel
    X0 = _dev(rng.standard_normal((M, n_expert * H_)).astype(np.float32))
    W0 = _dev(rng.standard_normal((M, n_before + ng)).astype(np.float32))
    dP = _dev(rng.standard_normal((M, n_out * H_)).astype(np.float32))
    # concatenated form
    Xa, Wa = X0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
    Ya = torch.cat([Xa, Wa[:, n_before:]], dim=1)
    da = ops.make_mix_desc(n_out, n_sel, H_, 0, n_expert * H_, n_sel, sel)
    Pa = ops.MoeMix.apply(Ya, da, Ya.shape[1])
    Pa.backward(dP)
    # separate form
    Xb, Wb = X0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
    blocks = ops.split_cols(Wb, [n_before, ng]) if n_before else (None, ops.split_cols(Wb, [ng])[0])
    db = ops.make_mix_desc(n_out, n_sel, H_, 0, 0, n_sel, sel)
    assert ops.moe_mix_separate_ok(Xb, blocks[1], db)
    Pb = ops.MoeMix.apply(Xb, db, Xb.shape[1], blocks[1])
    loss = (Pb * dP).sum()
    if n_before:
        loss = loss + (blocks[0] * 2.0).sum()          # the other block takes a gradient too
    loss.backward()
    assert torch.equal(Pa, Pb)
    assert torch.equal(Xa.grad, Xb.grad)
    assert torch.equal(Wa.grad[:, n_before:], Wb.grad[:, n_before:])
    if n_before:
        assert bool((Wb.grad[:, :n_before] == 2.0).all())


@pytest.mark.parametrize("M,H_,n_expert,sel,pad", [
    (1000, 32, 4, [[0, 1, 2, 3]] * 5, 0),                    # MMoE: every domain gate mixes every expert (16-byte path)
    (333, 16, 5, [[0, 1, 4], [2, 3, 4], [0, 1, 2]], 4),      # PLE-like subsets, spare columns in Y (stay zero)
    (257, 6, 3, [[0, 2], [1, 2]], 0),                        # H not a multiple of 4: row-staged kernel
    (1, 8, 2, [[0, 1]], 0),
    (131, 8, 3, [[0, 1, 2]] * 2, 4),                         # identity selection, rows not filling the last wave
    (77, 64, 6, [[0, 1, 2, 3, 4, 5]] * 3, 0)])
def test_moe_mix_forward_backward(M, H_, n_expert, sel, pad):
    """pooled_o = sum_j gate[o][j] * expert[sel[o][j]] and its gradients against numpy fp64 (mmoe.py:48-49)."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M + H_)
    n_out, n_sel = len(sel), len(sel[0])
    x_col, g_col = 0, n_expert * H_ + pad
    width = g_col + n_out * n_sel
    Y = rng.standard_normal((M, width)).astype(np.float32)
    desc = ops.make_mix_desc(n_out, n_sel, H_, x_col, g_col, n_sel, sel)
    Yd = _dev(Y).requires_grad_(True)
    P = ops.MoeMix.apply(Yd, desc, width)
    X = Y[:, :n_expert * H_].reshape(M, n_expert, H_).astype(np.float64)
    G = Y[:, g_col:].reshape(M, n_out, n_sel).astype(np.float64)
    want = np.stack([sum(G[:, o, j, None] * X[:, sel[o][j]] for j in range(n_sel)) for o in range(n_out)], axis=1)
    np.testing.assert_allclose(P.detach().cpu().numpy().reshape(M, n_out, H_), want, rtol=0, atol=4e-6)
    dP = rng.standard_normal((M, n_out * H_)).astype(np.float32)
    P.backward(_dev(dP))
    got = Yd.grad.cpu().numpy()
    dPn = dP.reshape(M, n_out, H_).astype(np.float64)
    dX = np.zeros((M, n_expert, H_))
    dG = np.zeros((M, n_out, n_sel))
    for o in range(n_out):
        for j in range(n_sel):
            dX[:, sel[o][j]] += G[:, o, j, None] * dPn[:, o]
            dG[:, o, j] = (dPn[:, o] * X[:, sel[o][j]]).sum(axis=1)
    np.testing.assert_allclose(got[:, :n_expert * H_].reshape(M, n_expert, H_), dX, rtol=0, atol=8e-6)
    np.testing.assert_allclose(got[:, g_col:].reshape(M, n_out, n_sel), dG, rtol=0, atol=3e-5)
    if pad:
        assert not np.any(got[:, n_expert * H_:g_col])


@pytest.mark.parametrize("M,D,ydt", [(5000, 5, torch.float32), (1, 3, torch.float32), (4097, 2, torch.int64)])
def test_fused_select_bce_is_bitwise_the_two_step_path(M, D, ydt):
    """swr_select_bce_fwd / _bwd against swr_select_fwd -> swr_bce_fwd and their backward kernels: identical bits for
    p, the loss and dV (same arithmetic, same summation trees); out-of-range domain ids give p = 0 exactly."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(M)
    V0 = torch.randn(M, D, device="cuda", generator=g) * 3
    dom = torch.randint(0, D + 1, (M,), device="cuda", generator=g)           # D itself: no tower -> p = 0
    y = (torch.rand(M, device="cuda", generator=g) < 0.3).to(ydt)
    Va = V0.clone().requires_grad_(True)
    pa = ops.domain_select(Va, dom)
    la = ops.bce_mean(pa, y)
    la.backward()
    Vb = V0.clone().requires_grad_(True)
    with ops.fused_bce(y) as f:
        pb = ops.domain_select(Vb, dom)
    lb = f.loss_for(pb)
    assert lb is not None and f.loss_for(pb * 1.0) is None
    lb.backward()
    assert torch.equal(pa, pb) and torch.equal(la, lb) and torch.equal(Va.grad, Vb.grad)
    for _ in range(3):                                                         # the ticket word is left clean
        with ops.fused_bce(y) as f2:
            p2 = ops.domain_select(V0, dom)
        assert torch.equal(f2.loss_for(p2), la)
    # a gradient arriving on p as well is added through the plain select backward
    Vc = V0.clone().requires_grad_(True)
    with ops.fused_bce(y) as f3:
        pc = ops.domain_select(Vc, dom)
    (f3.loss_for(pc) + pc.sum() * 0.5).backward()
    Vd = V0.clone().requires_grad_(True)
    pd_ = ops.domain_select(Vd, dom)
    (ops.bce_mean(pd_, y) + pd_.sum() * 0.5).backward()
    torch.testing.assert_close(Vc.grad, Vd.grad, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("M,G,K,Hd", [(1000, 5, 32, 16), (257, 3, 8, 4), (64, 2, 16, 32), (4099, 1, 16, 8)])
def test_tower_head_matches_the_layerwise_path(M, G, K, Hd):
    """The fused per-domain tower kernels (csrc/tower.hip) against the same towers evaluated layer by layer
    (grouped GEMM + BatchNorm + ReLU + GEMM, SWR_TOWER_HEAD=0): outputs, input gradient, every parameter gradient and
    the BatchNorm running statistics, plus an fp64 numpy check of the forward."""
    import os
    from scenario_wise_rec.basic.layers import MLP, mlp_bank_forward
    from scenario_wise_rec.basic.module import SwrModule

    class Towers(SwrModule):
        def __init__(self):
            super().__init__()
            self.towers = torch.nn.ModuleList(MLP(K, True, [Hd]) for _ in range(G))

        def forward(self, x):
            return mlp_bank_forward(list(self.towers), x, shared_input=False)

    torch.manual_seed(M + G)
    ref, fused = Towers(), Towers()
    fused.load_state_dict(ref.state_dict())
    for mod in (ref, fused):
        mod.to("cuda").train()
        for p in mod.parameters():                      # away from the degenerate init (beta = 0, gamma = 1)
            p.data.add_(0.1 * torch.randn_like(p))
    fused.load_state_dict(ref.state_dict())
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(M, G * K, device="cuda", generator=g)
    dV = torch.randn(M, G, device="cuda", generator=g)
    res = []
    for mod, flag in ((ref, "0"), (fused, "1")):
        os.environ["SWR_TOWER_HEAD"] = flag
        try:
            x = x0.clone().requires_grad_(True)
            V = mod(x)
            V.backward(dV)
        finally:
            os.environ.pop("SWR_TOWER_HEAD", None)
        res.append((V.detach(), x.grad, {n: p.grad.clone() for n, p in mod.named_parameters()},
                    {n: b.clone() for n, b in mod.named_buffers()}))
    (Va, dxa, ga, ba), (Vb, dxb, gb, bb) = res
    torch.testing.assert_close(Vb, Va, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dxb, dxa, rtol=1e-4, atol=1e-5 * float(dxa.abs().max()) + 1e-7)
    for n in ga:
        atol = 2e-5 * float(ga[n].abs().max()) + 1e-6
        if n.endswith("mlp.0.bias"):          # a bias in front of BatchNorm: zero in exact arithmetic, rounding noise here
            atol = 1e-3 * float(ga[n.replace("bias", "weight")].abs().max())
        torch.testing.assert_close(gb[n], ga[n], rtol=1e-4, atol=atol, msg=n)
    for n in ba:
        torch.testing.assert_close(bb[n].float(), ba[n].float(), rtol=1e-5, atol=1e-6, msg=n)
    # forward against numpy fp64
    sd = {k: v.detach().cpu().double().numpy() for k, v in ref.state_dict().items()}
    xs = x0.cpu().double().numpy()
    cols = []
    for t in range(G):
        W1, b1 = sd[f"towers.{t}.mlp.0.weight"], sd[f"towers.{t}.mlp.0.bias"]
        gm, be = sd[f"towers.{t}.mlp.1.weight"], sd[f"towers.{t}.mlp.1.bias"]
        w2, b2 = sd[f"towers.{t}.mlp.4.weight"], sd[f"towers.{t}.mlp.4.bias"]
        z = xs[:, t * K:(t + 1) * K] @ W1.T + b1
        a1 = np.maximum((z - z.mean(0)) / np.sqrt(z.var(0) + 1e-5) * gm + be, 0)
        cols.append(a1 @ w2.T + b2)
    np.testing.assert_allclose(Vb.cpu().numpy(), np.concatenate(cols, axis=1), rtol=0, atol=3e-5)


@pytest.mark.parametrize("B,ne,Hh,D", [(1000, 4, 32, 5), (130, 2, 16, 4), (257, 3, 32, 4), (64, 8, 16, 4), (100, 2, 16, 3)])   # last: width not 16-byte, layer-wise
def test_bnmix_path_matches_the_layerwise_path(B, ne, Hh, D):
    """MMOE with the fused BatchNorm + ReLU / softmax + gate-mix kernels (csrc/bnmix.hip) against the same model run
    layer by layer (SWR_BNMIX=0): probabilities, every parameter gradient and the BatchNorm running statistics."""
    import os
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.models.multi_domain import MMOE
    feats = [SparseFeature("a", 50, 8), SparseFeature("b", 7, 8), DenseFeature("x0"), DenseFeature("x1")]
    torch.manual_seed(B + ne)
    mods = [MMOE(feats, D, ne, {"dims": [Hh]}, {"dims": [8]}) for _ in range(2)]
    mods[1].load_state_dict(mods[0].state_dict())
    g = torch.Generator().manual_seed(3)
    x = {"a": torch.randint(0, 50, (B,), generator=g), "b": torch.randint(0, 7, (B,), generator=g),
         "x0": torch.rand(B, generator=g), "x1": torch.rand(B, generator=g),
         "domain_indicator": torch.randint(0, D, (B,), generator=g)}
    x = {k: v.cuda() for k, v in x.items()}
    y = (torch.rand(B, generator=g) < 0.3).float().cuda()
    res = []
    for mod, flag in zip(mods, ("0", "1")):
        mod.cuda().train()
        for p in mod.parameters():
            if p.dim() == 1:
                p.data.add_(0.1 * torch.randn(p.shape, generator=g).cuda())      # gamma / beta / biases off their init
        os.environ["SWR_BNMIX"] = flag
        try:
            mod.zero_grad()
            p_ = mod(x)
            ops_loss = torch.nn.functional.binary_cross_entropy(p_, y)
            ops_loss.backward()
        finally:
            os.environ.pop("SWR_BNMIX", None)
        res.append((p_.detach(), {n: q.grad.clone() for n, q in mod.named_parameters() if q.grad is not None},
                    {n: b.clone() for n, b in mod.named_buffers()}))
    # the two models were perturbed with different draws: redo with identical parameters
    mods[1].load_state_dict(mods[0].state_dict())
    res = []
    for mod, flag in zip(mods, ("0", "1")):
        os.environ["SWR_BNMIX"] = flag
        try:
            mod.zero_grad()
            p_ = mod(x)
            torch.nn.functional.binary_cross_entropy(p_, y).backward()
        finally:
            os.environ.pop("SWR_BNMIX", None)
        res.append((p_.detach(), {n: q.grad.clone() for n, q in mod.named_parameters() if q.grad is not None},
                    {n: b.clone() for n, b in mod.named_buffers()}))
    (pa, ga, ba), (pb, gb, bb) = res
    torch.testing.assert_close(pb, pa, rtol=1e-5, atol=2e-6)
    assert set(ga) == set(gb)
    for n in ga:
        scale = float(ga[n].abs().max())
        atol = 3e-5 * scale + 1e-7
        if n.endswith("mlp.0.bias"):              # bias in front of BatchNorm: zero in exact arithmetic
            atol = 1e-3 * float(ga[n.replace("bias", "weight")].abs().max()) + 1e-7
        torch.testing.assert_close(gb[n], ga[n], rtol=2e-4, atol=atol, msg=n)
    for n in ba:
        torch.testing.assert_close(bb[n].float(), ba[n].float(), rtol=1e-5, atol=1e-6, msg=n)


def test_rowmat_and_batch_standardize_against_torch():
    """The two glue-free pieces of HAMUR's adapter / STAR's partitioned norm: per-sample row x matrix products and the
    whole-batch standardisation (+ affine), forward and every gradient against torch's own autograd in fp64."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    B, D, k = 301, 3, 35
    T = torch.randn(B, D, k, device="cuda", generator=g, requires_grad=True)
    Hm = torch.randn(B, k, k, device="cuda", generator=g, requires_grad=True)
    dO = torch.randn(B, D, k, device="cuda", generator=g)
    out = ops.RowMat.apply(T, Hm)
    out.backward(dO)
    T64, H64 = T.detach().double().requires_grad_(True), Hm.detach().double().requires_grad_(True)
    ref = torch.einsum("bdi,bij->bdj", T64, H64)
    ref.backward(dO.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(T.grad.double(), T64.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(Hm.grad.double(), H64.grad, rtol=1e-5, atol=1e-5)
    M, N = 777, 38
    x = (torch.randn(M, N, device="cuda", generator=g) * 3 + 1).requires_grad_(True)
    gam = (torch.rand(N, device="cuda", generator=g) + 0.5).requires_grad_(True)
    bet = torch.randn(N, device="cuda", generator=g).requires_grad_(True)
    dy = torch.randn(M, N, device="cuda", generator=g)
    y = ops.batch_standardize(x, 1e-6, gam, bet)
    y.backward(dy)
    x64, g64, b64 = (t.detach().double().requires_grad_(True) for t in (x, gam, bet))
    cen = x64 - x64.mean(0)
    r = g64 * cen / torch.sqrt((cen * cen).mean(0) + 1e-6) + b64
    r.backward(dy.double())
    torch.testing.assert_close(y.double(), r, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.double(), x64.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(gam.grad.double(), g64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bet.grad.double(), b64.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("world,n,dim,vocab,A", [(1, 300, 16, 50, 1000), (2, 1000, 16, 400, 4099), (8, 513, 8, 200, 0),
                                                   (4, 2048, 64, 100000, 37), (3, 64, 16, 5, 8)])
def test_dp_finish_merges_rank_lists_like_a_sort(world, n, dim, vocab, A):
    """The exchange step's device half (swr_dp_finish): rank-ordered mean of the gradient arenas (bit-exact against
    the same fp32 additions in numpy) and the sort-free merge of the ranks' row lists: every distinct row listed
    once, by the lowest rank that holds it, with the holders' gradients added in rank order (bit-exact)."""
    from scenario_wise_rec import parallel
    rng = np.random.default_rng(world * 1000 + n)
    msgs, lists = [], []
    for r in range(world):
        # a rank's list in swr_embed_bwd's mode-1 format: n looked-up rows in row order, first of a run = the row id
        # (with the run's summed gradient), the others = ~row (zero gradient)
        ids = np.sort(rng.integers(0, vocab, size=n)).astype(np.int32)
        head = np.ones(n, bool)
        head[1:] = ids[1:] != ids[:-1]
        urow = np.where(head, ids, ~ids).astype(np.int32)
        ugrad = (rng.standard_normal((n, dim)) * head[:, None]).astype(np.float32)
        dense = rng.standard_normal(A).astype(np.float32)
        lists.append((urow, ugrad))
        msgs.append((dense, urow, ugrad))
    total = (A + n + n * dim + 3) // 4 * 4
    recv = np.zeros((world, total), np.float32)
    for r, (dense, urow, ugrad) in enumerate(msgs):
        recv[r, :A] = dense
        recv[r, A:A + n] = urow.view(np.float32)
        recv[r, A + n:A + n + n * dim] = ugrad.reshape(-1)
    dense_out = torch.zeros(max(A, 1), device="cuda")
    merged = parallel.finish(dense_out[:A] if A else None, (_dev(recv).reshape(-1), A, [(A, A + n, n, dim, vocab)], total), world)
    scale = np.float32(1.0 / world)
    if A:
        want = msgs[0][0].copy()
        for r in range(1, world):
            want = want + msgs[r][0]
        np.testing.assert_array_equal(dense_out[:A].cpu().numpy(), want * scale)
    out_row, out_grad = (t.cpu().numpy() for t in merged[0])
    assert out_row.shape == (world * n,) and out_grad.shape == (world * n, dim)
    want_sum = {}
    owner = {}
    for r, (urow, ugrad) in enumerate(lists):
        for i in np.nonzero(urow >= 0)[0]:
            k = int(urow[i])
            if k in want_sum:
                want_sum[k] = want_sum[k] + ugrad[i]
            else:
                want_sum[k] = ugrad[i].copy()
                owner[k] = r * n + i
    listed = np.nonzero(out_row >= 0)[0]
    assert sorted(out_row[listed].tolist()) == sorted(want_sum)            # every distinct row exactly once
    for pos in listed:
        k = int(out_row[pos])
        assert owner[k] == pos
        np.testing.assert_array_equal(out_grad[pos], want_sum[k] * scale)
    assert np.all(out_grad[out_row < 0] == 0)


@pytest.mark.parametrize("n,D,ties,ydt,ddt", [(1000, 5, 50, np.float32, np.int64), (7, 2, 3, np.int64, np.int8),
                                              (100003, 8, 1000, np.float16, np.int32), (65536, 3, 0, np.float32, np.int64),
                                              (300000, 5, 0, np.float32, np.int16)])
def test_eval_metrics_match_sklearn(n, D, ties, ydt, ddt):
    """Device log-loss / AUC (csrc/metrics.hip) against sklearn.metrics on the same arrays: counts exact, 2U an exact
    integer (AUC compared at 1e-12: sklearn integrates a float trapezoid), log-loss at 1e-12 relative (fp64 sums).
    Quantised scores give heavy ties; domain D - 1 is empty; ids outside [0, D) count only in the overall figure."""
    from sklearn.metrics import log_loss, roc_auc_score
    from scenario_wise_rec import ops
    rng = np.random.default_rng(n + D)
    p = rng.random(n).astype(np.float32)
    if ties:
        p = (np.floor(p * ties) / ties).astype(np.float32)          # includes exact 0.0 (clipped in the log-loss)
    y = (rng.random(n) < 0.3).astype(ydt)
    dom = rng.integers(-1, D, size=n).astype(ddt)                    # -1: outside; D - 1 never drawn below
    dom[dom == D - 1] = 0
    if n >= 7:
        y[:2] = [0, 1]
        dom[:2] = 0                                                  # domain 0 has both classes
    rows, pos, two_u, ll = ops.eval_metrics(_dev(p), _dev(y), _dev(dom), D)
    d64 = dom.astype(np.int64)
    yb = (y.astype(np.float64) > 0.5)
    for d in list(range(D)) + [D]:
        sel = np.ones(n, bool) if d == D else d64 == d
        assert rows[d] == int(sel.sum()) and pos[d] == int(yb[sel].sum())
        if rows[d] == 0:
            assert two_u[d] == 0 and ll[d] == 0.0
            continue
        P, N = pos[d], rows[d] - pos[d]
        pd_, yd = p[sel].astype(np.float64), yb[sel].astype(np.float64)
        want_ll = -np.sum(yd * np.log(np.clip(pd_, 2.0 ** -52, 1 - 2.0 ** -52)) +
                          (1 - yd) * np.log(1 - np.clip(pd_, 2.0 ** -52, 1 - 2.0 ** -52)))
        np.testing.assert_allclose(ll[d], want_ll, rtol=1e-12, atol=1e-12)
        if P and N:
            np.testing.assert_allclose(ll[d] / rows[d], log_loss(yd.tolist(), pd_.tolist()), rtol=1e-12)
            np.testing.assert_allclose(two_u[d] / (2.0 * P * N), roc_auc_score(yd.tolist(), pd_.tolist()), rtol=1e-12, atol=1e-15)
    assert rows[D - 1] == 0 if n >= 7 else True


def test_trainer_device_metrics_equal_the_sklearn_path(monkeypatch):
    """CTRTrainer.evaluate / evaluate_multi_domain_loss with the device metrics == the reference's host path
    (`.tolist()` + sklearn) on the same model and loader; empty domain -> None."""
    from _golden import Case, build_product_model, to_device
    from scenario_wise_rec.trainers import CTRTrainer
    c = Case("mmoe")
    model = build_product_model(c, device="cuda:0")
    tr = CTRTrainer(model, "t", device="cuda:0")
    x, y = c.batch(0)
    x = {k: torch.from_numpy(v) for k, v in x.items()}
    yt = torch.from_numpy(y)
    B = len(y)
    loader = [({k: v[i:i + 100] for k, v in x.items()}, yt[i:i + 100]) for i in range(0, B, 100)]
    D = int(x["domain_indicator"].max()) + 2          # one more domain than the data has: it must report None
    dev = tr.evaluate_multi_domain_loss(model, loader, D), tr.evaluate(model, loader)
    monkeypatch.setenv("SWR_DEVICE_METRICS", "0")
    host = tr.evaluate_multi_domain_loss(model, loader, D), tr.evaluate(model, loader)
    assert dev[0][0][-1] is None and dev[0][1][-1] is None and host[0][0][-1] is None
    for a, b in zip(dev[0][0][:-1] + dev[0][1][:-1] + [dev[0][2], dev[0][3]] + list(dev[1]),
                    host[0][0][:-1] + host[0][1][:-1] + [host[0][2], host[0][3]] + list(host[1])):
        np.testing.assert_allclose(a, b, rtol=1e-10)


@pytest.mark.parametrize("D,I,O,first", [(3, 37, 20, True), (3, 64, 32, False), (1, 5, 1, True), (8, 16, 8, False)])
def test_star_layer_weights_against_torch(D, I, O, first):
    """csrc/star.hip (one launch per layer each way) against the reference's elementwise formulation
    (`star.py:91-107`) differentiated by torch autograd, fp32: values to 1e-6, gradients to 1e-5 relative."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(D * 100 + I)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).requires_grad_(True)
    Ws, bs = mk(I, O), mk(O)
    gs, be = mk(I), mk(I)
    Wd, bd = [mk(I, O) for _ in range(D)], [mk(O) for _ in range(D)]
    gd, bed = [mk(I) for _ in range(D)], [mk(I) for _ in range(D)]
    params = [Ws, bs] + ([gs, be] if first else []) + Wd + bd + ((gd + bed) if first else [])
    eff = ops.star_layer_weights(first, D, *params)
    cot = [torch.randn(e.shape, device="cuda", generator=g) for e in eff]
    got = torch.autograd.grad(eff, params, cot)
    want_eff = []
    for d in range(D):
        w = Ws * Wd[d]
        b = bs + bd[d]
        if first:
            b = b + (be + bed[d]) @ w
            w = (gs * gd[d]).unsqueeze(1) * w
        want_eff.append((w.t().contiguous(), b))
    want_out = [w for w, _ in want_eff] + [b for _, b in want_eff]
    want = torch.autograd.grad(want_out, params, cot)
    for a, b in zip(eff, want_out):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    for a, b, p in zip(got, want, params):
        scale = float(b.abs().max()) + 1e-6
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5 * scale)


@pytest.mark.parametrize("dims", [[37, 20, 12, 6], [64, 32, 16, 8, 8, 4, 4], [5, 1]])
def test_star_stack_weights_is_bitwise_the_layer_by_layer_form(dims):
    """ops.star_stack_weights (swr_star_layers_fwd / bwd: four layers per launch, every layer of the stack in one call each
    way) against one ops.star_layer_weights call per layer: same kernels bodies, so values and gradients bit for bit --
    incl. a 7-layer stack, which takes two launches each way."""
    from scenario_wise_rec import ops
    D = 3
    g = torch.Generator(device="cuda").manual_seed(len(dims))
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).requires_grad_(True)
    layers = []
    for l in range(len(dims) - 1):
        I, O = dims[l], dims[l + 1]
        first = l == 0
        layers.append([mk(I, O), mk(O)] + ([mk(I), mk(I)] if first else []) + [mk(I, O) for _ in range(D)] +
                      [mk(O) for _ in range(D)] + ([mk(I) for _ in range(2 * D)] if first else []))
    flat = [p for ps in layers for p in ps]
    stacked = ops.star_stack_weights(D, layers)
    single = [ops.star_layer_weights(l == 0, D, *ps) for l, ps in enumerate(layers)]
    outs_a = [t for eff in stacked for t in eff]
    outs_b = [t for eff in single for t in eff]
    assert len(outs_a) == len(outs_b) == 2 * D * len(layers)
    cot = [torch.randn(t.shape, device="cuda", generator=g) for t in outs_a]
    ga = torch.autograd.grad(outs_a, flat, cot)
    gb = torch.autograd.grad(outs_b, flat, cot)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)


def test_device_data_loader_serves_the_same_batches():
    """DeviceDataLoader (row f3): without shuffle the batches are the DataLoader(TorchDataset) batches bit for bit (ragged
    last batch, drop_last); with shuffle every epoch is a permutation of the rows applied consistently to all columns."""
    from scenario_wise_rec.utils.data import DeviceDataLoader, TorchDataset
    from torch.utils.data import DataLoader
    rng = np.random.default_rng(5)
    n = 1003
    x = {"a": rng.integers(0, 1 << 40, n).astype(np.int64), "b": rng.integers(0, 100, n).astype(np.int8),
         "c": rng.random(n).astype(np.float32), "d": rng.integers(0, 3000, n).astype(np.int16), "e": rng.random(n).astype(np.float64),
         "domain_indicator": rng.integers(0, 3, n).astype(np.int32)}
    y = (rng.random(n) < 0.3).astype(np.float32)
    ref = DataLoader(TorchDataset({k: torch.from_numpy(v) for k, v in x.items()}, torch.from_numpy(y)), batch_size=128)
    got = DeviceDataLoader(x, y, 128)
    assert len(got) == len(ref) == 8
    for (xr, yr), (xg, yg) in zip(ref, got):
        assert torch.equal(yr, yg.cpu())
        for k in x:
            assert xg[k].dtype == xr[k].dtype and torch.equal(xr[k], xg[k].cpu()), k
    assert len(DeviceDataLoader(x, y, 128, drop_last=True)) == 7
    sh = DeviceDataLoader(x, y, 100, shuffle=True, generator=torch.Generator(device="cuda").manual_seed(1))
    key = x["a"].astype(np.float64) * 7 + x["c"]                      # identifies a row
    order = {float(v): i for i, v in enumerate(key)}
    epochs = []
    for _ in range(2):
        rows = []
        for xb, yb in sh:
            a, c = xb["a"].cpu().numpy(), xb["c"].cpu().numpy()
            idx = np.array([order[float(v)] for v in a.astype(np.float64) * 7 + c])
            for k in x:
                np.testing.assert_array_equal(xb[k].cpu().numpy(), x[k][idx])      # all columns moved together
            np.testing.assert_array_equal(yb.cpu().numpy(), y[idx])
            rows.append(idx)
        rows = np.concatenate(rows)
        assert sorted(rows.tolist()) == list(range(n))                               # a permutation
        epochs.append(rows)
    assert not np.array_equal(epochs[0], epochs[1]) and not np.array_equal(epochs[0], np.arange(n))


def test_gemm_n_compute_skips_trailing_column_tiles():
    """swr_gemm_args.n_compute: columns below it are the ordinary product (bit for bit), columns from it on are stored as
    zeros without being computed (dX of the layer on the embedding concat: dense-feature columns take no gradient)."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(3)
    M, N, K = 4100, 516, 148
    A, W = _dev(rng.standard_normal((M, K)).astype(np.float32)), _dev(rng.standard_normal((N, K)).astype(np.float32))
    full = torch.empty((M, N), device="cuda")
    part = torch.full((M, N), 7.0, device="cuda")
    ops.gemm("nt", A, W, full, M, N, K)
    ops.gemm("nt", A, W, part, M, N, K, n_compute=512)
    assert torch.equal(part[:, :512], full[:, :512])
    assert torch.count_nonzero(part[:, 512:]) == 0


def test_fit_with_device_loader_and_device_metrics(tmp_path):
    """The reference's driver loop (`CTRTrainer.fit`, ctr_trainer.py:79-97) end to end on the widened path: batches from
    DeviceDataLoader (HBM-resident columns, shuffled per epoch), validation AUC / log-loss from the device metrics, early
    stopper and checkpoint as in the reference; the loss goes down and the checkpoint reloads."""
    from _golden import Case, build_product_model
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.utils.data import DeviceDataLoader
    c = Case("mmoe")
    model = build_product_model(c, device="cuda:0")
    tr = CTRTrainer(model, "t", optimizer_params={"lr": 1e-2, "weight_decay": 1e-5}, n_epoch=3, device="cuda:0",
                    model_path=str(tmp_path))
    x, y = c.batch(0)
    train = DeviceDataLoader(x, y, 64, shuffle=True, generator=torch.Generator(device="cuda").manual_seed(0))
    val = DeviceDataLoader(x, y, 100)
    auc0, ll0 = tr.evaluate(model, val)
    tr.fit(train, val)
    auc1, ll1 = tr.evaluate(model, val)
    assert ll1 < ll0 and auc1 > auc0 and 0.0 < auc1 <= 1.0           # three epochs on its own data: it must fit
    saved = [f for f in tmp_path.iterdir() if f.suffix == ".pth"]
    assert len(saved) == 1
    state = torch.load(saved[0], map_location="cpu")
    assert set(state) == set(model.state_dict())
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), state[k]), k


@pytest.mark.parametrize("name", ["mmoe", "mmoe_out_of_range_domain", "mmoe_empty_domain", "mmoe_narrow_dtypes", "mmoe_b1000"])
def test_routed_eval_equals_the_dense_eval_path(name, monkeypatch):
    """Eval-mode MMoE: the routed head (own gate mix + own tower per row, csrc/routed.hip) against the dense all-domain
    path + select of the reference, same model and batch: within fp32 reassociation (1e-6 on probabilities), exact 0.0
    for out-of-range domain ids; both against the reference's golden eval probabilities."""
    from _golden import Case, build_product_model, to_device, assert_probs_close
    c = Case(name)
    model = build_product_model(c).eval()
    x, _ = c.batch(0)
    xd = to_device(x)
    with torch.no_grad():
        routed = model(xd).cpu().numpy()
        monkeypatch.setenv("SWR_ROUTED_EVAL", "0")
        dense = model(xd).cpu().numpy()
    np.testing.assert_allclose(routed, dense, rtol=0, atol=1e-6)
    dom = np.asarray(x["domain_indicator"]).astype(np.int64)
    D = model.domain_num
    assert np.all(routed[(dom < 0) | (dom >= D)] == 0.0)
    assert_probs_close(routed, c.z["eval_probs"], tol=1e-4)


@pytest.mark.parametrize("B,L,dim,pooling,pad,idx_dtype,limit", [
    (250, 7, 16, "sum", 0, np.int64, 1 << 30), (1000, 5, 8, "mean", 3, np.int16, 1 << 30), (64, 12, 32, "mean", None, np.int32, 1 << 30),
    (333, 3, 6, "sum", 2, np.int64, 1 << 30),        # dim not a multiple of 4: scalar path
    (129, 4, 16, "concat", None, np.int64, 1 << 30), (700, 9, 16, "mean", 0, np.int64, 1024)])      # last: row-sparse table
def test_sequence_pooled_lookup(B, L, dim, pooling, pad, idx_dtype, limit, monkeypatch):
    """SequenceFeature lookup (basic/layers.py:73-87) next to a plain sparse and a dense feature: sum / mean pooling with
    masked padding (fp32 sums vs an fp64 reference), concat = pure copy (bit-exact); backward = masked, scaled scatter of the
    pooled gradient into the table (vs np.add.at in fp64), including rows that are entirely padding."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    from scenario_wise_rec.basic.module import SwrModule
    monkeypatch.setattr(SwrModule, "dense_table_limit_bytes", limit)
    rng = np.random.default_rng(B + L)
    V = 120
    feats = [SparseFeature("s0", 17, dim), SequenceFeature("h", V, dim, pooling=pooling, padding_idx=pad), DenseFeature("d0")]
    layer = EmbeddingLayer(feats)
    t_s0 = rng.standard_normal((17, dim)).astype(np.float32)
    t_h = rng.standard_normal((V, dim)).astype(np.float32)
    with torch.no_grad():
        layer.embed_dict["s0"].weight.copy_(torch.from_numpy(t_s0))
        layer.embed_dict["h"].weight.copy_(torch.from_numpy(t_h))
    layer.to("cuda")
    ids = rng.integers(0, V, size=(B, L))
    if pad is not None and pooling != "concat":
        lens = rng.integers(0, L + 1, size=B)
        ids = np.where(ids == pad, (pad + 1) % V, ids)
        ids[np.arange(L)[None, :] >= lens[:, None]] = pad
    x = {"s0": rng.integers(0, 17, size=B), "h": ids.astype(idx_dtype), "d0": rng.random(B).astype(np.float32)}
    out = layer({k: _dev(v) for k, v in x.items()}, feats, squeeze_dim=True)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(_dev(g))
    torch.cuda.synchronize()
    H.check_errors()
    emb = t_h.astype(np.float64)[ids]                                        # [B, L, dim]
    if pooling == "concat":
        pooled, wts = emb.reshape(B, L * dim), np.ones((B, L))
    else:
        mask = (ids != (pad if pad is not None else -1)).astype(np.float64)
        pooled = (emb * mask[:, :, None]).sum(1)
        wts = mask
        if pooling == "mean":
            pooled = pooled / (mask.sum(1, keepdims=True) + 1e-16)
            wts = mask / (mask.sum(1, keepdims=True) + 1e-16)
    want = np.concatenate([t_s0[x["s0"]], pooled, x["d0"][:, None]], axis=1)
    got = out.detach().cpu().numpy()
    assert got.shape == want.shape
    if pooling == "concat":
        assert np.array_equal(got, want.astype(np.float32))
    else:
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    w = width = pooled.shape[1]
    gh = g[:, dim:dim + width].astype(np.float64)
    gh = gh.reshape(B, L, dim) if pooling == "concat" else gh[:, None, :] * wts[:, :, None]
    want_grad = np.zeros((V, dim))
    np.add.at(want_grad, ids.reshape(-1), gh.reshape(-1, dim))
    p = layer.embed_dict["h"].weight
    if p.grad is not None:
        got_grad = p.grad.cpu().numpy()
    else:
        r, gg = (t.cpu().numpy() for t in p._swr_sparse_grad)
        got_grad = np.zeros((V, dim))
        np.add.at(got_grad, r[r >= 0], gg[r >= 0].astype(np.float64))
    np.testing.assert_allclose(got_grad, want_grad, rtol=0, atol=2e-6 * max(1.0, np.abs(want_grad).max()))


@pytest.mark.parametrize("fold", [False, True, "fused"], ids=["block", "folded", "fused"])
def test_gather_one_hot_block_of_small_tables(fold, monkeypatch):
    """`onehot=True` (training): behind the concat, ONE-HOT columns of the tables with <= 16 rows -- out[b, oh + off_t + v]
    = (id_t(b) == v), exact 0.0 / 1.0; the alignment columns in between are zero.  `block`: the embeddings themselves are
    unchanged; `folded`: only [embeddings of the other tables | dense | 0 | one-hot] is written, behind the ordinary
    columns (the consuming layer folds the small tables into its weights), in 16-column groups; `fused`: nothing is written
    at all (csrc/first_layer.hip) -- the same block appears when a consumer asks for it (OneHotInfo.materialize), and the
    keys / one-hot bits the fused products read are checked here against the ids."""
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.basic.layers import EmbeddingLayer
    monkeypatch.setattr(ops, "FOLD", bool(fold))
    monkeypatch.setattr(ops, "FUSED_LOOKUP", fold == "fused")
    rng = np.random.default_rng(5)
    B, vocabs = 333, [2, 200, 7, 16, 17, 3]
    feats = [SparseFeature(f"s{i}", v, 16) for i, v in enumerate(vocabs)] + [DenseFeature("d0"), DenseFeature("d1"), DenseFeature("d2")]
    layer = EmbeddingLayer(feats).to("cuda").train()
    with torch.no_grad():
        for i in range(len(vocabs)):
            layer.embed_dict[f"s{i}"].weight.normal_(0, 0.3)
    x = {f"s{i}": rng.integers(0, v, size=B) for i, v in enumerate(vocabs)}
    x.update({f"d{i}": rng.random(B).astype(np.float32) for i in range(3)})
    xd = {k: _dev(v) for k, v in x.items()}
    plain = layer(xd, feats, squeeze_dim=True)
    out = layer(xd, feats, squeeze_dim=True, onehot=True)
    info = out._swr_onehot
    assert out.shape[1] == 6 * 16 + 3 and info.oh_width == (32 if fold else 28)    # 2 + 7 + 16 + 3 = 28 one-hot columns
    want = np.zeros((B, info.oh_width), np.float32)
    if fold:
        assert (info.wide is None) == (fold == "fused") and (info.fl is not None) == (fold == "fused")
        wide = info.materialize().detach().cpu().numpy() if fold == "fused" else info.wide.detach().cpu().numpy()
        pl = plain.detach().cpu().numpy()
        # compact block from column 100: tables s1 (200 rows) and s4 (17 rows), the three dense features, 13 pad columns
        assert info.col0 == 100 and info.Kp == 48 and info.oh_col == 148 and wide.shape[1] == 180
        assert np.array_equal(wide[:, 100:116], pl[:, 16:32]) and np.array_equal(wide[:, 116:132], pl[:, 64:80])
        assert np.array_equal(wide[:, 132:135], pl[:, 96:99]) and np.array_equal(wide[:, 135:148], np.zeros((B, 13), np.float32))
        off = 0
        for i, v in enumerate(vocabs):
            if v <= 16:
                want[np.arange(B), off + x[f"s{i}"]] = 1.0
                off += v
        assert np.array_equal(wide[:, 148:], want)
        if fold == "fused":
            f = info.fl
            ws, o = f["ws"], f["offs"]
            keys = ws[o.keys:o.keys + 4 * 2 * B].view(torch.int32).cpu().numpy().reshape(2, B)
            assert np.array_equal(keys[0], x["s1"]) and np.array_equal(keys[1], x["s4"])          # the slots that keep an embedding
            mask = ws[o.mask:o.mask + 16 * B].cpu().numpy().reshape(B, 16)
            bits = np.unpackbits(mask, axis=1, bitorder="little")[:, :32]
            assert np.array_equal(bits.astype(np.float32), want)
            mt = ws[o.mask_t:o.mask_t + 16 * B].view(torch.int32).cpu().numpy().reshape(4, B)
            assert np.array_equal(mt.T.copy().view(np.uint8).reshape(B, 16), mask)
            df = ws[o.densef:o.densef + 4 * o.nd4 * B].view(torch.float32).cpu().numpy().reshape(B, o.nd4)
            assert o.nd4 == 4 and np.array_equal(df[:, :3], pl[:, 96:99]) and not df[:, 3].any()
        return
    assert torch.equal(out, plain)
    assert info.oh_col == 100                                    # behind column 99 (+1 pad)
    wide = torch.as_strided(out.detach(), (B, info.oh_col + info.oh_width), (out.stride(0), 1)).cpu().numpy()
    assert np.array_equal(wide[:, 99], np.zeros(B, np.float32))
    off = 0
    for i, v in enumerate(vocabs):
        if v <= 16:
            want[np.arange(B), off + x[f"s{i}"]] = 1.0
            off += v
    assert np.array_equal(wide[:, 100:], want)
    assert sorted(t[1] for t in info.tables_p) == [2, 3, 7, 16]
    layer.eval()
    assert getattr(layer(xd, feats, squeeze_dim=True, onehot=True), "_swr_onehot", None) is None


@pytest.mark.parametrize("M,N,K,ex", [(4096, 148, 276, 148), (1000, 148, 276, 148), (250, 33, 200, 64), (4096, 64, 96, 32),
                                      (300, 148, 128, 1), (64, 160, 260, 200)])
def test_gemm_exact_columns_give_the_same_bits(M, N, K, ex):
    """`a_exact_from` (swr.h): columns of A that hold 0 / 1 need three of the six bf16 products; the three left out are
    products with zero terms, so the result is bitwise the full kernel's.  Also against fp64."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M + N + K + ex)
    A = rng.standard_normal((M, K)).astype(np.float32)
    A[:, ex:] = (rng.random((M, K - ex)) < 0.1).astype(np.float32)          # the one-hot block
    W = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dA, dW, db = _dev(A), _dev(W), _dev(b)
    full = torch.empty((M, N), device="cuda")
    fast = torch.empty((M, N), device="cuda")
    ops.gemm("nt", dA, dW, full, M, N, K, bias=db)
    ops.gemm("nt", dA, dW, fast, M, N, K, bias=db, a_exact_from=ex)
    assert torch.equal(full, fast)
    bound = 2e-6 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1)
    assert np.all(np.abs(fast.cpu().numpy() - (A.astype(np.float64) @ W.astype(np.float64).T + b)) <= bound)


@pytest.mark.parametrize("M,G,K,Hd,ddt,ydt,use_p", [(1000, 5, 32, 16, torch.int64, torch.float32, False),
                                                      (257, 3, 8, 4, torch.int8, torch.int64, False),
                                                      (4099, 4, 16, 32, torch.int32, torch.float32, True),
                                                      (64, 2, 16, 8, torch.int64, torch.float32, False)])
def test_tower_head_select_bce_is_bitwise_the_three_launch_path(M, G, K, Hd, ddt, ydt, use_p, monkeypatch):
    """Towers -> sigmoid -> domain select -> mean BCE: the one-launch output layer (own tower per row) with the gradient
    implied inside the backward kernels (ops.TowerHeadSelectBCE) against TowerHead -> SelectBCE (SWR_TOWER_SELECT off):
    probabilities, loss, input gradient, every parameter gradient and the running statistics bit for bit -- also with
    out-of-range domain ids (probability 0, no gradient) and with an extra gradient arriving on the probabilities."""
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.layers import MLP, mlp_bank_select
    from scenario_wise_rec.basic.module import SwrModule

    class Towers(SwrModule):
        def __init__(self):
            super().__init__()
            self.towers = torch.nn.ModuleList(MLP(K, True, [Hd]) for _ in range(G))

        def forward(self, x, dom):
            return mlp_bank_select(list(self.towers), x, dom)

    torch.manual_seed(M + G)
    two, one = Towers(), Towers()
    for mod in (two, one):
        mod.to("cuda").train()
    for p in two.parameters():
        p.data.add_(0.1 * torch.randn_like(p))
    one.load_state_dict(two.state_dict())
    g = torch.Generator(device="cuda").manual_seed(2)
    x0 = torch.randn(M, G * K, device="cuda", generator=g)
    dom = torch.randint(-1 if ddt != torch.int64 else 0, G + 1, (M,), device="cuda", generator=g).to(ddt)   # some ids out of range
    y = (torch.rand(M, device="cuda", generator=g) < 0.3).to(ydt)
    wp = torch.randn(M, device="cuda", generator=g)
    res = []
    for mod, flag in ((two, False), (one, True)):
        monkeypatch.setattr(ops, "TOWER_SELECT", flag)
        x = x0.clone().requires_grad_(True)
        with ops.fused_bce(y) as f:
            p = mod(x, dom)
        loss = f.loss_for(p)
        assert loss is not None
        total = loss + (p * wp).sum() * 1e-3 if use_p else loss
        total.backward()
        res.append((p.detach(), loss.detach(), x.grad, {n: q.grad.clone() for n, q in mod.named_parameters()},
                    {n: b.clone() for n, b in mod.named_buffers()}))
    (pa, la, dxa, ga, ba), (pb, lb, dxb, gb, bb) = res
    assert torch.equal(pa, pb) and torch.equal(la, lb)
    assert torch.equal(dxa, dxb)
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n
    for n in ba:
        assert torch.equal(ba[n], bb[n]), n
    oob = (dom.long() < 0) | (dom.long() >= G)
    assert bool((pb[oob] == 0).all()) and float(lb) > 0


@pytest.mark.parametrize("B,D,k,offset", [(4096, 1, 35, 0), (4099, 1, 35, 0), (1000, 3, 8, 0), (513, 2, 35, 1), (3, 1, 5, 0),
                                          (40000, 1, 35, 0), (32768, 8, 35, 0), (4099, 8, 35, 0), (4099, 8, 35, 1), (4100, 11, 20, 3)])
def test_rowmat_sample_blocks(B, D, k, offset):
    """swr_rowmat_fwd / _bwd stage four samples per workgroup with 16-byte accesses when the tensors allow it: full and
    ragged last groups, operands that are NOT 16-byte aligned (views at an odd element offset), more samples than the grid
    has workgroups; against torch's einsum in fp64."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(B + D + k)

    def view(shape):
        n = int(np.prod(shape))
        buf = torch.randn(n + offset, device="cuda", generator=g)
        return buf[offset:].view(shape).detach()

    T, Hm, dO = view((B, D, k)).requires_grad_(True), view((B, k, k)).requires_grad_(True), view((B, D, k))
    out = ops.RowMat.apply(T, Hm)
    out.backward(dO)
    T64, H64 = T.detach().double().requires_grad_(True), Hm.detach().double().requires_grad_(True)
    ref = torch.einsum("bdi,bij->bdj", T64, H64)
    ref.backward(dO.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(T.grad.double(), T64.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(Hm.grad.double(), H64.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,D,k,offset", [(32768, 8, 35, 0), (4099, 8, 35, 1), (4100, 11, 20, 3), (4096, 1, 35, 0), (5000, 3, 8, 2)])
def test_rowmat_pipelined_forward_is_bitwise_the_plain_one(B, D, k, offset):
    """swr_rowmat_fwd takes `rowmat_fwd2_kernel` (next sample prefetched into registers, DPP row-broadcast FMAs) from 4 096 samples
    on and `rowmat_fwd_kernel` below: same sums in the same order, so the same bits -- checked by running the long batch whole
    (pipelined) and in slices of 2 048 samples (plain), at HAMUR's D = 8 (register-tile rows 1..7) and a D past one tile; the
    first / last sample of an unaligned buffer take the element-wise edge loads.  Guard words around the output stay untouched."""
    from scenario_wise_rec import _hip as H
    g = torch.Generator(device="cuda").manual_seed(B * 31 + D * 7 + k)
    bufT = torch.randn(B * D * k + offset, device="cuda", generator=g)
    bufH = torch.randn(B * k * k + offset, device="cuda", generator=g)
    T, Hm = bufT[offset:].view(B, D, k), bufH[offset:].view(B, k, k)
    guard = 64
    full = torch.full((B * D * k + 2 * guard,), 7.25, device="cuda")
    out = full[guard:guard + B * D * k].view(B, D, k)
    H.check(H.lib.swr_rowmat_fwd(H.ptr(T), H.ptr(Hm), H.ptr(out), B, D, k, H.stream()), "swr_rowmat_fwd")
    ref = torch.empty(B, D, k, device="cuda")
    step = 2048
    for b0 in range(0, B, step):
        n = min(step, B - b0)
        H.check(H.lib.swr_rowmat_fwd(H.ptr(T[b0:b0 + n]), H.ptr(Hm[b0:b0 + n]), H.ptr(ref[b0:b0 + n]), n, D, k, H.stream()), "swr_rowmat_fwd")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert bool((full[:guard] == 7.25).all()) and bool((full[guard + B * D * k:] == 7.25).all())
    want = torch.einsum("bdi,bij->bdj", T[:64].double(), Hm[:64].double())
    torch.testing.assert_close(out[:64].double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,D,k,offset", [(32768, 8, 35, 0), (4099, 8, 35, 1), (515, 5, 35, 3), (777, 1, 36, 0), (1000, 3, 8, 2), (64, 8, 4, 0),
                                          (4100, 4, 17, 1)])
def test_rowmat_matrix_pipe_forms_are_bitwise_the_vector_ones(B, D, k, offset, monkeypatch):
    """swr_rowmat_fwd / _bwd on v_mfma_f32_4x4x1_16B_f32 (csrc/moe.hip rowmat_mfma_kernel / _dh_kernel; D <= 8, k <= 36): every
    output is the same k-ordered fp32 fmaf chain as in the vector kernels (SWR_ROWMAT_MFMA=0), so out, dT and dHm -- written and
    accumulated -- agree BIT FOR BIT; ragged tails (B not a multiple of 8, k not a multiple of 4, D < 8), unaligned operands, guard
    words around every output."""
    from scenario_wise_rec import _hip as H
    g = torch.Generator(device="cuda").manual_seed(B * 13 + D * 5 + k)

    def view(n):
        return torch.randn(n + offset, device="cuda", generator=g)[offset:]
    T, Hm, dO = view(B * D * k), view(B * k * k), view(B * D * k)
    base_dH = view(B * k * k).clone()
    guard = 32
    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SWR_ROWMAT_MFMA", mode)
        bufs = [torch.full((n + 2 * guard,), 3.5, device="cuda") for n in (B * D * k, B * D * k, B * k * k, B * k * k)]
        out, dT, dH, dHa = (b[guard:guard + n] for b, n in zip(bufs, (B * D * k, B * D * k, B * k * k, B * k * k)))
        dHa.copy_(base_dH)
        H.check(H.lib.swr_rowmat_fwd(H.ptr(T), H.ptr(Hm), H.ptr(out), B, D, k, H.stream()), "fwd")
        H.check(H.lib.swr_rowmat_bwd(H.ptr(dO), H.ptr(T), H.ptr(Hm), H.ptr(dT), H.ptr(dH), 0, B, D, k, H.stream()), "bwd")
        H.check(H.lib.swr_rowmat_bwd(H.ptr(dO), H.ptr(T), H.ptr(Hm), None, H.ptr(dHa), 1, B, D, k, H.stream()), "bwd acc")
        torch.cuda.synchronize()
        for b, n in zip(bufs, (B * D * k, B * D * k, B * k * k, B * k * k)):
            assert bool((b[:guard] == 3.5).all()) and bool((b[guard + n:] == 3.5).all())
        res.append([t.clone() for t in (out, dT, dH, dHa)])
    for name, a, b in zip(("out", "dT", "dHm", "dHm accumulated"), *res):
        assert torch.equal(a, b), name
    T3, H3, G3 = T.view(B, D, k)[:128].double(), Hm.view(B, k, k)[:128].double(), dO.view(B, D, k)[:128].double()
    torch.testing.assert_close(res[1][0].view(B, D, k)[:128].double(), torch.einsum("bdi,bij->bdj", T3, H3), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[1][1].view(B, D, k)[:128].double(), torch.einsum("bdj,bij->bdi", G3, H3), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[1][2].view(B, k, k)[:128].double(), torch.einsum("bdi,bdj->bij", T3, G3), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("uses", [2, 3])
def test_rowmat_shared_factor_collects_its_gradient_in_one_buffer(uses, monkeypatch):
    """One H_b feeding several RowMat products (HAMUR's adapter cell): the backward passes add into one buffer and the last
    one hands it to autograd -- the same bits as letting autograd sum the separate gradients; a product left out of the
    backward pass raises instead of dropping the collected gradient."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(uses)
    B, D, k = 515, 4, 35
    Ts = [torch.randn(B, D, k, device="cuda", generator=g) for _ in range(uses)]
    dOs = [torch.randn(B, D, k, device="cuda", generator=g) for _ in range(uses)]
    H0 = torch.randn(B, k, k, device="cuda", generator=g)
    res = []
    for share in (False, True):
        monkeypatch.setattr(ops, "ROWMAT_SHARE", share)
        Hm = H0.clone().requires_grad_(True)
        Tr = [t.clone().requires_grad_(True) for t in Ts]
        outs = [ops.RowMat.apply(t, Hm) for t in Tr]
        torch.autograd.backward(outs, dOs)
        res.append((Hm.grad.clone(), [t.grad.clone() for t in Tr]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    ref = sum(torch.einsum("bdi,bdj->bij", t.double(), d.double()) for t, d in zip(Ts, dOs))
    torch.testing.assert_close(res[1][0].double(), ref, rtol=1e-5, atol=1e-5)
    # second backward pass over a retained graph, and a product that takes no part
    monkeypatch.setattr(ops, "ROWMAT_SHARE", True)
    Hm = H0.clone().requires_grad_(True)
    outs = [ops.RowMat.apply(t, Hm) for t in Ts]
    with pytest.raises(H.SwrError, match="not all of them"):
        outs[0].backward(dOs[0])


def test_split_cols_collects_the_block_gradients_in_one_tensor():
    """ops.split_cols against plain slicing: same values, same gradient (bitwise: every element has one contributor), also
    with an unused block and a tail of columns outside the blocks."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x0 = torch.randn(513, 100, device="cuda", generator=g)
    ws = [24, 40, 16]                                      # 20 columns stay outside
    w = [torch.randn(513, k, device="cuda", generator=g) for k in ws]
    res = []
    for split in (False, True):
        x = x0.clone().requires_grad_(True)
        y = x * 2.0
        if split:
            a, b, c = ops.split_cols(y, ws)
        else:
            a, b, c = y[:, :24], y[:, 24:64], y[:, 64:80]
        assert a.shape == (513, 24) and c.shape == (513, 16)
        ((a * w[0]).sum() + (c * w[2]).sum()).backward()   # block b takes no part
        res.append((a.detach().clone(), c.detach().clone(), x.grad.clone()))
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
    assert bool((res[1][2][:, 24:64] == 0).all()) and bool((res[1][2][:, 80:] == 0).all())


def test_split_cols_blocks_written_in_place_by_grouped_layers():
    """Grouped layers reading split_cols blocks write their input gradient straight into the shared gradient tensor
    (ops._grad_dst_view): same bits as plain slicing, whose gradients autograd assembles itself."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(13)
    M, G, K, N = 777, 4, 16, 8
    widths = [G * K, G * K, 32]
    x0 = torch.randn(M, sum(widths), device="cuda", generator=g)
    Ws = [[torch.nn.Parameter(torch.randn(N, K, device="cuda", generator=g) * 0.3) for _ in range(G)] for _ in range(2)]
    tail_w = torch.randn(M, 32, device="cuda", generator=g)
    res = []
    for split in (False, True):
        x = x0.clone().requires_grad_(True)
        y = x * 1.5
        blocks = ops.split_cols(y, widths) if split else (y[:, :G * K], y[:, G * K:2 * G * K], y[:, 2 * G * K:])
        outs = [ops.linear_bn_act(blocks[i], Ws[i], None, bn=None, acts=("sigmoid" if i else "relu"), groups=G,
                                  training=True) for i in range(2)]
        for w_ in Ws[0] + Ws[1]:
            w_.grad = None
        (outs[0].sum() + (outs[1] * outs[1]).sum() + (blocks[2] * tail_w).sum()).backward()
        res.append((x.grad.clone(), [w_.grad.clone() for w_ in Ws[0] + Ws[1]]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,offset", [(4099 * 33, 0), (1000, 1), (7, 0)])
def test_mul_backward_in_one_pass(n, offset):
    """ops.mul's backward (swr_mul_scale_bwd: both gradients from one pass over dC) against the two products it replaces."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(n)
    buf = [torch.randn(n + offset, device="cuda", generator=g)[offset:] for _ in range(3)]
    a, b = buf[0].clone().requires_grad_(True), buf[1].clone().requires_grad_(True)
    dc, s = buf[2], 1.7
    ops.mul(a, b, s).backward(dc)
    assert torch.equal(a.grad, dc * (b.detach() * s)) and torch.equal(b.grad, dc * (a.detach() * s))
    a2 = buf[0].clone().requires_grad_(True)
    ops.mul(a2, b.detach(), s).backward(dc)                   # one-sided: the forward kernel on (dC, b)
    assert torch.equal(a2.grad, a.grad)
    H.check_errors()


@pytest.mark.parametrize("n,offset", [(4099 * 17, 0), (1001, 1), (5, 0)])
def test_mul_sigmoid_against_torch(n, offset):
    """ops.mul_sigmoid (GateNU sigmoid + gating product in one pass each way) against torch in fp64."""
    from scenario_wise_rec import ops
    g = torch.Generator(device="cuda").manual_seed(n + 3)
    buf = [torch.randn(n + offset, device="cuda", generator=g)[offset:] * 3 for _ in range(3)]
    a, z = buf[0].clone().requires_grad_(True), buf[1].clone().requires_grad_(True)
    dc, s = buf[2], 2.0
    c = ops.mul_sigmoid(a, z, s)
    c.backward(dc)
    a64, z64 = a.detach().double().requires_grad_(True), z.detach().double().requires_grad_(True)
    ref = a64 * (s * torch.sigmoid(z64))
    ref.backward(dc.double())
    torch.testing.assert_close(c.double(), ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(a.grad.double(), a64.grad, rtol=1e-6, atol=1e-6)
    # (1 - y) is formed in fp32 from y, as in the two-kernel path and in torch's own sigmoid backward: absolute, not relative
    torch.testing.assert_close(z.grad.double(), z64.grad, rtol=0, atol=2e-6 * float(z64.grad.abs().max()) + 1e-7)


@pytest.mark.parametrize("M,N,K", [(4096, 148, 276), (5000, 96, 132)])
def test_gemm_presplit_weights_give_the_same_bits(M, N, K):
    """ops.split_weights + gemm(B_split=...): the weights' bf16 planes made once (swr_split_weights) instead of inside every
    workgroup -- the same three terms per value, so the product must be bit-identical; the W^T planes serve the dX form."""
    from scenario_wise_rec import ops
    rng = np.random.default_rng(M + K)
    A = _dev(rng.standard_normal((M, K)).astype(np.float32))
    W = _dev(rng.standard_normal((N, K)).astype(np.float32))
    planes, planes_t = ops.split_weights(W, True)
    C0, C1 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    ops.gemm("nt", A, W, C0, M, N, K)
    ops.gemm("nt", A, W, C1, M, N, K, B_split=planes)
    assert torch.equal(C0, C1)
    G = _dev(rng.standard_normal((M, N)).astype(np.float32))
    D0, D1 = torch.empty((M, K), device="cuda"), torch.empty((M, K), device="cuda")
    ops.gemm("nt", G, W.t().contiguous(), D0, M, K, N)
    ops.gemm("nt", G, W, D1, M, K, N, ldb=N, B_split=planes_t)
    assert torch.equal(D0, D1)


def _fused_case(family, E, vocabs, n_dense, B, limit, seed):
    """One training step of a model whose first layer reads the lookup alone: (probabilities, loss, every gradient)."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec.basic.features import DenseFeature, SparseFeature
    from scenario_wise_rec.models.multi_domain import MMOE, SharedBottom
    from scenario_wise_rec.trainers import CTRTrainer
    from _golden import perturb_product
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    feats = [SparseFeature(f"s{i}", v, E) for i, v in enumerate(vocabs)] + [DenseFeature(f"d{i}") for i in range(n_dense)]
    if family == "MMOE":
        model = MMOE(feats, domain_num=5, n_expert=4, expert_params={"dims": [32]}, tower_params={"dims": [16]})
    else:
        model = SharedBottom(feats, domain_num=3, bottom_params={"dims": [128]}, tower_params={"dims": [8]})
    perturb_product(model, seed)
    if limit is not None:
        model.set_dense_table_limit(limit)
    x = {f"s{i}": rng.integers(0, v, size=B).astype([np.int64, np.int32, np.int16][i % 3] if v < 30000 else np.int64)
         for i, v in enumerate(vocabs)}
    x.update({f"d{i}": (rng.random(B).astype(np.float32) if i % 2 == 0 else rng.integers(0, 5, size=B).astype(np.float16))
              for i in range(n_dense)})
    x["domain_indicator"] = rng.integers(0, 5 if family == "MMOE" else 3, size=B)
    y = (rng.random(B) < 0.3).astype(np.float32)
    trainer = CTRTrainer(model, "fused-case", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
    model.train()
    xd = {k: _dev(v) for k, v in x.items()}
    p = model(xd)
    loss = trainer.criterion(p, _dev(y))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    H.check_errors()
    grads = {}
    for k, prm in model.named_parameters():
        sg = getattr(prm, "_swr_sparse_grad", None)
        grads[k] = (sg[0].cpu().numpy(), sg[1].cpu().numpy()) if sg is not None else prm.grad.cpu().numpy().copy()
    return p.detach().cpu().numpy(), float(loss.detach()), grads, model, x, y


@pytest.mark.parametrize("family,E,vocabs,n_dense,B,limit", [
    ("MMOE", 16, [1000, 5000, 8, 2, 3, 51, 1472, 16, 35, 4, 119, 455, 6, 3, 200, 300], 4, 4096 + 37, 65536),   # KuaiRand-like: 9 + 1 real groups incl. a row-sparse table, 5 one-hot groups
    ("MMOE", 16, [7, 50, 3000, 2], 2, 1000, None),                    # (the smoke shape) odd group counts: 3 real + 1 one-hot
    ("MMOE", 16, [40, 50, 5], 0, 250, None),                          # no dense features
    ("SharedBottom", 8, [6040, 3706, 2, 21, 3439, 18], 1, 4096, None),   # MovieLens shape: E = 8 (two tables per group), N = 128
    ("SharedBottom", 32, [300, 9, 1000], 3, 65, 20000),               # E = 32: four pieces per table; one tile and a bit
], ids=["kuairand_like", "smoke_shape", "no_dense", "movielens_e8", "e32_ragged"])
def test_fused_lookup_step_is_bitwise_the_written_layout(family, E, vocabs, n_dense, B, limit, monkeypatch):
    """csrc/first_layer.hip against the path it replaces: with SWR_FUSED_LOOKUP off the lookup WRITES the folded block
    [E_big | dense | 0 | one-hot] and swr_gemm_nt multiplies it; with it on, nothing is written and the product fetches
    table rows through the keys.  Same k order, same six / three bf16 products per group, same epilogue: every
    probability, the loss and every gradient must agree BIT FOR BIT (the written path itself is pinned against the
    reference-generated fixtures and the oracle by the family tests)."""
    from scenario_wise_rec import ops
    monkeypatch.setattr(ops, "FUSED_LOOKUP", False)
    p0, l0, g0, *_ = _fused_case(family, E, vocabs, n_dense, B, limit, 3)
    monkeypatch.setattr(ops, "FUSED_LOOKUP", True)
    calls = []
    real = ops.lib.swr_fl_fwd
    monkeypatch.setattr(ops.lib, "swr_fl_fwd", lambda *a: (calls.append(1), real(*a))[1])
    p1, l1, g1, *_ = _fused_case(family, E, vocabs, n_dense, B, limit, 3)
    assert calls, "the fused product was not taken"
    assert np.array_equal(p0, p1) and l0 == l1
    assert set(g0) == set(g1)
    for k in g0:
        if isinstance(g0[k], tuple):
            assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]), k
        else:
            assert np.array_equal(g0[k], g1[k]), f"{k}: max diff {np.abs(g0[k] - g1[k]).max():.3e}"


@pytest.mark.parametrize("family,E,vocabs,n_dense,B,limit", [
    ("MMOE", 16, [1000, 5000, 8, 2, 3, 51, 1472, 16, 35, 4, 119, 455, 6, 3, 200, 300], 4, 8192 + 37, 65536),
    ("MMOE", 16, [7, 50, 3000, 2], 2, 1000, None),
    ("SharedBottom", 8, [6040, 3706, 2, 21, 3439, 18], 1, 4096, None),
    ("SharedBottom", 32, [300, 9, 1000], 3, 65, 20000),
], ids=["kuairand_like_shard", "smoke_shape", "movielens_e8", "e32_ragged"])
def test_short_batch_column_split_products_are_bitwise_the_whole_row_ones(family, E, vocabs, n_dense, B, limit, monkeypatch):
    """Short batches spread the first layer's forward product and its BatchNorm-backward + dX product over the column tiles too
    (csrc/first_layer.hip, the NA = 1 forms: one 32-column tile per workgroup, grid y = tiles; chosen by batch size, forced here
    through SWR_FL_SPLIT): every output element is the same chain of products, so the step -- probabilities, loss, every
    gradient, the row lists -- must not change by a bit."""
    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SWR_FL_SPLIT", mode)
        res.append(_fused_case(family, E, vocabs, n_dense, B, limit, 3))
    (p0, l0, g0, *_), (p1, l1, g1, *_) = res
    assert np.array_equal(p0, p1) and l0 == l1
    assert set(g0) == set(g1)
    for k in g0:
        if isinstance(g0[k], tuple):
            assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]), k
        else:
            assert np.array_equal(g0[k], g1[k]), f"{k}: max diff {np.abs(g0[k] - g1[k]).max():.3e}"


def test_bn_backward_inside_the_dx_product_is_bitwise_the_two_launches(monkeypatch):
    """swr_bn_bwd_dx (csrc/first_layer.hip): dZ = ca dY + cb (Z - mean) + cc computed in the A fragment of the first layer's dX
    product and written out for the weight gradient, against swr_act_bwd_apply + swr_gemm_nt -- same operations in the same
    order: every gradient BIT FOR BIT.  (A batch of 32 768: the weight-gradient product is forked behind dX; on single-stream
    steps the same launch runs in front of the weight gradient, ops.FUSE_BN_DX_SINGLE.)"""
    from scenario_wise_rec import ops
    vocabs = [1000, 5000, 8, 2, 3, 51, 1472, 16, 35, 4, 119, 455, 6, 3, 200, 300]
    prev = ops.lib.swr_dw_tr_mode(0)          # (the weight gradient in the form whose batch splits are the written-dZ product's)
    try:
        monkeypatch.setattr(ops, "FUSE_BN_DX", False)
        p0, l0, g0, *_ = _fused_case("MMOE", 16, vocabs, 4, 32768, 65536, 5)
        monkeypatch.setattr(ops, "FUSE_BN_DX", True)
        calls = []
        real = ops.lib.swr_bn_bwd_dx
        monkeypatch.setattr(ops.lib, "swr_bn_bwd_dx", lambda *a: (calls.append(1), real(*a))[1])
        p1, l1, g1, *_ = _fused_case("MMOE", 16, vocabs, 4, 32768, 65536, 5)
    finally:
        ops.lib.swr_dw_tr_mode(prev)
    assert calls, "the fused BatchNorm-backward + dX product was not taken"
    assert np.array_equal(p0, p1) and l0 == l1
    for k in g0:
        if isinstance(g0[k], tuple):
            assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]), k
        else:
            assert np.array_equal(g0[k], g1[k]), f"{k}: max diff {np.abs(g0[k] - g1[k]).max():.3e}"


def test_fused_lookup_step_against_the_oracle():
    """The fused lookup + first layer directly against the fp64 oracle (KuaiRand-like shape, a row-sparse table, fp16 / int
    dense features): logits within 1e-4, loss, every gradient."""
    from _golden import assert_probs_close
    from oracle.models import OracleModel
    from oracle.nn import Dense, Sparse
    vocabs = [1000, 5000, 8, 2, 3, 51, 1472, 16, 35, 4, 119, 455, 6, 3, 200, 300]
    from scenario_wise_rec import ops
    assert ops.FUSED_LOOKUP
    p, loss, grads, model, x, y = _fused_case("MMOE", 16, vocabs, 4, 2048 + 5, 65536, 9)
    # (perturb_product ran before the step: the state the step started from = the current parameters, untouched by backward)
    state0 = {k: (v.detach().cpu().numpy().astype(np.float64) if v.dtype.is_floating_point else v.cpu().numpy().copy())
              for k, v in model.state_dict().items()}
    # running statistics moved in the forward pass: the oracle's train-mode forward does not read them
    ofe = [Sparse(f"s{i}", v, 16) for i, v in enumerate(vocabs)] + [Dense(f"d{i}") for i in range(4)]
    om = OracleModel("MMOE", dict(features=ofe, domain_num=5, n_expert=4, expert_params={"dims": [32]}, tower_params={"dims": [16]}),
                     state0, dtype=np.float64)
    xo = {k: (v.astype(np.float64) if k.startswith("d") and k != "domain_indicator" else v) for k, v in x.items()}
    op, oloss, ograds = om.loss_and_grads(xo, y)
    assert_probs_close(p, op, tol=1e-4)
    assert abs(loss - oloss) < 2e-6 * max(1.0, abs(oloss))
    for k, g in ograds.items():
        if isinstance(grads[k], tuple):
            r, gg = grads[k]
            got = np.zeros(g.shape)
            np.add.at(got, r[r >= 0], gg[r >= 0].astype(np.float64))
        else:
            got = grads[k]
        np.testing.assert_allclose(got, g, rtol=0, atol=2e-4 * max(1e-6, float(np.abs(g).max())) + 3e-7, err_msg=k)


@pytest.mark.parametrize("M,G", [(1000, 5), (64, 1), (4099, 3), (65536, 5), (33000, 6)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_tower_weight_gradients_in_one_pass(M, G, accumulate):
    """swr_tower_dw (csrc/tower.hip): dW1_g = dZ1_g^T X_g and db1_g = column sums of dZ1_g for all towers in one pass, against
    fp64 torch; twice the same bits (fixed-order partial sums); rows that do not fill the last 64-row tile."""
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec._hip import lib
    K, Hd = 32, 16
    assert lib.swr_tower_dw_supported(K, Hd, G) == 1 and lib.swr_tower_dw_supported(8, 4, G) == 0
    g = torch.Generator(device="cuda").manual_seed(M + G)
    dZ = torch.randn(M, G * Hd, device="cuda", generator=g)
    xw = torch.randn(M, G * K + 8, device="cuda", generator=g)
    x = xw[:, :G * K]                                          # a view with a row pitch of its own
    init_w = torch.randn(G * Hd, K, device="cuda", generator=g)
    init_b = torch.randn(G * Hd, device="cuda", generator=g)
    nb = lib.swr_tower_dw_workspace_bytes(M, K, Hd, G)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        dW, db = init_w.clone(), init_b.clone()
        H.check(lib.swr_tower_dw(H.ptr(dZ), dZ.stride(0), H.ptr(x), x.stride(0), M, K, Hd, G, H.ptr(dW), H.ptr(db), accumulate,
                                 H.ptr(ws), nb, H.stream()), "swr_tower_dw")
        outs.append((dW, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref_w = torch.stack([dZ[:, t * Hd:(t + 1) * Hd].double().t() @ x[:, t * K:(t + 1) * K].double() for t in range(G)]).reshape(G * Hd, K)
    ref_b = dZ.double().sum(0)
    if accumulate:
        ref_w, ref_b = ref_w + init_w.double(), ref_b + init_b.double()
    scale = float(M) ** 0.5
    torch.testing.assert_close(outs[0][0].double(), ref_w, rtol=0, atol=2e-6 * scale * 4)
    torch.testing.assert_close(outs[0][1].double(), ref_b, rtol=0, atol=2e-6 * scale * 4)


def test_weight_gradient_that_recomputes_dz_is_bitwise_the_written_dz(monkeypatch):
    """swr_fl_dw_bn + swr_bn_bwd_dx(dZ = NULL) (the wide weight-gradient kernel applies the BatchNorm backward to dY and Z while it
    stages them; dZ is never written) against the same step with dZ written by the dX launch and read back by swr_fl_dw: every
    gradient BIT FOR BIT.  Config 2's column layout (160 real + 128 one-hot columns: the shape the wide kernel is built for), a
    batch that does not fill the last 16-row stage of a split.  (swr_dw_tr_mode(0): the wide form; the transpose-read form splits
    the batch differently -- test_transpose_read_weight_gradient.)"""
    from scenario_wise_rec import ops
    vocabs = [1000, 5000, 8, 2, 3, 2, 8, 8, 7, 7, 3, 8, 51, 1472, 16, 35, 4, 119, 455, 8, 6, 6, 3, 3, 3, 3, 3, 3, 3, 8, 200, 300]
    B = 32768 + 8
    prev = ops.lib.swr_dw_tr_mode(0)
    try:
        monkeypatch.setattr(ops, "DZ_FREE", False)
        p0, l0, g0, *_ = _fused_case("MMOE", 16, vocabs, 4, B, 65536, 7)
        monkeypatch.setattr(ops, "DZ_FREE", True)
        calls = []
        real = ops.lib.swr_fl_dw_bn
        monkeypatch.setattr(ops.lib, "swr_fl_dw_bn", lambda *a: (calls.append(1), real(*a))[1])
        p1, l1, g1, *_ = _fused_case("MMOE", 16, vocabs, 4, B, 65536, 7)
    finally:
        ops.lib.swr_dw_tr_mode(prev)
    assert calls, "the weight-gradient product that recomputes dZ was not taken"
    assert np.array_equal(p0, p1) and l0 == l1
    for k in g0:
        if isinstance(g0[k], tuple):
            assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]), k
        else:
            assert np.array_equal(g0[k], g1[k]), f"{k}: max diff {np.abs(g0[k] - g1[k]).max():.3e}"


@pytest.mark.parametrize("B", [32768 + 8, 65536, 4096 + 1, 8192 + 40])
def test_transpose_read_weight_gradient(B):
    """csrc/dw_tr.hip (swr_dw_tr_mode(1), the default): the first layer's weight gradient with both operands stored in LDS as
    they arrive and read back through ds_read_b64_tr_b16, A' from the pre-split pieces by LDS-DMA.  One training step at config 2's
    column layout against the same step through the wide kernel (mode 0): forward and every gradient that does not pass through
    the product bit for bit, the product's outputs (first-layer dW / db, the small tables' gradients) within fp32 summation noise
    -- same bf16 terms, same six / three products, other batch splits; twice the same bits; ragged batches (a last stage of 8 / 1
    valid rows); the oracle test of the fused step runs in this mode (test_fused_lookup_step_against_the_oracle)."""
    from scenario_wise_rec import ops
    vocabs = [1000, 5000, 8, 2, 3, 2, 8, 8, 7, 7, 3, 8, 51, 1472, 16, 35, 4, 119, 455, 8, 6, 6, 3, 3, 3, 3, 3, 3, 3, 8, 200, 300]
    prev = ops.lib.swr_dw_tr_mode(0)
    try:
        p0, l0, g0, *_ = _fused_case("MMOE", 16, vocabs, 4, B, 65536, 11)
        ops.lib.swr_dw_tr_mode(1)
        p1, l1, g1, *_ = _fused_case("MMOE", 16, vocabs, 4, B, 65536, 11)
        p2, l2, g2, *_ = _fused_case("MMOE", 16, vocabs, 4, B, 65536, 11)
    finally:
        ops.lib.swr_dw_tr_mode(prev)
    assert np.array_equal(p0, p1) and l0 == l1
    n_diff = 0
    for k in g0:
        if isinstance(g0[k], tuple):
            assert np.array_equal(g0[k][0], g1[k][0]) and np.array_equal(g0[k][1], g1[k][1]), k
            assert np.array_equal(g1[k][1], g2[k][1]), k
            continue
        assert np.array_equal(g1[k], g2[k]), f"{k}: not deterministic"
        # (a bias in front of a BatchNorm has a mathematically zero gradient: what both kernels return is the summation noise of the
        # column sums of dZ -- measured against the layer's weight gradient, not against itself)
        kw = k[:-4] + "weight" if k.endswith("bias") else k
        scale = max(float(np.abs(g0[k]).max()), float(np.abs(g0[kw]).max()) if kw in g0 and not isinstance(g0[kw], tuple) else 0.0, 1e-12)
        err = float(np.abs(g0[k] - g1[k]).max()) / scale
        assert err < 2e-6, f"{k}: {err:.3e} of the largest entry"
        n_diff += int(err > 0)
    # (at B = 65 536 both kernels split the batch into the same 256 x 16-sample steps and the sums come out identical)
    print(f"B {B}: {n_diff} of {len(g0)} gradients differ in the last bits")
