"""Stream-skew harness (SURVEY.md section 5 "race detection").  The training step runs on two to three
HIP streams with hand-placed edges (the sort half of the embedding backward and the one-shot jobs forked after the lookup,
the first layer's weight-gradient chain forked after dX).  A missing edge does not fail on an idle GPU -- the reader just
happens to start after the writer -- so the harness makes the timing adversarial: an idle-spinning kernel of pseudo-random
length (0 .. 400 us, longer than any kernel of the step) is injected on the forked stream and on the forking stream at
every fork (ops._skew, csrc/moe.hip swr_spin_us).  Three steps of every model family, eager and replayed from a captured
graph, must give the same BITS as the undisturbed run for every seed."""
import numpy as np
import pytest
import torch

from _golden import Case, build_product_model, to_device

pytestmark = pytest.mark.gpu
FAMILIES = ["mmoe", "sharedbottom", "ple", "star", "ppnet", "epnet", "hamur_small", "m3oe", "mmoe_seq"]


def _three_steps(case, seed, graphed):
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec.basic.module import SwrModule
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    old = SwrModule.dense_table_limit_bytes
    SwrModule.dense_table_limit_bytes = 2048           # tables above 32 rows x 16 take the row-sparse path: sort fork + lazy rows
    ops.set_skew(seed)
    try:
        model = build_product_model(case)
        tr = CTRTrainer(model, "skew", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
        tr.use_graph = False
        model.train()
        batches = [(to_device(case.batch(s % 3)[0]), torch.from_numpy(case.batch(s % 3)[1]).cuda()) for s in range(3)]
        losses = []
        if graphed:
            g = GraphedStep(tr, batches[0][0], batches[0][1], warmup=2)      # two eager steps on batch 0, capture, one replay each
            for x, y in batches[1:]:
                g.load(x, y)
                losses.append(float(g.replay().clone()))
        else:
            for x, y in (batches[0], batches[0], batches[1], batches[2]):
                losses.append(float(tr.train_step(x, y).detach()))
            losses = losses[2:]
        torch.cuda.synchronize()
        H.check_errors()
        return {k: v.cpu().numpy() for k, v in model.state_dict().items()}, losses
    finally:
        ops.set_skew(None)
        SwrModule.dense_table_limit_bytes = old


@pytest.mark.parametrize("name", FAMILIES)
def test_results_do_not_depend_on_stream_timing(name):
    c = Case(name)
    ref, ref_losses = _three_steps(c, None, False)
    for seed in (1, 2, 3):
        for graphed in (False, True):
            got, losses = _three_steps(c, seed, graphed)
            assert losses == ref_losses, f"seed {seed}, graphed {graphed}: losses {losses} vs {ref_losses}"
            for k in ref:
                assert np.array_equal(got[k], ref[k]), \
                    f"seed {seed}, graphed {graphed}: {k} differs (max {np.abs(got[k].astype(np.float64) - ref[k]).max():.3e})"


def test_skewed_step_at_the_benched_shape():
    """Config 2 at a batch where the dW chain forks onto its own stream (2e9 <= flop < 2e10) and the 20 000-row table sorts
    on the side stream: the forks the bench line really has."""
    import bench
    from scenario_wise_rec import _hip as H
    from scenario_wise_rec import ops
    from scenario_wise_rec.trainers import CTRTrainer
    from scenario_wise_rec.trainers.graph import GraphedStep
    from _golden import perturb_product
    from test_baseline_shapes_gpu import small_config
    cfg = small_config(2, 20000, 16384)

    def run(seed, graphed):
        ops.set_skew(seed)
        try:
            model, _f = bench.build_model(cfg, seed=3)
            perturb_product(model, 9)
            tr = CTRTrainer(model, "skew2", optimizer_params={"lr": 1e-3, "weight_decay": 1e-5}, device="cuda")
            tr.use_graph = False
            model.train()
            bs = []
            for j in range(3):
                x, y = bench.synth_batch(cfg, cfg["batch"], seed=50 + j)
                bs.append(({k: torch.from_numpy(v).cuda() for k, v in x.items()}, torch.from_numpy(y).cuda()))
            if graphed:
                g = GraphedStep(tr, bs[0][0], bs[0][1], warmup=2)
                for x, y in bs[1:]:
                    g.load(x, y)
                    g.replay()
            else:
                for x, y in (bs[0], bs[0], bs[1], bs[2]):
                    tr.train_step(x, y)
            torch.cuda.synchronize()
            H.check_errors()
            return {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        finally:
            ops.set_skew(None)
    ref = run(None, False)
    for seed in (1, 2):
        for graphed in (False, True):
            got = run(seed, graphed)
            for k in ref:
                assert np.array_equal(got[k], ref[k]), f"seed {seed}, graphed {graphed}: {k}"
