"""Data-parallel path (N > 1): world_size-2 / -8 process groups over gloo.

CPU test: the product's exchange step (flat all-reduce + row-sparse all-gather/merge) reproduces the
reference's DataParallel gradients recorded in tests/golden/mmoe_dp{2,8}.npz from per-shard local gradients.
GPU test (-m gpu): two ranks sharing cuda:0 run the whole HIP path for one step and land on the golden state."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from _golden import Case, assert_state_close

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


GPU_FAULT_RETRIES = []      # (mode, case, world, rank, first line of the fault) of every retried worker group
# every retry is also appended to this file, so that a run whose LAST test does not execute (-k, -x) still leaves a trace
RETRY_LOG = os.path.join(tempfile.gettempdir(), f"swr_gpu_fault_retries_{os.getpid()}.log")


def run_workers(mode, case, world, extra=(), env_extra=None, _attempt=0):
    """Launch `world` worker processes and wait for them.

    A rank that dies of a GPU FAULT raised by the runtime (`HSA_STATUS_ERROR_*`, "Memory access fault": the process is
    aborted by the driver) -- not of a Python exception, not of a wrong number -- makes the whole group run again, at most
    twice, and the retry is recorded (GPU_FAULT_RETRIES, printed).  History: round 2 saw 9 such aborts + 2 garbage-gradient
    runs in 360 runs of the 8-rank test on the shared test pool and answered with this retry and with ranks taking turns on
    the GPU.  Round 3 looked for a cause instead: (i) the stream-skew harness (tests/test_skew_gpu.py, and the `rows_skewed`
    case below) stretches every fork of the step's stream graph against the others -- all families, eager and captured,
    bit-identical; (ii) tools/dp8_soak.py ran this 8-rank step 145 times un-serialised, 60 times with the skew harness on
    and a PyTorch-only control 145 times, no retry: 0 faults, 0 failures in 2 800 process launches
    (profiles/r03_dp8_soak.txt).  The ranks therefore no longer take turns; the retry stays for runtime aborts only, as a
    guard against the pool, and a numerical mismatch is never retried.  A retry that fires is NOT silent: it is recorded
    and `test_no_gpu_fault_retry_fired` (last test of this module) fails the run."""
    out = tempfile.mkdtemp()
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), mode, case, out, *extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    failed = [(r, log) for r, (p, log) in enumerate(zip(procs, logs)) if p.returncode != 0]
    faults = [(r, line) for r, log in failed for line in log.splitlines()
              if "HSA_STATUS_ERROR" in line or "Memory access fault" in line]
    if faults and _attempt < 2:
        GPU_FAULT_RETRIES.append((mode, case, world) + faults[0])
        with open(RETRY_LOG, "a") as f:
            f.write(repr(GPU_FAULT_RETRIES[-1]) + "\n")
        print(f"[test_parallel] GPU fault in rank {faults[0][0]} ({faults[0][1][-120:]}); running the group again", file=sys.stderr)
        return run_workers(mode, case, world, extra, env_extra, _attempt + 1)
    # the rank that failed FIRST is the interesting one: the others die of "Connection closed by peer"
    failed.sort(key=lambda rl: "Connection closed by peer" in rl[1] or "Connection reset" in rl[1])
    assert not failed, f"{len(failed)} rank(s) failed; rank {failed[0][0]}:\n{failed[0][1][-3000:]}"
    return out


@pytest.mark.parametrize("name,world,allreduce", [("mmoe_dp2", 2, False), ("mmoe_dp8", 8, False), ("mmoe_dp2", 2, True), ("mmoe_dp8", 8, True)])
def test_exchange_step_reproduces_dataparallel_gradients(name, world, allreduce):
    """allreduce: the gradient arena travels in ONE all-reduce (arenas above parallel.allreduce_min_bytes(); forced here),
    the row lists in the all-gather; otherwise one all-gather carries both."""
    out = run_workers("exchange-cpu", name, world, env_extra={"SWR_DP_ALLREDUCE_BYTES": "0"} if allreduce else None)
    got = np.load(os.path.join(out, "exchanged.npz"))
    c = Case(name)
    for k, g in c.group("grad").items():
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(got[k], g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,limit,allreduce", [("mmoe_dp2", 2, None, False), ("mmoe_dp2", 2, 2048, False),
                                                        ("mmoe_dp8", 8, 2048, False), ("mmoe_dp2", 2, 2048, True)])
def test_two_ranks_full_hip_step(name, world, limit, allreduce):
    """limit=2048 bytes forces every table above 32 rows x 16 onto the row-sparse exchange; the 8-rank case (all ranks
    on cuda:0, gloo) runs the split backward, both all-gathers and the sort-free merge at the world size of a full
    node against the reference's 8-shard DataParallel result."""
    env = {}
    if allreduce:
        env["SWR_DP_ALLREDUCE_BYTES"] = "0"        # the arena goes through dist.all_reduce (parallel.allreduce_min_bytes)
    out = run_workers("full-gpu", name, world, extra=() if limit is None else (str(limit),), env_extra=env or None)
    c = Case(name)
    # 0. what the ranks sent each other: every rank received the same bytes, and rank r's local gradient arena is the
    #    oracle's gradient of the local mean loss on shard r (locates a failure: a rank's kernels, or the exchange)
    from _golden import make_oracle
    recv = [np.load(os.path.join(out, f"received_rank{r}.npz")) for r in range(world)] if not allreduce else [None]
    for r in range(1, world if not allreduce else 0):
        for k in recv[0].files:
            assert np.array_equal(recv[r][k], recv[0][k]), f"rank {r} received different bytes for {k}"
    x, y = c.batch(0)
    sh = len(y) // world
    for r in range(world if not allreduce else 0):
        _, _, og = make_oracle(c).loss_and_grads({k: v[r * sh:(r + 1) * sh] for k, v in x.items()}, y[r * sh:(r + 1) * sh])
        for k in recv[0].files:
            scale = max(1e-6, float(np.abs(og[k]).max()))
            np.testing.assert_allclose(recv[0][k][r], og[k], rtol=0, atol=2e-4 * scale + 3e-7,
                                       err_msg=f"local gradient of rank {r}: {k}")
    # 1. the exchanged gradients (what the optimizer consumed) against the reference's summed replica gradients
    grads = [np.load(os.path.join(out, f"grads_rank{r}.npz")) for r in range(world)]
    want = c.group("grad")
    assert set(grads[0].files) == set(want)
    for k, g in want.items():
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(grads[0][k], g, rtol=0, atol=2e-4 * scale + 3e-7, err_msg="grad " + k)
    # 2. replicas are bitwise equal: gradients and every parameter (BatchNorm running statistics are per shard)
    states = [np.load(os.path.join(out, "state1.npz" if r == 0 else f"state1_rank{r}.npz")) for r in range(world)]
    for r in range(1, world):
        for k in grads[0].files:
            assert np.array_equal(grads[r][k], grads[0][k]), f"rank {r} gradient {k} differs from rank 0"
        for k in states[0].files:
            if "running_" not in k and "num_batches_tracked" not in k:
                assert np.array_equal(states[r][k], states[0][k]), f"rank {r} state {k} differs from rank 0"
    # 3. rank 0's state after the step (rank 0's running statistics are the model's, ctr_trainer.py:45-47)
    got = states[0]
    for k, v in c.group("state1").items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v)
        else:
            assert_state_close(got[k], v, c, k, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("limit,env", [(None, {}), (2048, {}), (2048, {"SWR_DP_ALLREDUCE_BYTES": "0"}), (2048, {"SWR_SKEW": "4"})],
                         ids=["dense", "rows", "rows_allreduce", "rows_skewed"])
def test_two_rank_captured_step_matches_eager(limit, env):
    """DataParallelStep.capture (three hipGraphs around eager collectives) vs three eager steps: identical state -- also
    with the arena all-reduced, and with the stream-skew harness stretching the merge / compute streams (ops._skew)."""
    extra = () if limit is None else (str(limit),)
    a = np.load(os.path.join(run_workers("graph-gpu", "mmoe_dp2", 2, extra=extra, env_extra=env or None), "state1.npz"))
    b = np.load(os.path.join(run_workers("graph-gpu", "mmoe_dp2", 2, extra=extra, env_extra=dict(env, DP_EAGER_REFERENCE="1", SWR_SKEW="")),
                             "state1.npz"))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("one_stream", ["0", "1"], ids=["branch", "one_stream"])
@pytest.mark.parametrize("limit", [None, 2048], ids=["dense", "rows"])
def test_one_graph_step_over_rccl_matches_eager(limit, one_stream):
    """Over RCCL the captured data-parallel step is ONE hipGraph with both collectives inside it (parallel._capture_one): the
    rows all-gather on a parallel branch (long batches) or on the step's one stream (short batches: SWR_DP_ONE_STREAM, a
    multi-stream graph is host-bound there).  World size 1 over the nccl backend -- what a one-GPU box can run of the real
    transport: two eager warm-up steps + capture + one replay land bitwise where three eager steps land.  SWR_DP_ONE_GRAPH=1
    FORCES the one-graph capture: a capture that fails raises (it used to fall back to three graphs and pass)."""
    extra = () if limit is None else (str(limit),)
    env = {"DP_WORKER_BACKEND": "nccl", "SWR_DP_ONE_GRAPH": "1", "SWR_DP_ONE_STREAM": one_stream}
    if one_stream == "1":
        env["SWR_SIDE_STREAM"] = "auto"          # the product default (the suite forces the forks on): a short-batch step on one stream
    a = np.load(os.path.join(run_workers("graph-gpu", "mmoe_dp2", 1, extra=extra, env_extra=env), "state1.npz"))
    b = np.load(os.path.join(run_workers("graph-gpu", "mmoe_dp2", 1, extra=extra, env_extra=dict(env, DP_EAGER_REFERENCE="1")), "state1.npz"))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_ctrtrainer_gpus_argument_runs_data_parallel():
    """`CTRTrainer(gpus=[0, 1])` (reference: single-process nn.DataParallel, ctr_trainer.py:45-47) = one process per GPU
    here: every process feeds the WHOLE batch, rank r trains on row chunk r.  One epoch of one batch lands on the
    reference's 2-shard DataParallel state; 4 epochs (eager, eager, capture + replay, replay) equal 4 eager ones
    bitwise and keep the replicas identical."""
    out = run_workers("trainer-gpu", "mmoe_dp2", 2)
    c = Case("mmoe_dp2")
    got = np.load(os.path.join(out, "state1.npz"))
    for k, v in c.group("state1").items():
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v)
        else:
            assert_state_close(got[k], v, c, k, 1)
    a = run_workers("trainer-gpu", "mmoe_dp2", 2, extra=("4",))
    b = run_workers("trainer-gpu", "mmoe_dp2", 2, extra=("4",), env_extra={"DP_EAGER_REFERENCE": "1"})
    ga, gb = np.load(os.path.join(a, "state1.npz")), np.load(os.path.join(b, "state1.npz"))
    ra = np.load(os.path.join(a, "state1_rank1.npz"))
    for k in ga.files:
        assert np.array_equal(ga[k], gb[k]), k
        if "running_" not in k and "num_batches_tracked" not in k:
            assert np.array_equal(ga[k], ra[k]), f"replicas differ: {k}"


@pytest.mark.gpu
def test_no_gpu_fault_retry_fired():
    """Runs last in this module: a worker group that had to be run again after a runtime abort (run_workers) turns the
    run red here instead of vanishing in the stderr of a passing test."""
    assert not GPU_FAULT_RETRIES, f"worker groups were retried after GPU faults: {GPU_FAULT_RETRIES}"
    assert not os.path.exists(RETRY_LOG), open(RETRY_LOG).read()
