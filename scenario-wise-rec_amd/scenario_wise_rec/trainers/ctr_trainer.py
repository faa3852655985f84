"""CTRTrainer (reference: `trainers/ctr_trainer.py:10-165`): same constructor, `fit`,
`train_one_epoch`, `evaluate`, `evaluate_multi_domain_loss`, `predict`.

The training step is the reference's (to(device) -> forward -> BCE -> zero_grad -> backward ->
optimizer step, `ctr_trainer.py:62-77`) on the HIP path: fused model forward/backward (ops.py), fused BCE,
FusedAdam.  `torch.optim.Adam` (the reference default) is mapped to FusedAdam, which implements the same
update; any other `optimizer_fn` is used as given (every table then takes a dense gradient).  The per-step
`loss.item()` host sync of the reference (`:74`) is kept only at `log_interval` boundaries.

hipGraph: `train_one_epoch` launches the first two batches of a shape eagerly, captures the step on the third
(trainers/graph.py) and from then on copies each batch into the captured input buffers and replays -- one graph
launch per step instead of ~40-150 kernel launches; a batch of another shape (the ragged tail of an epoch) runs
eagerly.  Every batch is applied exactly once either way, bit-identical to eager (tests/test_graph_gpu.py).
`SWR_TRAINER_GRAPH=0` or `trainer.use_graph = False` turns it off.

`gpus=[...]` (the reference wraps the model in single-process nn.DataParallel, `ctr_trainer.py:45-47`): here one
process per GPU -- launch the unchanged script under `python -m torch.distributed.run --nproc-per-node N`; every
process builds the same model and DataLoader (same seed), rank r trains on its row chunk of every batch
(`torch.chunk` semantics, like DataParallel's scatter) through parallel.DataParallelStep: per-shard BatchNorm
statistics, global-mean loss, summed gradients, rank 0's running statistics.
"""
import os
import time

import torch
import tqdm
from sklearn.metrics import log_loss, roc_auc_score

from .. import _hip as H
from .. import ops
from ..basic.callback import EarlyStopper
from ..optim import FusedAdam


class BCELoss(torch.nn.Module):
    """torch.nn.BCELoss(reduction='mean') on the fused HIP kernels."""

    def forward(self, y_pred, y):
        return ops.bce_mean(y_pred, y)


class CTRTrainer(object):
    def __init__(self, model, data_set_type, optimizer_fn=torch.optim.Adam, optimizer_params=None, scheduler_fn=None,
                 scheduler_params=None, n_epoch=10, earlystop_patience=10, device="cpu", gpus=None, model_path="./"):
        self.model = model
        self.data_set_type = data_set_type
        if gpus is None:
            gpus = []
        self.gpus = gpus
        self._dp = None
        self.device = torch.device(device)
        if len(gpus) > 1:
            self.device = self._init_data_parallel(gpus)
        self.model.to(self.device)
        if optimizer_params is None:
            optimizer_params = {"lr": 1e-3, "weight_decay": 1e-5}
        if optimizer_fn is torch.optim.Adam and not optimizer_params.get("amsgrad", False):
            optimizer_fn = FusedAdam
        if not (isinstance(optimizer_fn, type) and issubclass(optimizer_fn, FusedAdam)) and hasattr(self.model, "set_dense_table_limit"):
            # any other optimizer reads `.grad`: every table takes a dense gradient in the arena (the reference's
            # behaviour, nn.Embedding sparse=False); row-sparse gradients are a FusedAdam-only representation
            self.model.set_dense_table_limit(1 << 62)
        self.optimizer = optimizer_fn(self.model.parameters(), **optimizer_params)
        if isinstance(self.optimizer, FusedAdam):
            # this loop is zero_grad -> backward -> step (`ctr_trainer.py:71-73`) and reads no gradient after the step: the
            # update zeroes what it consumed and zero_grad has nothing left to launch (set False to inspect `.grad` after a step)
            self.optimizer.clear_grads = os.environ.get("SWR_CLEAR_GRADS", "1") != "0"
        self.scheduler = None
        if scheduler_fn is not None:
            self.scheduler = scheduler_fn(self.optimizer, **scheduler_params)
        self.criterion = BCELoss()
        self.evaluate_fn = roc_auc_score
        self.n_epoch = n_epoch
        self.early_stopper = EarlyStopper(patience=earlystop_patience)
        self.model_path = model_path
        self.use_graph = os.environ.get("SWR_TRAINER_GRAPH", "1") != "0"
        self._graph = None            # captured step (GraphedStep, or the DataParallelStep after capture)
        self._eager_sig, self._eager_run = None, 0
        if len(gpus) > 1:
            from ..parallel import DataParallelStep
            self._dp = DataParallelStep(self, self._world)

    # ---- gpus=[...]: one process per GPU (`ctr_trainer.py:45-47` is single-process nn.DataParallel) -------------
    def _init_data_parallel(self, gpus):
        import torch.distributed as dist
        if not dist.is_initialized():
            if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) != len(gpus):
                raise RuntimeError(
                    f"CTRTrainer(gpus={list(gpus)}): the MI355X build runs one process per GPU over RCCL.  Launch the same "
                    f"script with `python -m torch.distributed.run --nnodes=1 --nproc-per-node {len(gpus)} "
                    f"--master-addr 127.0.0.1 <script>`; every process then trains on its row chunk of each batch "
                    f"(WORLD_SIZE must equal len(gpus)).")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(os.environ.get("SWR_DP_BACKEND", "nccl"), rank=int(os.environ["RANK"]),
                                    world_size=int(os.environ["WORLD_SIZE"]))
        if dist.get_world_size() != len(gpus):
            raise RuntimeError(f"CTRTrainer(gpus={list(gpus)}) under a process group of {dist.get_world_size()} ranks")
        self._world, self._rank = dist.get_world_size(), dist.get_rank()
        dev = torch.device("cuda", int(gpus[self._rank]))
        torch.cuda.set_device(dev)
        return dev

    def _my_rows(self, x_dict, y):
        """This rank's chunk of a batch: `torch.chunk(dim 0)` like DataParallel's scatter (equal chunks required: the
        exchange averages the ranks' gradients with equal weights)."""
        B = y.shape[0]
        if B % self._world:
            raise ValueError(f"batch of {B} rows does not split evenly over {self._world} GPUs (use drop_last or a batch "
                             f"size divisible by the number of GPUs)")
        n = B // self._world
        lo = self._rank * n
        return {k: v[lo:lo + n] for k, v in x_dict.items()}, y[lo:lo + n]

    # ---- one optimisation step (`ctr_trainer.py:67-73`) ---------------------------------------------
    def forward_backward(self, x_dict, y):
        """forward -> criterion -> zero_grad -> backward (`ctr_trainer.py:69-72`); returns the loss tensor."""
        # zero_grad rides the forward pass's side-stream fork (its fill is pure launch latency on the main stream);
        # if the model forks nothing it runs right after the forward, where the reference has it
        ops._stamp("m_begin")
        ops.add_side_job(self.model.zero_grad)
        if isinstance(self.criterion, BCELoss):
            # the model's final domain select and the criterion in one launch when the model output IS the selected
            # probabilities (every multi-domain model here); otherwise the criterion runs the ordinary way
            with ops.fused_bce(y) as f:
                y_pred = self.model(x_dict)
            loss = f.loss_for(y_pred)
            if loss is None:
                loss = self.criterion(y_pred, y)
        else:
            y_pred = self.model(x_dict)
            loss = self.criterion(y_pred, y)
        ops.flush_loss_rider()       # the optimizer's step bookkeeping, unless the fused loss launch carried it
        ops.run_side_jobs()          # zero_grad, unless the fork already ran it
        ops._stamp("m_fwd_end")
        ops.join_side_extras()       # zero_grad / W^T copies forked during the forward pass: needed from the first backward
        loss.backward(gradient=self._one(loss))     # kernel on; the sort behind them is joined by the embedding backward
        ops._stamp("m_bwd_end")
        ops.join_side_streams()      # (no-op unless no embedding backward ran: nothing forked outlives the step)
        ops._stamp("m_joined")
        return loss

    def train_step(self, x_dict, y):
        if self._dp is not None and not getattr(self._dp, "_inside", False):
            self._dp._inside = True
            try:
                return self._dp.train_step(x_dict, y)
            finally:
                self._dp._inside = False
        return self._local_step(x_dict, y)

    def _local_step(self, x_dict, y):
        if hasattr(self.optimizer, "advance_rider"):
            # the step-counter launch leaves the step: it rides the fused select + BCE launch (or, where the model has none,
            # runs as a side job after the forward pass)
            ops.offer_loss_rider(self.optimizer.advance_rider, self.optimizer.advance_early)
        elif hasattr(self.optimizer, "advance_early"):
            ops.add_side_job(self.optimizer.advance_early, backward_needs=False)
        loss = self.forward_backward(x_dict, y)
        self.optimizer.step()
        ops._stamp("m_end")
        return loss

    def _one(self, loss):
        # d(loss)/d(loss): autograd would fill a fresh ones tensor every step (one more launch on a launch-bound path)
        one = getattr(self, "_grad_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            one = self._grad_one = torch.ones_like(loss)
        return one

    def train_one_epoch(self, data_loader, log_interval=10):
        self.model.train()
        total_loss = None
        tk0 = tqdm.tqdm(data_loader, desc="train", smoothing=0, mininterval=1.0)
        for i, (x_dict, y) in enumerate(tk0):
            if self._dp is not None:
                x_dict, y = self._my_rows(x_dict, y)
            loss = self._step_maybe_graphed(x_dict, y)
            # (a replayed step returns its static loss buffer, which the next replay overwrites: the first loss of a log
            # interval is copied, the sums are new tensors)
            total_loss = loss.clone() if total_loss is None else total_loss + loss
            if (i + 1) % log_interval == 0:
                tk0.set_postfix(loss=total_loss.item() / log_interval)      # the only host sync of the loop
                H.check_errors()
                total_loss = None
        H.check_errors()

    def _graph_ok(self):
        from ..optim import FusedAdam as _FA
        return (self.use_graph and self.device.type == "cuda" and isinstance(self.optimizer, _FA)
                and hasattr(self.model, "arena") and self.model.arena() is not None and self.model.training)

    def _step_maybe_graphed(self, x_dict, y):
        """One batch, applied exactly once: replay of the captured step when the batch has the captured shape; capture
        after two consecutive eager steps of one shape; eager otherwise (first batches, ragged tail).  -> detached loss"""
        from .graph import GraphedStep, batch_signature
        if not self._graph_ok():
            x_dict = {k: v.to(self.device, non_blocking=True) for k, v in x_dict.items()}
            return self.train_step(x_dict, y.to(self.device, non_blocking=True)).detach()
        sig = batch_signature(x_dict, y)
        g = self._graph
        if g is not None and g.signature == sig:
            g.load(x_dict, y)                              # host or device batch -> the captured input buffers
            return g.replay()
        x_dev = {k: v.to(self.device, non_blocking=True) for k, v in x_dict.items()}
        y_dev = y.to(self.device, non_blocking=True)
        if g is None and self._eager_sig == sig and self._eager_run >= 2:
            if self._dp is not None:
                self._dp.signature = sig
                g = self._graph = self._dp.capture(x_dev, y_dev, warmup=0)
            else:
                g = self._graph = GraphedStep(self, x_dev, y_dev, warmup=0, step_fn=self._local_step)
            return g.replay()
        loss = self.train_step(x_dev, y_dev).detach()
        self._eager_run = self._eager_run + 1 if self._eager_sig == sig else 1
        self._eager_sig = sig
        return loss

    def fit(self, train_dataloader, val_dataloader=None):
        for epoch_i in range(self.n_epoch):
            print('epoch:', epoch_i)
            self.train_one_epoch(train_dataloader)
            if self.scheduler is not None:
                if epoch_i % self.scheduler.step_size == 0:
                    print("Current lr : {}".format(self.optimizer.state_dict()['param_groups'][0]['lr']))
                self.scheduler.step()
            if val_dataloader:
                self._sync_running_stats()      # (gpus=[...]) every rank evaluates rank 0's model: one decision for all ranks
                auc, logloss = self.evaluate(self.model, val_dataloader)
                print(f'epoch:{epoch_i} | val auc: {auc} | val logloss: {logloss}')
                if self.early_stopper.stop_training(auc, self.model.state_dict()):
                    print(f'validation: best auc: {self.early_stopper.best_auc}')
                    self.model.load_state_dict(self.early_stopper.best_weights)
                    break
        time_now = time.strftime('%m_%d_%H_%M', time.localtime(int(round(time.time() * 1000)) / 1000))
        name = self.model.__class__.__name__ + "_" + self.data_set_type + "_" + time_now + ".pth"
        state = self.model.state_dict()
        if self._dp is None or self._rank == 0:         # replicas are identical; rank 0's running statistics are the model's
            torch.save(state, os.path.join(self.model_path, name))

    def _sync_running_stats(self):
        """Data parallel: parameters are replica-identical by construction, BatchNorm running statistics are per shard
        (nn.DataParallel keeps replica 0's, `ctr_trainer.py:45-47`).  Before anything is DECIDED from an evaluation -- early
        stopping, the best weights, the checkpoint -- every rank takes rank 0's buffers, so all ranks compute the same
        validation AUC, leave `fit` in the same epoch (a rank that broke out alone would leave the others waiting in the
        next epoch's collectives) and keep the same best weights."""
        if self._dp is None:
            return
        import torch.distributed as dist
        a = self.model.arena() if hasattr(self.model, "arena") else None
        bufs = [a["b"], a["i"]] if a is not None else [b for b in self.model.buffers()]
        for b in bufs:
            if b.numel():
                dist.broadcast(b, src=0)

    def _forward_all(self, model, data_loader, desc, with_domain=False):
        model.eval()
        ys, ps, ds = [], [], []
        with torch.no_grad():
            for x_dict, y in tqdm.tqdm(data_loader, desc=desc, smoothing=0, mininterval=1.0):
                x_dict = {k: v.to(self.device, non_blocking=True) for k, v in x_dict.items()}
                ps.append(model(x_dict).reshape(-1))
                ys.append(y.reshape(-1))
                if with_domain:
                    ds.append(x_dict["domain_indicator"].reshape(-1))
        H.check_errors()
        if not ps:
            return [], [], []
        p = torch.cat(ps).cpu().tolist()
        t = torch.cat([y.cpu() for y in ys]).tolist()
        d = torch.cat(ds).cpu().tolist() if with_domain else []
        return t, p, d

    # ---- metrics on the device: no `.tolist()` of the predictions, no host-side sort ------------
    def _device_metrics_ok(self):
        return self.device.type == "cuda" and self.evaluate_fn is roc_auc_score and os.environ.get("SWR_DEVICE_METRICS", "1") != "0"

    def _forward_all_device(self, model, data_loader, desc):
        model.eval()
        ys, ps, ds = [], [], []
        with torch.no_grad():
            for x_dict, y in tqdm.tqdm(data_loader, desc=desc, smoothing=0, mininterval=1.0):
                x_dict = {k: v.to(self.device, non_blocking=True) for k, v in x_dict.items()}
                ps.append(model(x_dict).reshape(-1))
                ys.append(y.reshape(-1).to(self.device, non_blocking=True))
                ds.append(x_dict["domain_indicator"].reshape(-1))
        H.check_errors()
        if not ps:
            return None
        return torch.cat(ps), torch.cat(ys), torch.cat(ds)

    @staticmethod
    def _metric_pair(rows, pos, two_u, ll_sum):
        """(logloss, auc) of one group from the device sums, with sklearn's conventions for degenerate groups."""
        if rows == 0:
            return None, None
        neg = rows - pos
        if pos == 0 or neg == 0:
            # sklearn.metrics.log_loss cannot infer the two classes from one label (ValueError, as in the reference run)
            raise ValueError("y_true contains only one label (%s). Please provide the list of all expected class labels "
                             "explicitly through the labels argument." % (1.0 if pos else 0.0))
        return ll_sum / rows, two_u / (2.0 * pos * neg)

    def evaluate(self, model, data_loader, mode="val"):
        if self._device_metrics_ok():
            got = self._forward_all_device(model, data_loader, "validation")
            if got is not None:
                rows, pos, two_u, ll = ops.eval_metrics(got[0], got[1], got[2], 1)
                logloss, auc = self._metric_pair(rows[1], pos[1], two_u[1], ll[1])
                return auc, logloss
        targets, predicts, _ = self._forward_all(model, data_loader, "validation")
        return self.evaluate_fn(targets, predicts), log_loss(targets, predicts)

    def evaluate_multi_domain_loss(self, model, data_loader, domain_num):
        """-> (logloss per domain, auc per domain, total logloss, total auc); None for empty domains
        (`ctr_trainer.py:113-152`)."""
        if self._device_metrics_ok():
            got = self._forward_all_device(model, data_loader, "validation")
            if got is None:
                return [None] * domain_num, [None] * domain_num, None, None
            rows, pos, two_u, ll = ops.eval_metrics(got[0], got[1], got[2], domain_num)
            pairs = [self._metric_pair(rows[d], pos[d], two_u[d], ll[d]) for d in range(domain_num + 1)]
            return [p[0] for p in pairs[:-1]], [p[1] for p in pairs[:-1]], pairs[-1][0], pairs[-1][1]
        targets, predicts, domains = self._forward_all(model, data_loader, "validation", with_domain=True)
        domain_logloss, domain_auc = [], []
        for d in range(domain_num):
            t = [a for a, dd in zip(targets, domains) if dd == d]
            p = [a for a, dd in zip(predicts, domains) if dd == d]
            domain_logloss.append(log_loss(t, p) if t else None)
            domain_auc.append(self.evaluate_fn(t, p) if t else None)
        total_logloss = log_loss(targets, predicts) if predicts else None
        total_auc = self.evaluate_fn(targets, predicts) if predicts else None
        return domain_logloss, domain_auc, total_logloss, total_auc

    def predict(self, model, data_loader):
        _, predicts, _ = self._forward_all(model, data_loader, "predict")
        return predicts
