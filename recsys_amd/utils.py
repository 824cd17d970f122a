"""Train / eval loop with the call contract of recsys/utils.py:61-117.

``train_loop(model, data, optimizer, num_epochs, clip_norm, k=10)`` runs the reference's step order
(zero_grad, loss, backward, global clip, step) and the same pre-training + per-epoch evaluation,
with two differences that do not change results: the loss is accumulated on the device (one host
sync per epoch instead of two per step, recsys/utils.py:105-106), and evaluation ranks on the
device (rx_topk_codes) instead of numpy argsort on the host (recsys/utils.py:40-58).

``optimizer`` is either a torch optimizer (dense gradients, the reference's way) or a
``recsys_amd.optim.FusedOptimizer`` (clip handled inside its two-phase update).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .autograd import backward_from_loss
from .dataset import EpochLoader
from .metrics import compute_metrics
from .optim import FusedOptimizer


def topk(scores, array=None, subset=None, k=10):
    """Reference semantics (recsys/utils.py:40-58) for numpy arrays or GPU tensors: indices of the
    k highest scores per row, mapped through ``array`` (or ``subset``) when given."""
    if isinstance(scores, torch.Tensor) and scores.is_cuda:
        if subset is not None:
            subset = torch.as_tensor(subset, device=scores.device).long().contiguous()
            scores = torch.gather(scores, 1, subset)
            array = subset
        codes = None if array is None else torch.as_tensor(array, device=scores.device).long().contiguous()
        if scores.shape[1] > ops.TOPK_MAX_CANDIDATES:
            # wider than the one-wavefront-per-row kernel takes (e.g. full-catalogue scores): a stable
            # ascending sort read backwards, which is the reference's tie order (larger column first)
            order = torch.sort(scores.float(), dim=1, stable=True).indices.flip(1)[:, :k].contiguous()
            return order if codes is None else torch.gather(codes, 1, order)
        return ops.topk_codes(scores.float().contiguous(), codes, k)
    scores = np.asarray(scores)
    if subset is not None:
        scores = np.take_along_axis(scores, subset, axis=1)
        array = subset
    order = np.argsort(scores, axis=1)[:, ::-1][:, :k]
    return order if array is None else np.take_along_axis(array, order, axis=1)


def load_model(model, device, **kwargs):
    return model(**kwargs).to(device)


def eval_loop(model, data, k, epoch=None, num_epochs=None, epoch_loss=None):
    model.eval()
    logits = []
    with torch.no_grad():
        for inputs in data.test_dataloader:
            logits.append(model(inputs))
        scores = torch.cat(logits, dim=0)
        y_pred = topk(scores, array=data.test_codes, k=k)
    if epoch:
        print(f"epoch [{epoch + 1}/{num_epochs}], loss: {epoch_loss:.4f}")
    return compute_metrics(data.test_true, y_pred)


def train_step(model, optimizer, inputs, clip_norm):
    """One batch of recsys/utils.py:96-103.  Returns the loss tensor (still on the device)."""
    if isinstance(optimizer, FusedOptimizer):
        # the backward pass reads the clip setting (one-phase in-place update without a clip,
        # compact gradients + norm partials with one), so it has to be in place before the loss
        optimizer.clip_norm = clip_norm
    optimizer.zero_grad()
    from . import autograd as rxa
    rxa.DEFER_LOSS_FINAL[0] = True          # the backward pass follows at once: its reduce may write the loss
    try:
        loss = model.loss(inputs)
        backward_from_loss(loss)
    finally:
        rxa.DEFER_LOSS_FINAL[0] = False
    if clip_norm is not None and not isinstance(optimizer, FusedOptimizer):
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
    optimizer.step()
    return loss.detach()


class LoopGraph:
    """Replays the train step as one HIP graph inside ``train_epoch``: the first ``eager_steps``
    batches of the first epoch run eagerly (they are ordinary train steps and leave every lazily
    sized buffer at its final size), the step is then recorded once on a static copy of a batch, and
    every later full-size batch is replayed.  A short last batch runs eagerly.

    Two ways to feed the recorded step.  Called per batch (any iterable of batches), each batch is
    copied into the static copy from the host (``GraphedTrainStep.copy_batch``).  Given a whole
    ``dataset.EpochLoader`` (``run_epoch``; ``train_epoch`` does that), the epoch's shuffled tensors
    stay on the device and the recorded step itself ends with the hand-over (rx_epoch_advance: running
    loss, copy of coming slices, cursor), so that the host's share of a step is one graph launch; two
    static batches take turns there, and the graph of one prepares the indices of the other beside its
    own step.  Both ways train on the same batches in the same order."""

    def __init__(self, model, optimizer, eager_steps: int = 2):
        from .graph import GraphedTrainStep
        if not isinstance(optimizer, FusedOptimizer):
            raise TypeError("graph replay of the train step needs a FusedOptimizer")
        self.step = GraphedTrainStep(model, optimizer, warmup=0)
        self.eager_left, self.slot, self.rows = eager_steps, None, None
        # run_epoch: static epoch tensors, its own recording, device-side cursor / running loss
        self.epoch, self.epoch_slot, self.epoch_shape = None, None, None
        self.cursor = self.ticket = self.running = None
        self.paired = False

    def run_epoch(self, model, optimizer, loader, clip_norm):
        """One pass over ``loader`` (a dataset.EpochLoader on the GPU); returns the sum of the batch
        losses as a device tensor, added in batch order like ``train_epoch`` does."""
        tensors, n, B = loader.tensors, loader.length, loader.batch_size
        dev = next(iter(tensors.values())).device
        shape = (n, B, tuple((k, tuple(v.shape[1:]), v.dtype) for k, v in tensors.items()))
        if self.epoch_shape != shape:       # first epoch, or a loader of another shape: new static tensors and recording
            self.epoch = {k: torch.empty_like(v.contiguous()) for k, v in tensors.items()}
            self.epoch_shape, self.epoch_slot = shape, None
            self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
            self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
            self.running = torch.zeros(1, dtype=torch.float64, device=dev)
        if loader.shuffle:
            perm = torch.randperm(n, device=dev, generator=loader.generator)
            for k, v in tensors.items():
                torch.index_select(v, 0, perm, out=self.epoch[k])
        else:
            for k, v in tensors.items():
                self.epoch[k].copy_(v)
        n_full, i = n // B, 0
        cut = lambda j: {k: v[j * B:(j + 1) * B] for k, v in self.epoch.items()}
        running = torch.zeros((), dtype=torch.float64, device=dev)
        while i < n_full and self.eager_left > 0:
            running += train_step(model, optimizer, cut(i), clip_norm)
            self.eager_left -= 1
            i += 1
        if i < n_full:
            if self.epoch_slot is None:
                optimizer.clip_norm = clip_norm
                # Two static batches that take turns -- while the step runs on one, the index preparation
                # (sort, dedup) of the other, which already holds the next batch, runs beside it on a forked
                # stream, and the step's hand-over refills its own batch with the batch after next.  A model
                # that already runs a branch of its own on a second stream (NeuMF's GMF half) gets that
                # preparation on the branch's stream (GraphedTrainStep._shared_branch): graphs with a third
                # concurrent branch have crashed the HIP runtime in hipGraphLaunch here (intermittently, on
                # replay).  RX_LOOP_ONE_BATCH=1 keeps such a model on ONE static batch and one graph per step.
                self.paired = not (getattr(model, "_branch_streams", False) and os.environ.get("RX_LOOP_ONE_BATCH") == "1")
                if self.paired:
                    pair = [{k: v.clone() for k, v in cut(i).items()} for _ in range(2)]
                    hand_over = [(lambda loss, b=b: ops.epoch_advance(self.epoch, b, n_full, self.cursor, self.ticket,
                                                                      loss.detach(), self.running, ahead=2)) for b in pair]
                    self.epoch_slot = tuple(self.step.add_pair(pair, hand_over))   # records; runs only the index preparation
                else:
                    batch = {k: v.clone() for k, v in cut(i).items()}
                    hand_over = lambda loss: ops.epoch_advance(self.epoch, batch, n_full, self.cursor, self.ticket,
                                                               loss.detach(), self.running)
                    self.epoch_slot = (self.step.add_batch(batch, epilogue=hand_over),)   # records, executes nothing
            self.step.copy_batch(self.epoch_slot[0], cut(i))
            if self.paired:
                if i + 1 < n_full:
                    self.step.copy_batch(self.epoch_slot[1], cut(i + 1))
                self.step.refresh_prepared(0)
            self.cursor.fill_(i)
            self.running.copy_(running.reshape(1))
            if self.paired:
                self.step.run_pair_steps(n_full - i)
            else:
                for _ in range(n_full - i):
                    self.step(self.epoch_slot[0])
            running = self.running[0].clone()
        if n % B:
            running += train_step(model, optimizer, {k: v[n_full * B:] for k, v in self.epoch.items()}, clip_norm)
        return running

    def __call__(self, model, optimizer, inputs, clip_norm):
        rows = next(iter(inputs.values())).shape[0]
        if self.eager_left > 0 or (self.rows is not None and rows != self.rows):
            self.eager_left -= 1
            return train_step(model, optimizer, inputs, clip_norm)
        if self.slot is None:
            optimizer.clip_norm = clip_norm
            self.rows = rows
            self.slot = self.step.add_batch({k: v.clone() for k, v in inputs.items()})   # records, executes nothing
        self.step.copy_batch(self.slot, inputs)
        return self.step(self.slot).detach()


def train_epoch(model, optimizer, train_dl, clip_norm, step_fn=None):
    """One pass over ``train_dl`` (recsys/utils.py:93-106 without its two host reads per step);
    returns the sum of the batch losses as a device tensor."""
    step_fn = step_fn or train_step
    if isinstance(step_fn, LoopGraph) and isinstance(train_dl, EpochLoader) and train_dl.length > 0 \
            and all(v.is_cuda and v.element_size() == 8 for v in train_dl.tensors.values()):
        return step_fn.run_epoch(model, optimizer, train_dl, clip_norm)
    running = None
    for inputs in train_dl:
        loss = step_fn(model, optimizer, inputs, clip_norm)
        running = loss.clone() if running is None else running.add_(loss)
    return running


def train_loop(model, data, optimizer, num_epochs, clip_norm, k=10, graph=False):
    """recsys/utils.py:84-117.  ``graph=True`` (GPU): the step is replayed as a HIP graph from the third
    batch on (``LoopGraph``).  That needs a FusedOptimizer; a plain ``torch.optim.Adam`` / ``SGD`` over all
    parameters -- what the reference's pipelines pass (recsys/pipelines/neuralcf/neumf.py:10) -- is
    replaced by its fused equivalent (``FusedOptimizer.from_torch``: same arithmetic, the torch object's
    own state is not advanced)."""
    if graph and not isinstance(optimizer, FusedOptimizer):
        fused = FusedOptimizer.from_torch(optimizer, model, clip_norm) if next(model.parameters()).is_cuda else None
        if fused is None:
            raise TypeError("train_loop(graph=True) takes a FusedOptimizer, or a fresh torch.optim.Adam / SGD without "
                            "weight decay or momentum over all of model.parameters()")
        optimizer = fused
    history = []
    data.create_test_dataloader()
    eval_loop(model, data, k, epoch=None, num_epochs=None, epoch_loss=None)
    on_gpu = next(model.parameters()).is_cuda
    if on_gpu:
        ops.set_index_check("deferred")
    step_fn = LoopGraph(model, optimizer) if graph else None
    try:
        for epoch in range(num_epochs):
            model.train()
            train_dl = data.create_negative_sampled_train_dataloader()
            running = train_epoch(model, optimizer, train_dl, clip_norm, step_fn)
            if on_gpu:
                ops.raise_if_index_error()
            epoch_loss = float(running.item()) / len(train_dl)
            metrics = eval_loop(model, data, k, epoch=epoch, num_epochs=num_epochs, epoch_loss=epoch_loss)
            history.append((epoch_loss,) + tuple(metrics))
    finally:
        if on_gpu:
            ops.set_index_check("sync")
    return history
