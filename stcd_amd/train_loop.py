"""The script-shaped loop: counterpart of ``train_cd_epoch`` / ``Poly`` in /root/reference/train_pse_cd.py
(:199-301, :385-402), i.e. the loop that actually runs with ``SiamUnet_diff.SiamUnet_diff(3, 1)`` (:424):
Adam(1e-3), Poly LR stepped per ITERATION, sigmoid + BCE+Dice, F1/IoU of class 1, best-by-val-IoU checkpoints.

Differences, all on the host side: the loss is the fused HIP kernel (sigmoid inside), the confusion matrix stays on the
device (no ``.cpu()`` per step, train_pse_cd.py:231), TensorBoard is optional (absent in this image)."""
from __future__ import annotations

import contextlib
import os
from copy import deepcopy

import torch

from .losses import bce_dice_with_logits
from .modules import frozen_weights
from .metrics import SegmentationMetric


@contextlib.contextmanager
def quiet_gc():
    """Keep Python's cyclic garbage collector off the step loop's back.

    A full collection walks every container object of the process; with a module tree of several hundred parameter tensors,
    BatchNorm buffers, gradient views and holder modules that is milliseconds of host time, and it lands in the middle of a
    step's launch sequence: the GPU drains its queue and idles.  Measured on SegCD-resnet50 (16 pairs, 256^2): 10.35 -> 9.47 ms per
    step; at 8 pairs 8.10 -> 6.42 ms (`bench.py`, STCD_BENCH_GC=default reproduces the slow case).  ``gc.freeze()`` moves
    everything alive at loop entry into the permanent generation, so the collector only ever looks at the few objects a step
    creates; collection itself stays enabled, and the objects are handed back on exit."""
    import gc
    gc.collect()
    gc.freeze()
    try:
        yield
    finally:
        gc.unfreeze()


class Poly:
    """lr = base * (1 - T/N)^0.9 with T = epoch*iters_per_epoch + cur_iter (train_pse_cd.py:385-402)."""

    def __init__(self, optimizer, num_epochs, iters_per_epoch, warmup_epochs=0):
        self.optimizer = optimizer
        self.iters_per_epoch = iters_per_epoch
        self.cur_iter = 0
        self.N = num_epochs * iters_per_epoch
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _apply(self):
        T = self.last_epoch * self.iters_per_epoch + self.cur_iter
        factor = pow((1 - 1.0 * T / self.N), 0.9)
        if self.warmup_iters > 0 and T < self.warmup_iters:
            factor = 1.0 * T / self.warmup_iters
        self.cur_iter %= self.iters_per_epoch
        self.cur_iter += 1
        assert factor >= 0, "error in lr_scheduler"
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * factor

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self._apply()


def _unwrap(out):
    if isinstance(out, (list, tuple)):
        return out[-1]
    return out


def train_cd_epoch(model, trainloader, valloader, optimizer, args, device="cuda:0", writer=None, on_epoch_end=None):
    """Returns (best_model, history) where history is a list of per-epoch dicts (loss, train/val F1 and IoU)."""
    with quiet_gc():
        return _train_cd_epoch(model, trainloader, valloader, optimizer, args, device, writer, on_epoch_end)


def _train_cd_epoch(model, trainloader, valloader, optimizer, args, device, writer, on_epoch_end):
    previous_best = 0.0
    best_model = None
    history = []
    lr_scheduler = Poly(optimizer=optimizer, num_epochs=args.n_epochs, iters_per_epoch=len(trainloader))
    iters = 0
    for epoch in range(1, args.n_epochs + 1):
        train_acc = SegmentationMetric(numClass=2, device=device)
        model.train()
        total = torch.zeros((), device=device)
        n_it = 0
        for image_A, image_B, cd_label in trainloader:
            image_A = image_A.to(device, non_blocking=True)
            image_B = image_B.to(device, non_blocking=True)
            cd_label = cd_label.to(device, non_blocking=True).unsqueeze(1)
            optimizer.zero_grad()
            diff = _unwrap(model(image_A, image_B))
            cd_loss = bce_dice_with_logits(diff, cd_label.float())
            train_acc.add_logits(diff.detach(), cd_label)
            cd_loss.backward()
            optimizer.step()
            total += cd_loss.detach()
            n_it += 1
            iters += 1
            lr_scheduler.step(epoch=epoch - 1)
        rec = {"epoch": epoch, "cd_loss": (total / max(n_it, 1)).item(),
               "train_f1": float(train_acc.F1score()[1]), "train_iou": float(train_acc.IntersectionOverUnion()[1])}
        model.eval()
        with torch.no_grad(), frozen_weights(model):      # the validation loop does not touch the weights: pack the filters once
            cd_acc = SegmentationMetric(numClass=2, device=device)
            vtot, vn = torch.zeros((), device=device), 0
            for batch in valloader:
                image_A, image_B, cd_label = batch[0].to(device), batch[1].to(device), batch[2].to(device).unsqueeze(1)
                diff = _unwrap(model(image_A, image_B))
                vtot += bce_dice_with_logits(diff, cd_label.float())
                vn += 1
                cd_acc.add_logits(diff, cd_label)
            rec.update(val_loss=(vtot / max(vn, 1)).item(), val_f1=float(cd_acc.F1score()[1]),
                       val_iou=float(cd_acc.IntersectionOverUnion()[1]), val_oa=float(cd_acc.OverallAccuracy()),
                       val_pre=float(cd_acc.Precision()[1]), val_rec=float(cd_acc.Recall()[1]))
        if writer is not None:
            for k, v in rec.items():
                if k != "epoch":
                    writer.add_scalar(k, v, epoch)
        history.append(rec)
        save_name = getattr(args, "save_name", None)
        cd_iou = rec["val_iou"]
        if cd_iou > previous_best:
            if save_name:
                if previous_best != 0 and os.path.exists(os.path.join(save_name, "%.2f_best_model.pth" % previous_best)):
                    os.remove(os.path.join(save_name, "%.2f_best_model.pth" % previous_best))
                os.makedirs(save_name, exist_ok=True)
                sd = (model.module if hasattr(model, "module") else model).state_dict()
                torch.save(sd, os.path.join(save_name, "%.2f_best_model.pth" % cd_iou))
            previous_best = cd_iou
            best_model = deepcopy(model)
        if save_name and epoch % 10 == 0:
            sd = (model.module if hasattr(model, "module") else model).state_dict()
            torch.save(sd, os.path.join(save_name, "%.2f_model.pth" % epoch))
        if on_epoch_end is not None:
            on_epoch_end(rec)
    return best_model, history


def train_seg_epoch(model, trainloader, valloader, optimizer, args, device="cuda:0", writer=None, on_epoch_end=None):
    """The supervised segmentation loop of /root/reference/train_sup.py:112-185 (``pred = model(image)``; sigmoid + criterion;
    Poly LR; validation F1 / IoU of class 1; best-by-IoU checkpoint) with the per-step host syncs removed: the loss is
    accumulated on the device and the metric is the device-side confusion matrix.  Returns (best_model, history)."""
    with quiet_gc():
        return _train_seg_epoch(model, trainloader, valloader, optimizer, args, device, writer, on_epoch_end)


def _train_seg_epoch(model, trainloader, valloader, optimizer, args, device, writer, on_epoch_end):
    previous_best, best_model, history = 0.0, None, []
    lr_scheduler = Poly(optimizer=optimizer, num_epochs=args.n_epochs, iters_per_epoch=len(trainloader))
    for epoch in range(1, args.n_epochs + 1):
        model.train()
        total, n_it = torch.zeros((), device=device), 0
        for image, label in trainloader:
            image, label = image.to(device, non_blocking=True), label.to(device, non_blocking=True).unsqueeze(1)
            optimizer.zero_grad()
            seg_loss = bce_dice_with_logits(_unwrap(model(image)), label.float())      # train_sup.py:133-135
            seg_loss.backward()
            optimizer.step()
            total += seg_loss.detach()
            n_it += 1
            lr_scheduler.step(epoch=epoch - 1)
        rec = {"epoch": epoch, "seg_loss": (total / max(n_it, 1)).item()}
        model.eval()
        with torch.no_grad(), frozen_weights(model):
            acc = SegmentationMetric(numClass=2, device=device)
            vtot, vn = torch.zeros((), device=device), 0
            for image, label in valloader:
                image, label = image.to(device), label.to(device).unsqueeze(1)
                pred = _unwrap(model(image))
                vtot += bce_dice_with_logits(pred, label.float())
                vn += 1
                acc.add_logits(pred, label)                                            # pred > 0.5 after the sigmoid (train_sup.py:162)
            rec.update(val_loss=(vtot / max(vn, 1)).item(), val_f1=float(acc.F1score()[1]), val_iou=float(acc.IntersectionOverUnion()[1]))
        if writer is not None:
            for k, v in rec.items():
                if k != "epoch":
                    writer.add_scalar(k, v, epoch)
        history.append(rec)
        save_name = getattr(args, "save_name", None)
        iou = rec["val_iou"]
        if iou > previous_best:                                                        # train_sup.py:174-181
            if save_name:
                if previous_best != 0 and os.path.exists(os.path.join(save_name, "%.2f_best_model.pth" % previous_best)):
                    os.remove(os.path.join(save_name, "%.2f_best_model.pth" % previous_best))
                os.makedirs(save_name, exist_ok=True)
                torch.save((model.module if hasattr(model, "module") else model).state_dict(), os.path.join(save_name, "%.2f_best_model.pth" % iou))
            previous_best = iou
            best_model = deepcopy(model)
        if on_epoch_end is not None:
            on_epoch_end(rec)
    return best_model, history


class GraphedTrainStep:
    """One training step -- zero_grad, forward, loss, backward, fused Adam -- captured ONCE as a hipGraph and replayed.

    An engine step is hundreds of kernel launches (SegCD-resnet50: 326) and the host needs ~10 us to enqueue each, i.e. ~3.3 ms per
    step whatever the batch; a replayed graph costs the host one call (0.36 ms), which frees the CPU for the input pipeline.  (The
    round-2 review expected it to shorten small-batch steps as well; measured below: it does not.)  What makes the step capturable: forward and backward are sync-free stream work on
    caller-provided buffers; the only per-step host values are the optimizer's scalars (learning rate, bias corrections), which
    the captured Adam launch reads from device memory -- the host writes them into a ring of pinned slots and enqueues the copy in
    front of each replay (``FlatAdam.prepare_graph_step`` / ``step_graph``, ``stcd_adam_step_dev``).  Learning-rate schedulers keep
    working: they change ``param_groups`` on the host between replays.

    MEASURED (MI355X, SegCD-resnet50, bf16, 256 x 256; gpurun_out/segcd_graph.jsonl): host enqueue time falls from 3.7 / 5.3 / 7.8 ms to
    0.36 ms per step at 2 / 8 / 16 pairs -- and the step time does not move (4.39 -> 4.51, 6.29 -> 6.40, 9.29 -> 9.37 ms).  The eager
    step was never waiting for the host (enqueue < step time in every case): what bounds a small-batch step is the GPU-side cost
    of ~330 dependent dispatches (~13 us each), which a graph replays one by one all the same.  The remedy is fewer launches.

    Limits (raised loudly): families whose dropout seed is a launch argument (FC-Siam, ChangeFormer) would replay ONE mask set --
    refused unless their dropout is off; fixed input shapes; the optimizer must be ``FlatAdam`` / ``FlatAdamW``.

        step = GraphedTrainStep(model, opt, lambda outs, y: bce_dice_with_logits(outs[-1], y), (A, B), target)
        for A, B, y in loader: loss = step(A, B, y); sched.step()
    """

    GRAPH_SAFE = ("segcd", "unetseg", "ffctlcd", "snunet")

    def __init__(self, model, optimizer, loss_fn, example_inputs, example_target, warmup: int = 3):
        from ._lib import StcdError
        from .modules import HipChangeDetector
        from .optim import _FlatAdamBase
        inner = model.module if hasattr(model, "module") and not isinstance(model, HipChangeDetector) else model
        if not isinstance(inner, HipChangeDetector) or not isinstance(optimizer, _FlatAdamBase):
            raise StcdError("GraphedTrainStep drives an engine module with a FlatAdam / FlatAdamW optimizer")
        arch = inner._engine.arch
        if not arch.startswith(self.GRAPH_SAFE) and getattr(inner, "_drop_p", 0.2) > 0.0:
            raise StcdError(f"{arch}: the dropout seed of this family is a launch argument -- a captured step would replay one mask set; "
                            "set_dropout_p(0) first, or train it eagerly")
        if inner.grad_stage_hook is not None:
            raise StcdError("a data-parallel gradient hook is installed: collectives are not captured; use the eager step")
        inner._engine.set_wgrad_side(False)                  # a captured step stays on one stream: plan the FC-Siam decoder's weight gradients for it
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self._x = [t.detach().clone() for t in example_inputs]
        self._y = example_target.detach().clone()
        model.train()
        inner._ensure_flat(self._x[0].device)
        optimizer._ensure_state()
        # the warm-up steps below are real training steps on the example batch: snapshot the training state and put it back, so
        # constructing the object leaves model and optimizer exactly where they were
        snap = (inner._flat_params.clone(), inner._flat_bn.clone(), inner._nbt.clone(), optimizer._exp_avg.clone(),
                optimizer._exp_avg_sq.clone(), optimizer._step, inner._steps)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up on a side stream (first-use uploads, allocator, autograd buffers)
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            inner._flat_params.copy_(snap[0]); inner._flat_bn.copy_(snap[1]); inner._nbt.copy_(snap[2])
            optimizer._exp_avg.copy_(snap[3]); optimizer._exp_avg_sq.copy_(snap[4])
        optimizer._step, inner._steps = snap[5], snap[6]
        self.graph = torch.cuda.CUDAGraph()
        optimizer.prepare_graph_step()                       # allocates the scalar buffers the captured launch points at
        optimizer._step -= 1                                 # (the first replay prepares its own scalars)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn(model(*self._x), self._y)
            loss.backward()
            optimizer.step_graph()
            self._loss = loss.detach()

    def _eager_step(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(*self._x), self._y)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, *args):
        *xs, y = args
        for dst, src in zip(self._x, xs):
            dst.copy_(src, non_blocking=True)
        self._y.copy_(y, non_blocking=True)
        self.opt.prepare_graph_step()                        # host scalars of THIS step -> device buffer (stream-ordered before the replay)
        self.graph.replay()
        return self._loss
