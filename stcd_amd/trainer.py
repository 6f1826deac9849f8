"""``CDTrainer`` -- the step loop of /root/reference/models/trainer.py (:24-127 set-up, :280-313 forward/backward,
:316-370 epoch loop, :130-163 / :178-186 / :250-264 checkpoints) driving the HIP engine.

The reference file cannot even be imported (its ``utils`` / ``misc.*`` modules are missing from the repository, SURVEY.md
R3) and breaks on tensor-returning models (R4: ``G_pred[-1]``); this counterpart keeps its attribute names, args fields,
checkpoint keys and epoch order, brings its own ``Logger`` / ``Timer`` / ``ConfuseMatrixMeter``, and normalises
tensor-vs-list model outputs.  Differences that are deliberate and visible:
  * the confusion matrix is accumulated on the device; the host only syncs when a log line is due (every 100 batches)
    instead of ``.cpu()`` every step (trainer.py:205);
  * with a torch.distributed process group, gradients are averaged by ``stcd_amd.ddp.FlatGradReducer``.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from .modules import frozen_weights
from .optim import FlatAdam, FlatAdamW

from . import losses
from .metrics import ConfuseMatrixMeter, SegmentationMetric
from .networks import define_G, get_scheduler


class Logger:
    def __init__(self, path, enabled=True):
        self.path = path
        self.enabled = enabled       # data-parallel runs: only rank 0 prints and appends to the shared log file

    def write(self, message):
        if not self.enabled:
            return
        print(message, end="")
        with open(self.path, "a") as f:
            f.write(message)

    def write_dict_str(self, d):
        self.write("".join("%s: %s\n" % (k, v) for k, v in d.items()))


class Timer:
    def __init__(self):
        self.start = time.time()
        self.progress = 0.0

    def update_progress(self, p):
        self.progress = p

    def get_stage_elapsed(self):
        return max(time.time() - self.start, 1e-9)

    def estimated_remaining(self):
        if self.progress <= 0:
            return 0.0
        return self.get_stage_elapsed() * (1 - self.progress) / self.progress / 3600.0


def _is_engine(net):
    from .modules import HipChangeDetector
    return isinstance(getattr(net, "module", net), HipChangeDetector)


def _as_list(pred):
    return list(pred) if isinstance(pred, (list, tuple)) else [pred]


class CDTrainer:
    def __init__(self, args, dataloaders):
        self.args = args
        self.dataloaders = dataloaders
        self.n_class = args.n_class
        self.net_G = define_G(args=args, gpu_ids=args.gpu_ids)
        self.device = torch.device("cuda:%s" % args.gpu_ids[0] if torch.cuda.is_available() and len(args.gpu_ids) > 0 else "cpu")
        print(self.device)
        self.lr = args.lr
        params = self.net_G.parameters()
        if args.optimizer == "sgd":          # trainer.py:41-50
            self.optimizer_G = optim.SGD(params, lr=self.lr, momentum=0.99, weight_decay=5e-4)
        elif args.optimizer == "adam":       # one fused launch over the flat buffers (stcd_amd.optim) on the engine modules
            self.optimizer_G = (FlatAdam(self.net_G, lr=self.lr, weight_decay=0) if _is_engine(self.net_G)
                                else optim.Adam(params, lr=self.lr, weight_decay=0))
        elif args.optimizer == "adamw":
            self.optimizer_G = (FlatAdamW(self.net_G, lr=self.lr, betas=(0.9, 0.999), weight_decay=0.01) if _is_engine(self.net_G)
                                else optim.AdamW(params, lr=self.lr, betas=(0.9, 0.999), weight_decay=0.01))
        else:
            raise NotImplementedError(args.optimizer)
        self.exp_lr_scheduler_G = get_scheduler(self.optimizer_G, args)
        self.running_metric = ConfuseMatrixMeter(n_class=2)
        self._dev_metric = None
        # one process per GPU: every rank trains, rank 0 alone owns the files (log, checkpoints, curves); the confusion
        # matrix is summed over the ranks before any score is formed, so all ranks agree on "best"
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self._rank = torch.distributed.get_rank() if dist_on else 0
        self._world = torch.distributed.get_world_size() if dist_on else 1
        os.makedirs(args.checkpoint_dir, exist_ok=True)
        os.makedirs(args.vis_dir, exist_ok=True)
        self.logger = Logger(os.path.join(args.checkpoint_dir, "log.txt"), enabled=self._rank == 0)
        self.logger.write_dict_str(args.__dict__)
        self.timer = Timer()
        self.batch_size = args.batch_size
        self.epoch_acc = 0
        self.best_val_acc = 0.0
        self.best_epoch_id = 0
        self.epoch_to_start = 0
        self.max_num_epochs = args.max_epochs
        self.global_step = 0
        self.steps_per_epoch = len(dataloaders["train"])
        self.total_steps = (self.max_num_epochs - self.epoch_to_start) * self.steps_per_epoch
        self.G_pred = None
        self.pred_vis = None
        self.batch = None
        self.G_loss = None
        self.is_training = False
        self.batch_id = 0
        self.epoch_id = 0
        self.checkpoint_dir = args.checkpoint_dir
        self.vis_dir = args.vis_dir
        self.shuffle_AB = getattr(args, "shuffle_AB", False)
        self.multi_scale_train = getattr(args, "multi_scale_train", "False")
        self.multi_scale_infer = getattr(args, "multi_scale_infer", "False")
        if self.multi_scale_train == "True" and hasattr(self.net_G, "set_multi_scale_train"):
            self.net_G.set_multi_scale_train(True)      # ChangeFormer: plan the auxiliary heads' backward (trainer.py:300-309)
        self.weights = tuple(getattr(args, "multi_pred_weights", (1.0,)))
        if args.loss == "ce":                # trainer.py:92-114
            self._pxl_loss = losses.cross_entropy
        elif args.loss == "bce":
            self._pxl_loss = F.binary_cross_entropy
        elif args.loss == "cd_loss":
            self._pxl_loss = losses.cd_loss
        elif args.loss in ("fl", "miou", "mmiou"):
            raise NotImplementedError("loss '%s' is outside the accelerated hot path of this build" % args.loss)
        else:
            raise NotImplementedError(args.loss)
        self.VAL_ACC = np.array([], np.float32)
        if os.path.exists(os.path.join(self.checkpoint_dir, "val_acc.npy")):
            self.VAL_ACC = np.load(os.path.join(self.checkpoint_dir, "val_acc.npy"))
        self.TRAIN_ACC = np.array([], np.float32)
        if os.path.exists(os.path.join(self.checkpoint_dir, "train_acc.npy")):
            self.TRAIN_ACC = np.load(os.path.join(self.checkpoint_dir, "train_acc.npy"))
        self._reducer = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            from .ddp import FlatGradReducer, broadcast_parameters
            broadcast_parameters(self.net_G)
            self._reducer = FlatGradReducer(self.net_G)

    # ------------------------------------------------------------------ checkpoints (trainer.py:130-163, 178-186)
    def _barrier(self):
        if self._world > 1:
            torch.distributed.barrier()

    def _load_checkpoint(self, ckpt_name="last_ckpt.pt"):
        self._barrier()              # rank 0 has finished writing before anyone reads
        path = os.path.join(self.checkpoint_dir, ckpt_name)
        if os.path.exists(path):
            self.logger.write("loading last checkpoint...\n")
            checkpoint = torch.load(path, map_location=self.device, weights_only=False)
            self.net_G.load_state_dict(checkpoint["model_G_state_dict"])
            self.optimizer_G.load_state_dict(checkpoint["optimizer_G_state_dict"])
            self.exp_lr_scheduler_G.load_state_dict(checkpoint["exp_lr_scheduler_G_state_dict"])
            self.net_G.to(self.device)
            self.epoch_to_start = checkpoint["epoch_id"] + 1
            self.best_val_acc = checkpoint["best_val_acc"]
            self.best_epoch_id = checkpoint["best_epoch_id"]
            self.total_steps = (self.max_num_epochs - self.epoch_to_start) * self.steps_per_epoch
            self.logger.write("Epoch_to_start = %d, Historical_best_acc = %.4f (at epoch %d)\n\n" %
                              (self.epoch_to_start, self.best_val_acc, self.best_epoch_id))
        elif getattr(self.args, "pretrain", None) is not None:
            print("Initializing backbone weights from: " + self.args.pretrain)
            self.net_G.load_state_dict(torch.load(self.args.pretrain, map_location=self.device), strict=False)
            self.net_G.to(self.device)
            self.net_G.eval()
        else:
            print("training from scratch...")

    def _save_checkpoint(self, ckpt_name):
        if self._rank != 0:
            return
        torch.save({
            "epoch_id": self.epoch_id,
            "best_val_acc": self.best_val_acc,
            "best_epoch_id": self.best_epoch_id,
            "model_G_state_dict": self.net_G.state_dict(),
            "optimizer_G_state_dict": self.optimizer_G.state_dict(),
            "exp_lr_scheduler_G_state_dict": self.exp_lr_scheduler_G.state_dict(),
        }, os.path.join(self.checkpoint_dir, ckpt_name))

    # ------------------------------------------------------------------ bookkeeping
    def _timer_update(self):
        self.global_step = (self.epoch_id - self.epoch_to_start) * self.steps_per_epoch + self.batch_id
        self.timer.update_progress((self.global_step + 1) / max(self.total_steps, 1))
        est = self.timer.estimated_remaining()
        imps = (self.global_step + 1) * self.batch_size / self.timer.get_stage_elapsed()
        return imps, est

    def _visualize_pred(self):
        return torch.argmax(self.G_final_pred, dim=1, keepdim=True) * 255

    def _update_lr_schedulers(self):
        self.exp_lr_scheduler_G.step()

    def _update_metric(self):
        """Accumulate the 2x2 confusion matrix.  On the GPU this is one kernel and no sync; the running mean-F1 is
        materialised only when asked for (``_running_mf1``)."""
        target = self.batch["L"].to(self.device).detach()
        G_pred = self.G_final_pred.detach()
        assert self.args.n_class == G_pred.shape[1]
        if G_pred.is_cuda:
            if self._dev_metric is None:
                self._dev_metric = SegmentationMetric(2, self.device)
            logits = G_pred if self.args.n_class > 1 else (G_pred - 0.5)   # n_class==1: pred = (p >= 0.5), trainer.py:200-203
            self._dev_metric.add_logits(logits.contiguous().float(), target)
            return None
        pred = torch.argmax(G_pred, dim=1) if self.args.n_class > 1 else (G_pred >= 0.5).long()
        return self.running_metric.update_cm(pr=pred.cpu().numpy(), gt=target.cpu().numpy())

    def _flush_dev_metric(self):
        if self._dev_metric is not None:
            cm = self._dev_metric.confusionMatrix.cpu().numpy()
            self._dev_metric.reset()
            return self.running_metric.add_cm(cm)
        return float(np.nanmean([self.running_metric.get_scores()["f1_0"], self.running_metric.get_scores()["f1_1"]]))

    def _collect_running_batch_states(self):
        running_acc = self._update_metric()
        m = len(self.dataloaders["train"]) if self.is_training else len(self.dataloaders["val"])
        imps, est = self._timer_update()
        if np.mod(self.batch_id, 100) == 1:
            if running_acc is None:
                running_acc = self._flush_dev_metric()
            loss = self.G_loss.item() if self.G_loss is not None else float("nan")
            self.logger.write("Is_training: %s. [%d,%d][%d,%d], imps: %.2f, est: %.2fh, G_loss: %.5f, running_mf1: %.5f\n" %
                              (self.is_training, self.epoch_id, self.max_num_epochs - 1, self.batch_id, m,
                               imps * self.batch_size, est, loss, running_acc))

    def _sync_epoch_metric(self):
        """Sum the epoch's confusion matrix over the ranks (each rank saw its own shard of the data)."""
        if self._world <= 1:
            return
        cm = torch.as_tensor(np.asarray(self.running_metric.cm, np.float64))
        if self.device.type == "cuda" and torch.distributed.get_backend() == "nccl":
            cm = cm.to(self.device)
        torch.distributed.all_reduce(cm, op=torch.distributed.ReduceOp.SUM)
        self.running_metric.cm = cm.cpu().numpy()

    def _collect_epoch_states(self):
        self._flush_dev_metric()
        self._sync_epoch_metric()
        scores = self.running_metric.get_scores()
        self.epoch_acc = scores["mf1"]
        self.logger.write("Is_training: %s. Epoch %d / %d, epoch_mF1= %.5f\n" %
                          (self.is_training, self.epoch_id, self.max_num_epochs - 1, self.epoch_acc))
        self.logger.write("".join("%s: %.5f " % (k, v) for k, v in scores.items()) + "\n\n")

    def _update_checkpoints(self):
        self._save_checkpoint(ckpt_name="last_ckpt.pt")
        self.logger.write("Lastest model updated. Epoch_acc=%.4f, Historical_best_acc=%.4f (at epoch %d)\n\n"
                          % (self.epoch_acc, self.best_val_acc, self.best_epoch_id))
        if self.epoch_acc > self.best_val_acc:
            self.best_val_acc = self.epoch_acc
            self.best_epoch_id = self.epoch_id
            self._save_checkpoint(ckpt_name="best_ckpt.pt")
            self.logger.write("*" * 10 + "Best model updated!\n\n")

    def _update_training_acc_curve(self):
        self.TRAIN_ACC = np.append(self.TRAIN_ACC, [self.epoch_acc])
        if self._rank == 0:
            np.save(os.path.join(self.checkpoint_dir, "train_acc.npy"), self.TRAIN_ACC)

    def _update_val_acc_curve(self):
        self.VAL_ACC = np.append(self.VAL_ACC, [self.epoch_acc])
        if self._rank == 0:
            np.save(os.path.join(self.checkpoint_dir, "val_acc.npy"), self.VAL_ACC)

    def _clear_cache(self):
        self.running_metric.clear()
        if self._dev_metric is not None:
            self._dev_metric.reset()

    # ------------------------------------------------------------------ the step (trainer.py:280-313)
    def _forward_pass(self, batch):
        self.batch = batch
        img_in1 = batch["A"].to(self.device)
        img_in2 = batch["B"].to(self.device)
        self.G_pred = _as_list(self.net_G(img_in1, img_in2))     # tensor-returning models are wrapped
        if self.multi_scale_infer == "True":
            final = torch.zeros_like(self.G_pred[-1])
            for pred in self.G_pred:
                if pred.size(2) != self.G_pred[-1].size(2):
                    final = final + F.interpolate(pred, size=self.G_pred[-1].size(2), mode="nearest")
                else:
                    final = final + pred
            self.G_final_pred = final / len(self.G_pred)
        else:
            self.G_final_pred = self.G_pred[-1]

    def _backward_G(self):
        gt = self.batch["L"].to(self.device).float()
        if self.multi_scale_train == "True":
            temp_loss = 0.0
            for i, pred in enumerate(self.G_pred):
                if pred.size(2) != gt.size(2):
                    temp_loss = temp_loss + self.weights[i] * self._pxl_loss(pred, F.interpolate(gt, size=pred.size(2), mode="nearest"))
                else:
                    temp_loss = temp_loss + self.weights[i] * self._pxl_loss(pred, gt)
            self.G_loss = temp_loss
        else:
            self.G_loss = self._pxl_loss(self.G_pred[-1], gt)
        self.G_loss.backward()

    def train_models(self):
        from .train_loop import quiet_gc
        self._load_checkpoint()
        with quiet_gc():        # keep full cyclic collections over the module tree out of the step loop (train_loop.quiet_gc)
            self._train_epochs()

    def _train_epochs(self):
        for self.epoch_id in range(self.epoch_to_start, self.max_num_epochs):
            self._clear_cache()
            self.is_training = True
            self.net_G.train()
            self.logger.write("lr: %0.7f\n \n" % self.optimizer_G.param_groups[0]["lr"])
            for self.batch_id, batch in enumerate(self.dataloaders["train"], 0):
                self._forward_pass(batch)
                self.optimizer_G.zero_grad()
                self._backward_G()
                self.optimizer_G.step()
                self._collect_running_batch_states()
                self._timer_update()
            weight_dir = getattr(self.args, "weight_dir", None)
            if weight_dir and ((self.max_num_epochs == 100 and self.epoch_id > 50 and (self.epoch_id + 1) % 10 == 0) or
                               (self.max_num_epochs == 200 and self.epoch_id > 100 and (self.epoch_id + 1) % 20 == 0)):
                if self._rank == 0:
                    torch.save(self.net_G, os.path.join(weight_dir, str(self.epoch_id) + ".pth"))
            self._collect_epoch_states()
            self._update_training_acc_curve()
            self._update_lr_schedulers()

            self.logger.write("Begin evaluation...\n")
            self._clear_cache()
            self.is_training = False
            self.net_G.eval()
            with frozen_weights(self.net_G):       # nothing updates the weights during validation: the filters are packed once
                for self.batch_id, batch in enumerate(self.dataloaders["val"], 0):
                    with torch.no_grad():
                        self._forward_pass(batch)
                    self._collect_running_batch_states()
            self._collect_epoch_states()
            self._update_val_acc_curve()
            self._update_checkpoints()
            self._barrier()
