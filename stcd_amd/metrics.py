"""Confusion-matrix metrics of the path.

``SegmentationMetric`` keeps the 2x2 confusion matrix ON THE DEVICE (one kernel per batch, no ``.cpu()`` sync per
step as in /root/reference/train_pse_cd.py:230-233) and exposes the reference's accessors
(train_pse_cd.py:313-350).  ``ConfuseMatrixMeter`` is the counterpart of the reference's missing
``misc.metric_tool.ConfuseMatrixMeter`` as ``CDTrainer`` uses it (/root/reference/models/trainer.py:55,205,240):
``update_cm(pr, gt)`` returns the running mean-F1, ``get_scores()`` a dict with ``mf1`` and per-class entries.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def scores_from_cm(cm: np.ndarray) -> dict:
    """cm[label, pred] (float64).  Formulae of train_pse_cd.py:313-350."""
    cm = np.asarray(cm, dtype=np.float64)
    diag = np.diag(cm)
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = diag / cm.sum(0)
        recall = diag / cm.sum(1)
        f1 = 2 * precision * recall / (precision + recall)
        iou = diag / (cm.sum(1) + cm.sum(0) - diag)
        oa = diag.sum() / cm.sum()
    return {"precision": precision, "recall": recall, "f1": f1, "iou": iou, "oa": oa}


class SegmentationMetric:
    def __init__(self, numClass=2, device="cuda:0"):
        assert numClass == 2, "binary change detection"
        self.numClass = numClass
        self.device = torch.device(device)
        self.reset()

    def reset(self):
        self._cm = torch.zeros(4, dtype=torch.int64, device=self.device)

    def add_logits(self, logits, label):
        """logits [B,1|2,H,W] fp32 on the device, label [B,H,W] or [B,1,H,W] int64 in {0,1}."""
        B, Cn = logits.shape[:2]
        label = label.long().contiguous()
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().stcd_confusion_update(C.c_void_p(logits.data_ptr()), C.c_void_p(label.data_ptr()), B, Cn,
                                                        logits.numel() // (B * Cn), C.c_void_p(self._cm.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    @property
    def confusionMatrix(self):
        return self._cm.reshape(2, 2).double()          # one sync, when somebody actually looks

    def _s(self):
        return scores_from_cm(self.confusionMatrix.cpu().numpy())

    def OverallAccuracy(self): return self._s()["oa"]
    def Precision(self): return self._s()["precision"]
    def Recall(self): return self._s()["recall"]
    def F1score(self): return self._s()["f1"]
    def IntersectionOverUnion(self): return self._s()["iou"]
    def meanIntersectionOverUnion(self): return float(np.mean(self._s()["iou"]))


class ConfuseMatrixMeter:
    def __init__(self, n_class=2):
        self.n_class = n_class
        self.clear()

    def clear(self):
        self.cm = np.zeros((self.n_class, self.n_class), dtype=np.float64)

    def update_cm(self, pr, gt):
        pr = np.asarray(pr).astype(np.int64).ravel()
        gt = np.asarray(gt).astype(np.int64).ravel()
        ok = (gt >= 0) & (gt < self.n_class)
        self.cm += np.bincount(self.n_class * gt[ok] + pr[ok], minlength=self.n_class ** 2).reshape(self.n_class, self.n_class)
        return float(np.nanmean(scores_from_cm(self.cm)["f1"]))

    def add_cm(self, cm):
        self.cm += np.asarray(cm, dtype=np.float64).reshape(self.n_class, self.n_class)
        return float(np.nanmean(scores_from_cm(self.cm)["f1"]))

    def get_scores(self):
        s = scores_from_cm(self.cm)
        out = {"acc": float(s["oa"]), "miou": float(np.nanmean(s["iou"])), "mf1": float(np.nanmean(s["f1"]))}
        for k in ("iou", "f1", "precision", "recall"):
            for c in range(self.n_class):
                out[f"{k}_{c}"] = float(s[k][c])
        return out
