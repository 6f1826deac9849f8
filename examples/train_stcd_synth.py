#!/usr/bin/env python3
"""The semi-supervised loop of the reference (/root/reference/train_stcd.py:400-460) on the HIP engine, on synthetic tiles:

    python examples/train_stcd_synth.py --n_epochs 3 --batch_size 4 --img_height 128 --img_width 128

Per step, as the reference: labelled pairs and pseudo-change pairs are concatenated into ONE forward of
``SegCD(encoder_name="resnet50")``; loss = BCE+Dice(sigmoid(mask_t1)[:B], building label) + BCE+Dice(sigmoid(change), labels) +
contrastive_loss(sigmoid(change), cd_label, pseudo_label); Adam(1e-3) + Poly; F1 / IoU of the change class.  Everything after
the uint8 tiles runs through the engine's C ABI: pair assembly (stcd_pseudo_pair), ColorJitter / grayscale / blur
(stcd_augment), the network, the three losses, the confusion matrix and the optimizer.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from stcd_amd import synth
from stcd_amd.augment import augment_pair
from stcd_amd.losses import cd_loss, contrastive_loss
from stcd_amd.metrics import SegmentationMetric
from stcd_amd.optim import FlatAdam
from stcd_amd.pseudo import pseudo_change_pairs
from stcd_amd.segcd import SegCD
from stcd_amd.train_loop import Poly

parser = argparse.ArgumentParser()      # the flags of train_stcd.py that matter here
parser.add_argument("--n_epochs", type=int, default=3)
parser.add_argument("--batch_size", type=int, default=4, help="labelled pairs per step (as many pseudo pairs are added)")
parser.add_argument("--img_height", type=int, default=128)
parser.add_argument("--img_width", type=int, default=128)
parser.add_argument("--train_tiles", type=int, default=32)
parser.add_argument("--encoder_weights", type=str, default=None)
parser.add_argument("--no_augment", action="store_true")


def main():
    args = parser.parse_args()
    assert torch.cuda.is_available(), "the engine needs a GPU (no CPU fallback)"
    dev = "cuda:0"
    n, h, w, bs = args.train_tiles, args.img_height, args.img_width, args.batch_size
    sets = []
    for seed in (200, 201):                 # labelled set, pseudo-change set
        a, donor, mask = synth.make_pairs_u8(n, h, w, seed)
        sets.append([torch.from_numpy(t).to(dev) for t in (a, donor, (mask * 255).astype(np.uint8))])
    model = SegCD(encoder_name="resnet50", encoder_weights=args.encoder_weights).to(dev)        # train_stcd.py:631-638
    optimizer = FlatAdam(model, lr=0.001, betas=(0.9, 0.999))
    ipe = n // bs
    sched = Poly(optimizer, args.n_epochs, ipe)
    rng = np.random.default_rng(7)
    history = []
    t0 = time.time()
    for epoch in range(1, args.n_epochs + 1):
        model.train()
        acc = SegmentationMetric(2, dev)
        tot = np.zeros(3)
        order = rng.permutation(n)
        for it in range(ipe):
            idx = torch.from_numpy(order[it * bs:(it + 1) * bs]).to(dev)
            ones = torch.ones(bs, dtype=torch.uint8, device=dev)
            (a, d, mk), (ca, cd, cmk) = sets
            image_A, image_B, cd_label, s_label_A, _ = pseudo_change_pairs(a[idx], d[idx], mk[idx], ones, seed=epoch * 1000 + it)
            CA, CB, CL, _, _ = pseudo_change_pairs(ca[idx], cd[idx], cmk[idx], ones, seed=epoch * 1000 + it + 500)
            if not args.no_augment:
                CA, CB = augment_pair(CA, CB, seed=epoch * 1000 + it)
            cd_label, s_label_A, CL = cd_label.unsqueeze(1), s_label_A.unsqueeze(1), CL.unsqueeze(1)
            optimizer.zero_grad()
            seg_A, seg_B, diff = model(torch.cat((image_A, CA)), torch.cat((image_B, CB)))     # train_stcd.py:421-426
            c_label_data = torch.cat((cd_label, CL))
            seg_loss_A = cd_loss(torch.sigmoid(seg_A)[:bs], s_label_A.float())
            change_prediction = torch.sigmoid(diff)
            loss_cd = cd_loss(change_prediction, c_label_data.float())
            loss_ct = contrastive_loss(change_prediction, cd_label, CL)
            (seg_loss_A + loss_cd + loss_ct).backward()
            optimizer.step()
            sched.step(epoch=epoch - 1)
            acc.add_logits(diff.detach(), c_label_data.squeeze(1))
            tot += np.array([seg_loss_A.item(), loss_cd.item(), loss_ct.item()])
        rec = {"epoch": epoch, "seg_loss": tot[0] / ipe, "cd_loss": tot[1] / ipe, "ct_loss": tot[2] / ipe,
               "train_f1": float(acc.F1score()[1]), "train_iou": float(acc.IntersectionOverUnion()[1])}
        history.append(rec)
        print("epoch %d: Seg_Loss %.3f  CD_Loss %.3f  CT_loss %.3f  train F1 %.3f  IoU %.3f" %
              (epoch, rec["seg_loss"], rec["cd_loss"], rec["ct_loss"], rec["train_f1"], rec["train_iou"]), flush=True)
    torch.cuda.synchronize()
    pairs = args.n_epochs * ipe * 2 * bs
    print("done: %d pairs in %.2f s (%.0f pairs/s)" % (pairs, time.time() - t0, pairs / (time.time() - t0)))
    return history


if __name__ == "__main__":
    main()
