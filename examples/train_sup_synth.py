#!/usr/bin/env python3
"""The reference's supervised pre-training script (/root/reference/train_sup.py:31-44 CLI, :112-185 loop, :303-309 model +
optimizer) on the HIP engine: ``UnetSeg(encoder_name="resnet50")`` learns building masks from single images, and its
checkpoint then initialises the change detector (``SegCD`` has the same ``state_dict``) -- the hand-over the paper's pipeline
relies on.  Tiles are synthetic (stcd_amd.synth); ToTensor / Normalize and the labels are produced on the device.

    python examples/train_sup_synth.py --n_epochs 3 --batch_size 8 --encoder resnet34
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from stcd_amd import synth
from stcd_amd.optim import FlatAdam
from stcd_amd.pseudo import pseudo_change_pairs
from stcd_amd.segcd import SegCD, UnetSeg
from stcd_amd.train_loop import train_seg_epoch

parser = argparse.ArgumentParser()      # the flags of train_sup.py:31-44 that matter here
parser.add_argument("--n_epochs", type=int, default=3)
parser.add_argument("--save_name", type=str, default="", help="checkpoint directory ('' = do not save)")
parser.add_argument("--batch_size", type=int, default=8)
parser.add_argument("--img_height", type=int, default=256)
parser.add_argument("--img_width", type=int, default=256)
parser.add_argument("--train_tiles", type=int, default=64)
parser.add_argument("--val_tiles", type=int, default=16)
parser.add_argument("--encoder", type=str, default="resnet50", help="train_sup.py:303 uses resnet50; any of resnet18/34/50/101/152")
parser.add_argument("--encoder_weights", type=str, default=None, help="None | path | imagenet, as in smp")


class DeviceTiles:
    """Batches of (image, label): the date-A tile normalised on the device and its building mask; iterable like a DataLoader."""

    def __init__(self, n, h, w, batch, seed, device, shuffle):
        a, donor, mask = synth.make_pairs_u8(n, h, w, seed)
        self.a, self.donor = torch.from_numpy(a).to(device), torch.from_numpy(donor).to(device)
        self.mask = torch.from_numpy((mask * 255).astype(np.uint8)).to(device)
        self.zero = torch.zeros(n, dtype=torch.uint8, device=device)
        self.batch, self.shuffle, self.rng, self.n = batch, shuffle, np.random.default_rng(seed + 1), n

    def __len__(self):
        return self.n // self.batch

    def __iter__(self):
        order = self.rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        for i in range(len(self)):
            idx = torch.from_numpy(order[i * self.batch:(i + 1) * self.batch]).to(self.a.device)
            x1, _, _, s_label, _ = pseudo_change_pairs(self.a[idx], self.donor[idx], self.mask[idx], self.zero[idx])
            yield x1, s_label


def main(argv=None):
    args = parser.parse_args(argv)
    assert torch.cuda.is_available(), "the engine needs a GPU (no CPU fallback)"
    device = "cuda:0"
    model = UnetSeg(encoder_name=args.encoder, encoder_weights=args.encoder_weights).to(device)      # train_sup.py:303
    optimizer = FlatAdam(model, lr=0.001, betas=(0.9, 0.999))                                          # train_sup.py:308
    train = DeviceTiles(args.train_tiles, args.img_height, args.img_width, args.batch_size, 200, device, True)
    val = DeviceTiles(args.val_tiles, args.img_height, args.img_width, min(4, args.batch_size), 201, device, False)
    args.save_name = args.save_name or None
    t0 = time.time()
    best, history = train_seg_epoch(model, train, val, optimizer, args, device=device,
                                    on_epoch_end=lambda r: print("epoch %d: loss %.4f  val F1 %.3f  val IoU %.3f" %
                                                                 (r["epoch"], r["seg_loss"], r["val_f1"], r["val_iou"]), flush=True))
    torch.cuda.synchronize()
    n = args.n_epochs * len(train) * args.batch_size
    print("done: %d training images in %.2f s (%.0f images/s incl. validation)" % (n, time.time() - t0, n / (time.time() - t0)))
    # the hand-over: the supervised weights initialise the change detector (same state_dict keys)
    cd = SegCD(encoder_name=args.encoder).to(device)
    cd.load_state_dict((best or model).state_dict())
    cd.eval()
    with torch.no_grad():
        x, _ = next(iter(val))
        m1, m2, change = cd(x, x)
    print("SegCD initialised from the supervised checkpoint: |change| on identical dates = %.3g" % change.abs().max().item())
    return history


if __name__ == "__main__":
    main()
