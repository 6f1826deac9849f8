#!/usr/bin/env python3
"""The reference's script-shaped training loop (/root/reference/train_pse_cd.py:31-43 CLI, :199-301 loop, :416-433 model +
optimizer) on the HIP engine, with pseudo-change pairs synthesised ON THE DEVICE (stcd_amd.pseudo) instead of read from
WHU-A files: a runnable stand-in for `python train_pse_cd.py` when no dataset is mounted.

    python examples/train_pse_cd_synth.py --n_epochs 3 --batch_size 8 --img_height 256 --img_width 256

Tiles are synthetic (stcd_amd.synth: smoothed noise with re-textured rectangles as the "in-painted" donor image and
their union as the building mask); everything after the uint8 tiles -- pair assembly, ToTensor/Normalize, model,
sigmoid + BCE+Dice, Adam + Poly, F1/IoU, best-by-IoU checkpoint -- runs through the engine's C ABI.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from stcd_amd import synth
from stcd_amd.modules import SiamUnet_diff
from stcd_amd.optim import FlatAdam
from stcd_amd.pseudo import pseudo_change_pairs
from stcd_amd.train_loop import train_cd_epoch

parser = argparse.ArgumentParser()      # same flags as train_pse_cd.py:31-43 (paths are unused here)
parser.add_argument("--n_epochs", type=int, default=3, help="number of epochs of training")
parser.add_argument("--root_path", type=str, default="", help="unused: tiles are synthetic")
parser.add_argument("--dataset_name", type=str, default="synthetic")
parser.add_argument("--CDdataset_name", type=str, default="synthetic")
parser.add_argument("--save_name", type=str, default="", help="checkpoint directory ('' = do not save)")
parser.add_argument("--batch_size", type=int, default=4, help="size of the batches")
parser.add_argument("--n_cpu", type=int, default=0)
parser.add_argument("--img_height", type=int, default=256)
parser.add_argument("--img_width", type=int, default=256)
parser.add_argument("--load_path", type=str, default="")
parser.add_argument("--train_tiles", type=int, default=64)
parser.add_argument("--val_tiles", type=int, default=16)
parser.add_argument("--change_fraction", type=float, default=0.75, help="share of tiles that are in the change list")
parser.add_argument("--net", type=str, default="SiamUnet_diff", choices=["SiamUnet_diff", "SegCD"],
                    help="SegCD = smp.SegCD(encoder_name='resnet50'), the model train_pse_cd.py:426 builds (its third output, the change "
                         "map, feeds the loss as at :224-228); --encoder_weights as in smp (None | path | imagenet)")
parser.add_argument("--encoder_weights", type=str, default=None)


class DevicePairs:
    """Batches of (image_A, image_B, cd_label) assembled on the GPU from uint8 tiles; iterable like a DataLoader."""

    def __init__(self, n, h, w, batch, seed, change_fraction, device, shuffle):
        a, donor, mask = synth.make_pairs_u8(n, h, w, seed)
        rng = np.random.default_rng(seed + 1)
        self.a = torch.from_numpy(a).to(device)
        self.donor = torch.from_numpy(donor).to(device)
        self.mask = torch.from_numpy((mask * 255).astype(np.uint8)).to(device)
        self.change = torch.from_numpy((rng.random(n) < change_fraction).astype(np.uint8)).to(device)
        self.batch, self.shuffle, self.rng, self.n = batch, shuffle, rng, n

    def __len__(self):
        return self.n // self.batch

    def __iter__(self):
        order = self.rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        for i in range(len(self)):
            idx = torch.from_numpy(order[i * self.batch:(i + 1) * self.batch]).to(self.a.device)
            x1, x2, c_label, _, _ = pseudo_change_pairs(self.a[idx], self.donor[idx], self.mask[idx], self.change[idx])
            yield x1, x2, c_label


def main():
    args = parser.parse_args()
    assert torch.cuda.is_available(), "the engine needs a GPU (no CPU fallback)"
    device = "cuda:0"
    if args.net == "SegCD":                                           # train_pse_cd.py:426
        from stcd_amd.segcd import SegCD
        model = SegCD(encoder_name="resnet50", encoder_weights=args.encoder_weights).to(device)
    else:
        model = SiamUnet_diff(3, 1).to(device)                        # train_pse_cd.py:424
    if args.load_path:
        model.load_state_dict(torch.load(args.load_path, map_location="cpu"), strict=False)
    optimizer = FlatAdam(model, lr=0.001, betas=(0.9, 0.999))          # train_pse_cd.py:431, one fused launch per step
    train = DevicePairs(args.train_tiles, args.img_height, args.img_width, args.batch_size, 100, args.change_fraction, device, True)
    val = DevicePairs(args.val_tiles, args.img_height, args.img_width, min(4, args.batch_size), 101, args.change_fraction, device, False)
    args.save_name = args.save_name or None
    t0 = time.time()
    _, history = train_cd_epoch(model, train, val, optimizer, args, device=device,
                                on_epoch_end=lambda r: print("epoch %d: loss %.4f  train F1 %.3f  val F1 %.3f  val IoU %.3f" %
                                                             (r["epoch"], r["cd_loss"], r["train_f1"], r["val_f1"], r["val_iou"]), flush=True))
    torch.cuda.synchronize()
    pairs = args.n_epochs * len(train) * args.batch_size
    print("done: %d training pairs in %.2f s (%.0f pairs/s incl. validation)" % (pairs, time.time() - t0, pairs / (time.time() - t0)))
    return history


if __name__ == "__main__":
    main()
