#!/usr/bin/env python3
"""F1 parity fixture (BASELINE.json 'F1 on a LEVIR-CD slice within 0.2 pt of the reference'):
the REFERENCE's own SiamUnet_diff(3, 1) trained on the synthetic LEVIR-CD-shaped slice -- 256 train / 64 val pairs of
256x256, 20 epochs, batch 16, Adam(1e-3) + Poly per iteration, sigmoid + cd_loss (the script loop,
/root/reference/train_pse_cd.py:199-249) -- on the CPU, with the Dropout2d masks supplied (the engine replays the same
masks).  Stores the loss curve, the per-epoch validation F1 / IoU of the change class and the final confusion matrix.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_f1_fixture.py        (authoring container only; ~15 min on 8 cores)
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch

from models.SiamUnet_diff import SiamUnet_diff      # noqa: E402  (reference)
from models import losses as ref_losses             # noqa: E402

from oracle import fcsiam_ref                        # noqa: E402  (synth_state / synth_masks / metrics only)
from stcd_amd import synth                           # noqa: E402
from stcd_amd.train_loop import Poly                 # noqa: E402
from tests.golden.make_golden import install_masks   # noqa: E402

N_TRAIN, N_VAL, SIZE, BS, EPOCHS = 256, 64, 256, 16, 20
# seed 41 is the round-2 fixture g9_f1.npz; `make_f1_fixture.py 42 43 44 45` adds g9_f1_s42.npz ... : K = 5 reference runs that
# differ in the initial weights and the dropout masks (same data), for a statistical F1 comparison (tests/test_trainer_gpu.py)
SEEDS = [int(v) for v in sys.argv[1:]] or [41]
torch.set_num_threads(int(os.environ.get("F1_THREADS", "8")))
# F1_REPLICA=r (round 4): the same protocol with the pairs INSIDE every batch permuted (images, labels, mask rows alike; the
# permutation stream of tests/test_trainer_gpu.py::_f1_run(replica=r)) -- mathematically the same steps, another fp32 summation order:
# an independent sample of the run's chaotic trajectory (a single 320-step run moves a seed's final F1 by +-1.5 pt).  Written to
# g9r<r>_f1_s<seed>.npz; the test averages the reference's replicas per seed, as it averages the engine's.
REPLICA = int(os.environ.get("F1_REPLICA", "0"))


def main():
    for seed in SEEDS:
        run(seed)


def run(SEED):
    a, b, lab = synth.make_batch(N_TRAIN, SIZE, SIZE, seed=900)
    va, vb, vlab = synth.make_batch(N_VAL, SIZE, SIZE, seed=901)
    A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab)
    VA, VB, VL = torch.from_numpy(va), torch.from_numpy(vb), torch.from_numpy(vlab)
    m = SiamUnet_diff(3, 1)
    m.load_state_dict(fcsiam_ref.synth_state("diff", 3, 1, SEED))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    ipe = N_TRAIN // BS
    sched = Poly(opt, EPOCHS, ipe)
    losses_, f1s, ious, cm = [], [], [], None
    t0 = time.time()
    prng = np.random.default_rng(7919 * REPLICA + SEED) if REPLICA else None
    for ep in range(EPOCHS):
        m.train()
        for it in range(ipe):
            sl = slice(it * BS, (it + 1) * BS)
            install_masks(m, fcsiam_ref.synth_masks("diff", BS, SEED + 1000 * ep + it)) if ep == 0 and it == 0 else None
            masks = fcsiam_ref.synth_masks("diff", BS, SEED + 1000 * ep + it)
            a_, b_, l_ = A[sl], B[sl], L[sl]
            if prng is not None:
                perm = torch.from_numpy(prng.permutation(BS))
                a_, b_, l_ = a_[perm], b_[perm], l_[perm]
                masks = {k: (torch.cat([v[perm], v[BS + perm]]) if v.shape[0] == 2 * BS else v[perm]) for k, v in masks.items()}
            for name, mk in masks.items():
                getattr(m, name).mask, getattr(m, name).pos = mk, 0
            opt.zero_grad()
            logits = m(a_, b_)
            loss = ref_losses.cd_loss(torch.sigmoid(logits), l_.float().unsqueeze(1))
            loss.backward()
            opt.step()
            sched.step(epoch=ep)
            losses_.append(loss.item())
        m.eval()
        with torch.no_grad():
            pred = torch.cat([(m(VA[i:i + 16], VB[i:i + 16])[:, 0] > 0).long() for i in range(0, N_VAL, 16)])
        cm = fcsiam_ref.confusion_matrix(pred, VL)
        sc = fcsiam_ref.scores_from_cm(cm)
        f1s.append(float(sc["f1"][1])); ious.append(float(sc["iou"][1]))
        print(f"epoch {ep:2d}  loss {np.mean(losses_[-ipe:]):.4f}  val F1 {f1s[-1]:.4f}  IoU {ious[-1]:.4f}  ({time.time() - t0:.0f} s)", flush=True)
    fname = ("g9_f1.npz" if SEED == 41 else f"g9_f1_s{SEED}.npz") if REPLICA == 0 else f"g9r{REPLICA}_f1_s{SEED}.npz"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), fname), replica=REPLICA,
                        n_train=N_TRAIN, n_val=N_VAL, size=SIZE, batch=BS, epochs=EPOCHS, seed=SEED, data_seed_train=900,
                        data_seed_val=901, losses=np.array(losses_), val_f1=np.array(f1s), val_iou=np.array(ious), cm=cm.numpy())


if __name__ == "__main__":
    main()
