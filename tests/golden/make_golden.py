#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Run in the authoring container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures hold DATA only -- inputs (or the seeds that regenerate them through
oracle.*.synth_state / stcd_amd.synth), outputs, and gradient summaries.  Weights are
not stored: both the generator and the tests rebuild them from the same numpy seed.

Groups:
  G1  per-op vectors through the torch.nn modules exactly as the reference configures them
  G2  whole-model eval/train forward + loss + all parameter gradients (diff/conc/sub/SNUNet)
  G3  config-1 step: SiamUnet_diff(3,2) [2,3,256,256], CE and sigmoid+cd_loss, Adam/AdamW deltas
  G4  5-step loss trajectory
  G5  confusion-matrix metrics (independent check through scikit-learn)
  G6  odd-size (100x100) forward -- ReplicationPad2d branch
  G7  train-mode step at 2x128x128 (diff / conc / SNUNet): the bar of the bf16 production path
  G8  contrastive_loss of train_stcd.py (the reference's own function, compiled from its file)
  G10 SegCD (ResNet-50 UNet, the model the scripts train): eval / train outputs, loss, sampled gradients
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from models.SiamUnet_diff import SiamUnet_diff      # noqa: E402  (reference)
from models.SiamUnet_conc import SiamUnet_conc      # noqa: E402
from models.SiamUnet_sub import SiamUnet_sub        # noqa: E402
from models.SNUNet import SNUNet_ECAM               # noqa: E402
from models.Unet import Unet as RefUnet             # noqa: E402  (FC-EF)
from models.SiamUnet_crossconc import SiamUnet_cross_conc as RefXConc   # noqa: E402
from models import losses as ref_losses             # noqa: E402

from oracle import fcsiam_ref, snunet_ref           # noqa: E402  (only for synth_state / synth_masks)
from stcd_amd import synth                          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_CLS = {"diff": SiamUnet_diff, "conc": SiamUnet_conc, "sub": SiamUnet_sub, "fcef": RefUnet, "xconc": RefXConc}
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def t2n(t):
    return t.detach().cpu().numpy().copy()   # copy: .numpy() aliases tensors that are later updated in place


class MaskFeeder(nn.Module):
    """Stands in for nn.Dropout2d: multiplies by externally supplied [N,C] masks, B rows per call."""

    def __init__(self, mask):
        super().__init__()
        self.mask, self.pos = mask, 0

    def forward(self, x):
        if not self.training:
            return x
        b = x.shape[0]
        m = self.mask[self.pos:self.pos + b]
        self.pos += b
        return x * m[:, :, None, None]


def install_masks(model, masks):
    for name, m in masks.items():
        assert isinstance(getattr(model, name), nn.Dropout2d)
        setattr(model, name, MaskFeeder(m))


def unwrap(out):
    return out[-1] if isinstance(out, (list, tuple)) else out


def grad_summary(model, full_for=None):
    """Per-parameter [sum, l2, first 8, 24 strided samples] ("gs/") and the reference gradient itself ("gf/"): the whole
    tensor up to tests/_util.GF_FULL elements, a fixed random sample of that many elements above (tests/_util.gf_index)."""
    from tests._util import gf_index
    out = {}
    for name, p in model.named_parameters():
        g = p.grad.detach().flatten().double()
        n = g.numel()
        idx = (np.arange(24) * max(n // 24, 1)) % n
        summ = np.concatenate([[g.sum().item(), g.norm().item()], t2n(g[:8]) if n >= 8 else np.pad(t2n(g), (0, 8 - n)),
                               t2n(g[idx])])
        out["gs/" + name] = summ
        out["gf/" + name] = t2n(p.grad).ravel()[gf_index(name, n)]
    return out


def rand_pair(seed, n, h, w):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((n, 3, h, w))).astype(np.float32)
    return torch.from_numpy(a), torch.from_numpy(b)


# ------------------------------------------------------------------------------ G1
def g1_ops():
    print("G1 per-op")
    rng = np.random.default_rng(101)
    r = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    d = {}

    def run(tag, mod, x, extra=None):
        x = x.clone().requires_grad_(True)
        y = mod(x)
        gy = r(*y.shape)
        y.backward(gy)
        d[f"{tag}/x"], d[f"{tag}/y"], d[f"{tag}/gy"], d[f"{tag}/dx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)
        for n_, p in mod.named_parameters():
            d[f"{tag}/{n_}"] = t2n(p)
            d[f"{tag}/d{n_}"] = t2n(p.grad)

    run("conv_3_16", nn.Conv2d(3, 16, kernel_size=3, padding=1), r(2, 3, 8, 8))            # SiamUnet_diff.py:18
    run("conv_16_32", nn.Conv2d(16, 32, kernel_size=3, padding=1), r(2, 16, 12, 10))       # :25
    run("convT_s1_32_16", nn.ConvTranspose2d(32, 16, kernel_size=3, padding=1), r(2, 32, 8, 8))  # :87
    run("convT_s2_16_16", nn.ConvTranspose2d(16, 16, kernel_size=3, padding=1, stride=2, output_padding=1),
        r(2, 16, 6, 5))                                                                      # :85
    run("convT_k2s2_32", nn.ConvTranspose2d(32, 32, 2, stride=2), r(2, 32, 4, 4))          # SNUNet.py:38
    run("conv1x1_128_2", nn.Conv2d(128, 2, kernel_size=1), r(2, 128, 6, 6))                # SNUNet.py:106

    # BatchNorm2d train step incl. running stats (SiamUnet_diff.py:19)
    bn = nn.BatchNorm2d(16)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * r(16)); bn.bias.copy_(0.1 * r(16))
        bn.running_mean.copy_(0.1 * r(16)); bn.running_var.copy_(1 + 0.2 * r(16).abs())
    d["bn/rm0"], d["bn/rv0"] = t2n(bn.running_mean), t2n(bn.running_var)
    bn.train()
    run("bn", bn, 1.5 * r(2, 16, 8, 8) + 0.3)
    d["bn/rm1"], d["bn/rv1"], d["bn/nbt1"] = t2n(bn.running_mean), t2n(bn.running_var), t2n(bn.num_batches_tracked)
    bn.eval()
    d["bn/y_eval"] = t2n(bn(torch.from_numpy(d["bn/x"])))

    # max-pool with ties (post-ReLU zeros tie constantly)
    x = torch.relu(r(2, 4, 8, 8)); x[0, 0, 0:2, 0:2] = 0.7
    x = x.requires_grad_(True)
    y = F.max_pool2d(x, kernel_size=2, stride=2); gy = r(*y.shape); y.backward(gy)
    d["pool/x"], d["pool/y"], d["pool/gy"], d["pool/dx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)
    x = r(1, 2, 7, 5).requires_grad_(True)   # odd size: floor
    y = F.max_pool2d(x, kernel_size=2, stride=2); gy = r(*y.shape); y.backward(gy)
    d["pool_odd/x"], d["pool_odd/y"], d["pool_odd/gy"], d["pool_odd/dx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)

    # skip fusions (SiamUnet_diff.py:150, SiamUnet_sub.py:150)
    a = r(2, 8, 4, 4).requires_grad_(True); b = r(2, 8, 4, 4)
    with torch.no_grad():
        b[0, 0] = a[0, 0]                     # exact ties -> abs' = 0
    b = b.requires_grad_(True)
    g = r(2, 8, 4, 4)
    torch.abs(a - b).backward(g)
    d["fuse/a"], d["fuse/b"], d["fuse/g"] = t2n(a), t2n(b), t2n(g)
    d["fuse/abs"], d["fuse/abs_da"], d["fuse/abs_db"] = t2n(torch.abs(a - b)), t2n(a.grad), t2n(b.grad)

    # ReplicationPad2d((0,1,0,1)) (SiamUnet_diff.py:149)
    x = r(1, 3, 4, 5).requires_grad_(True)
    y = nn.ReplicationPad2d((0, 1, 0, 1))(x); gy = r(*y.shape); y.backward(gy)
    d["rpad/x"], d["rpad/y"], d["rpad/gy"], d["rpad/dx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)

    # losses (models/losses.py:6-21 and :24-34)
    lg = (2 * r(2, 2, 8, 8)).requires_grad_(True)
    tg = torch.from_numpy(rng.integers(0, 2, size=(2, 8, 8))).long()
    tg[0, 0, :3] = 255
    loss = ref_losses.cross_entropy(lg, tg.float().unsqueeze(1)); loss.backward()
    d["ce/logits"], d["ce/target"], d["ce/loss"], d["ce/dlogits"] = t2n(lg), t2n(tg), loss.item(), t2n(lg.grad)
    lg = (3 * r(2, 1, 8, 8)).requires_grad_(True)
    tg = torch.from_numpy(rng.integers(0, 2, size=(2, 1, 8, 8))).float()
    loss = ref_losses.cd_loss(torch.sigmoid(lg), tg); loss.backward()
    d["cd/logits"], d["cd/target"], d["cd/loss"], d["cd/dlogits"] = t2n(lg), t2n(tg), loss.item(), t2n(lg.grad)
    lg = torch.tensor([[[[60.0, -60.0, 120.0, -120.0]]]], requires_grad=True)  # saturation / clamp(-100) branch
    tg = torch.tensor([[[[0.0, 1.0, 0.0, 1.0]]]])
    loss = ref_losses.cd_loss(torch.sigmoid(lg), tg); loss.backward()
    d["cd_sat/logits"], d["cd_sat/target"], d["cd_sat/loss"], d["cd_sat/dlogits"] = t2n(lg), t2n(tg), loss.item(), t2n(lg.grad)
    save("g1_ops.npz", **d)


# ------------------------------------------------------------------------------ G2

def g2_fcsiam(archs=("diff", "conc", "sub")):
    for arch in archs:
        for label in (1, 2):
            print(f"G2 {arch} label={label}")
            seed = 200 + 10 * fcsiam_ref.ARCHS.index(arch) + label
            d = {"seed": seed, "label": label}
            x1, x2 = rand_pair(seed + 1, 2, 32, 32)
            d["x1"], d["x2"] = t2n(x1), t2n(x2)
            # eval with perturbed running stats
            m = REF_CLS[arch](3, label)
            m.load_state_dict(fcsiam_ref.synth_state(arch, 3, label, seed, perturb_running=True))
            m.eval()
            with torch.no_grad():
                d["logits_eval"] = t2n(unwrap(m(x1, x2)))
                y1, y2 = rand_pair(seed + 2, 1, 64, 64)
                d["y1"], d["y2"] = t2n(y1), t2n(y2)
                d["logits_eval_64"] = t2n(unwrap(m(y1, y2)))
            # train step with supplied dropout masks
            m = REF_CLS[arch](3, label)
            m.load_state_dict(fcsiam_ref.synth_state(arch, 3, label, seed))
            install_masks(m, fcsiam_ref.synth_masks(arch, 2, seed + 3))
            m.train()
            logits = unwrap(m(x1, x2))
            rng = np.random.default_rng(seed + 4)
            tgt = torch.from_numpy((rng.random((2, 32, 32)) < 0.2).astype(np.int64))
            d["target"] = t2n(tgt)
            if label == 2:
                loss = ref_losses.cross_entropy(logits, tgt)
            else:
                loss = ref_losses.cd_loss(torch.sigmoid(logits), tgt.float().unsqueeze(1))
            loss.backward()
            d["logits_train"], d["loss"] = t2n(logits), loss.item()
            d.update(grad_summary(m))
            sd = m.state_dict()
            for k in ("bn11", "bn22", "bn43", "bn43d", "bn12d"):
                d[f"rs/{k}.running_mean"] = t2n(sd[f"{k}.running_mean"])
                d[f"rs/{k}.running_var"] = t2n(sd[f"{k}.running_var"])
                d[f"rs/{k}.num_batches_tracked"] = t2n(sd[f"{k}.num_batches_tracked"])
            save(f"g2_{arch}_{label}.npz", **d)


def g2_snunet():
    for label in (1, 2):
        print(f"G2 snunet label={label}")
        seed = 260 + label
        d = {"seed": seed, "label": label}
        x1, x2 = rand_pair(seed + 1, 2, 32, 32)
        d["x1"], d["x2"] = t2n(x1), t2n(x2)
        m = SNUNet_ECAM(3, label)
        m.load_state_dict(snunet_ref.synth_state(3, label, seed, perturb_running=True))
        m.eval()
        with torch.no_grad():
            d["logits_eval"] = t2n(m(x1, x2))
        m = SNUNet_ECAM(3, label)
        m.load_state_dict(snunet_ref.synth_state(3, label, seed))
        m.train()
        logits = m(x1, x2)
        rng = np.random.default_rng(seed + 4)
        tgt = torch.from_numpy((rng.random((2, 32, 32)) < 0.2).astype(np.int64))
        d["target"] = t2n(tgt)
        if label == 2:
            loss = ref_losses.cross_entropy(logits, tgt)
        else:
            loss = ref_losses.cd_loss(torch.sigmoid(logits), tgt.float().unsqueeze(1))
        loss.backward()
        d["logits_train"], d["loss"] = t2n(logits), loss.item()
        d.update(grad_summary(m))
        sd = m.state_dict()
        for k in ("conv0_0.bn1", "conv3_0.bn2", "conv4_0.bn1", "conv0_4.bn2"):
            d[f"rs/{k}.running_mean"] = t2n(sd[f"{k}.running_mean"])
            d[f"rs/{k}.running_var"] = t2n(sd[f"{k}.running_var"])
            d[f"rs/{k}.num_batches_tracked"] = t2n(sd[f"{k}.num_batches_tracked"])
        save(f"g2_snunet_{label}.npz", **d)


# ------------------------------------------------------------------------------ G3
def g3_cfg1():
    """BASELINE.json configs[0]: SiamUnet_diff 3-ch 256x256 synthetic pair, batch=2, one step."""
    print("G3 config-1 step")
    a, b, lab = synth.make_batch(2, 256, 256, seed=1337)
    A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab)
    d = {"data_seed": 1337, "label_pos_frac": float(lab.mean())}
    samp = (np.arange(4096) * 31) % (2 * 256 * 256)
    for tag, label, opt_name in (("ce", 2, "adamw"), ("cd", 1, "adam")):
        seed = 300 + label
        m = SiamUnet_diff(3, label)
        m.load_state_dict(fcsiam_ref.synth_state("diff", 3, label, seed))
        install_masks(m, fcsiam_ref.synth_masks("diff", 2, seed + 3))
        m.train()
        if opt_name == "adamw":   # trainer.py:48-50
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        else:                     # train_pse_cd.py:431
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        opt.zero_grad()
        logits = m(A, B)
        if label == 2:
            loss = ref_losses.cross_entropy(logits, L)
        else:
            loss = ref_losses.cd_loss(torch.sigmoid(logits), L.float().unsqueeze(1))
        loss.backward()
        d[f"{tag}/seed"], d[f"{tag}/loss"] = seed, loss.item()
        lf = logits.detach().flatten()
        d[f"{tag}/logits_sum"], d[f"{tag}/logits_l2"] = lf.double().sum().item(), lf.double().norm().item()
        d[f"{tag}/logits_sample_idx"] = samp[:lf.numel() if lf.numel() < 4096 else 4096] % lf.numel()
        d[f"{tag}/logits_sample"] = t2n(lf[torch.from_numpy(d[f"{tag}/logits_sample_idx"])])
        pred = (logits.argmax(1) if label == 2 else (torch.sigmoid(logits[:, 0]) > 0.5).long())
        d[f"{tag}/mask_packed"] = np.packbits(t2n(pred).astype(np.uint8))
        for k, v in grad_summary(m).items():
            d[f"{tag}/{k}"] = v
        opt.step()
        for k in ("conv11.weight", "bn33.weight", "conv12d.weight"):
            d[f"{tag}/delta/{k}"] = t2n(dict(m.named_parameters())[k].detach() - before[k])
        sd = m.state_dict()
        for k in ("bn11", "bn43", "bn12d"):
            d[f"{tag}/rs/{k}.running_mean"] = t2n(sd[f"{k}.running_mean"])
            d[f"{tag}/rs/{k}.running_var"] = t2n(sd[f"{k}.running_var"])
    save("g3_cfg1.npz", **d)


# ------------------------------------------------------------------------------ G4
def g4_traj():
    print("G4 5-step trajectory")
    a, b, lab = synth.make_batch(2, 64, 64, seed=4242)
    A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab)
    seed = 400
    m = SiamUnet_diff(3, 2)
    m.load_state_dict(fcsiam_ref.synth_state("diff", 3, 2, seed))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    losses = []
    for step in range(5):
        masks = fcsiam_ref.synth_masks("diff", 2, seed + 10 + step)
        for name, mk in masks.items():
            setattr(m, name, MaskFeeder(mk))
        m.train()
        opt.zero_grad()
        loss = ref_losses.cross_entropy(m(A, B), L)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("   losses", losses)
    save("g4_traj.npz", seed=seed, data_seed=4242, losses=np.array(losses))


# ------------------------------------------------------------------------------ G5
def g5_metric():
    """The reference's SegmentationMetric lives in train_pse_cd.py (argparse + smp at import time, not
    importable).  The fixture pins the formulae (train_pse_cd.py:313-368) through an independent
    implementation: scikit-learn."""
    print("G5 metrics")
    from sklearn import metrics as skm

    rng = np.random.default_rng(500)
    lab = (rng.random((4, 1, 64, 64)) < 0.15).astype(np.int64)
    pred = np.where(rng.random(lab.shape) < 0.85, lab, 1 - lab)
    y, p = lab.ravel(), pred.ravel()
    cm = skm.confusion_matrix(y, p, labels=[0, 1]).astype(np.float64)   # rows = label, cols = pred
    save("g5_metric.npz", label=lab, pred=pred, cm=cm,
         f1=skm.f1_score(y, p, average=None, labels=[0, 1]),
         iou=skm.jaccard_score(y, p, average=None, labels=[0, 1]),
         precision=skm.precision_score(y, p, average=None, labels=[0, 1]),
         recall=skm.recall_score(y, p, average=None, labels=[0, 1]),
         oa=skm.accuracy_score(y, p))


# ------------------------------------------------------------------------------ G6
def g6_odd():
    print("G6 odd size 100x100")
    d = {}
    for arch in ("diff", "conc"):
        seed = 600 + fcsiam_ref.ARCHS.index(arch)
        x1, x2 = rand_pair(seed + 1, 1, 100, 100)
        m = REF_CLS[arch](3, 2)
        m.load_state_dict(fcsiam_ref.synth_state(arch, 3, 2, seed, perturb_running=True))
        m.eval()
        with torch.no_grad():
            d[f"{arch}/logits"] = t2n(m(x1, x2))
        d[f"{arch}/seed"] = seed
    save("g6_odd.npz", **d)


# ------------------------------------------------------------------------------ G7
def g7_train128(archs=("diff", "conc", "snunet")):
    """Train-mode step at 2 x 128 x 128 (bottleneck BatchNorm over 2*8*8 samples instead of G2's 2*2*2): the vectors the
    bf16 production path is held to -- logits, loss and every parameter's (sampled) gradient of diff / conc / SNUNet."""
    for arch in archs:
        print(f"G7 {arch} 128x128")
        seed = 700 + ("diff", "conc", "snunet", "fcef", "xconc").index(arch)
        d = {"seed": seed}
        x1, x2 = rand_pair(seed + 1, 2, 128, 128)
        rng = np.random.default_rng(seed + 4)
        tgt = torch.from_numpy((rng.random((2, 128, 128)) < 0.2).astype(np.int64))
        if arch == "snunet":
            m = SNUNet_ECAM(3, 2)
            m.load_state_dict(snunet_ref.synth_state(3, 2, seed))
        else:
            m = REF_CLS[arch](3, 2)
            m.load_state_dict(fcsiam_ref.synth_state(arch, 3, 2, seed))
            install_masks(m, fcsiam_ref.synth_masks(arch, 2, seed + 3))
        m.train()
        logits = unwrap(m(x1, x2))
        loss = ref_losses.cross_entropy(logits, tgt)
        loss.backward()
        d["loss"] = loss.item()
        lf = logits.detach().flatten()
        idx = (np.arange(8192) * 7) % lf.numel()
        d["logits_sample_idx"], d["logits_sample"] = idx, t2n(lf[torch.from_numpy(idx)])
        d["logits_absmean"] = lf.abs().mean().item()
        d.update(grad_summary(m))
        save(f"g7_{arch}_128.npz", **d)


# ------------------------------------------------------------------------------ G8
def g19_fcef():
    """FC-EF (`Unet`, models/Unet.py): the G2 pair (eval + train step with masks, label 1 and 2, 2 x 32 x 32) and the G7 step
    (2 x 128 x 128), from the reference's own class."""
    g2_fcsiam(("fcef",))
    g7_train128(("fcef",))


def g20_xconc():
    """SiamUnet_cross_conc (models/SiamUnet_crossconc.py): the G2 pair and the G7 step from the reference's own class."""
    g2_fcsiam(("xconc",))
    g7_train128(("xconc",))


def g8_contrastive():
    """contrastive_loss of /root/reference/train_stcd.py:334-385.  The module cannot be imported (argparse at import time,
    pytorch_grad_cam / smp / timm absent), so the reference's OWN function is compiled from its file (ast: that one
    FunctionDef, nothing is copied into this repository) and run on CPU with Tensor.cuda() made a no-op."""
    import ast
    print("G8 contrastive_loss")
    src = open("/root/reference/train_stcd.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "contrastive_loss"][0]
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "/root/reference/train_stcd.py", "exec"), ns)
    torch.Tensor.cuda = lambda self, *a, **k: self
    rng = np.random.default_rng(800)
    d = {}
    for tag, shape in (("a", (2, 1, 16, 16)), ("b", (3, 1, 9, 11))):
        b = shape[0]
        pred = torch.from_numpy(rng.random((2 * b,) + shape[1:]).astype(np.float32))
        pred[0, 0, 0, 0] = 1.0                     # |cd - 1| at its kink (sign 0)
        cd = torch.from_numpy(rng.integers(0, 2, size=shape)).long()
        ps = torch.from_numpy(rng.integers(0, 2, size=shape)).long()
        cd[0, 0, 0, 0], ps[0, 0, 0, 0] = 0, 1
        pred.requires_grad_(True)
        loss = ns["contrastive_loss"](pred, cd, ps, None)
        loss.backward()
        d[f"{tag}/pred"], d[f"{tag}/cd_label"], d[f"{tag}/pse_label"] = t2n(pred), t2n(cd), t2n(ps)
        d[f"{tag}/loss"], d[f"{tag}/dpred"] = loss.item(), t2n(pred.grad)
    # all labels equal -> N empty (0 / 1e-8 = 0)
    pred = torch.from_numpy(rng.random((4, 1, 8, 8)).astype(np.float32)).requires_grad_(True)
    lab = torch.from_numpy(rng.integers(0, 2, size=(2, 1, 8, 8))).long()
    loss = ns["contrastive_loss"](pred, lab, lab.clone(), None)
    loss.backward()
    d["same/pred"], d["same/cd_label"], d["same/pse_label"], d["same/loss"], d["same/dpred"] = t2n(pred), t2n(lab), t2n(lab), loss.item(), t2n(pred.grad)
    save("g8_contrastive.npz", **d)


# ------------------------------------------------------------------------------ G10
def _reference_segcd(classes=1, encoder="resnet50"):
    """The reference's own pieces, assembled exactly as SegCD.__init__ / forward state (decoders/unet/model.py:267-332).
    The package `segmentation_models_pytorch` cannot be imported (its __init__ pulls timm; encoders/resnet.py pulls
    torchvision), so: the ResNet is the reference's models/resnet.py (the torchvision code ResNetEncoder subclasses), and
    base/ (Conv2dReLU, SegmentationHead) and decoders/unet/decoder.py (UnetDecoder) are loaded as files under a bare
    package object -- no stand-in for any missing library is written."""
    import importlib.util
    import types
    from models.resnet import ResNet, Bottleneck, BasicBlock
    from oracle import segcd_ref
    expansion, layers = segcd_ref.ENCODERS[encoder]          # encoders/resnet.py:126-171 "params" of each registry entry
    root = "/root/reference/segmentation_models_pytorch"
    if "segmentation_models_pytorch" not in sys.modules:
        pkg = types.ModuleType("segmentation_models_pytorch")
        pkg.__path__ = [root]
        sys.modules["segmentation_models_pytorch"] = pkg
    from segmentation_models_pytorch.base import SegmentationHead
    spec = importlib.util.spec_from_file_location("_ref_unet_decoder", root + "/decoders/unet/decoder.py")
    dec_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dec_mod)

    class RefSegCD(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = ResNet(Bottleneck if expansion == 4 else BasicBlock, list(layers))
            del self.encoder.fc, self.encoder.avgpool                   # ResNetEncoder.__init__ (encoders/resnet.py:44-45)
            self.decoder = dec_mod.UnetDecoder(encoder_channels=segcd_ref.enc_out(encoder), decoder_channels=(256, 128, 64, 32, 16),
                                               n_blocks=5, use_batchnorm=True, center=False, attention_type=None)
            self.segmentation_head = SegmentationHead(in_channels=16, out_channels=classes, activation=None, kernel_size=3)

        def features(self, x):                                          # ResNetEncoder.get_stages / forward (:47-66)
            e = self.encoder
            f = [x]
            x = e.relu(e.bn1(e.conv1(x))); f.append(x)
            x = e.layer1(e.maxpool(x)); f.append(x)
            x = e.layer2(x); f.append(x)
            x = e.layer3(x); f.append(x)
            x = e.layer4(x); f.append(x)
            return f

        def forward(self, A, B):                                        # SegCD.forward (model.py:316-332)
            x1_decode = self.decoder(*self.features(A))
            x2_decode = self.decoder(*self.features(B))
            mask_t1 = self.segmentation_head(x1_decode)
            mask_t2 = self.segmentation_head(x2_decode)
            diffea = self.segmentation_head(torch.abs(x1_decode - x2_decode))
            diffseg = torch.abs(mask_t1 - mask_t2)
            return mask_t1, mask_t2, torch.min(diffea, diffseg)

    return RefSegCD()


def _segcd_fixture(fname, tag, seed, classes, B, H, W, encoder="resnet50", full_for=None):
    from oracle import segcd_ref
    print(tag)
    d = {"seed": seed}
    m = _reference_segcd(classes, encoder)
    names = [k for k in m.state_dict().keys()]
    assert names == [n for n, _, _ in segcd_ref.param_specs(3, classes, encoder)], "state_dict order differs from oracle.segcd_ref.param_specs"
    x1, x2 = rand_pair(seed + 1, B, H, W)
    d["x1"], d["x2"] = t2n(x1), t2n(x2)
    m.load_state_dict(segcd_ref.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
    m.eval()
    with torch.no_grad():
        o = m(x1, x2)
    d["eval/m1"], d["eval/m2"], d["eval/change"] = (t2n(t) for t in o)
    m = _reference_segcd(classes, encoder)
    m.load_state_dict(segcd_ref.synth_state(3, classes, seed, encoder=encoder))
    m.train()
    m1, m2, ch = m(x1, x2)
    rng = np.random.default_rng(seed + 4)
    tgt = torch.from_numpy((rng.random((B, classes, H, W)) < 0.2).astype(np.float32))
    seg = torch.from_numpy((rng.random((B, classes, H, W)) < 0.3).astype(np.float32))
    d["target"], d["seg_target"] = t2n(tgt), t2n(seg)
    # the semi-supervised stage's sum (train_stcd.py:427-445 without the contrastive term): seg loss on mask_t1 + cd loss
    loss = ref_losses.cd_loss(torch.sigmoid(m1), seg) + ref_losses.cd_loss(torch.sigmoid(ch), tgt) + 0.5 * m2.mean()
    loss.backward()
    d["train/m1"], d["train/m2"], d["train/change"], d["loss"] = t2n(m1), t2n(m2), t2n(ch), loss.item()
    d.update(grad_summary(m))
    sd = m.state_dict()
    last = f"encoder.layer4.{segcd_ref.ENCODERS[encoder][1][3] - 1}.bn{3 if segcd_ref.ENCODERS[encoder][0] == 4 else 2}"
    for k in ("encoder.bn1", "encoder.layer2.0.downsample.1" if encoder != "resnet50" else "encoder.layer1.0.downsample.1", last,
              "decoder.blocks.0.conv1.1", "decoder.blocks.4.conv2.1"):
        d[f"rs/{k}.running_mean"], d[f"rs/{k}.running_var"] = t2n(sd[f"{k}.running_mean"]), t2n(sd[f"{k}.running_var"])
        d[f"rs/{k}.num_batches_tracked"] = t2n(sd[f"{k}.num_batches_tracked"])
    save(fname, **d)


def g10_segcd():
    _segcd_fixture("g10_segcd.npz", "G10 SegCD (ResNet-50 UNet)", 1000, 1, 2, 64, 64)


def g11_segcd():
    """A second SegCD fixture: two classes (the 3x3 head with two output channels, the per-channel min), a non-square 96 x 64 map,
    batch 3 (18 samples per BatchNorm at the deepest stage)."""
    _segcd_fixture("g11_segcd_2cls.npz", "G11 SegCD, 2 classes, 3 x 96 x 64", 1100, 2, 3, 96, 64)


def g15_unetseg():
    """UnetSeg (decoders/unet/model.py:109-171), the model train_sup.py:303 trains: the reference's ResNet / UnetDecoder /
    SegmentationHead on one image batch; the step of train_sup.py:131-137 (sigmoid + criterion = cd_loss)."""
    from oracle import segcd_ref
    print("G15 UnetSeg")
    d = {}
    for tag, encoder, seed, classes, B, H, W in (("r50", "resnet50", 1500, 1, 4, 96, 96), ("r34", "resnet34", 1510, 2, 3, 96, 64)):
        x, _ = rand_pair(seed + 1, B, H, W)
        d[f"{tag}/x"], d[f"{tag}/seed"] = t2n(x), seed
        ref = _reference_segcd(classes, encoder)
        unet = lambda m, t_: m.segmentation_head(m.decoder(*m.features(t_)))            # UnetSeg.forward (model.py:165-171)
        ref.load_state_dict(segcd_ref.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
        ref.eval()
        with torch.no_grad():
            d[f"{tag}/eval"] = t2n(unet(ref, x))
        ref = _reference_segcd(classes, encoder)
        ref.load_state_dict(segcd_ref.synth_state(3, classes, seed, encoder=encoder))
        ref.train()
        out = unet(ref, x)
        rng = np.random.default_rng(seed + 4)
        tgt = torch.from_numpy((rng.random((B, classes, H, W)) < 0.3).astype(np.float32))
        loss = ref_losses.cd_loss(torch.sigmoid(out), tgt)
        loss.backward()
        d[f"{tag}/target"], d[f"{tag}/train"], d[f"{tag}/loss"] = t2n(tgt), t2n(out), loss.item()
        for k, v in grad_summary(ref).items():
            d[f"{tag}/{k}"] = v
        sd = ref.state_dict()
        for k in ("encoder.bn1", "decoder.blocks.0.conv1.1", "decoder.blocks.4.conv2.1"):
            d[f"{tag}/rs/{k}.running_mean"], d[f"{tag}/rs/{k}.running_var"] = t2n(sd[f"{k}.running_mean"]), t2n(sd[f"{k}.running_var"])
            d[f"{tag}/rs/{k}.num_batches_tracked"] = t2n(sd[f"{k}.num_batches_tracked"])
    save("g15_unetseg.npz", **d)


def g16_ffctlcd():
    """FFCTLCD (decoders/unet/model.py:335-423): the reference's ResNet / UnetDecoder / SegmentationHead wired as its forward
    states -- decoder(|f1 - f2|), decoder(f1), decoder(f2), in that order; the step of the semi-supervised stage like G10."""
    from oracle import segcd_ref
    print("G16 FFCTLCD")
    d = {}

    def ffc(m, A, B):                                                    # FFCTLCD.forward (model.py:407-423)
        f1, f2 = m.features(A), m.features(B)
        diffea = m.segmentation_head(m.decoder(*[torch.abs(a - b) for a, b in zip(f1, f2)]))
        m1 = m.segmentation_head(m.decoder(*f1))
        m2 = m.segmentation_head(m.decoder(*f2))
        return m1, m2, torch.min(diffea, torch.abs(m1 - m2))

    for tag, encoder, seed, classes, B, H, W in (("r34", "resnet34", 1600, 1, 3, 96, 96), ("r50", "resnet50", 1610, 2, 4, 96, 64)):
        x1, x2 = rand_pair(seed + 1, B, H, W)
        d[f"{tag}/x1"], d[f"{tag}/x2"], d[f"{tag}/seed"] = t2n(x1), t2n(x2), seed
        ref = _reference_segcd(classes, encoder)
        ref.load_state_dict(segcd_ref.synth_state(3, classes, seed, perturb_running=True, encoder=encoder))
        ref.eval()
        with torch.no_grad():
            o = ffc(ref, x1, x2)
        d[f"{tag}/eval/m1"], d[f"{tag}/eval/m2"], d[f"{tag}/eval/change"] = (t2n(t_) for t_ in o)
        ref = _reference_segcd(classes, encoder)
        ref.load_state_dict(segcd_ref.synth_state(3, classes, seed, encoder=encoder))
        ref.train()
        m1, m2, ch = ffc(ref, x1, x2)
        rng = np.random.default_rng(seed + 4)
        tgt = torch.from_numpy((rng.random((B, classes, H, W)) < 0.2).astype(np.float32))
        seg = torch.from_numpy((rng.random((B, classes, H, W)) < 0.3).astype(np.float32))
        loss = ref_losses.cd_loss(torch.sigmoid(m1), seg) + ref_losses.cd_loss(torch.sigmoid(ch), tgt) + 0.5 * m2.mean()
        loss.backward()
        d[f"{tag}/target"], d[f"{tag}/seg_target"] = t2n(tgt), t2n(seg)
        d[f"{tag}/train/m1"], d[f"{tag}/train/m2"], d[f"{tag}/train/change"], d[f"{tag}/loss"] = t2n(m1), t2n(m2), t2n(ch), loss.item()
        for k, v in grad_summary(ref).items():
            if tag == "r34" or k.startswith("gs/"):          # the second case keeps the per-tensor summaries only (fixture size)
                d[f"{tag}/{k}"] = v
        sd = ref.state_dict()
        for k in ("encoder.bn1", "encoder.layer2.0.downsample.1", "decoder.blocks.0.conv1.1", "decoder.blocks.2.conv2.1", "decoder.blocks.4.conv2.1"):
            d[f"{tag}/rs/{k}.running_mean"], d[f"{tag}/rs/{k}.running_var"] = t2n(sd[f"{k}.running_mean"]), t2n(sd[f"{k}.running_var"])
            d[f"{tag}/rs/{k}.num_batches_tracked"] = t2n(sd[f"{k}.num_batches_tracked"])
    save("g16_ffctlcd.npz", **d)


def g18_segcd_wide():
    """A better-conditioned SegCD fixture (round-2 review, weak #3): 4 x 128 x 128, i.e. 4 x 4 x 4 = 64 samples per channel at the deepest
    BatchNorm (G10 has 2 x 2 x 2 = 8), so the fp32 bars of the SegCD family can sit at the north_star's 1e-3 instead of 2e-3 / 5e-2."""
    _segcd_fixture("g18_segcd_128.npz", "G18 SegCD (ResNet-50 UNet), 4 x 128 x 128", 1800, 1, 4, 128, 128)


def g12_segcd_r18():
    """SegCD over the BasicBlock encoders of the registry (encoders/resnet.py:126-144): resnet18 ..."""
    _segcd_fixture("g12_segcd_r18.npz", "G12 SegCD resnet18", 1200, 1, 2, 64, 64, encoder="resnet18")


def g13_segcd_r34():
    """... resnet34 (two classes, 3 x 64 x 96) ..."""
    _segcd_fixture("g13_segcd_r34.npz", "G13 SegCD resnet34, 2 classes, 3 x 64 x 96", 1300, 2, 3, 64, 96, encoder="resnet34")


def g14_segcd_r101():
    """... and the deeper Bottleneck encoder resnet101 ([3, 4, 23, 3]); resnet152 differs from it by depths only."""
    _segcd_fixture("g14_segcd_r101.npz", "G14 SegCD resnet101, 2 x 128 x 128", 1400, 1, 2, 128, 128, encoder="resnet101")


def g17_cf_base():
    """The three classes of /root/reference/models/ChangeFormerBaseNetworks.py that ChangeFormerV6's decoder head is built from
    (ConvLayer :85-96, UpsampleConvLayer :99-106, ResidualBlock :109-120) -- the part of ChangeFormer that imports here (the model
    file itself needs timm): inputs, parameters, outputs, and every gradient for a fixed upstream gradient."""
    from models.ChangeFormerBaseNetworks import ConvLayer, ResidualBlock, UpsampleConvLayer
    torch.manual_seed(1700)
    d = {}
    cases = {"res": (ResidualBlock(16), (2, 16, 12, 10)),
             "up": (UpsampleConvLayer(16, 24, kernel_size=4, stride=2), (2, 16, 6, 5)),
             "conv": (ConvLayer(16, 2, kernel_size=3, stride=1, padding=1), (3, 16, 9, 7))}
    for tag, (mod, shape) in cases.items():
        x = torch.randn(*shape, requires_grad=True)
        y = mod(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        d[f"{tag}/x"], d[f"{tag}/y"], d[f"{tag}/gy"], d[f"{tag}/gx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)
        for k, v in mod.named_parameters():
            d[f"{tag}/p/{k}"], d[f"{tag}/g/{k}"] = t2n(v), t2n(v.grad)
    save("g17_cf_base.npz", **d)


# ------------------------------------------------------------------------------ G21
class _DropRec(nn.Module):
    """nn.Dropout replaced by an explicit element-wise mask (already divided by keep), recorded in the fixture"""
    def __init__(self, store, name, keep, gen):
        super().__init__()
        self.store, self.name, self.keep, self.gen = store, name, keep, gen

    def forward(self, x):
        if not self.training:
            return x
        m = (torch.rand(x.shape, generator=self.gen) < self.keep).float() / self.keep
        self.store[self.name] = m
        return x * m


def g21_cf_decoder():
    """ChangeFormer's decoder from the reference's OWN classes.  /root/reference/models/ChangeFormer.py does not import here
    (timm at :10-11), but `resize` (:238-257), `DWConv` (:512-523), `MLP` (:677-688), `conv_diff` / `make_prediction` (:1138-1157)
    and `DecoderTransformer_v3` (:1475-1631) use none of timm's names: their definitions are compiled from the file by `ast`
    (the G8 recipe: those six nodes, nothing is copied into this repository) in a namespace of torch / nn / F / warnings plus the
    importable models.ChangeFormerBaseNetworks classes.  No timm name is defined anywhere; the encoder classes (which call
    trunc_normal_ at construction) stay out.  Two feature pyramids; eval mode, train mode with Dropout p = 0, and train mode with
    the two nn.Dropout(0.6) sites of every conv_diff swapped for mask recorders: outputs (5 maps), a weighted-sum loss, every
    parameter's gradient, the gradients of the ten input features, BatchNorm running statistics after the training forward."""
    import ast
    import warnings
    from models.ChangeFormerBaseNetworks import ConvLayer, ResidualBlock, UpsampleConvLayer
    print("G21 ChangeFormer decoder (reference classes by ast)")
    path = "/root/reference/models/ChangeFormer.py"
    want = {"resize", "DWConv", "MLP", "conv_diff", "make_prediction", "DecoderTransformer_v3"}
    nodes = [n for n in ast.parse(open(path).read()).body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in nodes} == want
    ns = {"torch": torch, "nn": nn, "F": F, "warnings": warnings, "ConvLayer": ConvLayer, "ResidualBlock": ResidualBlock,
          "UpsampleConvLayer": UpsampleConvLayer}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    Dec, DW = ns["DecoderTransformer_v3"], ns["DWConv"]
    d = {}
    cases = {"a": dict(B=2, chans=[16, 24, 40, 48], emb=32, out=2, hw=(16, 16)),
             "b": dict(B=1, chans=[8, 16, 24, 32], emb=16, out=1, hw=(24, 16))}
    for tag, c in cases.items():
        torch.manual_seed(2100 + ord(tag))
        dec = Dec(input_transform="multiple_select", in_index=[0, 1, 2, 3], align_corners=False, in_channels=c["chans"],
                  embedding_dim=c["emb"], output_nc=c["out"], decoder_softmax=False, feature_strides=[2, 4, 8, 16])
        with torch.no_grad():       # BatchNorm / PReLU away from their trivial initial values
            for k, v in dec.state_dict().items():
                if k.endswith("running_mean"): v.normal_(0, 0.2)
                elif k.endswith("running_var"): v.uniform_(0.5, 1.5)
            for k, v in dec.named_parameters():
                if v.dim() == 1 and "conv2d" not in k and "proj" not in k and v.numel() > 1: v.add_(torch.randn_like(v) * 0.1)
        init = {k: v.clone() for k, v in dec.state_dict().items()}
        for k, v in init.items():
            d[f"{tag}/init/{k}"] = t2n(v)
        h, w = c["hw"]
        feats = [[torch.randn(c["B"], ch, h >> s, w >> s) for s, ch in enumerate(c["chans"])] for _ in range(2)]
        for i in range(2):
            for s in range(4):
                d[f"{tag}/f{i + 1}_{s}"] = t2n(feats[i][s])
        wts = [0.5, -0.25, 0.75, 1.0, 2.0]

        def run(mode):
            dec.load_state_dict(init)
            f = [[x.clone().requires_grad_(True) for x in fs] for fs in feats]
            outs = dec(f[0], f[1])
            for i, o in enumerate(outs):
                d[f"{tag}/{mode}/out{i}"] = t2n(o)
            if mode == "eval":
                return
            loss = sum(wt * (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum() for wt, o in zip(wts, outs)) / 100.0
            dec.zero_grad()
            loss.backward()
            d[f"{tag}/{mode}/loss"] = loss.item()
            for k, v in dec.named_parameters():
                d[f"{tag}/{mode}/g/{k}"] = t2n(v.grad)
            for i in range(2):
                for s in range(4):
                    d[f"{tag}/{mode}/gf{i + 1}_{s}"] = t2n(f[i][s].grad)
            for k, v in dec.state_dict().items():
                if "running" in k or "num_batches" in k:
                    d[f"{tag}/{mode}/bn/{k}"] = t2n(v)

        dec.eval(); run("eval")
        dec.train()
        for m_ in dec.modules():
            if isinstance(m_, nn.Dropout): m_.p = 0.0
        run("train_p0")
        store, gen = {}, torch.Generator().manual_seed(77)
        for s in (1, 2, 3, 4):
            seq = getattr(dec, f"diff_c{s}")
            for idx in (3, 7):
                seq[idx] = _DropRec(store, f"TDec_x2.diff_c{s}.{idx}", 0.4, gen)
        dec.train(); run("train_masks")
        for k, v in store.items():
            d[f"{tag}/train_masks/mask/{k}"] = t2n(v)
        d[f"{tag}/meta"] = np.array([c["B"], c["emb"], c["out"], h, w] + c["chans"], dtype=np.int64)
    # DWConv (the depth-wise step of Mix-FFN): tokens [B, N, C] -> [B, N, C]
    torch.manual_seed(2190)
    dw = DW(24)
    x = torch.randn(2, 6 * 5, 24, requires_grad=True)
    y = dw(x, 6, 5)
    gy = torch.randn_like(y)
    y.backward(gy)
    d["dw/x"], d["dw/y"], d["dw/gy"], d["dw/gx"] = t2n(x), t2n(y), t2n(gy), t2n(x.grad)
    d["dw/w"], d["dw/b"], d["dw/gw"], d["dw/gb"] = t2n(dw.dwconv.weight), t2n(dw.dwconv.bias), t2n(dw.dwconv.weight.grad), t2n(dw.dwconv.bias.grad)
    save("g21_cf_decoder.npz", **d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g2s", "g3", "g4", "g5", "g6", "g7", "g8", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21"]
    fn = {"g1": g1_ops, "g2": g2_fcsiam, "g2s": g2_snunet, "g3": g3_cfg1, "g4": g4_traj, "g5": g5_metric, "g6": g6_odd,
          "g7": g7_train128, "g8": g8_contrastive, "g10": g10_segcd, "g11": g11_segcd, "g12": g12_segcd_r18, "g13": g13_segcd_r34,
          "g14": g14_segcd_r101, "g15": g15_unetseg, "g16": g16_ffctlcd, "g17": g17_cf_base, "g18": g18_segcd_wide, "g19": g19_fcef, "g20": g20_xconc, "g21": g21_cf_decoder}
    for w in which:
        fn[w]()
