#!/usr/bin/env python3
"""bench.py -- image-pairs/sec of full SiamUnet training steps through the HIP engine (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = zero_grad -> forward -> cross-entropy -> backward -> (gradient all-reduce) -> AdamW, on a synthetic
LEVIR-CD-shaped batch already resident in HBM (stcd_amd/synth.py).  Weak scaling: the per-GPU batch is fixed.
Rank 0 prints ONE JSON line.  At N=1 the line also carries
  roofline      the dominant kernel class, timed live with HIP events on the engine's stream
                (stcd_profile_* in include/stcd_hip.h) over instrumented steps of the same workload,
  cpu_baseline  the oracle (oracle/fcsiam_ref.py: a CPU PyTorch restatement pinned to the reference's golden
                vectors) timed on this box's host cores over a bounded sample of the same workload.
"""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
MFMA_F32_PEAK_TF = 157.3
# ALGORITHMIC work of one training step per image pair at 256x256 (measured on the reference's
# modules with forward hooks: 2*MAC of every conv call; (in + out + weights) * 2 B per conv call; train = 3x forward)
ALG_PER_PAIR_256 = {"diff": (25.37e9, 169e6), "sub": (25.37e9, 169e6), "conc": (28.99e9, 182e6), "snunet": (279.6e9, 1118e6),
                    # FC-EF (Unet): SiamUnet_diff's table with ONE encoder stream on 6 input channels (same accounting)
                    "fcef": (18.57e9, 122.9e6),
                    # SiamUnet_cross_conc: SiamUnet_diff + a (pairwise depthwise 3x3, 3x3 C -> C) block on each of the 4 skips
                    "xconc": (29.21e9, 228e6),
                    # SegCD(resnet50): same accounting from the layer table (oracle/segcd_ref.py block_specs / decoder_specs), both dates
                    "segcd": (127.65e9, 912.2e6)}


SEGCD_ENCODERS = {"resnet18": (1, (2, 2, 2, 2)), "resnet34": (1, (3, 4, 6, 3)), "resnet50": (4, (3, 4, 6, 3)),
                  "resnet101": (4, (3, 4, 23, 3)), "resnet152": (4, (3, 8, 36, 3))}       # encoders/resnet.py:126-171


def segcd_alg(encoder, classes=1, H=256, W=256, dates=2, ffc=False):
    """SURVEY 8d's accounting over SegCD's layer table: per conv call 2 MACs and (input + output + weights) x 2 B, both dates,
    the head on 3 maps, x 3 for fwd + dgrad + wgrad.  resnet50: (127.65e9, 912.2e6) per pair at 256 x 256."""
    X, layers = SEGCD_ENCODERS[encoder]
    convs = [(3, 64, 7, H, W, 2)]                       # (cin, cout, k, h_in, w_in, stride)
    h, w, inpl = H // 4, W // 4, 64
    for li, (nb, pl) in enumerate(zip(layers, (64, 128, 256, 512))):
        for b in range(nb):
            s = 2 if (b == 0 and li > 0) else 1
            if X == 4:
                convs += [(inpl, pl, 1, h, w, 1), (pl, pl, 3, h, w, s), (pl, 4 * pl, 1, h // s, w // s, 1)]
            else:
                convs += [(inpl, pl, 3, h, w, s), (pl, pl, 3, h // s, w // s, 1)]
            if b == 0 and (s != 1 or inpl != pl * X):
                convs.append((inpl, pl * X, 1, h, w, s))
            h, w, inpl = h // s, w // s, pl * X
    enc, dec, cin = [512 * X, 256 * X, 128 * X, 64 * X, 64], [256, 128, 64, 32, 16], 512 * X
    nenc = len(convs)
    for i in range(5):
        h, w = 2 * h, 2 * w
        convs += [(cin + (enc[i + 1] if i < 4 else 0), dec[i], 3, h, w, 1), (dec[i], dec[i], 3, h, w, 1)]
        cin = dec[i]
    heads = 3 if dates == 2 else 1                      # SegCD: head on d1, d2, |d1 - d2|; UnetSeg (dates = 1): on d
    passes = [dates] * nenc + [3 if ffc else dates] * (len(convs) - nenc)      # FFCTLCD: the decoder also runs on |f1 - f2|
    fl = sum(2 * (hi // s) * (wi // s) * k * k * ci * co * n for (ci, co, k, hi, wi, s), n in zip(convs, passes))
    by = sum((hi * wi * ci + (hi // s) * (wi // s) * co + k * k * ci * co) * 2 * n for (ci, co, k, hi, wi, s), n in zip(convs, passes))
    fl += 2 * H * W * 9 * 16 * classes * heads
    by += (H * W * 16 + H * W * classes + 9 * 16 * classes) * 2 * heads
    return 3.0 * fl, 3.0 * by


def changeformer_alg(H=512, W=512, out_ch=2, E=(64, 128, 320, 512), depths=(3, 3, 4, 3), srs=(8, 4, 2, 1), D=256, mlp=4, k1=7, k=7):
    """SURVEY 8d's accounting over ChangeFormerV6's layer table (/root/reference/models/ChangeFormer.py:1342-1631): per matrix
    product / convolution call 2 MACs and (input + output + weights) x 2 B, attention 4 N Nkv C flops and (q + k + v + out) x 2 B,
    both dates through the encoder, x 3 for fwd + dgrad + wgrad.  Per image PAIR."""
    fl = by = 0.0

    def op(m, kk, n):                      # [m, kk] x [kk, n]
        nonlocal fl, by
        fl += 2.0 * m * kk * n
        by += (m * kk + m * n + kk * n) * 2.0

    h, w, cin = H, W, 3
    hs = []
    for s in range(4):
        kk, st = (k1, 4) if s == 0 else (k, 2)
        hi, wi = h, w
        h, w = (h + 2 * (kk // 2) - kk) // st + 1, (w + 2 * (kk // 2) - kk) // st + 1
        C, N = E[s], h * w
        Nk = (h // srs[s]) * (w // srs[s]) if srs[s] > 1 else N
        for _date in range(2):
            fl += 2.0 * N * kk * kk * cin * C
            by += (hi * wi * cin + N * C + kk * kk * cin * C) * 2.0
            for _ in range(depths[s]):
                op(N, C, C)                                     # q
                if srs[s] > 1:
                    op(Nk, C * srs[s] ** 2, C)                  # sr
                op(Nk, C, 2 * C)                                # kv
                fl += 4.0 * N * Nk * C
                by += (2 * N * C + 2 * Nk * C) * 2.0            # attention
                op(N, C, C)                                     # proj
                op(N, C, mlp * C)                               # fc1
                fl += 2.0 * 9 * N * mlp * C
                by += (2 * N * mlp * C) * 2.0                   # depthwise 3x3
                op(N, mlp * C, C)                               # fc2
        hs.append((h, w))
        cin = C
    def conv3(px, ci, co):                 # 3x3 stride-1 convolution over px pixels: every tensor counted once
        nonlocal fl, by
        fl += 2.0 * px * 9 * ci * co
        by += (px * ci + px * co + 9 * ci * co) * 2.0

    for s in range(4):
        hh, ww = hs[s]
        op(2 * hh * ww, E[s], D)                                # linear_c on both dates
        conv3(hh * ww, 2 * D, D)                                # conv_diff
        conv3(hh * ww, D, D)
        conv3(hh * ww, D, out_ch)                               # make_prediction
        conv3(hh * ww, out_ch, out_ch)
    h1, w1 = hs[0]
    op(h1 * w1, 4 * D, D)                                       # linear_fuse
    for sc in (1, 2):                                           # convd2x + dense_2x at 2 h1, convd1x + dense_1x at 4 h1
        hi, wi = h1 * sc, w1 * sc
        fl += 2.0 * hi * wi * 16 * D * D
        by += (hi * wi * D + 4 * hi * wi * D + 16 * D * D) * 2.0
        conv3(4 * hi * wi, D, D)
        conv3(4 * hi * wi, D, D)
    conv3(H * W, D, out_ch)                                     # change_probability
    return 3.0 * fl, 3.0 * by


NAMES = {"changeformer": "ChangeFormerV6", "diff": "SiamUnet_diff", "conc": "SiamUnet_conc", "sub": "SiamUnet_sub", "fcef": "Unet (FC-EF)", "xconc": "SiamUnet_cross_conc", "snunet": "SNUNet_ECAM", "segcd": "SegCD-resnet50"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="diff", choices=["diff", "conc", "sub", "fcef", "xconc", "snunet", "segcd", "unetseg", "ffctlcd", "changeformer"],
                    help="unetseg: the single-image UNet of train_sup.py (one 'pair' = one image); ffctlcd: SegCD's feature-level variant")
    ap.add_argument("--encoder", default="resnet50", choices=sorted(SEGCD_ENCODERS) + ["mit_b0"],
                    help="--model segcd: the ResNet encoder; --model changeformer --encoder mit_b0: the MiT-B0 widths BASELINE.json configs[4] names")
    ap.add_argument("--batch", type=int, default=16, help="image pairs per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--label", type=int, default=2)
    ap.add_argument("--pseudo", action="store_true",
                    help="BASELINE.json configs[3]: every step first builds its pairs on the device from resident uint8 tiles "
                         "(stcd_pseudo_pair: blend + normalise + labels), then trains on them")
    ap.add_argument("--graph", action="store_true", help="replay the step as ONE captured hipGraph (stcd_amd.train_loop.GraphedTrainStep): "
                                                          "small batches of the SegCD family are host-bound otherwise")
    ap.add_argument("--bucket-dtype", default="auto", choices=["auto", "fp32", "bf16"],
                    help="gradient buckets of the data-parallel all-reduce (N > 1): auto = bf16 payloads with fp32 accumulation for the families with "
                         "large gradients (SNUNet 48 MB, ChangeFormer 164 MB: SURVEY 8e), fp32 for the rest (FC-Siam 5-6 MB: latency-bound either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=16)     # bounded CPU sample: ~10-30 s of host work in total (16 = the GPU's batch)
    ap.add_argument("--cpu-steps", type=int, default=20)
    return ap.parse_args()


def cpu_baseline_segcd(size, pairs, steps, encoder="resnet50"):
    """oracle/segcd_ref.py timed on the host: fwd + BCE/Dice on sigmoid(change) + bwd + Adam, fp32, all host threads
    (steps capped at 4: one ResNet-50 UNet step on two dates is seconds of CPU work)."""
    from oracle import fcsiam_ref as R
    from oracle import segcd_ref as G
    from stcd_amd import synth
    steps = min(steps, 4)
    a, b, lab = synth.make_batch(pairs, size, size, seed=1337)
    A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab).float().unsqueeze(1)
    st = G.synth_state(3, 1, seed=1, encoder=encoder)
    params = [v.requires_grad_(True) for k, v in st.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999))
    times = []
    i = 0
    while i < steps + 1:
        t0 = time.perf_counter()
        opt.zero_grad()
        _, _, ch = G.forward(st, A, B, training=True)
        loss = R.cd_loss(torch.sigmoid(ch), L)
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
        else:
            steps = _bounded_steps(steps, time.perf_counter() - t0)
        i += 1
    med = float(np.median(times))
    return {"value": round(pairs / med, 3), "unit": "image-pairs/sec", **_host_cores(), "kind": "port",
            "sample": f"oracle/segcd_ref.py SegCD({encoder}) fp32, {pairs} pairs {size}x{size}, median of {steps} full steps after 1 warm-up "
                      f"({med * 1e3:.0f} ms/step)"}


def _bounded_steps(steps, warmup_seconds, budget_seconds=25.0):
    """The CPU baseline is a BOUNDED sample (about 10-30 s of host work): after the warm-up step, as many timed steps as fit the budget
    (SNUNet on 16 cores takes 16 s per 16-pair step: one timed step; SiamUnet_diff 1.6 s: 15)."""
    return max(1, min(steps, int(budget_seconds / max(warmup_seconds, 1e-3))))


def _cgroup_cpus():
    """CPU quota of this container (cgroup v2 cpu.max, v1 cfs quota), None when unlimited: a GPU box hands each job a SHARE of the
    host's cores -- running one thread per host core against a 16-core quota is a slowdown, not a baseline."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(float(q) / float(per) + 0.999))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, (q + per - 1) // per)
    except (OSError, ValueError):
        return None


def _host_cores():
    """all host cores the process may use (affinity mask and cgroup quota), the box's own core count stated
    beside it.  STCD_CPU_THREADS overrides."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpus()
    usable = min(ncpu, quota) if quota else ncpu
    env = os.environ.get("STCD_CPU_THREADS")
    torch.set_num_threads(max(1, int(env)) if env else max(1, usable))
    return {"cores": torch.get_num_threads(), "host_cpu_count": os.cpu_count(), "affinity_cpus": ncpu, "cgroup_cpus": quota}


def cpu_baseline_changeformer(size, pairs, steps):
    """oracle/changeformer_ref.py timed on the host: ChangeFormerV6 fwd + cross-entropy on cp + bwd + AdamW, fp32, all host threads,
    dropout masks drawn once (one 512 x 512 pair is ~2.5 TFLOP of CPU work per step: a one-pair, one-step sample)."""
    from oracle import changeformer_ref as CF
    from stcd_amd import synth
    cores = _host_cores()
    cfg = CF.CFConfig()
    st = CF.synth_state(cfg, seed=1)
    params = [v.requires_grad_(True) for k, v in st.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01)
    times = []
    for i, sz in enumerate([128] + [size] * steps):          # one small warm-up step (thread pools, allocator), then the timed ones
        a, b, lab = synth.make_batch(pairs, sz, sz, seed=1337)
        A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab)
        masks = CF.random_masks(cfg, pairs, sz, sz, seed=2)
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(CF.forward(cfg, st, A, B, True, masks)[-1], L)
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(pairs / med, 4), "unit": "image-pairs/sec", **cores, "kind": "port",
            "sample": f"oracle/changeformer_ref.py ChangeFormerV6(3,2) fp32, {pairs} pair(s) {size}x{size}, {steps} full step(s) after one 128x128 "
                      f"warm-up step ({med * 1e3:.0f} ms/step)"}


def cpu_baseline(arch, label, size, pairs, steps, encoder="resnet50"):
    """The oracle timed on the host: same step (fwd + CE + bwd + AdamW), fp32, all host threads."""
    from oracle import fcsiam_ref as R
    from oracle import snunet_ref as SN
    from stcd_amd import synth

    if arch == "changeformer":
        return cpu_baseline_changeformer(size, min(pairs, 1), min(steps, 1))
    cores = _host_cores()
    a, b, lab = synth.make_batch(pairs, size, size, seed=1337)
    A, B, L = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab)
    if arch == "segcd":
        return cpu_baseline_segcd(size, pairs, steps, encoder)
    st = SN.synth_state(3, label, seed=1) if arch == "snunet" else R.synth_state(arch, 3, label, seed=1)
    params = [v.requires_grad_(True) for k, v in st.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    masks = None if arch == "snunet" else R.synth_masks(arch, pairs, seed=2)
    times = []
    i = 0
    while i < steps + 1:
        t0 = time.perf_counter()
        opt.zero_grad()
        out = SN.forward(st, A, B, training=True) if arch == "snunet" else R.forward(arch, st, A, B, training=True, masks=masks)
        loss = R.cross_entropy(out, L)
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
        else:
            steps = _bounded_steps(steps, time.perf_counter() - t0)
        i += 1
    med = float(np.median(times))
    return {"value": round(pairs / med, 3), "unit": "image-pairs/sec", **cores, "kind": "port",
            "sample": f"oracle/{'snunet_ref.py SNUNet_ECAM' if arch == 'snunet' else 'fcsiam_ref.py SiamUnet_' + arch}(3,{label}) fp32, {pairs} pairs {size}x{size}, "
                      f"median of {steps} full steps after 1 warm-up ({med * 1e3:.0f} ms/step)"}


def _pmc_files(model):
    """Committed PMC summaries of THIS model's bench command, newest round first (diff's files carry no model tag)."""
    import glob
    files = glob.glob(os.path.join(REPO, "profiles", f"*_{model}_pmc_traffic.txt"))
    if model == "diff":
        files += glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_traffic.txt"))
    return sorted(files, reverse=True)


def pmc_step_traffic(model):
    """Whole-step HBM-side bytes (sum over every kernel of launches x bytes per launch / profiled steps) from the committed
    PMC summary of the same command for this model (tools/collect_profiles.py writes the '# step_total_B' line)."""
    for path in _pmc_files(model):
        for line in open(path):
            if line.startswith("# step_total_B"):
                return int(float(line.split()[-1])), os.path.relpath(path, REPO)
    return None, None


def pmc_traffic(kernel, model):
    """HBM-side bytes per launch of `kernel` from the committed PMC summary of this same command: the counters need
    their own rocprofv3 --pmc passes (tools/profile_round.sh; FETCH_SIZE doubled per the gfx950 correction), so they
    cannot be read inside the timed run.  None when no summary of this model holds the kernel."""
    for path in _pmc_files(model):
        for line in open(path):
            if line.startswith(kernel + " ") or (len(kernel) > 48 and line.startswith(kernel[:48] + " ")):
                try:
                    return int(float(line.split()[-1])), os.path.relpath(path, REPO)
                except ValueError:
                    pass
    # the engine's timers name a kernel CLASS ("k_bn_reduce<T, 1>"); rocprofv3 lists every template instantiation of it under
    # its own (differently spelled) name: launch-weighted mean over the rows of the same base name
    base = kernel.split("<")[0]
    for path in _pmc_files(model):
        tot = launches = 0
        for line in open(path):
            if line.startswith(base + "<") or line.startswith(base + " "):
                f = line.split()
                try:
                    n, b = int(f[-4]), float(f[-1])
                except (ValueError, IndexError):
                    continue
                tot += n * b
                launches += n
        if launches:
            return int(tot / launches), os.path.relpath(path, REPO)
    return None, None


def main():
    args = parse()
    from stcd_amd import synth
    from stcd_amd.ddp import FlatGradReducer, broadcast_parameters, init_distributed
    from stcd_amd.losses import bce_dice_with_logits, cross_entropy
    from stcd_amd.modules import SiamUnet_conc, SiamUnet_cross_conc, SiamUnet_diff, SiamUnet_sub, SNUNet_ECAM, Unet
    from stcd_amd.optim import FlatAdamW

    rank, local_rank, world = init_distributed()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU fallback"
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(1337)
    if args.model == "changeformer":   # BASELINE.json configs[4]: ChangeFormerV6, 512 x 512, batch 32 over 8 GPUs = 4 pairs per GPU
        from stcd_amd.changeformer import ChangeFormerV6
        args.label = 2
        if args.size == 256 and "--size" not in sys.argv:
            args.size = 512
        if args.batch == 16 and "--batch" not in sys.argv:
            args.batch = 4
        if args.encoder == "mit_b0":
            from stcd_amd.changeformer import MIT_B0
            model = ChangeFormerV6(3, 2, dtype=args.dtype, config=dict(MIT_B0)).to(dev).train()
            NAMES["changeformer"] = "ChangeFormer(MiT-B0 encoder)"
            args.no_cpu_baseline = True
        else:
            model = ChangeFormerV6(3, 2, dtype=args.dtype).to(dev).train()
    elif args.model == "segcd":       # train_pse_cd.py:426-431: SegCD(resnet50), 1 class, Adam(lr 1e-3), BCE+Dice on sigmoid(change)
        from stcd_amd.segcd import SegCD
        args.label = 1
        model = SegCD(encoder_name=args.encoder, encoder_weights=None, in_channels=3, classes=1, dtype=args.dtype).to(dev).train()
        NAMES["segcd"] = "SegCD-" + args.encoder
    elif args.model == "ffctlcd":   # the commented alternative of train_pse_cd.py:419: same step as SegCD
        from stcd_amd.segcd import FFCTLCD
        args.label = 1
        model = FFCTLCD(encoder_name=args.encoder, encoder_weights=None, in_channels=3, classes=1, dtype=args.dtype).to(dev).train()
        NAMES["ffctlcd"] = "FFCTLCD-" + args.encoder
        args.no_cpu_baseline = True
    elif args.model == "unetseg":   # train_sup.py:303-309: UnetSeg(resnet50), sigmoid + criterion on the mask; a unit is ONE image
        from stcd_amd.segcd import UnetSeg
        args.label = 1
        model = UnetSeg(encoder_name=args.encoder, encoder_weights=None, in_channels=3, classes=1, dtype=args.dtype).to(dev).train()
        NAMES["unetseg"] = "UnetSeg-" + args.encoder
        args.no_cpu_baseline = True
    else:
        cls = {"diff": SiamUnet_diff, "conc": SiamUnet_conc, "sub": SiamUnet_sub, "fcef": Unet, "xconc": SiamUnet_cross_conc, "snunet": SNUNet_ECAM}[args.model]
        model = cls(3, args.label, dtype=args.dtype).to(dev).train()
    broadcast_parameters(model)
    # torch.optim.AdamW semantics, one launch (weight_decay 0 == the Adam the SegCD script uses)
    opt = FlatAdamW(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0 if args.model in ("segcd", "unetseg", "ffctlcd") else 0.01)
    bucket_dtype = args.bucket_dtype if args.bucket_dtype != "auto" else ("bf16" if args.model in ("snunet", "changeformer") else "fp32")
    reducer = FlatGradReducer(model, bucket_dtype=bucket_dtype)  # noqa: F841  (installs the gradient hook when world > 1)

    a, b, lab = synth.make_batch(args.batch, args.size, args.size, seed=1337 + rank)
    A, B, L = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(lab).to(dev)
    Lf = L.float().unsqueeze(1)
    if args.pseudo:      # train_pse_cd.py's data path on the device: uint8 HWC tiles -> normalised pair + change label, per step
        from stcd_amd.pseudo import pseudo_change_pairs
        ta, td, tl = synth.make_pairs_u8(args.batch, args.size, args.size, seed=1337 + rank)
        TA, TD, TM = torch.from_numpy(ta).to(dev), torch.from_numpy(td).to(dev), torch.from_numpy((tl * 255).astype("uint8")).to(dev)
        CH = torch.ones(args.batch, dtype=torch.uint8, device=dev)
        CH[1::4] = 0                                                   # a quarter of the tiles take the no-change branch
        step_no = [0]

    def step():
        opt.zero_grad(set_to_none=True)
        if args.pseudo:
            step_no[0] += 1
            x1, x2, lab_, _, _ = pseudo_change_pairs(TA, TD, TM, CH, seed=step_no[0])
            out = model(x1, x2)
            out = out[-1] if isinstance(out, (list, tuple)) else out
            loss = cross_entropy(out, lab_) if args.label == 2 else bce_dice_with_logits(out, lab_.float().unsqueeze(1))
            loss.backward()
            opt.step()
            return loss
        out = model(A) if args.model == "unetseg" else model(A, B)
        out = out[-1] if isinstance(out, (list, tuple)) else out      # SiamUnet_sub: [logits]; SegCD: (mask_t1, mask_t2, change)
        loss = cross_entropy(out, L) if args.label == 2 else bce_dice_with_logits(out, Lf)
        loss.backward()
        opt.step()
        return loss

    if args.graph:
        from stcd_amd.train_loop import GraphedTrainStep
        assert world == 1 and not args.pseudo, "--graph: single GPU, resident inputs"
        lossf = (lambda out, y: cross_entropy(out[-1] if isinstance(out, (list, tuple)) else out, y)) if args.label == 2 else \
                (lambda out, y: bce_dice_with_logits(out[-1] if isinstance(out, (list, tuple)) else out, y))
        gstep = GraphedTrainStep(model, opt, lossf, (A,) if args.model == "unetseg" else (A, B), L if args.label == 2 else Lf)
        eager_step = step

        def step():                                       # noqa: F811  (the timed loop calls whatever `step` is)
            return gstep(*(((A,) if args.model == "unetseg" else (A, B)) + ((L if args.label == 2 else Lf),)))
        args.no_roofline = True                            # the per-kernel timers bracket launches: not meaningful inside a replay

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    # the step loop runs under stcd_amd.train_loop.quiet_gc() like the product's training loops do (a full cyclic collection over
    # the module tree in the middle of a launch sequence idles the GPU); STCD_BENCH_GC=default measures without it
    gc_guard = contextlib.ExitStack()
    if os.environ.get("STCD_BENCH_GC", "quiet") != "default":
        from stcd_amd.train_loop import quiet_gc
        gc_guard.enter_context(quiet_gc())
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue_s = time.perf_counter() - t0          # host time to enqueue the timed steps (the GPU may still be running them)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0        # this rank's own K steps (before it waits for the others at the barrier)
    barrier()
    elapsed = time.perf_counter() - t0
    gc_guard.close()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    last_loss = loss.item()
    # the host's OWN cost of a step: enqueue time into an idle queue (a synchronize in front of every step).  `host_enqueue_ms_per_step`
    # above is taken while the GPU is busy: once the hardware queue is full the enqueue blocks, so on a GPU-bound step it approaches
    # the step time whatever the host costs (ChangeFormer: 19.3 ms of 25.1 with 3.9 ms of real host work per step)
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        idle.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    host_idle_ms = sorted(idle)[len(idle) // 2] * 1e3
    my_rate = args.batch * args.steps / own_elapsed
    rank_rates = [my_rate]
    if world > 1:      # self-check of the first multi-GPU run: every rank's own pairs/s and the collective backend in use
        rr = torch.tensor([0.0] * world, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        rr[rank] = my_rate
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rank_rates = [round(v, 2) for v in rr.tolist()]

    result = {
        "metric": "image-pairs/sec (256x256 bf16 SiamUnet_diff train)" if (args.model, args.size, args.dtype) == ("diff", 256, "bf16")
        else f"{'images' if args.model == 'unetseg' else 'image-pairs'}/sec ({args.size}x{args.size} {args.dtype} {NAMES[args.model]} train)",
        "value": round(world * args.batch * args.steps / elapsed, 2),
        "unit": "images/sec" if args.model == "unetseg" else "image-pairs/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),     # << ms_per_step: the GPU is the bound, not the host
        "host_ms_per_step_idle_queue": round(host_idle_ms, 4),                  # the host's own work per step (median of 5 steps, GPU idle)
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{NAMES[args.model]}(3,{args.label}) {args.size}x{args.size} full training step "
                               f"(fwd + {'cross_entropy' if args.label == 2 else 'sigmoid+cd_loss'}{' on cp' if args.model == 'changeformer' else ''} + bwd + {'Adam' if args.model in ('segcd', 'unetseg', 'ffctlcd') else 'AdamW'}), "
                               f"{args.batch} pairs/GPU, " + ("pairs built every step by the on-device pseudo-change generator from uint8 tiles resident in HBM"
                                                             if args.pseudo else "synthetic LEVIR-CD-shaped pairs resident in HBM"),
                   "global_batch": world * args.batch, "image": args.size, "parallelism": f"dp{world}",
                   "last_loss": round(last_loss, 5)},
        "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else 0,
        "collective_backend": dist.get_backend() if world > 1 else None,
        "bucket_dtype": bucket_dtype if world > 1 else None,
        "pairs_per_sec_per_rank": [round(v, 2) for v in rank_rates],
        "graph_replay": bool(args.graph),
    }
    # whole-step roofline: SURVEY 8d's algorithmic figures per pair x pairs per step / measured step time (all ranks)
    seg_family = args.model in ("segcd", "unetseg", "ffctlcd")
    fl_pp, by_pp = (segcd_alg(args.encoder, dates=1 if args.model == "unetseg" else 2, ffc=args.model == "ffctlcd") if seg_family
                    else (changeformer_alg(args.size, args.size, E=(32, 64, 160, 256), depths=(2, 2, 2, 2), k=3) if args.encoder == "mit_b0" else changeformer_alg(args.size, args.size)) if args.model == "changeformer" else ALG_PER_PAIR_256[args.model])
    pmc_key = args.model if (not seg_family or args.encoder == "resnet50") else args.model + "_" + args.encoder
    sc = 1.0 if args.model == "changeformer" else (args.size / 256.0) ** 2
    step_s = elapsed / args.steps
    peak_tf_ = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
    by_scale = 1.0 if args.dtype == "bf16" else 2.0
    step_roof = {"alg_flops": fl_pp * sc * args.batch, "alg_bytes": by_pp * sc * by_scale * args.batch,
                 "hbm_frac": round(by_pp * sc * by_scale * args.batch / step_s / 1e9 / HBM_PEAK_GBS, 4),
                 "mfma_frac": round(fl_pp * sc * args.batch / step_s / 1e12 / peak_tf_, 4),
                 "note": "per GPU: algorithmic bytes / flops of one step (SURVEY 8d per-pair figures x pairs per GPU) over the measured step time"}

    if rank == 0 and world == 1 and not args.no_roofline:
        eng = model._engine
        nprof = 5
        # every instrumented step is read on its own and each kernel / class takes its MEDIAN over the steps, scaled back to nprof
        # steps: the event pair of a launch also sees a host hiccup between its two records (round 4's first run: one 4-ms stall in one
        # of three steps read as k_conv_res<2,32> = 1.59 ms per step, against 0.26 in the timed loop's rocprof trace)
        per_step_k, per_step_c = [], []
        for _ in range(nprof):
            eng.profile_enable(True)
            step()
            torch.cuda.synchronize()
            per_step_c.append(eng.profile_read())
            per_step_k.append(eng.profile_kernels())
        eng.profile_enable(False)

        def _median_steps(runs):
            out = {}
            for k_ in runs[0]:
                vals = sorted(r_[k_]["ms"] for r_ in runs if k_ in r_)
                v0 = runs[0][k_]
                out[k_] = {"ms": vals[len(vals) // 2] * nprof, "launches": v0["launches"] * nprof,
                           "flops": v0["flops"] * nprof, "bytes": v0["bytes"] * nprof}
            return out
        prof = _median_steps(per_step_c)
        kern = _median_steps(per_step_k)
        # the dominant KERNEL: template instantiations of one kernel (k_conv_res<2, 32, false>, <1, 64, false> ...) are ONE kernel
        # for this purpose -- merged by base name, so a kernel split over many instantiations cannot hide behind a smaller one
        merged = {}
        for k_, v_ in kern.items():
            mname = k_.split("<")[0]
            mm = merged.setdefault(mname, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "parts": []})
            for f_ in ("ms", "launches", "flops", "bytes"):
                mm[f_] += v_[f_]
            mm["parts"].append(k_)
        dom = max(merged, key=lambda k: merged[k]["ms"])
        p = merged[dom]
        # Per-launch time = the RAW HIP-event reading (an event pair around every launch, on the launch's own stream).  Round 3
        # subtracted half an empty event pair; the committed rocprofv3 --kernel-trace average of the same kernel equals the raw
        # figure (profiles/r03_kernel_stats.csv: 18.41 us vs 18.40 us raw), so nothing is subtracted any more.  The empty-pair
        # reading is still printed (event_pair_overhead_us) as a diagnostic only.
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a_, b_ in pairs:
            a_.record(); b_.record()
        torch.cuda.synchronize()
        ev_us = float(np.median([a_.elapsed_time(b_) for a_, b_ in pairs])) * 1e3
        raw_us = p["ms"] * 1e3 / max(p["launches"], 1)
        secs = raw_us * 1e-6 * max(p["launches"], 1)
        hbm_frac = p["bytes"] / secs / 1e9 / HBM_PEAK_GBS if secs > 0 else 0.0
        peak_tf = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
        mfma_frac = p["flops"] / secs / 1e12 / peak_tf if secs > 0 else 0.0
        bound = "mfma" if mfma_frac > hbm_frac else "hbm"
        if bound == "hbm":
            ach, peak, unit = p["bytes"] / secs / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = p["flops"] / secs / 1e12, peak_tf, "TFLOP/s"
        tot_ms = sum(v["ms"] for v in prof.values())
        traffic, traffic_src = pmc_traffic("stcd::" + dom, pmc_key)
        result["roofline"] = {
            "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "stcd::" + dom, "instantiations": sorted(p["parts"]), "avg_launch_us": round(secs * 1e6 / max(p["launches"], 1), 3),
            "avg_launch_us_raw": round(raw_us, 3), "event_pair_overhead_us": round(ev_us, 3),
            "launches_per_step": p["launches"] // nprof,
            "alg_bytes_per_launch": round(p["bytes"] / max(p["launches"], 1)),
            "alg_flops_per_launch": round(p["flops"] / max(p["launches"], 1)),
            "other_bound_frac": round(min(hbm_frac, mfma_frac), 4),
            "class_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in prof.items()},
            "kernel_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("STCD_BENCH_TOP_KERNELS", "8"))]},
            "instrumented_ms_per_step": round(tot_ms / nprof, 4),     # (instrumented steps run on ONE stream: no side-stream overlap)
            "launches_per_step_all_kernels": sum(v["launches"] for v in kern.values()) // nprof,
            "step": step_roof,
        }
        ts, ts_src = pmc_step_traffic(pmc_key)
        result["roofline"]["traffic_step"] = ts
        result["roofline"]["traffic_step_source"] = ts_src
        if ts:
            result["roofline"]["traffic_step_over_algorithmic"] = round(ts / step_roof["alg_bytes"], 3)
    elif rank == 0:
        result["roofline"] = {"step": step_roof}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.model, args.label, args.size, args.cpu_pairs, args.cpu_steps, args.encoder)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
