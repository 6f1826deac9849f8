"""The bf16 engine against a bf16-EMULATING oracle (oracle/fcsiam_bf16.py: the pinned fp32 oracle with bf16 rounding at exactly the
tensors the engine stores, forward and backward).  Round-2 review, weak #1: whole-network bf16 gradients sit at cosine 0.76-0.9
against the fp32 reference on random-init fixtures -- rounding noise of ~20 stored layers through ReLU / max-pool / |a - b| gates --
so a bound against the fp32 reference alone could hide a wrong term worth 20 % of a gradient.  Against the emulation the same
rounding happens on both sides: what is left is accumulation order plus the rare value within an fp32 ulp of a bf16 rounding
boundary, and every tensor must agree closely.  Second fixture: a partly-trained state on LEVIR-shaped inputs (the fp32 engine's own
100 AdamW steps from the synthetic initialisation), where the fp32-reference bound itself is meaningful."""
import numpy as np
import pytest
import torch

from oracle import fcsiam_bf16 as E
from oracle import fcsiam_ref as R
from stcd_amd import synth
from stcd_amd.modules import SiamUnet_conc, SiamUnet_diff, SiamUnet_sub
from stcd_amd.optim import FlatAdamW
from tests import _util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CLS = {"diff": SiamUnet_diff, "conc": SiamUnet_conc, "sub": SiamUnet_sub}


def _first(out):
    return out[0] if isinstance(out, (list, tuple)) else out


def _oracle_grads(arch, st, x1, x2, tgt, masks, emulate):
    ref = {k: v.clone() for k, v in st.items()}
    for k, v in ref.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    logits = E.forward(arch, ref, x1, x2, masks) if emulate else R.forward(arch, ref, x1, x2, training=True, masks=masks)
    loss = R.cross_entropy(logits, tgt)
    loss.backward()
    return loss.item(), logits.detach(), {k: v.grad for k, v in ref.items() if v.requires_grad}


def _engine_grads(arch, st, x1, x2, tgt, masks, dtype):
    m = CLS[arch](3, 2, dtype=dtype)
    m.load_state_dict(st)
    m.to(DEV).train()
    m.set_dropout_masks(masks)
    logits = _first(m(x1.to(DEV), x2.to(DEV)))
    loss = torch.nn.functional.cross_entropy(logits, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), logits.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}


def _cosines(got, ref):
    out = []
    for k, g in ref.items():
        if _util.zero_grad_by_construction(k) or float(g.abs().max()) < 1e-9:
            continue
        rel, cos = _util.rel_l2_cos(got[k].numpy(), g.numpy())
        out.append((cos, rel, k))
    return sorted(out)


@pytest.mark.parametrize("arch", ["diff", "conc", "sub"])
def test_bf16_engine_matches_the_bf16_emulating_oracle_at_random_init(arch):
    seed = 700
    rng = np.random.default_rng(seed + 1)
    a = rng.standard_normal((2, 3, 128, 128)).astype(np.float32)
    b = (a + 0.5 * rng.standard_normal((2, 3, 128, 128))).astype(np.float32)
    x1, x2 = torch.from_numpy(a), torch.from_numpy(b)
    tgt = torch.from_numpy((np.random.default_rng(seed + 4).random((2, 128, 128)) < 0.2).astype(np.int64))
    st = R.synth_state(arch, 3, 2, seed)
    masks = R.synth_masks(arch, 2, seed + 3)
    le, oe, ge = _engine_grads(arch, st, x1, x2, tgt, masks, "bf16")
    lm, om, gm = _oracle_grads(arch, st, x1, x2, tgt, masks, emulate=True)
    lf, of_, gf = _oracle_grads(arch, st, x1, x2, tgt, masks, emulate=False)
    vs_emul, vs_fp32, emul_vs_fp32 = _cosines(ge, gm), _cosines(ge, gf), _cosines(gm, gf)
    print(f"{arch} random init: engine vs emulation worst {vs_emul[0][0]:.4f} ({vs_emul[0][2]}) median {vs_emul[len(vs_emul) // 2][0]:.4f} | "
          f"engine vs fp32 oracle worst {vs_fp32[0][0]:.4f} median {vs_fp32[len(vs_fp32) // 2][0]:.4f} | "
          f"emulation vs fp32 oracle worst {emul_vs_fp32[0][0]:.4f} median {emul_vs_fp32[len(emul_vs_fp32) // 2][0]:.4f}")
    # logits / loss: the engine and the emulation round the same tensors
    assert float((oe - om).abs().max()) <= 2e-2 * float(om.abs().max()) and abs(le - lm) < 2e-3
    # Measured (MI355X): engine vs emulation worst 0.92-0.95 / median 0.98, while BOTH sit at worst 0.81-0.83 / median 0.94 against the
    # fp32 oracle: on this fixture (random-init weights, white-noise inputs) the network is chaotic enough that two bf16 evaluations
    # which differ only in accumulation order decorrelate a little -- but the engine is 3x closer to the emulation than either is to
    # fp32, and no further from fp32 than the emulation is (medians within 0.005; the single worst tensor, an extreme statistic of 48,
    # within 0.07).  The partly-trained fixture below carries the tight bounds.
    assert vs_emul[0][0] >= 0.90, vs_emul[:4]
    assert vs_emul[len(vs_emul) // 2][0] >= 0.97
    assert vs_fp32[0][0] >= emul_vs_fp32[0][0] - 0.10 and vs_fp32[len(vs_fp32) // 2][0] >= emul_vs_fp32[len(emul_vs_fp32) // 2][0] - 0.02, \
        "the engine's bf16 gradients are further from the fp32 reference than bf16 storage explains"


@pytest.mark.parametrize("arch", ["diff", "conc"])
def test_bf16_gradients_on_a_partly_trained_state(arch):
    """100 AdamW steps of the fp32 engine on LEVIR-shaped synthetic pairs (stcd_amd.synth.make_batch: smooth imagery, ~5 % change),
    then one step on a held-out batch: bf16 engine vs the emulation (tight) and vs the fp32 oracle (the meaningful bf16 bound)."""
    B, S = 8, 128
    a, b, lab = synth.make_batch(B, S, S, seed=77)
    A, Bt, L = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), torch.from_numpy(lab).to(DEV)
    m = CLS[arch](3, 2, dtype="fp32")
    m.load_state_dict(R.synth_state(arch, 3, 2, 31))
    m.to(DEV).train()
    opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    for _ in range(100):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(_first(m(A, Bt)), L).backward()
        opt.step()
    torch.cuda.synchronize()
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    a2, b2, lab2 = synth.make_batch(2, S, S, seed=78)
    x1, x2, tgt = torch.from_numpy(a2), torch.from_numpy(b2), torch.from_numpy(lab2)
    masks = R.synth_masks(arch, 2, 5)
    le, oe, ge = _engine_grads(arch, st, x1, x2, tgt, masks, "bf16")
    lm, om, gm = _oracle_grads(arch, st, x1, x2, tgt, masks, emulate=True)
    lf, of_, gf = _oracle_grads(arch, st, x1, x2, tgt, masks, emulate=False)
    vs_emul, vs_fp32 = _cosines(ge, gm), _cosines(ge, gf)
    print(f"{arch} trained state: loss engine {le:.4f} emulation {lm:.4f} fp32 {lf:.4f} | engine vs emulation worst {vs_emul[0][0]:.4f} "
          f"({vs_emul[0][2]}) median {vs_emul[len(vs_emul) // 2][0]:.4f} | engine vs fp32 oracle worst {vs_fp32[0][0]:.4f} ({vs_fp32[0][2]}) "
          f"median {vs_fp32[len(vs_fp32) // 2][0]:.4f}")
    # measured: vs emulation worst 0.9985 / 0.9992 (median 0.9998), vs the fp32 oracle worst 0.991 / 0.994 (median 0.999)
    assert vs_emul[0][0] >= 0.995, vs_emul[:4]
    assert vs_fp32[0][0] >= 0.98, vs_fp32[:4]
    assert abs(le - lf) < 5e-3


def test_snunet_bf16_gradients_against_its_bf16_emulating_oracle():
    """BASELINE.json configs[2] is SNUNet in bf16 (round-3 review, weak #2: its bf16 parity rested on fitted bounds -- rel-l2 0.32 /
    cosine 0.95 against G7).  oracle/snunet_bf16.py rounds exactly what the engine stores (Y1 -- also the identity branch --, A1, Y2,
    the block output, the transposed convs' outputs, Z; every stored gradient incl. the per-consumer contributions and their
    once-more-rounded sum); from a partly-trained state (60 AdamW steps of the fp32 engine on LEVIR-shaped pairs) the bf16 engine
    must agree with it tensor by tensor, and with the fp32 oracle at the bf16 bound."""
    from oracle import snunet_bf16 as ES
    from oracle import snunet_ref as S
    from stcd_amd.modules import SNUNet_ECAM
    B, Sz = 4, 64
    a, b, lab = synth.make_batch(B, Sz, Sz, seed=91)
    A, Bt, L = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), torch.from_numpy(lab).to(DEV)
    m = SNUNet_ECAM(3, 2, dtype="fp32")
    m.load_state_dict(S.synth_state(3, 2, 13))
    m.to(DEV).train()
    opt = FlatAdamW(m, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    for _ in range(60):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(_first(m(A, Bt)), L).backward()
        opt.step()
    torch.cuda.synchronize()
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    a2, b2, lab2 = synth.make_batch(2, Sz, Sz, seed=92)
    x1, x2, tgt = torch.from_numpy(a2), torch.from_numpy(b2), torch.from_numpy(lab2)

    def oracle(emulate):
        ref = {k: v.clone() for k, v in st.items()}
        for k, v in ref.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        logits = ES.forward(ref, x1, x2) if emulate else S.forward(ref, x1, x2, training=True)
        loss = R.cross_entropy(logits, tgt)
        loss.backward()
        return loss.item(), logits.detach(), {k: v.grad for k, v in ref.items() if v.requires_grad}

    e = SNUNet_ECAM(3, 2, dtype="bf16")
    e.load_state_dict(st)
    e.to(DEV).train()
    logits = _first(e(x1.to(DEV), x2.to(DEV)))
    loss = torch.nn.functional.cross_entropy(logits, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    le, oe, ge = loss.item(), logits.detach().cpu(), {k: p.grad.detach().cpu() for k, p in e.named_parameters()}
    lm, om, gm = oracle(True)
    lf, of_, gf = oracle(False)

    def cosines(got, ref):
        out = []
        for k, g in ref.items():
            # conv biases in front of a training-mode BatchNorm have mathematically zero gradients (conv2's; conv1's reaches the
            # loss through the identity branch only and is real)
            if (".conv2.bias" in k) or float(g.abs().max()) < 1e-9:
                continue
            rel, cos = _util.rel_l2_cos(got[k].numpy(), g.numpy())
            out.append((cos, rel, k))
        return sorted(out)
    vs_emul, vs_fp32, emul_vs_fp32 = cosines(ge, gm), cosines(ge, gf), cosines(gm, gf)
    print(f"snunet trained state: loss engine {le:.4f} emulation {lm:.4f} fp32 {lf:.4f} | engine vs emulation worst {vs_emul[0][0]:.4f} ({vs_emul[0][2]}) "
          f"median {vs_emul[len(vs_emul) // 2][0]:.4f} | engine vs fp32 worst {vs_fp32[0][0]:.4f} ({vs_fp32[0][2]}) median {vs_fp32[len(vs_fp32) // 2][0]:.4f} | "
          f"emulation vs fp32 worst {emul_vs_fp32[0][0]:.4f} median {emul_vs_fp32[len(emul_vs_fp32) // 2][0]:.4f}")
    assert float((oe - om).abs().max()) <= 2e-2 * float(om.abs().max()) and abs(le - lm) < 2e-3
    # Measured (MI355X): median 0.9998; 140 of the 146 tensors >= 0.995.  The ones below sit in the deepest nested-decoder blocks
    # (conv0_3 / conv1_2: bn2.weight 0.989, bn1.weight 0.991, conv2.weight 0.991, bn1.weight 0.994): their output gradient is the sum
    # of up to five separately stored (bf16-rounded) consumer contributions, itself rounded, after the longest chain of such sums in
    # the network, and the BatchNorm scale gradients d(gamma) = sum(dz * xhat) are sums of nearly cancelling terms on top -- two bf16
    # evaluations that differ only in accumulation order agree less there.  The emulation itself sits at 0.984 against the fp32
    # oracle on the same tensors and the engine is no further from fp32 than the emulation is (asserted below): rounding-order
    # noise, not a missing term (a wrong term worth 10 % of a gradient is a cosine of 0.995 on EVERY tensor it reaches).
    convs = [t for t in vs_emul if ".bn" not in t[2]]
    bns = [t for t in vs_emul if ".bn" in t[2]]
    print(f"  conv / attention weights worst {convs[0][0]:.4f} ({convs[0][2]}); BatchNorm parameters worst {bns[0][0]:.4f} ({bns[0][2]})")
    assert convs[0][0] >= 0.985 and sum(1 for t in convs if t[0] < 0.995) <= 3, convs[:6]
    assert bns[0][0] >= 0.985 and sum(1 for t in bns if t[0] < 0.995) <= 6, bns[:8]
    assert vs_fp32[0][0] >= emul_vs_fp32[0][0] - 0.02, (vs_fp32[:4], emul_vs_fp32[:4])
