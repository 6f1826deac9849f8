"""stcd_adam_step / stcd_amd.optim.FlatAdam[W] against torch.optim.Adam / AdamW (the optimizers the reference builds:
/root/reference/models/trainer.py:46-50, /root/reference/train_pse_cd.py:431)."""
import ctypes as C

import numpy as np
import pytest
import torch

from stcd_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("decoupled,wd", [(0, 0.0), (0, 5e-4), (1, 0.01)])
@pytest.mark.parametrize("n", [1, 7, 4096, 1350146])
def test_adam_step_matches_torch(decoupled, wd, n):
    rng = np.random.default_rng(n + decoupled)
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
    pt = p0.clone().requires_grad_(True)
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    ref = cls([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    l = _lib.lib()
    for step in range(1, 6):
        g = torch.from_numpy((rng.standard_normal(n) * 10.0 ** rng.uniform(-4, 0)).astype(np.float32)).to(DEV)
        pt.grad = g.clone()
        ref.step()
        _lib.check(l.stcd_adam_step(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                    C.c_void_p(v.data_ptr()), n, step, 1e-3, 0.9, 0.999, 1e-8, wd, decoupled,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    st = ref.state[pt]
    np.testing.assert_allclose(p.cpu().numpy(), pt.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].cpu().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)


def test_adam_step_rejects_bad_arguments():
    l = _lib.lib()
    t = torch.zeros(8, device=DEV)
    a = C.c_void_p(t.data_ptr())
    assert l.stcd_adam_step(a, a, a, None, 8, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) != 0
    assert l.stcd_adam_step(a, a, a, a, 8, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) != 0
    assert b"step" in l.stcd_last_error()


@pytest.mark.parametrize("kind", ["adam", "adamw"])
def test_flat_optimizer_trains_like_torch(kind):
    """A real engine module trained by FlatAdam[W]; a shadow copy of its parameters is stepped by torch.optim with the SAME
    gradients (whole-network gradients are discontinuous, so two separately trained twins drift chaotically: the
    optimizer is what is compared here).  LR schedulers drive param_groups[0]["lr"]; state_dict round-trips."""
    from stcd_amd import synth
    from stcd_amd.losses import cross_entropy
    from stcd_amd.modules import SiamUnet_diff
    from stcd_amd.optim import FlatAdam, FlatAdamW

    a, b, lab = synth.make_batch(4, 64, 64, seed=5)
    A, B, L = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), torch.from_numpy(lab).to(DEV)
    torch.manual_seed(3)
    m = SiamUnet_diff(3, 2, dtype="fp32").to(DEV).train()
    shadow = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    if kind == "adam":
        o1, o2 = torch.optim.Adam(shadow, lr=1e-3), FlatAdam(m, lr=1e-3)
    else:
        o1, o2 = torch.optim.AdamW(shadow, lr=1e-3, weight_decay=0.01), FlatAdamW(m, lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.StepLR(o2, step_size=2, gamma=0.5)
    sched1 = torch.optim.lr_scheduler.StepLR(o1, step_size=2, gamma=0.5)
    for _ in range(4):
        o2.zero_grad()
        cross_entropy(m(A, B), L).backward()
        for sp, p in zip(shadow, m.parameters()):
            sp.grad = p.grad.detach().clone()
        o2.step(); o1.step()
        sched.step(); sched1.step()
    for (n1, p), sp in zip(m.named_parameters(), shadow):
        np.testing.assert_allclose(p.detach().cpu().numpy(), sp.detach().cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg=n1)
    sd = o2.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(shadow)
    assert sd["state"][0]["exp_avg"].shape == shadow[0].shape
    np.testing.assert_allclose(sd["state"][3]["exp_avg_sq"].cpu().numpy(), o1.state[shadow[3]]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
    o3 = (FlatAdam if kind == "adam" else FlatAdamW)(m, lr=1e-3)
    o3.load_state_dict(sd)
    assert o3._step == 4 and torch.equal(o3._exp_avg, o2._exp_avg)
    assert o3.param_groups[0]["lr"] == o2.param_groups[0]["lr"]
