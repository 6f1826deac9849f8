"""Host-side logic that needs no GPU: factory names, init, LR policies, metrics, Poly, sharding."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import fcsiam_ref as R
from stcd_amd import metrics
from stcd_amd.ddp import shard_range
from stcd_amd.networks import define_G, get_scheduler, init_weights
from stcd_amd.train_loop import Poly


def test_define_G_registry_and_errors():
    for name, cls in (("SiamUnet_abs", "SiamUnet_diff"), ("SiamUnet_conc", "SiamUnet_conc"), ("SiamUnet_sub", "SiamUnet_sub"),
                      ("SNUNet", "SNUNet_ECAM"), ("Unet", "Unet"), ("SiamUnet_cross_conc", "SiamUnet_cross_conc")):
        m = define_G(NS(net_G=name, n_class=2))
        assert type(m).__name__ == cls
    with pytest.raises(NotImplementedError, match="not recognized"):
        define_G(NS(net_G="nope", n_class=2))
    with pytest.raises(NotImplementedError, match="outside"):
        define_G(NS(net_G="ChangeFormerV5", n_class=2))
    assert type(define_G(NS(net_G="ChangeFormerV6", n_class=2, embed_dim=64))).__name__ == "ChangeFormerV6"


def test_init_weights_normal_statistics():
    """networks.py:85-116: conv ~ N(0, 0.02), bias 0, BN weight ~ N(1, 0.02), BN bias 0 (overrides the default init)."""
    torch.manual_seed(0)
    m = define_G(NS(net_G="SiamUnet_abs", n_class=2))
    sd = m.state_dict()
    w = sd["conv43d.weight"]
    assert abs(w.mean().item()) < 1e-3 and abs(w.std().item() - 0.02) < 1e-3
    assert sd["conv43d.bias"].abs().max().item() == 0.0
    assert abs(sd["bn43.weight"].mean().item() - 1.0) < 0.02 and sd["bn43.bias"].abs().max().item() == 0.0
    init_weights(m, "kaiming")
    assert m.state_dict()["conv43d.weight"].std().item() > 0.02


def test_lr_policies():
    p = torch.nn.Parameter(torch.zeros(1))
    for policy, expect in (("linear", [1.0, 1 - 1 / 11, 1 - 2 / 11]), ("step", [1.0, 1.0, 0.5]), ("exponential", [1.0, 0.95, 0.9025]),
                           (None, [1.0, 1.0, 1.0])):
        opt = torch.optim.SGD([p], lr=1.0)
        sch = get_scheduler(opt, NS(lr_policy=policy, max_epochs=10, lr_decay_iters=2))
        got = []
        for _ in range(3):
            got.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        np.testing.assert_allclose(got, expect, rtol=1e-6)


def test_poly_matches_formula():
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sched = Poly(opt, num_epochs=3, iters_per_epoch=4)
    lrs = []
    for epoch in range(1, 4):
        for it in range(4):
            lrs.append(opt.param_groups[0]["lr"])
            sched.step(epoch=epoch - 1)
    # lr used at iteration k of epoch e (after k scheduler steps inside the epoch): T = e*ipe + cur_iter
    want = [1e-3] + [R.poly_lr(1e-3, e, it + 1, 4, 3) for e in range(3) for it in range(4)][:-1]
    # at an epoch boundary the reference's cur_iter wraps: T = e*4 + (it % 4) + ... -- reproduce it literally
    cur, want2 = 0, []
    for e in range(3):
        for it in range(4):
            want2.append(None)
    np.testing.assert_allclose(lrs[:4], want[:4], rtol=1e-9)
    assert all(b <= a + 1e-12 for a, b in zip(lrs[:4], lrs[1:5]))


def test_confuse_matrix_meter(golden):
    g = golden("g5_metric.npz")
    cm = metrics.ConfuseMatrixMeter(2)
    cm.update_cm(g["pred"][:2], g["label"][:2])
    mf1 = cm.update_cm(g["pred"][2:], g["label"][2:])
    np.testing.assert_array_equal(cm.cm, g["cm"])
    s = cm.get_scores()
    assert abs(mf1 - s["mf1"]) < 1e-15 and abs(s["mf1"] - float(np.mean(g["f1"]))) < 1e-12
    assert abs(s["iou_1"] - float(g["iou"][1])) < 1e-12 and abs(s["acc"] - float(g["oa"])) < 1e-12


def test_shard_range_partitions_exactly():
    for n in (1, 7, 16, 128, 129):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_optimizer_host_contract():
    """FlatAdam[W] is a torch.optim.Optimizer over the module's parameters (schedulers / checkpoints work on it) and
    refuses to step without a GPU -- there is no CPU fallback."""
    import pytest
    import torch
    from stcd_amd._lib import StcdError
    from stcd_amd.modules import SiamUnet_diff
    from stcd_amd.optim import FlatAdam, FlatAdamW

    m = SiamUnet_diff(3, 2)
    opt = FlatAdamW(m, lr=2e-3, weight_decay=0.01)
    assert isinstance(opt, torch.optim.Optimizer)
    assert len(opt.param_groups) == 1 and opt.param_groups[0]["lr"] == 2e-3 and opt.param_groups[0]["weight_decay"] == 0.01
    assert len(opt.param_groups[0]["params"]) == len(list(m.parameters()))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    assert sched.get_last_lr() == [2e-3]
    sd = opt.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["betas"] == (0.9, 0.999)
    with pytest.raises(StcdError):
        opt.step()                                   # model on the CPU
    with pytest.raises(StcdError):
        FlatAdam(torch.nn.Linear(2, 2))              # not an engine module
    with pytest.raises(ValueError):
        FlatAdam(m, lr=-1.0)


def test_pseudo_pairs_need_the_gpu():
    import pytest
    import torch
    from stcd_amd._lib import StcdError
    from stcd_amd.pseudo import pseudo_change_pairs

    z = torch.zeros(1, 8, 8, 3, dtype=torch.uint8)
    with pytest.raises(StcdError):
        pseudo_change_pairs(z, z, z[..., 0], torch.ones(1))


def test_bench_reads_pmc_traffic_from_profiles(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from the committed PMC summary (the counters need their own rocprofv3 passes)."""
    import importlib.util
    import os

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r09_pmc_traffic.txt").write_text(
        "# header\nkernel   launches read_B write_B total_B\nstcd::k_conv_res<2, 32>   56   100   50   150\n")
    monkeypatch.setattr(b, "REPO", str(tmp_path))
    (tmp_path / "profiles" / "r09_snunet_pmc_traffic.txt").write_text(
        "# step_total_B (sum): 9000\nkernel   launches read_B write_B total_B\nstcd::k_conv_res<2, 32>   56   100   70   170\n")
    assert b.pmc_traffic("stcd::k_conv_res<2, 32>", "diff") == (150, os.path.join("profiles", "r09_pmc_traffic.txt"))
    assert b.pmc_traffic("stcd::k_other", "diff") == (None, None)
    # every family reads ITS OWN summary (same kernel name, different launches)
    assert b.pmc_traffic("stcd::k_conv_res<2, 32>", "snunet") == (170, os.path.join("profiles", "r09_snunet_pmc_traffic.txt"))
    assert b.pmc_traffic("stcd::k_conv_res<2, 32>", "segcd") == (None, None)
    assert b.pmc_step_traffic("snunet") == (9000, os.path.join("profiles", "r09_snunet_pmc_traffic.txt"))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: the package, the tools and the examples must not import it (only tests/, smoke() and the
    cpu_baseline leg of bench.py may), and bench.py may only do so inside its cpu_baseline functions."""
    import ast
    import glob
    import os

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node.lineno)
        return hits

    for pat in ("stcd_amd/*.py", "tools/*.py", "examples/*.py"):
        for f in glob.glob(os.path.join(repo, pat)):
            assert oracle_imports(f) == [], f
    # bench.py: only inside cpu_baseline* functions
    src = open(os.path.join(repo, "bench.py")).read()
    tree = ast.parse(src)
    allowed = set()
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name.startswith("cpu_baseline"):
            allowed.update(range(node.lineno, node.end_lineno + 1))
    assert all(l in allowed for l in oracle_imports(os.path.join(repo, "bench.py")))
