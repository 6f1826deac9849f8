"""HIP loss / metric kernels (through stcd_amd.losses / metrics -> C ABI) against the reference vectors (G1, G5)
and the C oracle."""
import numpy as np
import pytest
import torch

from oracle import ops_c as O
from stcd_amd import losses, metrics
from tests._util import t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cross_entropy_value_and_gradient(golden):
    g = golden("g1_ops.npz")
    lg = t(g["ce/logits"]).to(DEV).requires_grad_(True)
    tgt = t(g["ce/target"]).to(DEV)
    loss = losses.cross_entropy(lg, tgt.float().unsqueeze(1))      # the trainer hands a float N*1*H*W label
    loss.backward()
    assert abs(loss.item() - float(g["ce/loss"])) < 1e-6
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g["ce/dlogits"], rtol=1e-5, atol=1e-8)
    # upstream gradient is honoured
    lg2 = t(g["ce/logits"]).to(DEV).requires_grad_(True)
    (3.0 * losses.cross_entropy(lg2, tgt)).backward()
    np.testing.assert_allclose(lg2.grad.cpu().numpy(), 3.0 * g["ce/dlogits"], rtol=1e-5, atol=1e-8)


def test_cross_entropy_large_random_vs_c_oracle():
    rng = np.random.default_rng(3)
    lg = (3 * rng.standard_normal((3, 2, 64, 48))).astype(np.float32)
    tg = rng.integers(0, 2, size=(3, 64, 48)).astype(np.int64)
    tg[rng.random(tg.shape) < 0.05] = 255
    ref_loss, ref_d = O.ce_fwd_bwd(lg, tg)
    x = t(lg).to(DEV).requires_grad_(True)
    loss = losses.cross_entropy(x, t(tg).to(DEV))
    loss.backward()
    assert abs(loss.item() - ref_loss) < 1e-5
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_d, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("tag", ["cd", "cd_sat"])
def test_cd_loss_probability_form_and_fused_logit_form(golden, tag):
    g = golden("g1_ops.npz")
    ref_loss = float(g[f"{tag}/loss"])
    # reference call form: cd_loss(sigmoid(x), y)
    lg = t(g[f"{tag}/logits"]).to(DEV).requires_grad_(True)
    tgt = t(g[f"{tag}/target"]).to(DEV)
    loss = losses.cd_loss(torch.sigmoid(lg), tgt)
    loss.backward()
    assert abs(loss.item() - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g[f"{tag}/dlogits"], rtol=2e-4, atol=1e-7)
    # fused form
    lg2 = t(g[f"{tag}/logits"]).to(DEV).requires_grad_(True)
    loss2 = losses.bce_dice_with_logits(lg2, tgt)
    loss2.backward()
    assert abs(loss2.item() - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    np.testing.assert_allclose(lg2.grad.cpu().numpy(), g[f"{tag}/dlogits"], rtol=2e-4, atol=1e-7)
    assert isinstance(losses.BCE_DICE()(torch.sigmoid(lg2.detach()), tgt).item(), float)


def test_dice_module_matches_its_formula(golden):
    """class Dice (train_pse_cd.py:436-447): 1 - (2*sum(p*t) + 1) / (sum(p) + sum(t) + 1), value and gradient against the
    same expression in fp64 autograd; and BCE_DICE == BCELoss(mean) + Dice on the same inputs."""
    g = golden("g1_ops.npz")
    p0 = torch.sigmoid(t(g["cd/logits"]).double())
    tgt = t(g["cd/target"]).double()
    pr = p0.clone().requires_grad_(True)
    want = 1 - (2.0 * (pr.view(-1) * tgt.view(-1)).sum() + 1.0) / (pr.sum() + tgt.sum() + 1.0)
    want.backward()
    pd = p0.float().to(DEV).requires_grad_(True)
    got = losses.Dice()(pd, tgt.float().to(DEV))
    got.backward()
    assert abs(got.item() - want.item()) < 1e-6
    np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.float().numpy(), rtol=2e-4, atol=1e-9)
    both = losses.BCE_DICE()(p0.float().to(DEV), tgt.float().to(DEV)).item()
    bce = torch.nn.functional.binary_cross_entropy(p0.float(), tgt.float()).item()
    assert abs(both - (bce + want.item())) < 1e-5


def test_confusion_matrix_on_device(golden):
    g = golden("g5_metric.npz")
    lab, pred = t(g["label"]).to(DEV), t(g["pred"]).to(DEV)
    # two-class logits whose argmax is `pred`, and one-class logits whose sign is `pred`
    l2 = torch.stack([1.0 - pred[:, 0].float(), pred[:, 0].float()], 1).contiguous()
    l1 = (pred.float() * 2 - 1).contiguous()
    for logits in (l2, l1):
        m = metrics.SegmentationMetric(2, DEV)
        m.add_logits(logits[:2], lab[:2])
        m.add_logits(logits[2:], lab[2:])          # accumulates across batches
        np.testing.assert_array_equal(m.confusionMatrix.cpu().numpy(), g["cm"])
        np.testing.assert_allclose(m.F1score(), g["f1"], rtol=1e-12)
        np.testing.assert_allclose(m.IntersectionOverUnion(), g["iou"], rtol=1e-12)
        np.testing.assert_allclose(m.Precision(), g["precision"], rtol=1e-12)
        np.testing.assert_allclose(m.Recall(), g["recall"], rtol=1e-12)
        assert abs(m.OverallAccuracy() - float(g["oa"])) < 1e-12
    cm = metrics.ConfuseMatrixMeter(2)
    cm.update_cm(g["pred"], g["label"])
    assert abs(cm.get_scores()["mf1"] - float(np.mean(g["f1"]))) < 1e-12


def test_full_size_loss_and_metric_properties():
    """Bench-sized tensors (16 x 2 x 256 x 256 logits): the confusion matrix is a checksum of its inputs (entries sum to
    the pixel count, rows to the label histogram, accumulation over half-batches equals one call), and the
    cross-entropy gradient sums to zero over the class axis at every valid pixel and to exactly zero at ignored ones."""
    rng = np.random.default_rng(3)
    B, H, W = 16, 256, 256
    logits = t(rng.standard_normal((B, 2, H, W)).astype(np.float32)).to(DEV)
    lab = t((rng.random((B, H, W)) < 0.2).astype(np.int64)).to(DEV)
    whole = metrics.SegmentationMetric(2, DEV)
    whole.add_logits(logits, lab.unsqueeze(1))
    halves = metrics.SegmentationMetric(2, DEV)
    halves.add_logits(logits[:8], lab[:8].unsqueeze(1))
    halves.add_logits(logits[8:], lab[8:].unsqueeze(1))
    cm = whole.confusionMatrix.cpu().numpy()
    np.testing.assert_array_equal(cm, halves.confusionMatrix.cpu().numpy())
    assert cm.sum() == B * H * W
    np.testing.assert_array_equal(cm.sum(axis=1), np.bincount(lab.cpu().numpy().ravel(), minlength=2))
    pred = logits.argmax(1).cpu().numpy().ravel()
    np.testing.assert_array_equal(cm.sum(axis=0), np.bincount(pred, minlength=2))

    lab_ig = lab.clone()
    lab_ig[:, :8, :] = 255                                    # ignore_index rows
    lg = logits.clone().requires_grad_(True)
    loss = losses.cross_entropy(lg, lab_ig)
    loss.backward()
    g = lg.grad
    assert torch.isfinite(loss) and float(g[:, :, :8, :].abs().max()) == 0.0
    assert float(g.sum(1).abs().max()) < 1e-9                 # softmax - onehot sums to zero over classes
    ref = torch.nn.functional.cross_entropy(logits.cpu().double(), lab_ig.cpu(), ignore_index=255).item()
    assert abs(loss.item() - ref) < 1e-6


def test_cross_entropy_out_of_range_label_is_loud():
    """A label outside [0, classes) that is not the ignore index (label 2 with 2 classes, a negative value, a 0/255 mask
    with ignore_index changed): F.cross_entropy raises a device assert; the kernel must never dereference it -- loss and
    the gradient at that pixel come back NaN, every other pixel's gradient stays finite."""
    rng = np.random.default_rng(8)
    logits = t(rng.standard_normal((2, 2, 16, 16)).astype(np.float32)).to(DEV)
    for bad, ignore in ((2, 255), (-1, 255), (255, 100), (10 ** 12, 255)):
        lab = t((rng.random((2, 16, 16)) < 0.3).astype(np.int64)).to(DEV)
        lab[1, 3, 5] = bad
        lg = logits.clone().requires_grad_(True)
        loss = losses.cross_entropy(lg, lab, ignore_index=ignore)
        loss.backward()
        assert torch.isnan(loss), (bad, ignore)
        g = lg.grad
        assert torch.isnan(g[1, :, 3, 5]).all()
        g2 = g.clone(); g2[1, :, 3, 5] = 0
        assert torch.isnan(g2).sum().item() == 0 or torch.isnan(g2).all()   # NaN 1/count scaling may poison all: still loud


def test_bce_dice_ignores_cutout_label_255():
    """stcd_pseudo_pair writes 255 into the change label of cutout pixels (data/dataset.py:24-57 cutout contract).  On the
    1-channel loss those pixels take no part: loss and gradient equal the loss over the remaining pixels (oracle on the
    valid subset), gradient exactly 0 at the cutout.  Any other target outside [0,1] is an error: NaN."""
    from oracle import ops_c as O
    rng = np.random.default_rng(9)
    x = (2 * rng.standard_normal((2, 1, 16, 16))).astype(np.float32)
    y = (rng.random((2, 1, 16, 16)) < 0.3).astype(np.float32)
    cut = np.zeros_like(y, bool); cut[:, :, 4:9, 2:7] = True
    y_cut = y.copy(); y_cut[cut] = 255.0
    lg = t(x).to(DEV).requires_grad_(True)
    loss = losses.bce_dice_with_logits(lg, t(y_cut).to(DEV))
    loss.backward()
    ref_loss, ref_dl = O.bce_dice_fwd_bwd(x[~cut], y[~cut])
    assert abs(loss.item() - ref_loss) < 1e-5
    g = lg.grad.cpu().numpy()
    assert np.abs(g[cut]).max() == 0.0
    np.testing.assert_allclose(g[~cut], ref_dl, rtol=1e-4, atol=1e-8)
    y_bad = y.copy(); y_bad[0, 0, 0, 0] = 7.0
    assert torch.isnan(losses.bce_dice_with_logits(t(x).to(DEV), t(y_bad).to(DEV)))


def test_contrastive_loss_against_reference_vectors(golden):
    """stcd_loss_contrastive (train_stcd.py:334-385) against the vectors produced by the reference's own function: value and
    the gradient of BOTH halves, incl. the |cd - 1| kink, an odd map and the all-labels-equal case (empty N mask)."""
    g = golden("g8_contrastive.npz")
    for tag in ("a", "b", "same"):
        pred = t(g[f"{tag}/pred"]).to(DEV).requires_grad_(True)
        loss = losses.contrastive_loss(pred, t(g[f"{tag}/cd_label"]).to(DEV), t(g[f"{tag}/pse_label"]).to(DEV), None)
        (2.0 * loss).backward()
        assert abs(loss.item() - float(g[f"{tag}/loss"])) < 1e-6
        np.testing.assert_allclose(pred.grad.cpu().numpy(), 2.0 * g[f"{tag}/dpred"], rtol=1e-5, atol=1e-8)
    # bench-sized: against the oracle
    from oracle import fcsiam_ref as R
    rng = np.random.default_rng(12)
    p = rng.random((16, 1, 256, 256)).astype(np.float32)
    cd, ps = rng.integers(0, 2, size=(8, 1, 256, 256)), rng.integers(0, 2, size=(8, 1, 256, 256))
    pr = t(p).requires_grad_(True)
    ref = R.contrastive_loss(pr, t(cd), t(ps)); ref.backward()
    pg = t(p).to(DEV).requires_grad_(True)
    got = losses.contrastive_loss(pg, t(cd).to(DEV), t(ps).to(DEV)); got.backward()
    assert abs(got.item() - ref.item()) < 1e-6
    np.testing.assert_allclose(pg.grad.cpu().numpy(), pr.grad.numpy(), rtol=1e-4, atol=1e-10)
