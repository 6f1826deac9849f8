"""The plain-C per-op oracle (oracle/ops_ref.c) against the G1 vectors captured from torch.nn modules
configured exactly as the reference configures them.  Pins every explicit backward formula."""
import numpy as np

from oracle import ops_c as O

TOL = dict(rtol=2e-5, atol=2e-5)


def test_conv_family(golden):
    g = golden("g1_ops.npz")
    for tag, pad in (("conv_3_16", 1), ("conv_16_32", 1), ("conv1x1_128_2", 0)):
        x, w, b, gy = g[f"{tag}/x"], g[f"{tag}/weight"], g[f"{tag}/bias"], g[f"{tag}/gy"]
        np.testing.assert_allclose(O.conv2d_fwd(x, w, b, pad), g[f"{tag}/y"], **TOL)
        dx, dw, db = O.conv2d_bwd(x, w, gy, pad)
        np.testing.assert_allclose(dx, g[f"{tag}/dx"], **TOL)
        np.testing.assert_allclose(dw, g[f"{tag}/dweight"], rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(db, g[f"{tag}/dbias"], rtol=2e-5, atol=1e-4)
    for tag, k, s, p, op in (("convT_s1_32_16", 3, 1, 1, 0), ("convT_s2_16_16", 3, 2, 1, 1), ("convT_k2s2_32", 2, 2, 0, 0)):
        x, w, b, gy = g[f"{tag}/x"], g[f"{tag}/weight"], g[f"{tag}/bias"], g[f"{tag}/gy"]
        np.testing.assert_allclose(O.convT2d_fwd(x, w, b, s, p, op), g[f"{tag}/y"], **TOL)
        dx, dw, db = O.convT2d_bwd(x, w, gy, s, p, op)
        np.testing.assert_allclose(dx, g[f"{tag}/dx"], **TOL)
        np.testing.assert_allclose(dw, g[f"{tag}/dweight"], rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(db, g[f"{tag}/dbias"], rtol=2e-5, atol=1e-4)


def test_batchnorm(golden):
    g = golden("g1_ops.npz")
    y, mean, invstd, rm, rv = O.bn_train_fwd(g["bn/x"], g["bn/weight"], g["bn/bias"], g["bn/rm0"], g["bn/rv0"])
    np.testing.assert_allclose(y, g["bn/y"], **TOL)
    np.testing.assert_allclose(rm, g["bn/rm1"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv, g["bn/rv1"], rtol=1e-6, atol=1e-7)
    dx, dg, db = O.bn_train_bwd(g["bn/x"], g["bn/gy"], g["bn/weight"], mean, invstd)
    np.testing.assert_allclose(dx, g["bn/dx"], **TOL)
    np.testing.assert_allclose(dg, g["bn/dweight"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(db, g["bn/dbias"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(O.bn_eval_fwd(g["bn/x"], g["bn/weight"], g["bn/bias"], g["bn/rm1"], g["bn/rv1"]),
                               g["bn/y_eval"], **TOL)


def test_pool_and_fusion(golden):
    g = golden("g1_ops.npz")
    for tag in ("pool", "pool_odd"):
        y, _ = O.maxpool2_fwd(g[f"{tag}/x"])
        np.testing.assert_array_equal(y, g[f"{tag}/y"])
        np.testing.assert_array_equal(O.maxpool2_bwd(g[f"{tag}/x"], g[f"{tag}/gy"]), g[f"{tag}/dx"])
    a, b, gg = g["fuse/a"], g["fuse/b"], g["fuse/g"]
    np.testing.assert_array_equal(O.fuse_fwd(a, b, 0), g["fuse/abs"])
    da, db = O.fuse_bwd(a, b, gg, 0)
    np.testing.assert_array_equal(da, g["fuse/abs_da"])
    np.testing.assert_array_equal(db, g["fuse/abs_db"])
    np.testing.assert_array_equal(O.fuse_fwd(a, b, 1), b - a)
    np.testing.assert_array_equal(O.rep_pad_fwd(g["rpad/x"], 5, 6), g["rpad/y"])
    np.testing.assert_allclose(O.rep_pad_bwd(g["rpad/gy"], 4, 5), g["rpad/dx"], rtol=1e-6, atol=1e-6)


def test_losses(golden):
    g = golden("g1_ops.npz")
    loss, dl = O.ce_fwd_bwd(g["ce/logits"], g["ce/target"])
    assert abs(loss - float(g["ce/loss"])) < 1e-6
    np.testing.assert_allclose(dl, g["ce/dlogits"], rtol=1e-5, atol=1e-8)
    for tag in ("cd", "cd_sat"):
        loss, dl = O.bce_dice_fwd_bwd(g[f"{tag}/logits"], g[f"{tag}/target"])
        assert abs(loss - float(g[f"{tag}/loss"])) < 1e-5 * max(1.0, abs(float(g[f"{tag}/loss"])))
        np.testing.assert_allclose(dl, g[f"{tag}/dlogits"], rtol=1e-4, atol=1e-8)
