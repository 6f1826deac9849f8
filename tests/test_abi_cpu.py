"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header declares,
the engine's parameter table is the reference's state_dict layout, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from oracle import fcsiam_ref as R
from stcd_amd import _lib
from stcd_amd.modules import SiamUnet_conc, SiamUnet_diff, SiamUnet_sub

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "stcd_hip.h")).read()
    declared = set(re.findall(r"\b(stcd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in stcd_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.lib().stcd_abi_version() == 2


@pytest.mark.parametrize("cls,arch", [(SiamUnet_diff, "diff"), (SiamUnet_conc, "conc"), (SiamUnet_sub, "sub")])
@pytest.mark.parametrize("label", [1, 2])
def test_state_dict_layout_is_the_reference_layout(cls, arch, label):
    m = cls(3, label)
    sd = m.state_dict()
    specs = R.param_specs(arch, 3, label)          # pinned against the reference by tests/golden (oracle tests)
    assert list(sd.keys()) == [n for n, _, _ in specs]
    for (n, shape, _), v in zip(specs, sd.values()):
        assert tuple(v.shape) == tuple(shape), n
    # loading a reference-shaped state dict works strictly, and round-trips
    st = R.synth_state(arch, 3, label, seed=5, perturb_running=True)
    m.load_state_dict(st, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, st[k]), k


def test_init_weights_style_apply_sees_reference_class_names():
    m = SiamUnet_diff(3, 2)
    seen = {"Conv": 0, "BatchNorm2d": 0}

    def fn(mod):
        name = mod.__class__.__name__
        if hasattr(mod, "weight") and name.find("Conv") != -1:
            seen["Conv"] += 1
        elif name.find("BatchNorm2d") != -1:
            seen["BatchNorm2d"] += 1

    m.apply(fn)
    assert seen == {"Conv": 24, "BatchNorm2d": 19}


def test_cpu_tensors_are_rejected_loudly():
    m = SiamUnet_diff(3, 2)
    x = torch.zeros(1, 3, 32, 32)
    with pytest.raises(_lib.StcdError):
        m(x, x)


def test_error_reporting_through_the_abi():
    l = _lib.lib()
    h = ctypes.c_void_p()
    assert l.stcd_create(99, 3, 2, 0, ctypes.byref(h)) != 0
    assert b"arch" in l.stcd_last_error()
    assert l.stcd_create(0, 3, 2, 0, ctypes.byref(h)) == 0
    assert l.stcd_configure(h, 1, 8, 8) != 0 and b">= 16" in l.stcd_last_error()
    assert l.stcd_configure(h, 2, 100, 100) == 0 and l.stcd_workspace_bytes(h) > 0
    l.stcd_destroy(h)


def test_snunet_state_dict_layout_is_the_reference_layout():
    from oracle import snunet_ref as S
    from stcd_amd.modules import SNUNet_ECAM

    for label in (1, 2):
        m = SNUNet_ECAM(3, label)
        sd = m.state_dict()
        specs = S.param_specs(3, label)           # pinned against the reference by tests/golden (oracle tests)
        assert list(sd.keys()) == [n for n, _, _ in specs] and len(sd) == 236
        for (n, shape, _), v in zip(specs, sd.values()):
            assert tuple(v.shape) == tuple(shape), n
        st = S.synth_state(3, label, seed=3, perturb_running=True)
        m.load_state_dict(st, strict=True)
        assert all(torch.equal(v, st[k]) for k, v in m.state_dict().items())


def test_header_is_plain_c_and_usable_without_python(tmp_path):
    """examples/abi_query.c: a C99 program including include/stcd_hip.h, linked against the library, reads the parameter
    layout and workspace size through the ABI (no Python, torch or C++ on the caller's side)."""
    import os
    import shutil
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("gcc not available")
    exe = str(tmp_path / "abi_query")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "examples", "abi_query.c"), "-o", exe, "-L", os.path.join(repo, "stcd_amd"),
                           "-lstcd_hip", "-Wl,-rpath," + os.path.join(repo, "stcd_amd")])
    out = subprocess.check_output([exe, "0", "2", "64"], text=True)
    assert "params 86 tensors / 1350148 floats" in out and "conv11.weight" in out and "shape 16 3 3 3" in out
    assert "workspace" in out and "rejected bad shape" in out
    out = subprocess.check_output([exe, "3", "1", "32"], text=True)
    assert "params 146 tensors / 12034980 floats" in out and "conv0_0.conv1.weight" in out


@pytest.mark.parametrize("cls_name,encoder,classes", [("SegCD", "resnet50", 1), ("SegCD", "resnet18", 2), ("SegCD", "resnet34", 1), ("SegCD", "resnet101", 1),
                                                      ("SegCD", "resnet152", 1), ("UnetSeg", "resnet50", 1), ("UnetSeg", "resnet34", 2),
                                                      ("FFCTLCD", "resnet34", 1), ("FFCTLCD", "resnet50", 2)])
def test_segcd_family_state_dict_layout_is_the_reference_layout(cls_name, encoder, classes):
    """SegCD / UnetSeg / FFCTLCD over every supported encoder: the module's state_dict keys, order and shapes are the reference's
    (oracle.segcd_ref.param_specs, pinned against the reference's own classes by the G10-G16 fixtures), the engine enumerates the
    same parameters (HipChangeDetector._check_layout ran in the constructor), and a reference-shaped state dict loads strictly."""
    from oracle import segcd_ref as G
    from stcd_amd import segcd
    m = getattr(segcd, cls_name)(encoder_name=encoder, classes=classes)
    specs = G.param_specs(3, classes, encoder)
    sd = m.state_dict()
    assert list(sd.keys()) == [n for n, _, _ in specs]
    for (n, shape, _), v in zip(specs, sd.values()):
        assert tuple(v.shape) == tuple(shape), n
    st = G.synth_state(3, classes, 5, perturb_running=True, encoder=encoder)
    m.load_state_dict(st, strict=True)
    for k in ("encoder.conv1.weight", "decoder.blocks.4.conv2.1.running_var", "segmentation_head.0.bias"):
        assert torch.equal(m.state_dict()[k], st[k]), k
    calls = {b.name: b.calls_per_forward for b in m._engine.bns}
    enc_calls, dec_calls = {"SegCD": (2, 2), "UnetSeg": (1, 1), "FFCTLCD": (2, 3)}[cls_name]
    assert calls["encoder.bn1"] == enc_calls and calls["decoder.blocks.0.conv1.1"] == dec_calls      # num_batches_tracked increments
