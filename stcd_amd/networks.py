"""Factory side of the boundary: same function names, argument meaning and error behaviour as
/root/reference/models/networks.py (``get_scheduler`` :26-57, ``init_weights`` :85-116, ``init_net`` :119-135,
``define_G`` :138-215) for the networks on the hot path.  The model classes returned are the HIP-engine modules of
``stcd_amd.modules``; names of networks outside the path raise NotImplementedError like an unknown name does."""
from __future__ import annotations

import torch
from torch.nn import init
from torch.optim import lr_scheduler

from .modules import SiamUnet_conc, SiamUnet_cross_conc, SiamUnet_diff, SiamUnet_sub, Unet

_ON_PATH = {
    "Unet": lambda n: Unet(input_nbr=3, label_nbr=n),                     # networks.py:144-145 (FC-EF)
    "SiamUnet_abs": lambda n: SiamUnet_diff(input_nbr=3, label_nbr=n),    # networks.py:148-149
    "SiamUnet_conc": lambda n: SiamUnet_conc(input_nbr=3, label_nbr=n),   # :150-151
    "SiamUnet_sub": lambda n: SiamUnet_sub(input_nbr=3, label_nbr=n),     # :146-147
    "SiamUnet_cross_conc": lambda n: SiamUnet_cross_conc(input_nbr=3, label_nbr=n),   # :152-153
}
_OFF_PATH = ("DTCDSCN", "IFNet", "base_resnet18", "base_transformer_pos_s4",
             "base_transformer_pos_s4_dd8", "base_transformer_pos_s4_dd8_dedim8", "ChangeFormerV1", "ChangeFormerV2",
             "ChangeFormerV3", "ChangeFormerV4", "ChangeFormerV5", "ChangeGNNV1", "ChangeGNNV2",
             "ChangeGNNV2_sub", "ChangeGNNV2_abs", "ChangeGNNV2_conc", "GNN")


def _register_snunet():
    try:
        from .modules import SNUNet_ECAM
    except ImportError:
        return
    _ON_PATH["SNUNet"] = lambda n: SNUNet_ECAM(in_ch=3, out_ch=n)        # networks.py:168-169


_register_snunet()


def get_scheduler(optimizer, args):
    """linear | step | exponential | None, stepped once per EPOCH by the trainer (trainer.py:350)."""
    if args.lr_policy == "linear":
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0 - epoch / float(args.max_epochs + 1))
    if args.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=args.lr_decay_iters, gamma=0.5)
    if args.lr_policy == "exponential":
        return lr_scheduler.ExponentialLR(optimizer, 0.95)
    if args.lr_policy is None:
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0)
    return NotImplementedError("learning rate policy [%s] is not implemented", args.lr_policy)   # returned, as the reference does


def init_weights(net, init_type="normal", init_gain=0.02):
    """Conv / Linear weights by ``init_type`` (bias 0); BatchNorm2d weight ~ N(1, gain), bias 0.
    Works on the engine modules because their holder sub-modules carry the reference's class names."""

    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    print("initialize network with %s" % init_type)
    net.apply(init_func)


def init_net(net, init_type="normal", init_gain=0.02, gpu_ids=[]):
    """Device placement + weight init.  One process drives one GPU here: for several GPUs launch one process per
    GPU (torchrun) and wrap with stcd_amd.ddp.FlatGradReducer instead of nn.DataParallel (networks.py:132-133)."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        if len(gpu_ids) > 1:
            raise NotImplementedError("multi-GPU runs use one process per GPU (torchrun + stcd_amd.ddp), not "
                                      "nn.DataParallel over gpu_ids=%s" % (gpu_ids,))
        net.to(gpu_ids[0])
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(args, init_type="normal", init_gain=0.02, gpu_ids=[]):
    name = args.net_G
    if name == "ChangeFormerV6":                           # networks.py:195-196: ChangeFormerV6(embed_dim=args.embed_dim), two classes
        from .changeformer import ChangeFormerV6
        net = ChangeFormerV6(embed_dim=getattr(args, "embed_dim", 256))
    elif name in _ON_PATH:
        net = _ON_PATH[name](args.n_class)
    elif name in _OFF_PATH:
        raise NotImplementedError("Generator model name [%s] is outside the accelerated hot path of this build" % name)
    else:
        raise NotImplementedError("Generator model name [%s] is not recognized" % name)
    return init_net(net, init_type, init_gain, gpu_ids)
