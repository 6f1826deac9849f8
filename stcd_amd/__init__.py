"""stcd_amd -- MI355X-native engine for the bi-temporal change-detection hot path of VCISwang/STCD.

The package holds only what that path needs: ``csrc/`` (HIP kernels + C ABI -> libstcd_hip.so) and the
host-side mirror of the reference's Python interface for the path (model classes, ``define_G``, losses,
``CDTrainer``).  Importing the package does not touch the GPU or load the library; the first model
construction does, and raises if the library has not been built.
"""
__all__ = ["SiamUnet_diff", "SiamUnet_conc", "SiamUnet_sub", "SNUNet_ECAM", "SegCD", "UnetSeg", "FFCTLCD"]


def __getattr__(name):
    if name in ("SegCD", "UnetSeg", "FFCTLCD"):
        from . import segcd
        return getattr(segcd, name)
    if name in __all__:
        from . import modules
        return getattr(modules, name)
    raise AttributeError(name)
