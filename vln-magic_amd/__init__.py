"""magic_amd: MI355X-native (gfx950) implementation of VLN-MAGIC's cross-modal transformer +
MAKD distillation training hot path.  Import as ``magic_amd`` (see /magic_amd.py shim)."""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: `magic_amd.wrap_model` / `magic_amd.DistributedDataParallel` (host/ddp.py) without importing torch at package import
    if name in ("wrap_model", "DistributedDataParallel", "TorchDDPWrapperError"):
        from .host import ddp
        return getattr(ddp, name)
    raise AttributeError(name)
