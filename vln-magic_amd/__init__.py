"""magic_amd: MI355X-native (gfx950) implementation of VLN-MAGIC's cross-modal transformer +
MAKD distillation training hot path.  Import as ``magic_amd`` (see /magic_amd.py shim)."""
__version__ = "0.1.0"
