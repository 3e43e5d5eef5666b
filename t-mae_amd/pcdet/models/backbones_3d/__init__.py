from tmae_amd.modules import registry
from tmae_amd.modules.siam_wca_mae import SiamWCA_MAE  # noqa: F401

__all__ = registry.BACKBONES_3D
