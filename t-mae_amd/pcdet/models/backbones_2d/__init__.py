from tmae_amd.modules import registry
from tmae_amd.modules.bev_backbone import SSTBEVBackbone  # noqa: F401

__all__ = registry.BACKBONES_2D
