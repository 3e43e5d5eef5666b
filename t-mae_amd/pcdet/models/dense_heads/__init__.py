from tmae_amd.modules import registry
from tmae_amd.modules.center_head import CenterHead  # noqa: F401

__all__ = registry.DENSE_HEADS
