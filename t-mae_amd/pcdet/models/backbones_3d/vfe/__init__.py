from tmae_amd.modules import registry
from tmae_amd.modules.vfe import TemporalDynVFE, VFETemplate  # noqa: F401

__all__ = registry.VFE
