from tmae_amd.modules import registry
from tmae_amd.modules.detector import Detector3DTemplate, TMAE  # noqa: F401

__all__ = registry.detectors()


def build_detector(model_cfg, num_class, dataset, logger=None):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)
