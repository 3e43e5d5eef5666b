"""Name -> class registries with the reference's keys: pcdet/models/detectors/__init__.py:11-29,
backbones_3d/__init__.py, backbones_3d/vfe/__init__.py.  Entries outside the pre-training hot path resolve to a
stub that says so."""
from .siam_wca_mae import SiamWCA_MAE
from .vfe import TemporalDynVFE


def _next_row(name):
    class _NotOnHotPath:
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name}: fine-tune path, not part of the pre-training hot path yet '
                                      f'(SURVEY.md 8f rank 1)')
    _NotOnHotPath.__name__ = name
    return _NotOnHotPath


VFE = {'TemporalDynVFE': TemporalDynVFE}
BACKBONES_3D = {'SiamWCA_MAE': SiamWCA_MAE, 'SiamWCA': _next_row('SiamWCA')}


def detectors():
    from .detector import Detector3DTemplate, TMAE
    return {'Detector3DTemplate': Detector3DTemplate, 'TMAE': TMAE, 'CenterPoint': _next_row('CenterPoint')}
