"""Name -> class registries with the reference's keys: pcdet/models/detectors/__init__.py:11-29,
backbones_3d/__init__.py, backbones_3d/vfe/__init__.py, backbones_2d/__init__.py, dense_heads/__init__.py."""
from .bev_backbone import SSTBEVBackbone
from .center_head import CenterHead
from .siam_wca import SiamWCA
from .siam_wca_mae import SiamWCA_MAE
from .vfe import TemporalDynVFE

VFE = {'TemporalDynVFE': TemporalDynVFE}
BACKBONES_3D = {'SiamWCA_MAE': SiamWCA_MAE, 'SiamWCA': SiamWCA}
BACKBONES_2D = {'SSTBEVBackbone': SSTBEVBackbone}
DENSE_HEADS = {'CenterHead': CenterHead}


def detectors():
    from .detector import CenterPoint, Detector3DTemplate, TMAE
    return {'Detector3DTemplate': Detector3DTemplate, 'TMAE': TMAE, 'CenterPoint': CenterPoint}
