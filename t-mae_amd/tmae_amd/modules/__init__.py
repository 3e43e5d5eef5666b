from .sparse import SparseConvTensor, SubMConv2d, SparseConv2d, SparseSequential, post_act_block  # noqa: F401
from .vfe import TemporalDynVFE  # noqa: F401
from .sst import SSTBlockV1, WCABlock, CosineMultiheadAttention  # noqa: F401
from .siam_wca_mae import SiamWCA_MAE  # noqa: F401
from .detector import Detector3DTemplate, TMAE  # noqa: F401
