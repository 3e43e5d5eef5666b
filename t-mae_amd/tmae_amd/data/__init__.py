from .pipeline import TemporalPairPipeline, prev_to_cur_transform  # noqa: F401
from .once_temporal import (ONCETemporalDataset, DeviceBatchLoader, EpochSampler, build_dataloader,  # noqa: F401
                            interval_list)
