from .optim import build_optimizer, build_scheduler, AdamOneCycle, OneCycle  # noqa: F401
from .synthetic import SyntheticEvalLoader, SyntheticTemporalDataset, synth_frame_pair, synth_gt_boxes  # noqa: F401
from .engine import build_model_from_cfg, settle_gc, train_one_step, unsettle_gc, wrap_ddp  # noqa: F401
