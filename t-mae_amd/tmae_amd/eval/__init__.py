from .eval_loop import eval_one_epoch, merge_results_dist, statistics_info  # noqa: F401
from .once_eval import get_evaluation_results, iou3d_with_heading  # noqa: F401
