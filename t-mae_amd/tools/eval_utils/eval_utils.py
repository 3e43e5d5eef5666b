"""tools/eval_utils/eval_utils.py of the reference: the same entry points, implemented in tmae_amd.eval."""
from tmae_amd.eval import eval_one_epoch, statistics_info  # noqa: F401
