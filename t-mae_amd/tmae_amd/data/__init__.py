from .pipeline import TemporalPairPipeline, prev_to_cur_transform  # noqa: F401
