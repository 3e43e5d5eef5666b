"""pcdet compatibility surface for the T-MAE pre-training hot path (OpenPCDet config + registry API).

Only what tools/train.py needs for MODEL.NAME == TMAE lives here; the implementation is in tmae_amd.
"""
__version__ = '0.5.1+tmae_amd'
