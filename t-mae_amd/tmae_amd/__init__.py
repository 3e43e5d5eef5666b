"""tmae_amd: MI355X-native T-MAE pre-training hot path (HIP kernels behind a C ABI + pcdet-style modules).

Importing this package loads libtmae_hip.so; there is no CPU / eager fallback (see _lib.py).
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension is missing)

__version__ = '0.1.0'
