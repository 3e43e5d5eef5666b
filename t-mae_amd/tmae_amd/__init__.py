"""tmae_amd: MI355X-native T-MAE pre-training hot path (HIP kernels behind a C ABI + pcdet-style modules).

Importing this package loads libtmae_hip.so; there is no CPU / eager fallback (see _lib.py).
"""
import os

# The dense decoder keeps one MIOpen convolution (3x3, 384 -> 128).  MIOpen's default find benchmarks every applicable
# solver on first use, including its reference 'naive' solvers that take ~90 s on this shape and never win; leave
# them out of the search unless the user has set the variables.
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    os.environ.setdefault(_k, '0')

from . import _lib  # noqa: F401,E402  (fails loudly when the HIP extension is missing)

__version__ = '0.1.0'
