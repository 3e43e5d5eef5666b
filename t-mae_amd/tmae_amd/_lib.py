"""ctypes binding of libtmae_hip.so (the C ABI of include/tmae_hip.h).

There is NO fallback: if the shared object is missing the import fails loudly, and every call checks the
library's return code.  Pointers are raw device addresses (tensor.data_ptr()), the stream is the HIP stream
of torch.cuda.current_stream() -- PyTorch is only the allocator / stream provider here.
"""
import ctypes as C
import os

import torch  # noqa: F401  MUST precede the CDLL below: torch bundles its own libamdhip64.so.7; loading ours first
#                     would put a second HIP runtime in the process (torch's pointers/streams would be foreign to it)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TMAE_LIB_PATH: another build of the same library (A/B runs of one kernel on one box); the default is the in-tree build
LIB_PATH = os.environ.get('TMAE_LIB_PATH') or os.path.join(_HERE, 'lib', 'libtmae_hip.so')

P, I, L, F, D, Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/tmae_hip.h one to one
SIGNATURES = {
    'tmae_abi_version': (I, []),
    'tmae_abi_hash': (I, []),
    'tmae_voxelize_workspace': (Z, [L, I, I, I, I]),
    'tmae_voxelize': (I, [P, I, L, I, F, F, F, F, F, F, I, I, I, P, P, P, P, P, P, Z, P]),
    'tmae_segment_csr_workspace': (Z, [L, L]),
    'tmae_segment_csr': (I, [P, L, L, P, P, P, Z, P]),
    'tmae_ingroup_rank_workspace': (Z, [L, L]),
    'tmae_ingroup_rank': (I, [P, L, L, P, P, Z, P]),
    'tmae_vfe_point_features': (I, [P, I, P, P, P, P, L, L, F, F, F, F, F, F, P, P, P]),
    'tmae_vfe_point_features_bf16x2': (I, [P, I, P, P, P, P, L, L, F, F, F, F, F, F, P, P, P, P]),
    'tmae_segment_max_fwd': (I, [P, I, L, L, I, P, P, P, P, P]),
    'tmae_segment_max_bwd': (I, [P, I, L, L, I, P, P, P, P]),
    'tmae_segment_max_bn_fwd': (I, [P, I, L, L, I, P, P, P, P, P, P, I, P, P, P]),
    'tmae_group_points': (I, [P, I, P, P, P, L, I, F, F, F, F, F, F, P, P, P]),
    'tmae_random_mask_workspace': (Z, [L, I]),
    'tmae_random_mask': (I, [P, P, L, I, D, P, P, P, P, Z, P]),
    'tmae_index_grid': (I, [P, L, I, I, I, P, P]),
    'tmae_window_bucket_workspace': (Z, [I, I, I, I, I, I]),
    'tmae_window_bucket': (I, [P, L, P, P, I, I, I, I, I, I, P, I, P, P, P, P, P, P, P, P, Z, P]),
    'tmae_win_attn_fwd': (I, [P, L, P, L, P, L, I, L, L, I, I, P, P, I, I, I, I, P, F, P, L, P, P, I, P]),
    'tmae_win_attn_zero_orphans': (I, [P, P, I, I, I, I, P, L, I, P, I, P, L, P, L, I, P]),
    'tmae_window_worklist_size': (Z, [I, I, I]),
    'tmae_window_worklist': (I, [P, P, I, I, I, I, P, P]),
    'tmae_win_attn_dtau': (I, [P, L, P, F, P, P, I, I, P]),
    'tmae_win_attn_num_blocks': (L, [I, I, I, I, I]),
    'tmae_win_attn_bwd': (I, [P, L, P, L, P, L, P, L, P, L, P, I, L, L, I, I, P, P, I, I, I, I, P, F,
                              P, L, P, L, P, L, P, P, I, P]),
    'tmae_add_pos_embed': (I, [P, I, L, I, P, I, I, I, P, P, P]),
    'tmae_spconv_down_outputs_workspace': (Z, [I, I, I]),
    'tmae_spconv_down_outputs': (I, [P, I, I, I, I, I, P, P, P, P, Z, P]),
    'tmae_spconv_neighbors': (I, [P, L, P, I, I, I, I, P, P]),
    'tmae_spconv_neighbors_t': (I, [P, L, P, I, I, I, I, P, P]),
    'tmae_spconv_fwd': (I, [P, L, L, I, P, L, P, I, P, L, P]),
    'tmae_spconv_bwd_data': (I, [P, L, L, I, P, L, P, I, P, L, P]),
    'tmae_dense_conv3x3': (I, [P, I, I, I, I, P, I, P, P]),
    'tmae_dense_conv3x3_dilated': (I, [P, I, I, I, I, P, I, I, P, P]),
    'tmae_dense_conv3x3_add': (I, [P, I, I, I, I, P, I, I, P, P, P]),
    'tmae_dense_conv3x3_sums_workspace': (Z, [I]),
    'tmae_dense_conv3x3_sums': (I, [P, I, I, I, I, P, I, P, I, P, P, P, Z, P]),
    'tmae_dense_conv3x3_wgrad_workspace': (Z, [I, I]),
    'tmae_dense_conv3x3_wgrad': (I, [P, P, I, I, I, I, I, I, P, P, Z, P]),
    'tmae_conv3x3_c64_workspace': (Z, []),
    'tmae_conv3x3_c64': (I, [P, L, I, I, I, P, I, I, P, L, P, Z, P]),
    'tmae_conv3x3_c64_wgrad_workspace': (Z, []),
    'tmae_conv3x3_c64_wgrad': (I, [P, L, P, L, I, I, I, P, P, Z, P]),
    'tmae_conv3x3_c64_narrow_fwd': (I, [P, L, I, I, I, P, P, I, P, P]),
    'tmae_conv3x3_c64_narrow_bwd_data_workspace': (Z, []),
    'tmae_conv3x3_c64_narrow_bwd_data': (I, [P, I, I, I, I, P, P, L, P, Z, P]),
    'tmae_conv3x3_c64_narrow_wgrad_workspace': (Z, [I]),
    'tmae_conv3x3_c64_narrow_wgrad': (I, [P, P, L, I, I, I, I, P, P, Z, P]),
    'tmae_spconv_gather': (I, [P, I, L, I, P, L, P, P]),
    'tmae_spconv_gather_t': (I, [P, I, L, I, P, L, P, P]),
    'tmae_sparse_to_dense': (I, [P, I, L, I, P, I, I, I, P, P]),
    'tmae_dense_gather': (I, [P, I, I, I, I, I, P, L, P, P]),
    'tmae_chamfer_fwd': (I, [P, P, P, L, I, I, P, P, P, P]),
    'tmae_chamfer_bwd': (I, [P, P, P, P, P, P, L, I, I, P, P]),
    'tmae_add_layernorm_fwd': (I, [P, P, I, L, I, P, P, F, P, P, P, P, P, P, P]),
    'tmae_layernorm_bwd_workspace': (Z, [L, I]),
    'tmae_layernorm_bwd': (I, [P, P, I, L, I, P, P, P, P, P, P, P, P, P, P, P, Z, P]),
    'tmae_bn_workspace': (Z, [L, I]),
    'tmae_bn_relu_fwd': (I, [P, I, L, I, P, P, F, I, P, P, P, P, P, Z, P]),
    'tmae_bn_relu_add_fwd': (I, [P, I, L, I, P, P, F, I, P, P, P, P, P, P, Z, P]),
    'tmae_bn_relu_bwd': (I, [P, P, I, L, I, P, P, P, P, I, P, P, P, P, Z, P]),
    'tmae_bn_relu_bwd2': (I, [P, P, P, I, L, I, P, P, P, P, I, P, P, P, P, Z, P]),
    'tmae_bn_relu_bwd_gathered': (I, [P, P, P, L, P, I, I, I, I, I, P, P, P, P, I, P, P, P, P, Z, P]),
    'tmae_bn_stats': (I, [P, I, L, I, D, F, P, P, P, P, Z, P]),
    'tmae_bn_apply': (I, [P, I, L, I, P, P, P, P, I, P, P]),
    'tmae_bn_bwd_sums': (I, [P, P, I, L, I, P, P, P, P, I, P, P, P, Z, P]),
    'tmae_bn_bwd_sums3': (I, [P, P, I, L, I, P, P, P, P, I, P, P, P, P, Z, P]),
    'tmae_deblock_bn_bwd_workspace': (Z, [L, I, I]),
    'tmae_deblock_bn_bwd': (I, [P, I, L, I, P, L, I, I, I, I, P, P, P, P, P, P, D, P, P, P, P, Z, P]),
    'tmae_bn_bwd_apply': (I, [P, P, I, L, I, P, P, P, P, I, P, P, D, P, P]),
    'tmae_deblock_scatter': (I, [P, I, P, I, I, I, I, I, P, P, P, P, P, I, I, P]),
    'tmae_deblock_scatter_multi': (I, [I, P, I, P, I, P, P, P, P, P, P, P, P, P, I, P]),
    'tmae_deblock_gather': (I, [P, I, I, I, P, L, I, I, I, I, P, P]),
    'tmae_column_sums_workspace': (Z, [L, I]),
    'tmae_column_sums': (I, [P, I, L, I, P, P, Z, P]),
    'tmae_deblock_bn_tail': (I, [P, P, P, P, P, P, P, P, I, P, P, P]),
    'tmae_centerhead_targets': (I, [P, I, I, I, P, I, I, I, I, F, F, F, F, F, I, D, I, P, P, P, P, P]),
    'tmae_focal_loss_workspace': (Z, [L]),
    'tmae_focal_loss_fwd': (I, [P, I, P, L, P, P, Z, P]),
    'tmae_focal_loss_bwd': (I, [P, I, P, L, P, P, P, P]),
    'tmae_boxes_pairwise': (I, [P, I, P, I, I, P, P]),
    'tmae_nms_bev_workspace': (Z, [I]),
    'tmae_nms_bev': (I, [P, I, F, P, P, P, Z, P]),
    'tmae_frame_prepare_workspace': (Z, [L]),
    'tmae_frame_prepare': (I, [P, I, L, P, P, F, I, I, F, F, F, F, F, F, F, I, P, P, P, Z, P]),
    'tmae_frame_prepare_boxes': (I, [P, I, L, P, P, F, I, I, F, F, F, F, F, F, F, I, P, I, P, P, P, Z, P]),
    'tmae_token_gemm': (I, [P, L, L, I, P, I, P, P, L, P]),
    'tmae_multi_cast_transpose': (I, [P, I, L, P]),
    'tmae_adam_step': (I, [P, P, L, F, F, F, F, F, P]),
    'tmae_bn_running_update': (I, [P, I, P, P, I, P]),
    'tmae_token_gemm_acc': (I, [P, L, L, I, P, I, P, P, L, P]),
    'tmae_token_gemm_res': (I, [P, L, L, I, P, I, P, P, P, L, P]),
    'tmae_token_gemm_dgelu': (I, [P, L, L, I, P, I, P, P, P, L, P]),
    'tmae_token_gemm_gelu': (I, [P, L, L, I, P, I, P, P, P, L, P]),
    'tmae_token_gemm_pos': (I, [P, L, L, I, P, I, P, P, P, L, P]),
    'tmae_window_cells': (I, [P, L, L, I, I, I, P, P, P]),
    'tmae_linear_wgrad_workspace': (Z, [L, I, I]),
    'tmae_linear_wgrad': (I, [P, L, P, L, L, I, I, P, P, P, Z, P]),
    'tmae_linear_wgrad_cells': (I, [P, L, P, L, L, I, I, P, I, P, P, P, P, P, Z, P]),
    'tmae_spconv_wgrad': (I, [P, L, P, L, P, L, I, I, P, P, Z, P]),
    'tmae_probe_copy': (I, [P, P, L, I, P]),
    'tmae_probe_mfma': (I, [I, P, P, P]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: build the HIP extension first (python t-mae_amd/build.py, or '
        f'__graft_entry__.build()).  tmae_amd has no CPU / eager fallback.')

def _strict(fn, name, nargs):
    """ctypes checks a cdecl call for TOO FEW arguments only; extra ones are passed on as 32-bit ints (a truncated pointer or
    stream handle).  Every entry point is therefore called through this arity check."""
    def call(*args):
        if len(args) != nargs:
            raise TypeError(f'{name} takes {nargs} arguments ({len(args)} given)')
        return fn(*args)
    call.__name__ = name
    call.raw = fn
    return call


lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args
    setattr(lib, _name, _strict(_fn, _name, len(_args)))

ABI_VERSION = 25            # = TMAE_ABI_VERSION of include/tmae_hip.h (hand-bumped with every change of the export list / a signature)
if lib.tmae_abi_version() != ABI_VERSION:
    raise ImportError(f'libtmae_hip.so ABI version {lib.tmae_abi_version()} != binding {ABI_VERSION}; rebuild with '
                      f't-mae_amd/build.py')

# signature fingerprint: the library carries the hash of the header it was compiled against, the table above yields the same
# canonical text (tmae_amd/_abi.py) -- any disagreement in a function's argument count, order or width ends the import here
from . import _abi  # noqa: E402
_CODE = {P: 'P', I: 'I', L: 'L', F: 'F', D: 'D', Z: 'Z'}
ABI_HASH = _abi.fnv1a31(_abi.canonical({n: (_CODE[r], [_CODE[a] for a in args]) for n, (r, args) in SIGNATURES.items()}))
if lib.tmae_abi_hash() != ABI_HASH:
    raise ImportError(f'libtmae_hip.so was built from a header whose prototypes differ from the ctypes table of {__file__} '
                      f'(fingerprint {lib.tmae_abi_hash()} != {ABI_HASH}): rebuild with t-mae_amd/build.py, and if that does '
                      f'not help compare SIGNATURES with include/tmae_hip.h (tests/test_abi_and_host.py names the function)')


class TmaeHipError(RuntimeError):
    pass


_ERR = {-1: 'bad argument', -2: 'workspace too small', -3: 'unsupported dtype'}


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f'hipError_t {rc}' if rc > 0 else f'error {rc}')
        raise TmaeHipError(f'{what}: {msg}')


class PinnedStager:
    """Host -> device upload of a small int64 table through pinned staging buffers, asynchronously.

    `upload(values, device_tensor)` writes `values` into the next of `depth` pinned buffers and enqueues the copy on
    the current stream.  Every buffer carries the event recorded behind its last copy and is rewritten only after that
    event has completed: a host running several steps ahead of the GPU (nothing else orders the two once the step's
    host syncs are gone) would otherwise overwrite a table whose copy is still queued, and the launch behind that copy
    would read pointers of a LATER step.  The wait is free in the steady state (the buffer's copy is `depth` uploads old).
    """

    def __init__(self, shape, depth=2):
        self._bufs = [torch.empty(shape, dtype=torch.int64).pin_memory() for _ in range(depth)]
        self._events = [None] * depth
        self._next = 0

    def upload(self, values, device_tensor):
        i = self._next
        self._next = (i + 1) % len(self._bufs)
        ev = self._events[i]
        if ev is not None:
            ev.synchronize()                       # the previous copy out of this buffer has run
        self._bufs[i].numpy()[...] = values
        device_tensor.copy_(self._bufs[i], non_blocking=True)
        if ev is None:
            ev = self._events[i] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device_tensor.device))
        return device_tensor
