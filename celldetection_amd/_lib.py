"""ctypes binding of libcpn_hip.so (the C ABI declared in include/cpn_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or does not export the expected symbols,
importing the binding raises, and every op raises ``RuntimeError`` on failure with ``cpn_last_error()``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CPN_HIP_LIB') or os.path.join(HERE, 'libcpn_hip.so')  # env: kernel A/B tuning only

ABI_VERSION = 14
PRECISION_BF16, PRECISION_F32, PRECISION_FP8 = 0, 1, 2
E_INVALID, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3

OP_INPUT, OP_CONV, OP_MAXPOOL, OP_BILINEAR, OP_CONV_DEFERRED, OP_INPUT_STEM, OP_STEM7, OP_CONV_PAIR, OP_CONV_BRIDGE, OP_ACT = \
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH_SCALED = 0, 1, 2, 3
ACT_LEAKY_RELU, ACT_SILU, ACT_GELU, ACT_ELU, ACT_TANH, ACT_HARDSWISH, ACT_MISH, ACT_SELU, ACT_SOFTPLUS = 4, 5, 6, 7, 8, 9, 10, 11, 12
SUBPIXEL_NONE, SUBPIXEL_HEAD, SUBPIXEL_PHASE, SUBPIXEL_LATERAL, SUBPIXEL_SCATTER = 0, 1, 2, 3, 4
SUBPIXEL_BL_HEAD, SUBPIXEL_BL_PHASE, SUBPIXEL_BL_FRAME = 5, 6, 7
OUT_SCORES, OUT_LOCATIONS, OUT_FOURIER, OUT_REFINEMENT, OUT_UNCERTAINTY = 0, 1, 2, 3, 4
NUM_OUTPUTS = 5


class TensorDesc(Structure):
    _fields_ = [('channels', c_int32), ('down', c_int32), ('scale', c_float)]


class OpDesc(Structure):
    _fields_ = [('op', c_int32), ('src0', c_int32), ('src1', c_int32), ('res', c_int32), ('dst', c_int32),
                ('up0', c_int32), ('up1', c_int32), ('res_up', c_int32), ('c0_used', c_int32),
                ('kh', c_int32), ('kw', c_int32), ('stride', c_int32), ('pad', c_int32),
                ('bundles', c_int32), ('cin_b', c_int32), ('cout_b', c_int32),
                ('weight_offset', c_int64), ('bias_offset', c_int64),
                ('act', c_int32), ('act_scale', c_float), ('out_index', c_int32), ('cout_real', c_int32),
                ('dst_coff', c_int32), ('in_channels', c_int32),
                ('fuse_weight_offset', c_int64), ('fuse_bias_offset', c_int64), ('fuse_cout', c_int32),
                ('fuse_act', c_int32), ('fuse_act_scale', c_float), ('mult_offset', c_int32), ('subpixel', c_int32), ('alt', c_int32)]


# every symbol include/cpn_hip.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ('cpn_last_error', c_char_p, []),
    ('cpn_abi_version', ctypes.c_int, []),
    ('cpn_plan_create', ctypes.c_int, [POINTER(c_void_p), POINTER(TensorDesc), c_int32, POINTER(OpDesc), c_int32,
                                       c_void_p, c_size_t, c_void_p, c_size_t, c_int32]),
    ('cpn_plan_destroy', None, [c_void_p]),
    ('cpn_plan_workspace_bytes', c_int64, [c_void_p, c_int32, c_int32, c_int32]),
    ('cpn_plan_output_dims', ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    ('cpn_plan_max_tensor_elements', c_int64, [c_void_p, c_int32, c_int32]),
    ('cpn_plan_executed_flops', c_double, [c_void_p, c_int32, c_int32, c_int32]),
    ('cpn_plan_run', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                                    POINTER(c_void_p), c_void_p, c_void_p]),
    ('cpn_plan_run_stats', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                                          POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    ('cpn_plan_num_ops', ctypes.c_int, [c_void_p]),
    ('cpn_plan_run_timed', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                                          POINTER(c_void_p), c_void_p, c_void_p, POINTER(c_float), POINTER(c_double)]),
    ('cpn_conv2d', ctypes.c_int, [POINTER(OpDesc), c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                  c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_conv2d_fp8', ctypes.c_int, [POINTER(OpDesc), c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                      c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                      c_void_p]),
    ('cpn_convert_input_stem', ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                              c_void_p]),
    ('cpn_stem7', ctypes.c_int, [POINTER(OpDesc), c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                 c_float, c_void_p]),
    ('cpn_conv_pair', ctypes.c_int, [POINTER(OpDesc), c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p]),
    ('cpn_conv_bridge', ctypes.c_int, [POINTER(OpDesc), c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                       c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_maxpool2d', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_void_p]),
    ('cpn_resize_bilinear', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_void_p]),
    ('cpn_convert_input', ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p]),
    ('cpn_compact_workspace_bytes', c_int64, [c_int32, c_int32, c_int32]),
    ('cpn_compact', ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    ('cpn_decode', ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_decode_gathered', ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                           c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_sparse_heads', ctypes.c_int, [POINTER(OpDesc), POINTER(OpDesc), c_void_p, c_int32, c_int32, c_int32, c_int32,
                                        c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ('cpn_plan_tensor_info', ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_int64),
                                            POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    ('cpn_fouriers2contours', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                             c_void_p, c_void_p]),
    ('cpn_local_refinement', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32,
                                            c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_class_scores', ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    ('cpn_certainty_mask', ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p,
                                          c_void_p]),
    ('cpn_gather_channels', ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                           c_void_p]),
    ('cpn_nms_workspace_bytes', c_int64, [c_int64, c_int64, c_int32]),
    ('cpn_nms', ctypes.c_int, [c_void_p, c_void_p, c_int64, POINTER(c_int64), c_void_p, c_int32, c_float, c_void_p,
                               c_void_p, c_void_p, c_int64, c_void_p]),
    ('cpn_box_votes', ctypes.c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    ('cpn_border_keep', ctypes.c_int, [c_void_p, c_int64, c_int32, c_float, c_float, c_float, c_float, c_float,
                                       c_int32, c_void_p, c_void_p]),
    ('cpn_resize_bilinear_f32', ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                               c_void_p]),
    ('cpn_resize_f32', ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    ('cpn_border_keep_batched', ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                               c_float, c_float, c_float, c_void_p, c_void_p]),
    ('cpn_histogram', ctypes.c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    ('cpn_rescale_to_uint8', ctypes.c_int, [c_void_p, c_int32, c_int64, c_double, c_double, c_void_p, c_void_p]),
    ('cpn_debug_clock_probe', ctypes.c_int, [POINTER(ctypes.c_uint64), c_int32]),
    ('cpn_window_any', ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    ('cpn_labels_prepare', ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p, c_void_p]),
    ('cpn_labels_bin', ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    ('cpn_labels_cell_bounds', ctypes.c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    ('cpn_labels_round', ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int32), c_int32, c_double,
                                        c_void_p]),
    ('cpn_nms_binned_workspace_bytes', c_int64, [c_int64, c_int64]),
    ('cpn_nms_binned', ctypes.c_int, [c_void_p, c_void_p, c_int64, c_float, c_int64, c_void_p, c_void_p,
                                      POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), c_void_p, c_int64,
                                      c_void_p]),
]

EXPORTED_SYMBOLS = tuple(s[0] for s in _SIGNATURES)

_lib = None


def load():
    """Loads libcpn_hip.so; raises RuntimeError (never falls back) when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: build the HIP extension first '
                           f'(python -m celldetection_amd.build); there is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in _SIGNATURES:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f'{LIB_PATH} does not export {name}; rebuild it.') from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.cpn_abi_version() != ABI_VERSION:
        raise RuntimeError(f'libcpn_hip.so ABI version {lib.cpn_abi_version()} != expected {ABI_VERSION}; rebuild it.')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().cpn_last_error()
        raise RuntimeError(f'libcpn_hip {what} failed (code {rc}): {msg.decode() if msg else "?"}')


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return c_void_p(0 if t is None else t.data_ptr())
