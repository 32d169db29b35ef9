"""ctypes binding of liba3d.so (include/a3d.h).  Fails loudly when the library is missing or lacks a symbol."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_void_p, POINTER

# A3D_LIB points the binding at another build of the same library (tools/bench_layers.py A/B runs); there is still no
# fallback: a missing file or symbol raises.
LIB_PATH = os.environ.get('A3D_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'liba3d.so')


class A3dError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """struct a3d_conv_desc"""
    _fields_ = [(n, c_int32) for n in ('n', 'h', 'w', 'c', 'k', 'r', 's', 'stride', 'pad_t', 'pad_l', 'ho', 'wo',
                                       'ldx', 'ldy', 'precision', 'storage', 'hints')]


class SecondOutput(ctypes.Structure):
    """struct a3d_second_output"""
    _fields_ = [('ptr', c_void_p), ('ld', c_int32), ('step', c_int32), ('offset', c_int32), ('bf16', c_int32), ('cols', c_int32)]


class ExampleView(ctypes.Structure):
    """struct a3d_example_view"""
    _fields_ = [('image_height', c_int64), ('image_width', c_int64), ('image_channels', c_int64),
                ('depth_height', c_int64), ('depth_width', c_int64), ('depth_channels', c_int64),
                ('image', c_void_p), ('image_bytes', c_size_t), ('depth', c_void_p), ('depth_bytes', c_size_t)]


class TimingRecord(ctypes.Structure):
    """struct a3d_timing_record"""
    _fields_ = [(n, c_int32) for n in ('mode', 'bm', 'bn', 'waves_m', 'nwaves', 'bk', 'avec', 'bvec', 'prec', 'lds_dma', 'splitk', 'm', 'n', 'k')] + \
               [('ms', c_float), ('flops', ctypes.c_double)]


_P = c_void_p
_D = POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/a3d.h declares
SIGNATURES = {
    'a3d_version': (c_char_p, []),
    'a3d_last_error': (c_int, [c_char_p, c_size_t]),
    'a3d_conv2d_fwd_ws_bytes': (c_size_t, [_D]),
    'a3d_conv2d_fwd': (c_int, [_D, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    'a3d_conv2d_bwd_data_ws_bytes': (c_size_t, [_D]),
    'a3d_conv2d_bwd_data': (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    'a3d_conv2d_bwd_filter_ws_bytes': (c_size_t, [_D]),
    'a3d_conv2d_bwd_filter': (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    'a3d_conv2d_fwd_prepared_filter_bytes': (c_size_t, [_D]),
    'a3d_conv2d_fwd_prepare_filter': (c_int, [_D, _P, _P, c_size_t, _P]),
    'a3d_conv2d_bwd_filter_pooled_ws_bytes': (c_size_t, [_D]),
    'a3d_conv2d_bwd_filter_pooled': (c_int, [_D, _P, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    'a3d_conv2d_bwd_both_supported': (c_int, [_D]),
    'a3d_conv2d_bwd_both_ws_bytes': (c_size_t, [_D]),
    'a3d_conv2d_bwd_both': (c_int, [_D, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    'a3d_dense_fwd_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'a3d_dense_fwd': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, c_size_t, _P]),
    'a3d_dense_bwd_data_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'a3d_dense_bwd_data': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P, c_size_t, _P]),
    'a3d_dense_bwd_filter_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'a3d_dense_bwd_filter': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'a3d_maxpool2x2_fwd': (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    'a3d_maxpool2x2_bwd': (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P]),
    'a3d_resize_bilinear_tf1': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P]),
    'a3d_resize_bilinear_tf1_pair': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P]),
    'a3d_resize_bilinear_tf1_ex': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int,
                                           _P, _P]),
    'a3d_extract_patches': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, _P]),
    'a3d_silog_loss_fwd': (c_int, [c_int, c_int, _P, _P, _P, _P, _P]),
    'a3d_silog_loss_bwd': (c_int, [c_int, c_int, _P, _P, _P, _P, _P]),
    'a3d_dropout_keep_mask': (c_int, [c_size_t, ctypes.c_uint64, ctypes.c_uint64, c_float, _P, _P]),
    'a3d_adam_apply_tf1': (c_int, [c_size_t, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                                   c_float, _P]),
    'a3d_adam_apply_tf1_flag': (c_int, [c_size_t, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                                        c_float, _P, _P]),
    'a3d_dense_fwd_ex': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, _P, c_size_t, _P]),
    'a3d_dense_bwd_data_ex': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, c_int, c_int, _P, c_size_t, _P]),
    'a3d_dense_fwd_ex2': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int, POINTER(SecondOutput), _P,
                                  c_size_t, _P]),
    'a3d_dense_bwd_data_ex2': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, c_int, c_int, POINTER(SecondOutput), _P,
                                       c_size_t, _P]),
    'a3d_conv2d_fwd_ex2': (c_int, [_D, _P, _P, _P, _P, c_int, POINTER(SecondOutput), _P, c_size_t, _P]),
    'a3d_silog_loss_bwd_ex': (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    'a3d_cast_bf16': (c_int, [c_size_t, _P, _P, c_int, _P]),
    'a3d_cast_rows': (c_int, [c_size_t, c_int, _P, c_int, c_int, _P, c_int, c_int, _P]),
    'a3d_pad_channels_bf16': (c_int, [c_size_t, c_int, _P, c_int, _P, _P]),
    'a3d_stream_create': (c_int, [c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'a3d_stream_destroy': (c_int, [_P]),
    'a3d_maxpool2x2_fwd_bf16': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P]),
    'a3d_maxpool2x2_bwd_bf16': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    'a3d_copy_channel_bf16': (c_int, [c_size_t, _P, c_int, c_int, _P, c_int, c_int, _P]),
    'a3d_maxpool2x2_bwd_idx_bf16': (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P]),
    'a3d_maxpool2x2_bwd_idx_bf16s': (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P]),
    'a3d_dense_bwd_filter_adam_tf1': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float,
                                              c_float, c_float, c_float, _P]),
    'a3d_dense_bwd_filter_adam_tf1_ex': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float,
                                                 c_float, c_float, c_float, c_int, _P]),
    'a3d_superpixel_mean': (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    'a3d_superpixel_hist': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P]),
    'a3d_pair_similarity': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P, c_float, _P, _P, _P]),
    'a3d_crf_loss': (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_float, _P, _P, _P, _P]),
    'a3d_sgd_apply': (c_int, [c_size_t, _P, _P, c_float, _P]),
    'a3d_conv2d_pool_fwd': (c_int, [_D, _P, _P, _P, _P, c_int, _P, c_int, _P, c_size_t, _P]),
    'a3d_maxpool2x2_bwd_idx': (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P]),
    'a3d_copy_channel': (c_int, [c_size_t, _P, c_int, c_int, _P, c_int, c_int, _P]),
    'a3d_comm_standin': (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_float, _P]),
    'a3d_timing_enable': (c_int, [c_int]),
    'a3d_timing_collect': (c_int, [POINTER(TimingRecord), c_int]),
    'a3d_timing_select': (c_int, [POINTER(TimingRecord)]),
    'a3d_crc32c': (c_uint32, [_P, c_size_t]),
    'a3d_masked_crc32c': (c_uint32, [_P, c_size_t]),
    'a3d_tfrecord_next': (c_int, [_P, c_size_t, c_int, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t)]),
    'a3d_example_parse': (c_int, [_P, c_size_t, POINTER(ExampleView)]),
    'a3d_decode_raw_plus_half': (c_int, [_P, c_size_t, _P]),
    'a3d_record_decode': (c_int, [_P, c_size_t, c_int, _P, c_size_t, _P, c_size_t, POINTER(ExampleView)]),
    'a3d_record_decode_u8': (c_int, [_P, c_size_t, c_int, _P, _P, c_size_t, _P, _P, c_size_t, POINTER(ExampleView),
                                     POINTER(c_int)]),
    'a3d_records_decode': (c_int, [POINTER(c_void_p), POINTER(c_size_t), c_int, c_int, POINTER(c_int64), _P, _P, _P, _P,
                                   POINTER(c_int32), c_int, POINTER(c_int32)]),
    'a3d_h2d_gather': (c_int, [_P, _P, POINTER(c_int32), c_int, c_int, c_size_t, _P]),
    'a3d_sizeof_conv_desc': (c_size_t, []),
    'a3d_example_write': (c_int64, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_size_t]),
}

_lib = None


def load():
    """Load liba3d.so once and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise A3dError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       f'or `make -C ann3depth_amd/csrc` (there is no fallback path)')
    # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that liba3d.so binds to
    # the runtime copy torch initialises: two HIP runtimes in one process cannot both own the device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise A3dError(f'liba3d.so lacks symbol {name}: stale build?') from e
        fn.restype = res
        fn.argtypes = args
    if lib.a3d_sizeof_conv_desc() != ctypes.sizeof(ConvDesc):
        raise A3dError(f'liba3d.so reads a {lib.a3d_sizeof_conv_desc()}-byte a3d_conv_desc, this binding passes '
                       f'{ctypes.sizeof(ConvDesc)} bytes: stale build?')
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().a3d_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def check(rc, what):
    if rc != 0:
        raise A3dError(f'{what} failed ({rc}): {last_error()}')
