"""ctypes binding of libpvhip.so (include/pvhip.h) and the DeviceTensor handle that flows through
the scheduler's ``inputs`` dicts instead of ndarrays.

No PyTorch, no numpy compute: this module only moves bytes and calls the C ABI.  If the shared
library is missing or no GPU is visible, every entry point raises -- there is no CPU fallback.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpvhip.so')
DIAG_LIB_PATH = os.path.join(_HERE, 'libpvhip_diag.so')      # diagnostic build (make diag): include/pvhip_diag.h
# PVHIP_LIBRARY=<path>: load another build of the library instead (A/B runs against a previous commit's build; the test file of the
# diagnostic build's extra kernels, tests/diag_variants.py, runs with the diagnostic library).  Never a CPU substitute: same C ABI.
if os.environ.get('PVHIP_LIBRARY'):
    LIB_PATH = os.path.abspath(os.environ['PVHIP_LIBRARY'])

MAX_RANK = 6
MAX_CONCAT = 16
UNIQUE_ID_BYTES = 128

_c = ctypes
_i64p = _c.POINTER(_c.c_int64)
_fp = _c.c_void_p  # device pointers travel as plain integers

# name -> (restype, argtypes); mirrors include/pvhip.h one to one (tests check the two agree)
SIGNATURES = {
    'pvhip_abi_version': (_c.c_int, []),
    'pvhip_last_error': (_c.c_char_p, []),
    'pvhip_device_count': (_c.c_int, [_c.POINTER(_c.c_int)]),
    'pvhip_init': (_c.c_int, [_c.c_int]),
    'pvhip_settings_reload': (_c.c_int, []),
    'pvhip_shutdown': (_c.c_int, []),
    'pvhip_device_name': (_c.c_int, [_c.c_char_p, _c.c_size_t]),
    'pvhip_malloc': (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_size_t]),
    'pvhip_free': (_c.c_int, [_c.c_void_p]),
    'pvhip_pool_release': (_c.c_int, []),
    'pvhip_pool_stats': (_c.c_int, [_c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
    'pvhip_memcpy_h2d': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t]),
    'pvhip_memcpy_d2h': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t]),
    'pvhip_host_alloc': (_c.c_void_p, [_c.c_size_t]),
    'pvhip_host_free': (_c.c_int, [_c.c_void_p]),
    'pvhip_memcpy_d2d': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t]),
    'pvhip_memset': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_size_t]),
    'pvhip_sync': (_c.c_int, []),
    'pvhip_pool_epoch_begin': (_c.c_int, [_c.POINTER(_c.c_int)]),
    'pvhip_pool_epoch_dispatched': (_c.c_int, []),
    'pvhip_pool_epoch_end': (_c.c_int, [_c.c_int]),
    'pvhip_stream_select': (_c.c_int, [_c.c_int]),
    'pvhip_stream_wait_event': (_c.c_int, [_c.c_void_p]),
    'pvhip_event_create_untimed': (_c.c_int, [_c.POINTER(_c.c_void_p)]),
    'pvhip_event_create': (_c.c_int, [_c.POINTER(_c.c_void_p)]),
    'pvhip_event_destroy': (_c.c_int, [_c.c_void_p]),
    'pvhip_event_record': (_c.c_int, [_c.c_void_p]),
    'pvhip_event_sync': (_c.c_int, [_c.c_void_p]),
    'pvhip_event_elapsed_ms': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_float)]),
    'pvhip_graph_begin_capture': (_c.c_int, []),
    'pvhip_graph_capture_status': (_c.c_int, [_c.POINTER(_c.c_int)]),
    'pvhip_graph_end_capture': (_c.c_int, [_c.POINTER(_c.c_void_p)]),
    'pvhip_graph_launch': (_c.c_int, [_c.c_void_p]),
    'pvhip_graph_destroy': (_c.c_int, [_c.c_void_p]),
    'pvhip_relu_f32': (_c.c_int, [_fp, _fp, _c.c_size_t]),
    'pvhip_clamp_f32': (_c.c_int, [_fp, _fp, _c.c_size_t, _c.c_float, _c.c_float]),
    'pvhip_sigmoid_f32': (_c.c_int, [_fp, _fp, _c.c_size_t]),
    'pvhip_add_f32': (_c.c_int, [_fp, _fp, _fp, _c.c_int, _i64p, _i64p, _i64p]),
    'pvhip_mul_f32': (_c.c_int, [_fp, _fp, _fp, _c.c_int, _i64p, _i64p, _i64p]),
    'pvhip_maxpool2d_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 14),
    'pvhip_avgpool2d_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 10),
    'pvhip_softmax_rows_f32': (_c.c_int, [_fp, _fp, _c.c_int, _c.c_int]),
    'pvhip_lrn_f32': (_c.c_int, [_fp, _fp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float]),
    'pvhip_lrn_maxpool_supported': (_c.c_int, [_c.c_int] * 5 + [_c.c_float, _c.c_float] + [_c.c_int] * 10),
    'pvhip_lrn_maxpool_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 5 + [_c.c_float] * 3 + [_c.c_int] * 10),
    'pvhip_maxpool_lrn_supported': (_c.c_int, [_c.c_int] * 15 + [_c.c_float, _c.c_float]),
    'pvhip_maxpool_lrn_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 15 + [_c.c_float] * 3),
    'pvhip_maxpool_lrn_conv1x1_supported': (_c.c_int, [_c.c_int] * 15 + [_c.c_float, _c.c_float, _c.c_int]),
    'pvhip_maxpool_lrn_conv1x1_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 15 + [_c.c_float] * 3 + [_c.c_int, _fp, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_concat_f32': (_c.c_int, [_c.c_int, _c.POINTER(_c.c_void_p), _i64p, _fp, _c.c_int64]),
    'pvhip_pad2d_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 8 + [_fp]),
    'pvhip_transpose_f32': (_c.c_int, [_fp, _fp, _c.c_int, _i64p, _i64p]),
    'pvhip_matmul_f32': (_c.c_int, [_fp, _fp, _fp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'pvhip_conv2d_pack_elems': (_c.c_size_t, [_c.c_int] * 4),
    'pvhip_conv2d_pack_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 6),
    'pvhip_conv2d_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 13 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_f16_pack_elems': (_c.c_size_t, [_c.c_int] * 4),
    'pvhip_conv2d_f16_pack': (_c.c_int, [_fp, _fp] + [_c.c_int] * 6),
    'pvhip_conv2d_f16': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 13 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_f16_dma_supported': (_c.c_int, [_c.c_int] * 3),
    'pvhip_conv2d_f16_span_supported': (_c.c_int, [_c.c_int] * 11),
    'pvhip_conv2d_f16_span_pack_elems': (_c.c_size_t, [_c.c_int] * 4),
    'pvhip_conv2d_f16_span_pack': (_c.c_int, [_fp, _fp] + [_c.c_int] * 4),
    'pvhip_conv2d_f16_span': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 13 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_f16_dma': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 13 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_matmul_f16': (_c.c_int, [_fp, _fp, _fp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'pvhip_c8_f16_elems': (_c.c_size_t, [_c.c_int] * 4),
    'pvhip_c8_f16_from_f32': (_c.c_int, [_fp, _c.c_void_p] + [_c.c_int] * 4),
    'pvhip_c8_f16_to_f32': (_c.c_int, [_c.c_void_p, _fp] + [_c.c_int] * 4),
    'pvhip_conv2d_f16_c8_supported': (_c.c_int, [_c.c_int] * 11),
    'pvhip_conv2d_f16_c8_pack_elems': (_c.c_size_t, [_c.c_int] * 4),
    'pvhip_conv2d_f16_c8_pack': (_c.c_int, [_fp, _fp] + [_c.c_int] * 4),
    'pvhip_conv2d_f16_c8_multi_supported': (_c.c_int, [_c.c_int] * 7),
    'pvhip_conv2d_f16_c8_multi': (_c.c_int, [_c.c_void_p, _fp] + [_c.c_int] * 7 + [_fp, _c.c_int, _c.c_float, _c.c_float, _c.c_int, _c.c_void_p]),
    'pvhip_maxpool3x3_c8': (_c.c_int, [_c.c_void_p, _c.c_void_p] + [_c.c_int] * 12),
    'pvhip_lrn_maxpool3x3_c8_supported': (_c.c_int, [_c.c_int] * 9),
    'pvhip_lrn_maxpool3x3_c8': (_c.c_int, [_c.c_void_p, _c.c_void_p] + [_c.c_int] * 5 + [_c.c_float] * 3 + [_c.c_int] * 8),
    'pvhip_avgpool_c8': (_c.c_int, [_c.c_void_p, _fp] + [_c.c_int] * 10),
    'pvhip_maxpool3x3_lrn_c8': (_c.c_int, [_c.c_void_p, _c.c_void_p] + [_c.c_int] * 13 + [_c.c_float] * 3),
    'pvhip_maxpool3x3_lrn_conv1x1_c8_supported': (_c.c_int, [_c.c_int] * 3),
    'pvhip_maxpool3x3_lrn_conv1x1_c8': (_c.c_int, [_c.c_void_p, _fp, _c.c_void_p] + [_c.c_int] * 13 + [_c.c_float] * 3 + [_c.c_int, _fp, _c.c_int]),
    'pvhip_conv2d_f16_stem_supported': (_c.c_int, [_c.c_int] * 12),
    'pvhip_conv2d_f16_stem_pack_elems': (_c.c_size_t, [_c.c_int]),
    'pvhip_conv2d_f16_stem_pack': (_c.c_int, [_fp, _fp, _c.c_int]),
    'pvhip_conv2d_f16_stem': (_c.c_int, [_fp, _fp, _c.c_void_p] + [_c.c_int] * 6 + [_fp, _c.c_int]),
    'pvhip_conv2d_f16_stem_direct_supported': (_c.c_int, [_c.c_int] * 12),
    'pvhip_conv2d_f16_stem_direct_pack': (_c.c_int, [_fp, _fp, _c.c_int]),
    'pvhip_conv2d_f16_stem_direct': (_c.c_int, [_fp, _fp, _c.c_void_p] + [_c.c_int] * 6 + [_fp, _fp, _c.c_int]),
    'pvhip_conv2d_f16_dma_c8': (_c.c_int, [_fp, _fp, _c.c_void_p] + [_c.c_int] * 13 + [_fp, _c.c_int]),
    'pvhip_conv2d_f16_c8': (_c.c_int, [_c.c_void_p, _fp, _fp] + [_c.c_int] * 7 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_kernel_kind': (_c.c_int, [_c.c_int] * 13),
    'pvhip_conv2d_stem_f32_supported': (_c.c_int, [_c.c_int] * 12),
    'pvhip_conv2d_stem_f32_pack_elems': (_c.c_size_t, [_c.c_int]),
    'pvhip_conv2d_stem_f32_pack': (_c.c_int, [_fp, _fp, _c.c_int]),
    'pvhip_conv2d_stem_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 6 + [_fp, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_stem_direct_supported': (_c.c_int, [_c.c_int] * 12),
    'pvhip_conv2d_stem_wino_supported': (_c.c_int, [_c.c_int] * 12),
    'pvhip_conv2d_stem_wino_pack_elems': (_c.c_long, []),
    'pvhip_conv2d_stem_wino_pack': (_c.c_int, [_fp, _fp, _c.c_int]),
    'pvhip_conv2d_stem_wino_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 6 + [_fp, _fp, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_stem_direct_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 6 + [_fp, _fp, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_pooled_supported': (_c.c_int, [_c.c_int] * 5),
    'pvhip_conv2d_pooled_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 5 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_pooled_f16': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 5 + [_fp, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_conv2d_multi_supported': (_c.c_int, [_c.c_int] * 8),
    'pvhip_conv2d_multi_f32': (_c.c_int, [_fp, _fp] + [_c.c_int] * 12 + [_fp, _c.c_int, _c.c_float, _c.c_float, _c.c_int, _c.c_void_p]),
    'pvhip_conv2d_multi_f16_dma': (_c.c_int, [_fp, _fp] + [_c.c_int] * 12 + [_fp, _c.c_int, _c.c_float, _c.c_float, _c.c_int, _c.c_void_p]),
    'pvhip_dwconv2d_f32': (_c.c_int, [_fp, _fp, _fp] + [_c.c_int] * 12 + [_fp, _c.c_int, _c.c_float, _c.c_float]),
    'pvhip_detection_output_f32': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_float, _c.c_float, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'pvhip_comm_unique_id': (_c.c_int, [_c.c_void_p]),
    'pvhip_comm_init': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    'pvhip_comm_allgather_f32': (_c.c_int, [_fp, _fp, _c.c_size_t]),
    'pvhip_comm_ranks': (_c.c_int, [_c.POINTER(_c.c_int)]),
    'pvhip_comm_destroy': (_c.c_int, []),
}

# entry points whose return value is not a status code
_NOT_STATUS = {'pvhip_host_alloc', 'pvhip_conv2d_stem_wino_supported', 'pvhip_conv2d_stem_wino_pack_elems', 'pvhip_maxpool3x3_lrn_conv1x1_c8_supported', 'pvhip_conv2d_f16_stem_direct_supported', 'pvhip_maxpool_lrn_conv1x1_supported', 'pvhip_conv2d_stem_direct_supported', 'pvhip_conv2d_stem_f32_supported', 'pvhip_conv2d_stem_f32_pack_elems', 'pvhip_conv2d_f16_stem_supported', 'pvhip_conv2d_f16_stem_pack_elems', 'pvhip_lrn_maxpool3x3_c8_supported', 'pvhip_conv2d_f16_c8_multi_supported', 'pvhip_c8_f16_elems', 'pvhip_conv2d_f16_c8_supported', 'pvhip_conv2d_f16_c8_pack_elems', 'pvhip_conv2d_f16_pack_elems', 'pvhip_conv2d_f16_dma_supported', 'pvhip_conv2d_f16_span_supported', 'pvhip_conv2d_f16_span_pack_elems', 'pvhip_conv2d_kernel_kind', 'pvhip_abi_version', 'pvhip_last_error', 'pvhip_conv2d_pack_elems', 'pvhip_lrn_maxpool_supported', 'pvhip_maxpool_lrn_supported',
               'pvhip_conv2d_multi_supported', 'pvhip_conv2d_pooled_supported'}


MAX_CONV_DESTS = 6


class ConvDest(_c.Structure):
    """pvhip_conv_dest of include/pvhip.h."""
    _fields_ = [('y', _c.c_void_p), ('k', _c.c_int), ('channel_offset', _c.c_int), ('channels_total', _c.c_int), ('layout', _c.c_int)]


class PvhipError(RuntimeError):
    """A libpvhip call returned a negative status."""


_lib = None
_initialised_device = None


def load_library():
    """dlopen libpvhip.so and declare every prototype.  Loud failure if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PvhipError('HIP extension {} is not built -- run `python -c "import __graft_entry__ as g; g.build()"` '
                         'or `make -C pyopenvino_amd/csrc`; there is no CPU fallback'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Invoke a status-returning entry point; raise PvhipError with the library's message on failure."""
    lib = load_library()
    rc = getattr(lib, name)(*args)
    if name not in _NOT_STATUS and rc != 0:
        raise PvhipError('{} failed ({}): {}'.format(name, rc, lib.pvhip_last_error().decode(errors='replace')))
    return rc


def device_count() -> int:
    n = _c.c_int(0)
    lib = load_library()
    rc = lib.pvhip_device_count(_c.byref(n))
    return n.value if rc == 0 else 0


def init(device: int = None) -> int:
    """Bind this process to one GPU (default: LOCAL_RANK or 0) and create the compute stream."""
    global _initialised_device
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
        if device_count() == 1:
            device = 0
    if _initialised_device is not None:
        if _initialised_device != device:
            raise PvhipError('already bound to GPU {}'.format(_initialised_device))
        return device
    call('pvhip_init', int(device))
    _initialised_device = device
    return device


settings_serial = 0      # bumped by every reload: host-side plans that bake kernel choices in (a captured hipGraph) key on it


conv_prepad = os.environ.get('PVHIP_CONV_PREPAD', '1') != '0'      # Convolution plugin: pad the input of a c-major layer in a pass of its own
conv_f16_dma = os.environ.get('PVHIP_CONV_F16_DMA', '1') != '0'    # Convolution plugin, FP16 IRs: the f16 form of the LDS-DMA kernel where it applies
conv_f16_span = int(os.environ.get('PVHIP_CONV_F16_SPAN', '1') or 0)  # ... and the span kernel before it: 1 = 3x3 / 5x5 layers, 2 = 1x1 too (slower there), 0 = never
conv_stem_direct = os.environ.get('PVHIP_CONV_STEM_DIRECT', '1') != '0'   # Convolution plugin: the row-span kernel of a 7x7 / 2 first convolution reads the image itself (no padding pass)
conv_f16_stem = os.environ.get('PVHIP_CONV_F16_STEM', '1') != '0'  # ... and the row-span kernel for a 7x7 / 2 first convolution over three channels with a blocked output
def _env_level(name, default):
    try:
        return int(os.environ.get(name, default) or 0)
    except ValueError:
        return int(default)


# ... and fp16 tensors blocked by eight channels: 0 = never, 1 = between a 1x1 convolution and the 3x3 / 5x5 behind it, 2 = whole modules and
# the stem (Executable_Network.plan_fusion / plan_c8_modules read THIS value, the plugins too: one source for both sides)
conv_f16_c8 = _env_level('PVHIP_CONV_F16_C8', '2')
fuse_poolconv = _env_level('PVHIP_FUSE_POOLCONV', '2')
fuse_stem_conv = _env_level('PVHIP_FUSE_STEM_CONV', '1')          # the 1x1 convolution behind MaxPool + LRN in the same launch (0 = two launches)             # MaxPool + pool_proj as one launch: 0 = never (what the plan reads; the library parses its own copy)


def reload_settings():
    """Make libpvhip read the PVHIP_* environment variables again (it parses them once, at pvhip_init or at the first
    query that needs them; no device needed)."""
    global settings_serial, conv_prepad, conv_f16_dma, conv_f16_span, conv_f16_c8, conv_f16_stem, fuse_poolconv, conv_stem_direct, fuse_stem_conv
    call('pvhip_settings_reload')
    conv_prepad = os.environ.get('PVHIP_CONV_PREPAD', '1') != '0'
    conv_f16_dma = os.environ.get('PVHIP_CONV_F16_DMA', '1') != '0'
    conv_f16_span = int(os.environ.get('PVHIP_CONV_F16_SPAN', '1') or 0)
    conv_f16_c8 = _env_level('PVHIP_CONV_F16_C8', '2')
    fuse_poolconv = _env_level('PVHIP_FUSE_POOLCONV', '2')
    fuse_stem_conv = _env_level('PVHIP_FUSE_STEM_CONV', '1')
    conv_f16_stem = os.environ.get('PVHIP_CONV_F16_STEM', '1') != '0'
    conv_stem_direct = os.environ.get('PVHIP_CONV_STEM_DIRECT', '1') != '0'
    settings_serial += 1


def ensure_init():
    if _initialised_device is None:
        init()


def is_initialised() -> bool:
    return _initialised_device is not None


def device_name() -> str:
    ensure_init()
    buf = ctypes.create_string_buffer(256)
    call('pvhip_device_name', buf, 256)
    return buf.value.decode()


def synchronize():
    ensure_init()
    call('pvhip_sync')


def current_device():
    """Index of the GPU this process is bound to (None before init)."""
    return _initialised_device


def pool_stats():
    a, b = _c.c_size_t(0), _c.c_size_t(0)
    call('pvhip_pool_stats', _c.byref(a), _c.byref(b))
    return a.value, b.value


class _Block:
    """Owner of one pooled device allocation; returns it to the pool when the last view dies."""
    __slots__ = ('ptr', 'nbytes')

    def __init__(self, nbytes: int):
        ensure_init()
        p = _c.c_void_p(0)
        call('pvhip_malloc', _c.byref(p), int(nbytes))
        self.ptr = p.value or 0
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.pvhip_free(_c.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = 0

    # One owner per device allocation: a copied _Block would free the same pointer twice (the second time possibly after
    # the pool has handed the block to another tensor).
    def __deepcopy__(self, memo=None):
        raise PvhipError('a device block cannot be copied: strip device tensors from the graph first')

    __copy__ = __deepcopy__

    def __reduce__(self):
        raise PvhipError('a device block cannot be pickled')


def _contig_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= int(d)
    return tuple(reversed(st))


_PINNED_MAX_BYTES = 8 << 20          # per array; larger read-backs (activations in tests) stay pageable
_PINNED_POOL_BYTES = 64 << 20        # page-locked memory this process keeps at most
_pinned_free = {}                    # rounded size -> [address, ...]
_pinned_total = 0


def _pinned_release(size, addr):
    _pinned_free.setdefault(size, []).append(addr)


def _pinned_empty(shape, dtype):
    """np.empty in page-locked memory when the pool allows it (PVHIP_PINNED_RESULTS=0: never), else plain np.empty."""
    global _pinned_total
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize if len(shape) else dtype.itemsize
    if nbytes == 0 or nbytes > _PINNED_MAX_BYTES or os.environ.get('PVHIP_PINNED_RESULTS', '1') == '0':
        return np.empty(shape, dtype=dtype)
    size = 1 << max(12, (nbytes - 1).bit_length())
    free = _pinned_free.get(size)
    if free:
        addr = free.pop()
    else:
        if _pinned_total + size > _PINNED_POOL_BYTES:
            return np.empty(shape, dtype=dtype)
        addr = call('pvhip_host_alloc', size)
        if not addr:
            return np.empty(shape, dtype=dtype)
        _pinned_total += size
    buf = (_c.c_char * size).from_address(addr)
    weakref.finalize(buf, _pinned_release, size, addr)          # the array (and every view of it) keeps `buf` alive
    return np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


class DeviceTensor:
    """A dense, C-contiguous tensor resident in HBM.

    Exposes ``shape`` / ``dtype`` / ``ndim`` / ``size`` like an ndarray so the reference's per-plugin
    validation (``data.dtype == ...``, ``data.shape == dims``) keeps working on it unchanged, and
    ``__array__`` so that host code that really wants the values (``np.asarray(t)``, the Result
    plugin) gets a device-to-host copy -- which synchronises the stream.
    """
    __slots__ = ('_block', 'shape', 'dtype', '__weakref__')
    __array_priority__ = 100

    def __init__(self, block: _Block, shape, dtype=np.float32):
        self._block = block
        self.shape = tuple(int(d) for d in shape)
        self.dtype = np.dtype(dtype)
        if self.nbytes > block.nbytes:
            raise PvhipError('view {} larger than its block'.format(self.shape))

    # ---- construction
    @classmethod
    def empty(cls, shape, dtype=np.float32):
        shape = tuple(int(d) for d in shape)
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        return cls(_Block(max(1, n) * np.dtype(dtype).itemsize), shape, dtype)

    @classmethod
    def from_numpy(cls, array, dtype=None):
        a = np.ascontiguousarray(array, dtype=dtype)
        t = cls.empty(a.shape, a.dtype)
        if a.nbytes:
            call('pvhip_memcpy_h2d', _c.c_void_p(t.ptr), a.ctypes.data_as(_c.c_void_p), a.nbytes)
        return t

    # ---- ndarray-like surface
    @property
    def ptr(self) -> int:
        return self._block.ptr

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    def reshape(self, *shape):
        """Metadata-only reshape: the new tensor shares the device block."""
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            shape = tuple(shape[0])
        shape = [int(d) for d in shape]
        if shape.count(-1) > 1:
            raise ValueError('can only specify one unknown dimension')
        if -1 in shape:
            known = 1
            for d in shape:
                if d != -1:
                    known *= d
            if known == 0 or self.size % known:
                raise ValueError('cannot reshape tensor of size {} into shape {}'.format(self.size, tuple(shape)))
            shape[shape.index(-1)] = self.size // known
        n = 1
        for d in shape:
            n *= d
        if n != self.size:
            raise ValueError('cannot reshape tensor of size {} into shape {}'.format(self.size, tuple(shape)))
        return DeviceTensor(self._block, shape, self.dtype)

    def numpy(self) -> np.ndarray:
        """Device-to-host copy (synchronises the compute stream).  Read-backs of up to a few MB (a Result) land in page-locked host memory from a
        small pool -- one DMA instead of the runtime's staged copy into pageable memory; the block returns to the pool when the array is collected."""
        out = _pinned_empty(self.shape, self.dtype)
        if out.nbytes:
            call('pvhip_memcpy_d2h', out.ctypes.data_as(_c.c_void_p), _c.c_void_p(self.ptr), out.nbytes)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __repr__(self):
        return 'DeviceTensor(shape={}, dtype={}, ptr=0x{:x})'.format(self.shape, self.dtype.name, self.ptr)


class ChannelSlice:
    """Channels [coff, coff + k) of an NCHW DeviceTensor, as written in place by a producer whose consumer is
    a channel Concat.  Has the ndarray surface (shape / dtype / numpy()) but is not dense: it is only handed
    to the Concat it belongs to (which is not dispatched) and to debugging code."""
    __slots__ = ('base', 'coff', 'shape', 'dtype')

    def __init__(self, base: DeviceTensor, coff: int, k: int):
        self.base, self.coff = base, int(coff)
        self.shape = (base.shape[0], int(k)) + tuple(base.shape[2:])
        self.dtype = base.dtype

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    def numpy(self):
        return np.ascontiguousarray(self.base.numpy()[:, self.coff:self.coff + self.shape[1]])

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)


class BlockedHalf:
    """An fp16 tensor in HBM with the channels blocked by eight -- [n][ceil16(c) / 8][h * w][8 halves], include/pvhip.h
    pvhip_c8_f16_* -- as the 3x3_reduce / 5x5_reduce convolutions of an FP16 IR hand it to the 3x3 / 5x5 convolution behind them
    (what the reference holds there is a float16 ndarray, common_def.py:13-17).  Has the ndarray surface of the LOGICAL tensor
    (shape (n, c, h, w); dtype float32, the precision its port is declared with) so that the reference's per-plugin validation keeps
    working; numpy() converts back on the device.  Only pvhip_conv2d_f16_c8 reads the blocked bytes."""
    __slots__ = ('buf', 'shape', 'dtype')

    def __init__(self, shape):
        n, c, h, w = (int(d) for d in shape)
        self.shape = (n, c, h, w)
        self.dtype = np.dtype(np.float32)
        self.buf = DeviceTensor.empty((max(1, int(call('pvhip_c8_f16_elems', n, c, h, w))),))

    @classmethod
    def from_dense(cls, t: 'DeviceTensor'):
        out = cls(t.shape)
        call('pvhip_c8_f16_from_f32', _c.c_void_p(t.ptr), _c.c_void_p(out.buf.ptr), *out.shape)
        return out

    @property
    def ptr(self) -> int:
        return self.buf.ptr

    @property
    def ndim(self):
        return 4

    @property
    def size(self):
        n, c, h, w = self.shape
        return n * c * h * w

    def dense(self) -> 'DeviceTensor':
        """The same values as an NCHW fp32 tensor (every one of them an fp16 value)."""
        t = DeviceTensor.empty(self.shape)
        call('pvhip_c8_f16_to_f32', _c.c_void_p(self.buf.ptr), _c.c_void_p(t.ptr), *self.shape)
        return t

    def numpy(self):
        return self.dense().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)


class BlockedChannelSlice:
    """Channels [coff, coff + k) of a BlockedHalf, as written in place by a producer whose consumer is a channel Concat whose buffer is
    blocked fp16 (coff and k are multiples of 8).  The ndarray surface of the logical slice; only handed to the Concat it belongs to (which
    is not dispatched) and to debugging code."""
    __slots__ = ('base', 'coff', 'shape', 'dtype')

    def __init__(self, base: 'BlockedHalf', coff: int, k: int):
        self.base, self.coff = base, int(coff)
        self.shape = (base.shape[0], int(k)) + tuple(base.shape[2:])
        self.dtype = base.dtype

    @property
    def ndim(self):
        return 4

    @property
    def size(self):
        n, c, h, w = self.shape
        return n * c * h * w

    def numpy(self):
        return np.ascontiguousarray(self.base.numpy()[:, self.coff:self.coff + self.shape[1]])

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)


def as_device(data, dtype=np.float32) -> DeviceTensor:
    """Accept what a predecessor plugin handed over: a DeviceTensor (ours) or an ndarray (when our
    plugins are mixed with host plugins, e.g. under the reference's own engine)."""
    if isinstance(data, DeviceTensor):
        return data
    if isinstance(data, ChannelSlice):
        return DeviceTensor.from_numpy(data.numpy())     # densify (debug paths only)
    if isinstance(data, BlockedHalf):
        return data.dense()                              # a reader that does not take the blocked layout
    if isinstance(data, BlockedChannelSlice):
        return DeviceTensor.from_numpy(data.numpy())     # densify (debug paths only)
    return DeviceTensor.from_numpy(np.asarray(data), dtype=dtype)


def i64_array(values):
    """Host int64 array in ctypes form (shapes, strides, permutations)."""
    arr = (_c.c_int64 * max(1, len(values)))(*[int(v) for v in values])
    return arr


def pool_epoch_begin() -> int:
    """Open an allocation epoch (one forward pass on several streams / in flight): see include/pvhip.h."""
    ensure_init()
    e = _c.c_int(0)
    call('pvhip_pool_epoch_begin', _c.byref(e))
    return e.value


def pool_epoch_dispatched():
    call('pvhip_pool_epoch_dispatched')


def pool_epoch_end(epoch: int):
    call('pvhip_pool_epoch_end', int(epoch))


MAX_STREAMS = 9      # PVHIP_MAX_STREAMS of include/pvhip.h: 8 compute streams + one for copies / the RCCL gather
COPY_STREAM = 8      # what InferRequest.wait() gathers and copies on when the batch is sharded: no request computes there


def select_stream(index: int):
    """Make stream `index` (0..8) the current one: every later launch and copy goes to it."""
    ensure_init()
    call('pvhip_stream_select', int(index))


class Event:
    """hipEvent on the current compute stream: device-side timing of the hot path (`timed=True`), or an
    ordering-only event that another stream waits for (`timed=False`)."""

    def __init__(self, timed: bool = True):
        ensure_init()
        h = _c.c_void_p(0)
        call('pvhip_event_create' if timed else 'pvhip_event_create_untimed', _c.byref(h))
        self.handle = h.value

    def wait(self):
        """The current stream waits (on the device) until this event's recorded work has finished."""
        call('pvhip_stream_wait_event', _c.c_void_p(self.handle))
        return self

    def record(self):
        call('pvhip_event_record', _c.c_void_p(self.handle))
        return self

    def synchronize(self):
        call('pvhip_event_sync', _c.c_void_p(self.handle))

    def elapsed_ms(self, end: 'Event') -> float:
        ms = _c.c_float(0.0)
        call('pvhip_event_elapsed_ms', _c.c_void_p(self.handle), _c.c_void_p(end.handle), _c.byref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.pvhip_event_destroy(_c.c_void_p(self.handle))
        except Exception:
            pass
        self.handle = None
