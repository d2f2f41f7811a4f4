"""Tests of what exists in the DIAGNOSTIC build only (libpvhip_diag.so: make -C pyopenvino_amd/csrc diag; include/pvhip_diag.h): the
predecessor convolution kernels kept for A/B runs (PVHIP_CONV_KERNEL=lds|wave, every PVHIP_CONV_TILE / PVHIP_CONV_WTILE), the opt-in
16-byte gather (PVHIP_CONV_PW) and the fp32-MFMA ceiling probe.  NOT collected by `pytest tests/` (the file name does not match
test_*.py): tests/test_hip_ops.py::test_diagnostic_build_variants_in_their_own_process runs it in a process of its own with
PVHIP_LIBRARY=pyopenvino_amd/libpvhip_diag.so, so that the main test process only ever loads the product library.  GPU only.

    PVHIP_LIBRARY=pyopenvino_amd/libpvhip_diag.so python -m pytest tests/diag_variants.py -q
"""
import os

import numpy as np
import pytest

import helpers
from helpers import assert_bit_exact, assert_close, first_out
from test_hip_ops import conv_data, hip_plugin, make_node, oracle_plugin, rnd, vs_oracle


def test_this_process_loads_the_diagnostic_build(hip):
    assert os.path.basename(hip.LIB_PATH) == 'libpvhip_diag.so', hip.LIB_PATH


def test_conv_every_tile_config(hip, monkeypatch):
    """Force each (BM, BN) instantiation on one shape that has ragged edges in both tile dimensions."""
    helpers.setenv(monkeypatch, 'PVHIP_CONV_WINOGRAD', '0')
    x = rnd(1, (3, 20, 13, 11))
    w = rnd(2, (150, 20, 3, 3), 0.1)
    x2 = rnd(3, (3, 32, 13, 11))
    w2 = rnd(4, (150, 32, 3, 3), 0.1)
    helpers.setenv(monkeypatch, 'PVHIP_CONV_KERNEL', 'lds')      # register-staged kernels: c-major (C=20) and (r,s)-major (C=32)
    for tile in ('32x128', '32x256', '64x128', '64x256', '128x128', '128x256'):
        helpers.setenv(monkeypatch, 'PVHIP_CONV_TILE', tile)
        vs_oracle('Convolution', [x, w], conv_data((1, 1), (1, 1), (1, 1)), 'tile ' + tile)
        vs_oracle('Convolution', [x2, w2], conv_data((1, 1), (1, 1), (1, 1)), 'rs tile ' + tile)




def test_conv_wave_direct_kernel_every_tile(hip, monkeypatch):
    """The LDS-free wave-direct kernel, every wave tile, on shapes with ragged edges, padding, stride 2,
    an odd number of reduction stages and a 7x7 (64-bit mask) window."""
    helpers.setenv(monkeypatch, 'PVHIP_CONV_KERNEL', 'wave')
    cases = [((3, 20, 13, 11), (150, 20, 3, 3), (1, 1), (1, 1), (1, 1)),
             ((2, 48, 7, 7), (24, 48, 1, 1), (1, 1), (0, 0), (0, 0)),
             ((1, 3, 37, 37), (16, 3, 7, 7), (2, 2), (3, 3), (3, 3)),
             ((2, 9, 10, 10), (70, 9, 5, 5), (1, 1), (2, 2), (2, 2))]
    for tile in ('1x1', '1x2', '2x1', '2x2', '4x1', '1x4'):
        helpers.setenv(monkeypatch, 'PVHIP_CONV_WTILE', tile)
        for xs, ws, st, pb, pe in cases:
            x = rnd(sum(xs), xs)
            w = rnd(sum(ws), ws, 0.1)
            vs_oracle('Convolution', [x, w], conv_data(st, pb, pe), 'wave tile {} {}'.format(tile, xs))




def test_conv_lds_dma_kernel_every_tile(hip, monkeypatch):
    """The LDS-DMA (buffer_load ... lds) kernel, both reduction orders, on every channel tile: zero padding
    through the out-of-range sentinel, stride 2, ragged pixel and channel tiles, one and many reduction stages."""
    helpers.setenv(monkeypatch, 'PVHIP_CONV_WINOGRAD', '0')
    cases = [((3, 32, 13, 11), (150, 32, 3, 3), (1, 1), (1, 1), (1, 1)),
             ((2, 48, 7, 7), (24, 48, 1, 1), (1, 1), (0, 0), (0, 0)),
             ((1, 16, 37, 37), (16, 16, 7, 7), (2, 2), (3, 3), (3, 3)),
             ((2, 16, 10, 10), (70, 16, 5, 5), (1, 1), (2, 2), (2, 2)),
             ((5, 192, 28, 28), (16, 192, 1, 1), (1, 1), (0, 0), (0, 0)),
             # c-major reduction order (C not a multiple of 16): per-row window-bit table
             ((3, 20, 13, 11), (150, 20, 3, 3), (1, 1), (1, 1), (1, 1)),
             ((1, 3, 37, 37), (16, 3, 7, 7), (2, 2), (3, 3), (3, 3)),
             ((2, 9, 10, 10), (70, 9, 5, 5), (1, 1), (2, 2), (2, 2)),
             ((2, 1, 12, 12), (8, 1, 3, 3), (1, 1), (0, 0), (0, 0))]
    for tile in ('32x128', '64x128', '128x128'):
        helpers.setenv(monkeypatch, 'PVHIP_CONV_TILE', tile)
        for xs, ws, st, pb, pe in cases:
            x = rnd(sum(xs), xs)
            w = rnd(sum(ws), ws, (2.0 / (ws[1] * ws[2] * ws[3])) ** 0.5)
            vs_oracle('Convolution', [x, w], conv_data(st, pb, pe), 'dma tile {} {}'.format(tile, xs))




def test_conv_pointwise_16byte_gather_variant(hip, monkeypatch):
    """The opt-in 16-byte gather of the (r,s)-major kernel for 1x1 / stride 1 / unpadded layers (PVHIP_CONV_PW=1;
    off by default because it measured slower) stays correct, ragged last pixel tile included."""
    helpers.setenv(monkeypatch, 'PVHIP_CONV_PW', '1')
    helpers.setenv(monkeypatch, 'PVHIP_CONV_KERNEL', 'lds')
    for xs, ws in [((3, 64, 14, 14), (96, 64, 1, 1)), ((2, 32, 6, 6), (40, 32, 1, 1)), ((5, 192, 28, 28), (16, 192, 1, 1))]:
        x = rnd(sum(xs), xs)
        w = rnd(sum(ws), ws, (2.0 / ws[1]) ** 0.5)
        vs_oracle('Convolution', [x, w], conv_data((1, 1), (0, 0), (0, 0)), 'pointwise {}'.format(xs))




def test_mfma_ceiling_probe_reports_a_plausible_rate(hip):
    """bench.py's roofline.sustained: fp32 MFMA alone is below the 157.3 TFLOP/s of 2.4 GHz and far above any convolution here; a
    VALU-only wave beside every MFMA wave takes a visible share away (the two do not overlap on a SIMD)."""
    import ctypes
    lib = hip.load_library()             # this process runs on the diagnostic build (PVHIP_LIBRARY), which exports the probe
    lib.pvhip_mfma_ceiling_f32.restype = ctypes.c_int
    lib.pvhip_mfma_ceiling_f32.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]

    def probe(partner, iters):
        tf, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
        assert lib.pvhip_mfma_ceiling_f32(int(partner), iters, ctypes.byref(tf), ctypes.byref(ghz)) == 0
        return tf.value, ghz.value

    alone = max(probe(False, 4000) for _ in range(3))
    shared = max(probe(True, 4000) for _ in range(2))
    assert 90.0 < alone[0] < 158.0 and 1.2 < alone[1] < 2.6, alone
    assert shared[0] < 0.85 * alone[0], (alone, shared)




@pytest.mark.parametrize('kernel', ['lds'])
def test_conv_fused_bias_and_activation_bit_exact(hip, monkeypatch, kernel):
    """Fused epilogues (bias, then ReLU or Clamp) of both convolution kernels and of the depthwise kernel equal
    the separate Add / ReLU / Clamp launches bit for bit."""
    if kernel != 'default':
        helpers.setenv(monkeypatch, 'PVHIP_CONV_KERNEL', kernel)
        helpers.setenv(monkeypatch, 'PVHIP_CONV_WINOGRAD', '0')
    cases = [('Convolution', (2, 32, 9, 9), (40, 32, 3, 3)),     # (r,s)-major kernel (LDS-DMA by default)
             ('Convolution', (2, 5, 9, 9), (70, 5, 3, 3)),       # c-major kernel
             ('GroupConvolution', (2, 24, 11, 11), (24, 1, 1, 3, 3))]
    for type_, xs, ws in cases:
        x, w = rnd(1, xs), rnd(2, ws, 0.2)
        b = rnd(3, (1, ws[0], 1, 1))
        node = make_node(type_, [x, w], conv_data((1, 1), (1, 1), (1, 1), 'same_upper' if type_ == 'GroupConvolution' else 'explicit'))
        plain = first_out(hip_plugin(type_).compute(node, {0: x, 1: w}))
        biased = first_out(hip_plugin('Add').compute(make_node('Add', [plain, b]), {0: plain, 1: b}))
        for act, ref_type, data in ((('relu',), 'ReLU', None), (('clamp', 0.0, 6.0), 'Clamp', {'min': '0', 'max': '6'})):
            want = first_out(hip_plugin(ref_type).compute(make_node(ref_type, [biased], data), {0: biased}))
            fused_node = dict(node)
            fused_node['_fuse_bias'] = hip.DeviceTensor.from_numpy(b)
            fused_node['_fuse_act'] = act
            got = first_out(hip_plugin(type_).compute(fused_node, {0: x, 1: w}))
            assert_bit_exact(got, want, '{} fused bias + {}'.format(type_, act[0]))

