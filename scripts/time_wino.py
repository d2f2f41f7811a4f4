#!/usr/bin/env python3
"""GPU-box tool: time the 3x3 / stride 1 / pad 1 and 5x5 / stride 1 / pad 2 GoogLeNet layers (batch 256) under the Winograd
variants and the direct kernel, through the Convolution plugin (fused bias + ReLU).
  python scripts/time_wino.py [substring of the layer name]"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('4a/3x3', (256, 96, 14, 14), 208, 3), ('4c/3x3', (256, 128, 14, 14), 256, 3), ('4d/3x3', (256, 144, 14, 14), 288, 3), ('5a/3x3', (256, 160, 7, 7), 320, 3), ('5b/3x3', (256, 192, 7, 7), 384, 3), ('5a/5x5', (256, 32, 7, 7), 128, 5), ('5b/5x5', (256, 48, 7, 7), 128, 5), ('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3a/3x3', (256, 96, 28, 28), 128, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3),
          ('4e/3x3', (256, 160, 14, 14), 320, 3),
          ('3a/5x5', (256, 16, 28, 28), 32, 5), ('3b/5x5', (256, 32, 28, 28), 96, 5), ('4a/5x5', (256, 16, 14, 14), 48, 5),
          ('4b/5x5', (256, 24, 14, 14), 64, 5), ('4d/5x5', (256, 32, 14, 14), 64, 5), ('4e/5x5', (256, 32, 14, 14), 128, 5)]
if os.environ.get('ABLATE'):          # the ablation paths exist only in the diagnostic build (make -C pyopenvino_amd/csrc diag)
    dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    gf = 2.0 * n * k * c * ks * ks * h * w / 1e9
    line = '{:10s} {:6.1f} GFLOP |'.format(name, gf)
    outs = {}
    variants = [('F(4x4)', {'PVHIP_CONV_WINOGRAD4': 'force'}), ('F(2x2)', {'PVHIP_CONV_WINOGRAD4': '0'}), ('direct', {'PVHIP_CONV_WINOGRAD': '0'})]
    if ks == 5:
        variants = [('F(2x2,5x5)', {'PVHIP_CONV_WINOGRAD5': 'force'}), ('direct', {'PVHIP_CONV_WINOGRAD5': '0'})]
    if os.environ.get('ABLATE'):
        variants = [('F(4x4)', {'PVHIP_CONV_WINOGRAD4': 'force'})] + [('abl%s' % a_, {'PVHIP_CONV_WINOGRAD4': 'force', 'PVHIP_WINO4_ABLATE': a_}) for a_ in os.environ['ABLATE'].split(',')] + [('direct', {'PVHIP_CONV_WINOGRAD': '0'}), ('F(2x2)', {'PVHIP_CONV_WINOGRAD4': '0'})]
    for tag, env in variants:
        for k_, v_ in env.items():
            os.environ[k_] = v_
        dev.reload_settings()
        node = {}
        pd = (ks // 2, ks // 2)
        run = lambda: Convolution.launch(node, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',))
        for _ in range(3):
            y = run()
        dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(5):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        outs[tag] = np.asarray(y)[:2]
        line += ' {}: {:.3f} ms {:5.1f} TF |'.format(tag, ms, gf / ms)
        for k_ in env:
            del os.environ[k_]
        dev.reload_settings()
    ref = outs['direct']
    sc = np.abs(ref).max()
    line += ' max |x - direct| / max: ' + ', '.join('{} {:.1e}'.format(k_, np.abs(v_ - ref).max() / sc) for k_, v_ in outs.items() if k_ != 'direct')
    print(line, flush=True)
