#!/usr/bin/env python3
"""GPU-box tool: the six-point Winograd layers of GoogLeNet (batch 256) on the two-workgroup form (conv_wino4_kernel) and on the
shared-V form (conv_wino4s_kernel, PVHIP_WINO_SHARED=2), alternating on one box; bits compared.
  python scripts/time_wino_shared.py [substring of the layer name]"""
import os, sys, statistics
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3a/3x3', (256, 96, 28, 28), 128, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3),
          ('4a/3x3', (256, 96, 14, 14), 208, 3), ('4b/3x3', (256, 112, 14, 14), 224, 3), ('4c/3x3', (256, 128, 14, 14), 256, 3),
          ('4d/3x3', (256, 144, 14, 14), 288, 3), ('4e/3x3', (256, 160, 14, 14), 320, 3), ('5a/3x3', (256, 160, 7, 7), 320, 3),
          ('5b/3x3', (256, 192, 7, 7), 384, 3),
          ('3a/5x5', (256, 16, 28, 28), 32, 5), ('3b/5x5', (256, 32, 28, 28), 96, 5), ('4a/5x5', (256, 16, 14, 14), 48, 5),
          ('4b/5x5', (256, 24, 14, 14), 64, 5), ('4c/5x5', (256, 24, 14, 14), 64, 5), ('4d/5x5', (256, 32, 14, 14), 64, 5),
          ('4e/5x5', (256, 32, 14, 14), 128, 5), ('5a/5x5', (256, 32, 7, 7), 128, 5), ('5b/5x5', (256, 48, 7, 7), 128, 5)]
dev.init(0)
only = sys.argv[1] if len(sys.argv) > 1 else ''
tot = {'two workgroups': 0.0, 'shared V': 0.0, 'best': 0.0}
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy((synth.normal(5, 6, k) * 0.1).astype(np.float32).reshape((1, k, 1, 1)))
    pd = (ks // 2, ks // 2)
    times, outs = {'two workgroups': [], 'shared V': [], 'shared V, no lag': [], 'shared V, no prio': [], 'shared V, young': []}, {}
    applies = ((k + 31) // 32) >= 2 and (c // 4) % 4 == 0
    for rnd in range(3):
        for tag, mode in (('two workgroups', '0'), ('shared V', '2'), ('shared V, no lag', '2'), ('shared V, no prio', '2'), ('shared V, young', '2')):
            os.environ['PVHIP_WINO_SHARED'] = mode
            os.environ['PVHIP_WINO_SHARED_LAG'] = '1' if 'no lag' in tag else '0'
            os.environ['PVHIP_WINO_SHARED_PRIO'] = '0' if 'no prio' in tag else '1'
            os.environ['PVHIP_WINO_SHARED_OLD'] = '0' if 'young' in tag else '1'
            os.environ['PVHIP_CONV_WINOGRAD4'] = 'force'
            os.environ['PVHIP_CONV_WINOGRAD5'] = 'force'
            dev.reload_settings()
            node = {}
            run = lambda: Convolution.launch(node, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',))
            for _ in range(2):
                y = run()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(5):
                run()
            e1 = dev.Event().record(); e1.synchronize()
            times[tag].append(e0.elapsed_ms(e1) / 5)
            outs[tag] = np.asarray(y)
    a_, s_ = statistics.median(times['two workgroups']), statistics.median(times['shared V'])
    tot['two workgroups'] += a_; tot['shared V'] += s_; tot['best'] += min(a_, s_)
    same = np.array_equal(outs['two workgroups'], outs['shared V'])
    print('{:10s} two workgroups {:.4f} ms | shared V {:.4f} ms ({}) | {:+.1f} % | with lag {:.4f} | no prio {:.4f} | young producers {:.4f} | same bits {} finite {}'.format(
        name, a_, s_, 'applies' if applies else 'falls back', 100.0 * (s_ / a_ - 1.0), statistics.median(times['shared V, no lag']), statistics.median(times['shared V, no prio']), statistics.median(times['shared V, young']), same, bool(np.isfinite(outs['shared V']).all())), flush=True)
print('sum: two workgroups {:.4f} ms, shared V where it applies {:.4f} ms, best of both per layer {:.4f} ms'.format(tot['two workgroups'], tot['shared V'], tot['best']))
