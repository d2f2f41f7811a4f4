#!/usr/bin/env python3
"""GPU-box tool: time the MaxPool entry on GoogLeNet's pooling shapes (batch 256) under several environment
settings (one column per setting; a setting is a comma-separated list of NAME=VALUE, 'base' = no overrides).
  python scripts/tune_pool.py base PVHIP_POOL3=0 PVHIP_POOL3_KB=24 PVHIP_POOL3_STAGE=0"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth

SHAPES = [((256, 64, 112, 112), 2, 0), ((256, 192, 56, 56), 2, 0), ((256, 480, 28, 28), 2, 0), ((256, 832, 14, 14), 2, 0),
          ((256, 192, 28, 28), 1, 1), ((256, 256, 28, 28), 1, 1), ((256, 480, 14, 14), 1, 1), ((256, 512, 14, 14), 1, 1),
          ((256, 528, 14, 14), 1, 1), ((256, 832, 7, 7), 1, 1)]
WEIGHT = [1, 1, 1, 1, 1, 1, 1, 3, 1, 2]          # launches per GoogLeNet forward pass
dev.init(0)
settings = sys.argv[1:] or ['base', 'PVHIP_POOL3=0']
tot = {s: 0.0 for s in settings}
mbt = 0.0
for (xs, s, p), wgt in zip(SHAPES, WEIGHT):
    n, c, h, w = xs
    oh = -(-(h + 2 * p - 3) // s) + 1
    x = dev.DeviceTensor.from_numpy(synth.normal(1, h, n * c * h * w).astype(np.float32).reshape(xs))
    y = dev.DeviceTensor.empty((n, c, oh, oh))
    mb = 4.0 * (x.size + y.size) / 1e6
    mbt += mb * wgt
    line = '{} s{} p{} {:7.1f} MB |'.format(xs, s, p, mb)
    for st in settings:
        keys = []
        if st != 'base':
            for kv in st.split(','):
                k, v = kv.split('=', 1)
                os.environ[k] = v
                keys.append(k)
        run = lambda: dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, oh, 3, 3, s, s, p, p, p, p)
        run(); dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(10):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 10
        tot[st] += ms * wgt
        line += ' {}: {:.3f}ms {:.0f}GB/s |'.format(st, ms, mb / ms)
        for k in keys:
            del os.environ[k]
    print(line, flush=True)
print('per forward pass ({:.0f} MB):'.format(mbt), {k: '{:.3f} ms {:.0f} GB/s'.format(v, mbt / v) for k, v in tot.items()})
