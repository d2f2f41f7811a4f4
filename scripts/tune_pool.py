#!/usr/bin/env python3
"""GPU-box tool: time the MaxPool kernel on GoogLeNet's pooling shapes (batch 256) for several LDS group sizes."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth

SHAPES = [((256, 64, 112, 112), 2, 0), ((256, 192, 56, 56), 2, 0), ((256, 480, 28, 28), 2, 0), ((256, 832, 14, 14), 2, 0),
          ((256, 192, 28, 28), 1, 1), ((256, 256, 28, 28), 1, 1), ((256, 480, 14, 14), 1, 1), ((256, 512, 14, 14), 1, 1),
          ((256, 832, 7, 7), 1, 1)]
dev.init(0)
kbs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '12,16,24,32,48,60').split(',')]
tot = {kb: 0.0 for kb in kbs}
for xs, s, p in SHAPES:
    n, c, h, w = xs
    oh = -(-(h + 2 * p - 3) // s) + 1
    x = dev.DeviceTensor.from_numpy(synth.normal(1, h, n * c * h * w).astype(np.float32).reshape(xs))
    y = dev.DeviceTensor.empty((n, c, oh, oh))
    mb = 4.0 * (x.size + y.size) / 1e6
    line = '{} s{} p{} {:7.1f} MB |'.format(xs, s, p, mb)
    for kb in kbs:
        os.environ['PVHIP_POOL_LDS_KB'] = str(kb)
        run = lambda: dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, oh, 3, 3, s, s, p, p, p, p)
        run(); dev.synchronize()
        e0 = dev.Event().record()
        for _ in range(5):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        tot[kb] += ms
        line += ' {}KB:{:.3f}ms({:.0f}GB/s)'.format(kb, ms, mb / ms)
    print(line, flush=True)
print('totals', {k: round(v, 3) for k, v in tot.items()})
