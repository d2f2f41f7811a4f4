#!/usr/bin/env python3
"""GPU-box tool: sweep maxpool3x3_cols_kernel tile geometries (PVHIP_POOL3_CFG=G,S,band / PVHIP_POOL3_WG) on one shape.
  python scripts/sweep_pool.py N C H W stride pad  cfg[:wg] ..."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
n, c, h, w, s, p = [int(v) for v in sys.argv[1:7]]
dev.init(0)
oh = -(-(h + 2 * p - 3) // s) + 1
x = dev.DeviceTensor.from_numpy(synth.normal(1, h, n * c * h * w).astype(np.float32).reshape((n, c, h, w)))
y = dev.DeviceTensor.empty((n, c, oh, oh))
# a 600 MB scratch tensor written between timed launches keeps the infinity cache cold
big = dev.DeviceTensor.empty((150 * 1000 * 1000,))
mb = 4.0 * (x.size + y.size) / 1e6
for cfg in sys.argv[7:]:
    parts = cfg.split(':')
    for k in ('PVHIP_POOL3_CFG', 'PVHIP_POOL3_WG', 'PVHIP_POOL3', 'PVHIP_POOL3_STAGE'):
        os.environ.pop(k, None)
    if parts[0] == 'old':
        os.environ['PVHIP_POOL3'] = '0'
    elif parts[0] != 'auto':
        os.environ['PVHIP_POOL3_CFG'] = parts[0]
    if len(parts) > 1 and parts[1]:
        os.environ['PVHIP_POOL3_WG'] = parts[1]
    if len(parts) > 2:
        os.environ['PVHIP_POOL3_STAGE'] = parts[2]
    run = lambda: dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, oh, 3, 3, s, s, p, p, p, p)
    run(); dev.synchronize()
    ts = []
    for _ in range(6):
        dev.call('pvhip_memset', ctypes.c_void_p(big.ptr), 0, big.size * 4)
        e0 = dev.Event().record(); run(); e1 = dev.Event().record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    print('{:>16s}: {:.4f} ms {:6.0f} GB/s (min {:.4f})'.format(cfg, ms, mb / ms, ts[0]), flush=True)
