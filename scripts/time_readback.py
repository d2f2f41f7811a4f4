#!/usr/bin/env python3
"""GPU-box tool: the read-back of a Result (256 x 1000 fp32 = 1 MB) into page-locked memory from device.py's pool (default) and into pageable memory
(PVHIP_PINNED_RESULTS=0), alternating."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev
dev.init(0)
t = dev.DeviceTensor.from_numpy(np.random.default_rng(0).normal(size=(256, 1000)).astype(np.float32))
for rep in range(3):
    for knob in ('1', '0'):
        os.environ['PVHIP_PINNED_RESULTS'] = knob
        for _ in range(5): t.numpy()
        t0 = time.perf_counter()
        for _ in range(200): a = t.numpy()
        print('pinned' if knob == '1' else 'pageable', '{:.1f} us per read-back'.format((time.perf_counter() - t0) / 200 * 1e6), flush=True)
