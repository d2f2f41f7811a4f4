#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): s_memtime accounts per 16-channel stage of MaxPool + pool_proj (conv_pool1x1_kernel, PVHIP_CONV_ABLATE=64) on GoogLeNet's
modules at batch 256 -- who waits for whom: the producers' pooling (incl. waiting for their loads) and barrier, the consumers' MFMA section, weight-copy wait and
barrier.  The diagnostic build carries the run-time ablation branches in the producers' loops: its stages are about twice as long as the product's
(4a: 5.4 k cycles here, 2.4 k with the same stamps patched into the product build, profiles/r05_stamps.md) -- read the proportions."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
dev.LIB_PATH = dev.DIAG_LIB_PATH
from pyopenvino_amd.op_plugins import Convolution
os.environ['PVHIP_CONV_ABLATE'] = '64'
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_poolconv_stamps.argtypes = [ctypes.c_void_p]
for name, (c, h, w), k in (('3a', (192, 28, 28), 32), ('3b', (256, 28, 28), 64), ('4a', (480, 14, 14), 64), ('4d', (512, 14, 14), 64), ('4e', (528, 14, 14), 128)):
    n = 256
    xs = (n, c, h, w)
    x = dev.DeviceTensor.from_numpy(np.maximum(synth.normal(1, 2, n * c * h * w), 0).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1)))
    for _ in range(5):
        Convolution.launch_pooled({}, x, wt)
    st = (ctypes.c_ulonglong * 8)()
    lib.pvhip_diag_poolconv_stamps(st)
    t = list(st); nk = max(1, t[0])
    print('{} ({} stages) per stage, cycles | producer: pooling incl. its loads {}, barrier {}, loop {} | consumer: MFMA section {}, weight copy {}, barrier {}, loop {}'.format(
        name, nk, t[1] // nk, t[2] // nk, t[3] // nk, t[4] // nk, t[5] // nk, t[6] // nk, t[7] // nk), flush=True)
