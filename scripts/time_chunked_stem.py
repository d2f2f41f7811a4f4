#!/usr/bin/env python3
"""GPU-box tool: GoogLeNet's stem -- padding pass (+ mean), conv1 (7x7 / 2, 3 -> 64), MaxPool 3x3 / 2 + LRN -- at batch 256 as three
whole-batch launches (what the pass does) and in chunks of 128 / 64 / 32 / 16 images (pad, conv1, pool + LRN per chunk): does the pool
kernel read conv1's output (3.2 MB per image) from the 256 MB Infinity Cache when the chunk fits?
  python scripts/time_chunked_stem.py"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
dev.init(0)
P = ctypes.c_void_p
N, C, H, W, K = 256, 3, 224, 224, 64
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (N, C, H, W)))
wt = (synth.normal(3, 4, K * C * 49) * (2.0 / (C * 49)) ** 0.5).astype(np.float32).reshape((K, C, 7, 7))
w_dev = dev.DeviceTensor.from_numpy(wt)
bias = dev.DeviceTensor.from_numpy(synth.normal(5, 6, K).astype(np.float32))
mean = dev.DeviceTensor.from_numpy(np.array([-104.0, -117.0, -123.0], dtype=np.float32))
pack = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_pack_elems', K, C, 7, 7)),))
dev.call('pvhip_conv2d_pack_f32', P(w_dev.ptr), P(pack.ptr), K, C, 7, 7, 230, 230)
xp = dev.DeviceTensor.empty((N, C, 230, 230))
y1 = dev.DeviceTensor.empty((N, K, 112, 112))
y2 = dev.DeviceTensor.empty((N, K, 56, 56))

def stem(chunk):
    for n0 in range(0, N, chunk):
        n = min(chunk, N - n0)
        xo, xpo = x.ptr + n0 * C * H * W * 4, xp.ptr + n0 * C * 230 * 230 * 4
        y1o, y2o = y1.ptr + n0 * K * 112 * 112 * 4, y2.ptr + n0 * K * 56 * 56 * 4
        dev.call('pvhip_pad2d_f32', P(xo), P(xpo), n, C, H, W, 3, 3, 3, 3, P(mean.ptr))
        dev.call('pvhip_conv2d_f32', P(xpo), P(pack.ptr), P(y1o), n, C, 230, 230, K, 7, 7, 112, 112, 2, 2, 0, 0, P(bias.ptr), 1, 0, 0, 0.0, 0.0)
        dev.call('pvhip_maxpool_lrn_f32', P(y1o), P(y2o), n, K, 112, 112, 56, 56, 3, 3, 2, 2, 0, 0, 1, 1, 5, 9.999999747378752e-05 , 0.75, 1.0)

def timed(run, reps=10):
    for _ in range(3):
        run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps

ref = None
for rnd in range(2):
    for chunk in (256, 128, 64, 32, 16):
        ms = timed(lambda: stem(chunk))
        out = np.asarray(y2)[::37]
        if ref is None:
            ref = out.copy()
        print('chunks of {:3d} images: {:.3f} ms   same bits as whole-batch: {}'.format(chunk, ms, bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32)))), flush=True)
