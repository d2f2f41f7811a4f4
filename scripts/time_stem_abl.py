#!/usr/bin/env python3
"""GPU-box tool (diagnostic build: PVHIP_LIBRARY=pyopenvino_amd/libpvhip_diag.so): what conv_stem_f32_kernel's time is made of -- the kernel
with one piece taken out (wrong results on purpose): its epilogue, its MFMAs, its copies."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('PVHIP_LIBRARY', os.path.join(REPO, 'pyopenvino_amd', 'libpvhip_diag.so'))
from pyopenvino_amd import device as dev, synth
dev.init(0)
n, c, h, w, k, ks = 256, 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9
wps = int(dev.call('pvhip_conv2d_stem_f32_supported', c, h, w, k, ks, ks, 2, 2, 3, 3, 112, 112))
xp = dev.DeviceTensor.empty((n, c, h + 6, wps))
dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, w, 3, 3, 3, wps - w - 3, ctypes.c_void_p(0))
wf = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_stem_f32_pack_elems', k)),))
dev.call('pvhip_conv2d_stem_f32_pack', ctypes.c_void_p(wt.ptr), ctypes.c_void_p(wf.ptr), k)
y = dev.DeviceTensor.empty((n, k, 112, 112))

def timed(run, reps=20):
    for _ in range(3):
        run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps

for rep in range(2):
    for tag, abl in (('as it is', '0'), ('no epilogue (bias, activation, stores)', '1'), ('no MFMAs', '2'), ('no copies after the first tile', '3'), ('stores to consecutive 16-byte pieces (wrong places)', '4')):
        os.environ['PVHIP_STEM_ABLATE'] = abl; dev.reload_settings()
        ms = timed(lambda: dev.call('pvhip_conv2d_stem_f32', ctypes.c_void_p(xp.ptr), ctypes.c_void_p(wf.ptr), ctypes.c_void_p(y.ptr), n, h + 6, wps, k, 112, 112,
                                    ctypes.c_void_p(b.ptr), 1, 0.0, 0.0))
        print('{:42s} {:.3f} ms  ({:.1f} TFLOP/s if it were the whole)'.format(tag, ms, gf / ms), flush=True)
