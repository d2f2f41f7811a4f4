#!/usr/bin/env python3
"""GPU-box tool: conv1 (7x7 / stride 2 / pad 3, 3 -> 64 channels, batch 256) on the general LDS-DMA kernel (PVHIP_CONV_STEM=0) and on the
row-span kernel (pvhip_conv2d_stem_f32), alternating on one box: whole launch (padding pass + kernel), the kernel alone, and whether the
two carry the same bits."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
n, c, h, w, k, ks = int(os.environ.get('BATCH', '256')), 3, 224, 224, 64, 7
x = dev.DeviceTensor.from_numpy(synth.uniform_pixels(7, (n, c, h, w)))
wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
b = dev.DeviceTensor.from_numpy(synth.normal(5, 6, k).astype(np.float32).reshape((1, k, 1, 1)))
gf = 2.0 * n * k * c * ks * ks * 112 * 112 / 1e9
mean = dev.DeviceTensor.from_numpy(np.array([-104.0, -117.0, -123.0], dtype=np.float32).reshape((1, 3, 1, 1)))

def timed(run, reps=20):
    for _ in range(3):
        y = run()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps, y

outs = {}
for rep in range(3):
    for tag, env, direct in (('general LDS-DMA kernel', '0', '1'), ('row-span kernel', '1', '0'), ('row-span kernel, no padding pass', '1', '1')):
        os.environ['PVHIP_CONV_STEM'] = env; os.environ['PVHIP_CONV_STEM_DIRECT'] = direct; dev.reload_settings()
        node = {'_pre_add': mean}
        ms, y = timed(lambda: Convolution.launch(node, x, wt, (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=('relu',)))
        outs[tag] = np.asarray(y)[:4]
        print('{:34s} whole launch, data/mean folded in {:.3f} ms  {:.1f} TFLOP/s'.format(tag, ms, gf / ms), flush=True)
# the kernels alone (the padded image made once)
wps = int(dev.call('pvhip_conv2d_stem_f32_supported', c, h, w, k, ks, ks, 2, 2, 3, 3, 112, 112))
xp = dev.DeviceTensor.empty((n, c, h + 6, wps))
dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, w, 3, 3, 3, wps - w - 3, ctypes.c_void_p(0))
wf = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_stem_f32_pack_elems', k)),))
dev.call('pvhip_conv2d_stem_f32_pack', ctypes.c_void_p(wt.ptr), ctypes.c_void_p(wf.ptr), k)
y = dev.DeviceTensor.empty((n, k, 112, 112))
for rep in range(3):
    ms, _ = timed(lambda: dev.call('pvhip_conv2d_stem_f32', ctypes.c_void_p(xp.ptr), ctypes.c_void_p(wf.ptr), ctypes.c_void_p(y.ptr), n, h + 6, wps, k, 112, 112,
                                   ctypes.c_void_p(b.ptr), 1, 0.0, 0.0))
    print('row-span kernel alone      {:.3f} ms  {:.1f} TFLOP/s ({:.3f} of 157.3)'.format(ms, gf / ms, gf / ms / 157.3), flush=True)
    ms, _ = timed(lambda: dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, w, 3, 3, 3, wps - w - 3, ctypes.c_void_p(0)))
    print('padding pass alone         {:.3f} ms  {:.2f} TB/s'.format(ms, (x.nbytes + xp.nbytes) / ms / 1e9), flush=True)
for rep in range(3):
    ms, _ = timed(lambda: dev.call('pvhip_conv2d_stem_direct_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wf.ptr), ctypes.c_void_p(y.ptr), n, h, w, k, 112, 112,
                                   ctypes.c_void_p(mean.ptr), ctypes.c_void_p(b.ptr), 1, 0.0, 0.0))
    print('row-span kernel on the image itself {:.3f} ms  {:.1f} TFLOP/s ({:.3f} of 157.3)'.format(ms, gf / ms, gf / ms / 157.3), flush=True)
    ms, _ = timed(lambda: dev.call('pvhip_conv2d_stem_direct_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wf.ptr), ctypes.c_void_p(y.ptr), n, h, w, k, 112, 112,
                                   ctypes.c_void_p(0), ctypes.c_void_p(b.ptr), 1, 0.0, 0.0))
    print('  ... without the folded Add         {:.3f} ms'.format(ms), flush=True)
a_, b_ = outs['general LDS-DMA kernel'], outs['row-span kernel, no padding pass']
print('same bits (padded copy):', bool((outs['row-span kernel'].view(np.uint32) == a_.view(np.uint32)).all()))
print('same bits:', bool((a_.view(np.uint32) == b_.view(np.uint32)).all()), ' max |difference|', float(np.abs(a_ - b_).max()), ' max |value|', float(np.abs(a_).max()))
