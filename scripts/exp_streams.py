"""GPU-box experiment: do independent kernels on two compute streams overlap?  Inception-3a-like pair:
a 3x3 convolution (MFMA-bound) on stream 0 and a 3x3/s1 MaxPool + 1x1 convolutions (HBM-bound) on stream 1."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth  # noqa: E402

dev.init(0)
N = 256


def conv_setup(c, h, w, k, kh, pad):
    x = dev.DeviceTensor.from_numpy(synth.normal(1, c, N * c * h * w).astype(np.float32).reshape(N, c, h, w))
    wt = dev.DeviceTensor.from_numpy((synth.normal(2, k, k * c * kh * kh) * 0.05).astype(np.float32).reshape(k, c, kh, kh))
    y = dev.DeviceTensor.empty((N, k, h, w))
    wp = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_pack_elems', k, c, kh, kh)),))
    dev.call('pvhip_conv2d_pack_f32', ctypes.c_void_p(wt.ptr), ctypes.c_void_p(wp.ptr), k, c, kh, kh, h, w)

    def run():
        dev.call('pvhip_conv2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wp.ptr), ctypes.c_void_p(y.ptr),
                 N, c, h, w, k, kh, kh, h, w, 1, 1, pad, pad, ctypes.c_void_p(0), 0, 0, 0, 0.0, 0.0)
    return run, (x, wt, y, wp)


def pool_setup(c, h, w):
    x = dev.DeviceTensor.from_numpy(synth.normal(3, c, N * c * h * w).astype(np.float32).reshape(N, c, h, w))
    y = dev.DeviceTensor.empty((N, c, h, w))

    def run():
        dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), N, c, h, w, h, w, 3, 3, 1, 1, 1, 1, 1, 1)
    return run, (x, y)


def timed(fn, reps=5):
    fn()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        fn()
    e1 = dev.Event().record()
    e1.synchronize()
    return e0.elapsed_ms(e1) / reps


for (c, h, k3r, k3, k1, kp) in [(192, 28, 96, 128, 64, 32), (480, 14, 96, 208, 192, 64), (832, 7, 192, 384, 384, 128)]:
    big, keep1 = conv_setup(k3r, h, h, k3, 3, 1)
    pool, keep2 = pool_setup(c, h, h)
    proj, keep3 = conv_setup(c, h, h, kp, 1, 0)
    one, keep4 = conv_setup(c, h, h, k1, 1, 0)
    red, keep5 = conv_setup(c, h, h, k3r, 1, 0)
    t_big, t_pool, t_proj, t_one, t_red = (timed(f) for f in (big, pool, proj, one, red))

    def serial():
        red(); big(); pool(); proj(); one()

    fork, join = dev.Event(timed=False), dev.Event(timed=False)

    def forked():
        fork.record()
        dev.select_stream(1)
        fork.wait()
        pool(); proj(); one()
        join.record()
        dev.select_stream(0)
        red(); big()
        join.wait()

    t_serial = timed(serial)
    t_fork = timed(forked)
    print('C={} {}x{}: 3x3r {:.3f} 3x3 {:.3f} pool {:.3f} proj {:.3f} 1x1 {:.3f} | sum {:.3f} serial {:.3f} two streams {:.3f}'.format(
        c, h, h, t_red, t_big, t_pool, t_proj, t_one, t_red + t_big + t_pool + t_proj + t_one, t_serial, t_fork), flush=True)
    dev.synchronize()
