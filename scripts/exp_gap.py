"""GPU-box experiment: stream time per back-to-back launch (tiny kernels), same kernel vs alternating kernels."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev  # noqa: E402

dev.init(0)
x = dev.DeviceTensor.from_numpy(np.ones((4096,), np.float32))
y = dev.DeviceTensor.empty((4096,))
big = dev.DeviceTensor.from_numpy(np.ones((64 << 20,), np.float32))
bigy = dev.DeviceTensor.empty((64 << 20,))


def relu(a, b):
    dev.call('pvhip_relu_f32', ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), a.size)


def sig(a, b):
    dev.call('pvhip_sigmoid_f32', ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), a.size)


def timed(fn, reps):
    fn()
    dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps):
        fn()
    e1 = dev.Event().record()
    e1.synchronize()
    return e0.elapsed_ms(e1) * 1e3 / reps


print('tiny relu x1 per iteration: {:.2f} us per launch'.format(timed(lambda: relu(x, y), 200)))
print('tiny relu,sigmoid alternating: {:.2f} us per launch'.format(timed(lambda: (relu(x, y), sig(y, x)), 100) / 2))
t1 = timed(lambda: relu(big, bigy), 20)
t2 = timed(lambda: (relu(big, bigy), sig(bigy, big)), 10) / 2
print('256 MB relu: {:.1f} us; relu,sigmoid alternating: {:.1f} us per launch ({:.0f} GB/s)'.format(t1, t2, 2 * 256e6 * 1.048576 / t1 / 1e3))
# with an untimed event record+wait between launches (what a cross-stream dependency costs on one stream)
ev = dev.Event(timed=False)
print('tiny relu + event record: {:.2f} us per launch'.format(timed(lambda: (relu(x, y), ev.record()), 200)))
tev = dev.Event()
print('tiny relu + TIMED event record: {:.2f} us per launch'.format(timed(lambda: (relu(x, y), tev.record()), 200)))
