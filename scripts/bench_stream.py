#!/usr/bin/env python3
"""GPU-box tool: achieved HBM bandwidth of the streaming kernels (ReLU, Add per-channel, Concat-like copy) at
several tensor sizes, through the C ABI."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev
dev.init(0)
def timeit(fn, reps=20):
    fn(); dev.synchronize()
    e0 = dev.Event().record()
    for _ in range(reps): fn()
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps
for shape in [(256, 64, 28, 28), (256, 64, 56, 56), (256, 192, 56, 56), (256, 64, 112, 112), (256, 128, 112, 112)]:
    n = int(np.prod(shape))
    x = dev.DeviceTensor.empty(shape); y = dev.DeviceTensor.empty(shape); b = dev.DeviceTensor.empty((1, shape[1], 1, 1))
    dev.call('pvhip_memset', ctypes.c_void_p(x.ptr), 0, n * 4); dev.call('pvhip_memset', ctypes.c_void_p(b.ptr), 0, shape[1] * 4)
    mb = 8.0 * n / 1e6
    t_relu = timeit(lambda: dev.call('pvhip_relu_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n))
    shp = dev.i64_array(shape); st_a = dev.i64_array([shape[1]*shape[2]*shape[3], shape[2]*shape[3], shape[3], 1]); st_b = dev.i64_array([0, 1, 0, 0])
    t_add = timeit(lambda: dev.call('pvhip_add_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(y.ptr), 4, shp, st_a, st_b))
    t_cp = timeit(lambda: dev.call('pvhip_memcpy_d2d', ctypes.c_void_p(y.ptr), ctypes.c_void_p(x.ptr), n * 4))
    print('{} {:7.1f} MB r+w | relu {:.3f} ms {:5.0f} GB/s | add(bias) {:.3f} ms {:5.0f} GB/s | hipMemcpyDtoD {:.3f} ms {:5.0f} GB/s'.format(
        shape, mb, t_relu, mb / t_relu, t_add, mb / t_add, t_cp, mb / t_cp), flush=True)
