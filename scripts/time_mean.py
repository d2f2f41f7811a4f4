import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from pyopenvino_amd import device as dev
dev.init(0)
def timeit(fn, reps=20):
    fn(); dev.synchronize()
    best=1e9
    for _ in range(3):
        e0 = dev.Event().record()
        for _ in range(reps): fn()
        e1 = dev.Event().record(); e1.synchronize()
        best=min(best, e0.elapsed_ms(e1) / reps)
    return best
for shape in [(256, 3, 224, 224), (256, 64, 56, 56)]:
    n = int(np.prod(shape))
    x = dev.DeviceTensor.empty(shape); y = dev.DeviceTensor.empty(shape); b = dev.DeviceTensor.empty((1, shape[1], 1, 1))
    dev.call('pvhip_memset', ctypes.c_void_p(x.ptr), 0, n * 4); dev.call('pvhip_memset', ctypes.c_void_p(b.ptr), 0, shape[1] * 4)
    shp = dev.i64_array(shape); st_a = dev.i64_array([shape[1]*shape[2]*shape[3], shape[2]*shape[3], shape[3], 1]); st_b = dev.i64_array([0, 1, 0, 0])
    for nt in ('0','1','2'):
        os.environ['PVHIP_STREAM_NT']=nt; dev.reload_settings()
        t = timeit(lambda: dev.call('pvhip_add_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(y.ptr), 4, shp, st_a, st_b))
        print(shape, 'nt', nt, 'add(bias) %.4f ms %.0f GB/s' % (t, 8.0*n/1e6/t))
