#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): where the waves of the F(2x2,3x3) kernel spend their cycles (s_memtime stamps, every 61st workgroup)."""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
LAYERS = [('4a/3x3', (256, 96, 14, 14), 208), ('4e/3x3', (256, 160, 14, 14), 320), ('5b/3x3', (256, 192, 7, 7), 384)]
dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_wino4_stamps.argtypes = [ctypes.c_void_p]
for name, xs, k in LAYERS:
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * 9) * (2.0 / (c * 9)) ** 0.5).astype(np.float32).reshape((k, c, 3, 3)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    for abl in ('0', '5'):
        os.environ['PVHIP_CONV_WINOGRAD4'] = '0'
        os.environ['PVHIP_WINO4_ABLATE'] = abl
        dev.reload_settings()
        node = {}
        run = lambda: Convolution.launch(node, x, wt, (1, 1), (1, 1), (1, 1), 'explicit', bias=b, act=('relu',))
        for _ in range(3):
            run()
        dev.synchronize()
        out = (ctypes.c_ulonglong * 64)()
        lib.pvhip_diag_wino4_stamps(out)
        e0 = dev.Event().record()
        for _ in range(5):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        lib.pvhip_diag_wino4_stamps(out)
        st = np.array(list(out), dtype=np.float64).reshape(8, 8)
        print('{} stamps={}: {:.3f} ms'.format(name, abl, ms), flush=True)
        if abl == '5':
            stages = c // 4
            for wv in range(8):
                cnt = st[wv, 7]
                if cnt:
                    per = st[wv, :3] / cnt / stages
                    print('  wave {}: per stage  gather issue + U DMA + MFMAs {:6.0f}  transform + waits {:6.0f}  barrier {:6.0f} | total {:6.0f} | main loop {:8.0f}  epilogue {:7.0f}  ({} workgroups)'.format(
                        wv, per[0], per[1], per[2], per.sum(), st[wv, :3].sum() / cnt, st[wv, 5] / cnt, int(cnt)))
