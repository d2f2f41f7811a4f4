#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): where the waves of conv_wino4s_kernel (shared-V six-point Winograd) spend their cycles.
  python scripts/stamps_wino4s.py [substring of the layer name]"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192, 3), ('3a/3x3', (256, 96, 28, 28), 128, 3), ('3b/3x3', (256, 128, 28, 28), 192, 3),
          ('4c/3x3', (256, 128, 14, 14), 256, 3), ('4e/3x3', (256, 160, 14, 14), 320, 3), ('5b/3x3', (256, 192, 7, 7), 384, 3),
          ('4e/5x5', (256, 32, 14, 14), 128, 5)]
dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_wino4s_stamps.argtypes = [ctypes.c_void_p]
lib.pvhip_diag_wino4s_simd.argtypes = [ctypes.c_void_p]
only = sys.argv[1] if len(sys.argv) > 1 else ''
os.environ['PVHIP_CONV_WINOGRAD4'] = 'force'
os.environ['PVHIP_CONV_WINOGRAD5'] = 'force'
os.environ['PVHIP_WINO_SHARED'] = '2'
dev.reload_settings()
for name, xs, k, ks in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    node = {}
    pd = (ks // 2, ks // 2)
    run = lambda: Convolution.launch(node, x, wt, (1, 1), pd, pd, 'explicit', bias=b, act=('relu',))
    for _ in range(3):
        run()
    dev.synchronize()
    out = (ctypes.c_ulonglong * 128)()
    lib.pvhip_diag_wino4s_stamps(out)
    lib.pvhip_diag_wino4s_simd((ctypes.c_ulonglong * 64)())
    e0 = dev.Event().record()
    for _ in range(5):
        run()
    e1 = dev.Event().record(); e1.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    lib.pvhip_diag_wino4s_stamps(out)
    st = np.array(list(out), dtype=np.float64).reshape(16, 8)
    sim = (ctypes.c_ulonglong * 64)()
    lib.pvhip_diag_wino4s_simd(sim)
    sim = np.array(list(sim)).reshape(16, 4)
    print('{}: {:.3f} ms (stamped build); SIMD of waves 0..15: {}'.format(name, ms, ' | '.join(','.join(str(int(v)) for v in sim[w_]) for w_ in range(16))))
    young = os.environ.get('PVHIP_WINO_SHARED_OLD') == '0'
    for wv0 in range(16):
        wv = wv0
        wg = max(st[wv, 7], 1.0)
        is_cons = (wv < 12) if young else (wv >= 4)
        cw = wv if young else wv - 4
        pw = wv - 12 if young else wv
        if is_cons:
            stages, tiles = max(st[wv, 5], 1.0), max(st[wv, 6], 1.0)
            print('  consumer g{} row {}: per stage: wait {:6.0f}  reads+MFMA {:6.0f} | per tile: epilogue work {:6.0f}  counter barriers {:6.0f} | stages/tile {:.0f}  tiles/wg {:.1f}  life {:.0f}'.format(
                cw // 6, cw % 6, st[wv, 0] / stages, st[wv, 1] / stages, st[wv, 2] / tiles, st[wv, 3] / tiles, stages / tiles, tiles / wg, st[wv, 4] / wg))
        else:
            cref = 0 if young else 4
            own = max(st[wv, 7], 1.0) * (c // 4) / 2.0 * max(st[cref, 6] / max(st[cref, 7], 1.0), 1.0) * 0.75     # own stages in the traced part of the loop (3 of every 4)
            print('  producer pair {} wave {}: per own stage: wait for the gathered data {:6.0f}  for a free buffer {:6.0f}  transform+writes {:6.0f}  gather issue {:6.0f}  one poll: LDS read {:.0f} cycles, v_readfirstlane {:.0f} | life {:.0f}'.format(
                pw // 2, pw % 2, st[wv, 3] / own, st[wv, 0] / own, st[wv, 1] / own, st[wv, 2] / own, st[wv, 6] / own, st[wv, 5] / own, st[wv, 4] / wg))
    if os.environ.get('TRACE'):
        lib.pvhip_diag_wino4s_trace.argtypes = [ctypes.c_void_p]
        tr = (ctypes.c_uint * (16 * 96 * 4))()
        lib.pvhip_diag_wino4s_trace(tr)
        tr = np.array(list(tr), dtype=np.int64).reshape(16, 96, 4)
        n_st = c // 4
        print('  trace of workgroup 3 (cycles since its start; consumers: saw image / finished; producers: began writing / signalled), stages 0..{}'.format(min(95, 2 * n_st + 7)))
        for q in range(min(96, 2 * n_st + 8)):
            pr = [w_ for w_ in range(12, 16) if tr[w_, q, 3] > 0]
            line = '  q{:3d} b{} | producers {} | g0 saw {:7d}..{:7d} done {:7d}..{:7d} | g1 saw {:7d}..{:7d} done {:7d}..{:7d}'.format(
                q, q & 3, ' '.join('{}:[{}] poll {} -> got {} -> signalled {}'.format(w_, tr[w_, q, 0], tr[w_, q, 1], tr[w_, q, 2], tr[w_, q, 3]) for w_ in pr) or '(tile boundary: not traced)',
                tr[0:6, q, 0].min(), tr[0:6, q, 0].max(), tr[0:6, q, 1].min(), tr[0:6, q, 1].max(),
                tr[6:12, q, 0].min(), tr[6:12, q, 0].max(), tr[6:12, q, 1].min(), tr[6:12, q, 1].max())
            print(line)
