#!/usr/bin/env python3
"""GPU-box tool (diagnostic build, `make -C pyopenvino_amd/csrc diag`): where the waves of conv_wino4_kernel<4> spend their cycles.
Runs the 3x3 layers with PVHIP_WINO4_ABLATE=5 (s_memtime stamps around the segments of a stage, every 61st workgroup) and prints
cycles per stage and wave.  The stamps cost time themselves (each is an s_memtime + lgkmcnt(0)): read the split, not the total.
  python scripts/stamps_wino4.py [substring of the layer name]"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution

LAYERS = [('conv2/3x3', (256, 64, 56, 56), 192), ('3a/3x3', (256, 96, 28, 28), 128), ('3b/3x3', (256, 128, 28, 28), 192)]
dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_wino4_stamps.argtypes = [ctypes.c_void_p]
lib.pvhip_diag_wino4_hw.argtypes = [ctypes.c_void_p]
lib.pvhip_diag_wino4_epilogue.argtypes = [ctypes.c_void_p]
only = sys.argv[1] if len(sys.argv) > 1 else ''
for name, xs, k in LAYERS:
    if only not in name:
        continue
    n, c, h, w = xs
    x = dev.DeviceTensor.from_numpy(synth.normal(1, 2, n * c * h * w).astype(np.float32).reshape(xs))
    wt = dev.DeviceTensor.from_numpy((synth.normal(3, 4, k * c * 9) * (2.0 / (c * 9)) ** 0.5).astype(np.float32).reshape((k, c, 3, 3)))
    b = dev.DeviceTensor.from_numpy(np.zeros((1, k, 1, 1), dtype=np.float32))
    for abl in ('0', '5'):
        os.environ['PVHIP_CONV_WINOGRAD4'] = 'force'
        os.environ['PVHIP_WINO4_ABLATE'] = abl
        dev.reload_settings()
        node = {}
        run = lambda: Convolution.launch(node, x, wt, (1, 1), (1, 1), (1, 1), 'explicit', bias=b, act=('relu',))
        for _ in range(3):
            run()
        dev.synchronize()
        out = (ctypes.c_ulonglong * 64)()
        lib.pvhip_diag_wino4_stamps(out)          # clear
        lib.pvhip_diag_wino4_epilogue(out)
        e0 = dev.Event().record()
        for _ in range(5):
            run()
        e1 = dev.Event().record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        lib.pvhip_diag_wino4_stamps(out)
        st = np.array(list(out), dtype=np.float64).reshape(8, 8)
        lib.pvhip_diag_wino4_epilogue(out)
        epi_ph = np.array(list(out), dtype=np.float64).reshape(8, 8)
        print('{} ablate={}: {:.3f} ms'.format(name, abl, ms), flush=True)
        if abl == '5' and os.environ.get('HW'):
            hw = (ctypes.c_uint * (64 * 8 * 2))()
            lib.pvhip_diag_wino4_hw(hw)
            hw = np.array(list(hw), dtype=np.uint32).reshape(64, 8, 2)
            for k_ in range(24):
                print('  wg ticket {:2d}: LDS_ALLOC {}  HW_ID simd {} cu {} sh {} se {} wave {}'.format(
                    k_, ' '.join('%08x' % v for v in sorted(set(hw[k_, :, 0].tolist()))),
                    [(int(v) >> 4) & 3 for v in hw[k_, :, 1]], sorted(set((int(v) >> 8) & 15 for v in hw[k_, :, 1])),
                    sorted(set((int(v) >> 12) & 1 for v in hw[k_, :, 1])), sorted(set((int(v) >> 13) & 7 for v in hw[k_, :, 1])),
                    [int(v) & 15 for v in hw[k_, :, 1]]))
        if abl == '5':
            stages = c // 4
            tiles = ((n * (h // 4) * (w // 4) + 31) // 32) * ((k + 31) // 32)
            per_wg = tiles / min(tiles, 512)
            for wv in range(8):
                cnt = st[wv, 7]
                if cnt == 0:
                    continue
                per = st[wv, :4] / cnt / stages / per_wg
                life, epi, head = st[wv, 4] / cnt, st[wv, 5] / cnt, st[wv, 6] / cnt
                if wv < 6:
                    print('  consumer {}: per stage  MFMA segment {:5.0f}  U wait {:4.0f}  barrier {:5.0f} | per tile: head {:6.0f}  main loop {:7.0f}  epilogue {:6.0f} | life {:8.0f} cycles = {:.1f} us: {:.2f} GHz, {:.1f} tiles'.format(
                        wv, per[0], per[1], per[2], head / per_wg, st[wv, :3].sum() / cnt / per_wg, epi / per_wg, life, st[wv, 3] / cnt / 100.0, life / max(1.0, st[wv, 3] / cnt) / 10.0, per_wg))
                else:
                    print('  producer {}: per stage (all but the last two of a tile)  gather issue {:4.0f}  gather wait {:4.0f}  transform+store {:5.0f}  barrier {:5.0f} | before the first tile {:6.0f}  epilogues per tile {:6.0f} | life {:8.0f}'.format(
                        wv - 6, per[0] * stages / max(1, stages - 2), per[1] * stages / max(1, stages - 2), per[2] * stages / max(1, stages - 2), per[3] * stages / max(1, stages - 2), head, epi / per_wg, life))
            print('  epilogue phases, cycles per tile: write 0 | barrier | read + store 0 | barrier | write 1 | barrier | read + store 1 | barrier')
            for wv in range(8):
                cnt = st[wv, 7]
                if cnt == 0:
                    continue
                print('    {} {}: '.format('consumer' if wv < 6 else 'producer', wv if wv < 6 else wv - 6) +
                      ' | '.join('{:5.0f}'.format(v) for v in epi_ph[wv] / cnt / per_wg))
