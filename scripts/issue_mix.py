#!/usr/bin/env python3
"""GPU-box tool (diagnostic build): what shares the SIMD's vector issue with what -- fp32 MFMA, bf16 MFMA and packed-fp32 vector
streams alone, interleaved in one wave, and on different waves of one SIMD (pvhip_diag_issue_mix).
  python scripts/issue_mix.py"""
import os, sys, ctypes
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev

dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
dev.init(0)
lib = ctypes.CDLL(dev.LIB_PATH)
lib.pvhip_diag_issue_mix.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
BLOCKS, ITERS = 256, 20000
out = dev.DeviceTensor.from_numpy(np.zeros(BLOCKS * 512, dtype=np.float32))

def run(mfma, valu, split):
    for _ in range(2):
        assert lib.pvhip_diag_issue_mix(out.ptr, mfma, valu, split, ITERS, BLOCKS) == 0
    dev.synchronize()
    e0 = dev.Event().record()
    assert lib.pvhip_diag_issue_mix(out.ptr, mfma, valu, split, ITERS, BLOCKS) == 0
    e1 = dev.Event().record(); e1.synchronize()
    return e0.elapsed_ms(e1)

NAMES = {0: 'no MFMA', 1: 'fp32 32x32x2 (8 per iteration)', 2: 'bf16 32x32x16 (16 per iteration)'}
VNAMES = {1: '64 v_pk_fma_f32', 2: '128 v_fma_f32 (the same arithmetic unpacked)', 3: '64 v_fma_f32', 4: '64 v_max3_f32', 5: '64 v_mov_b32 dpp wave_shr:1'}
print('256 workgroups x 8 waves (two waves per SIMD), {} iterations; per iteration: 8 fp32 MFMAs = 512 cycles or 16 bf16 MFMAs, and a vector stream'.format(ITERS))
for valu in (1, 2, 3, 4, 5):
    for split in (1, 0):
        print('--- vector stream: {}; '.format(VNAMES[valu]) + ('MFMAs on waves 0-3, vector stream on waves 4-7 (one of each per SIMD)' if split else 'every wave runs both, interleaved'))
        tv = run(0, valu, split)
        print('  vector stream alone              {:8.3f} ms  ({:.1f} cycles per iteration and wave at 2.4 GHz)'.format(tv, tv * 2.4e6 / ITERS))
        for mfma in (1, 2):
            tm = run(mfma, 0, split)
            tb = run(mfma, valu, split)
            flop = {1: 8 * 32 * 32 * 2 * 2, 2: 16 * 32 * 32 * 16 * 2}[mfma] * ITERS * BLOCKS * (4 if split else 8)
            print('  {:32s} alone {:8.3f} ms ({:7.1f} TFLOP/s)   with the vector stream {:8.3f} ms   sum {:8.3f}  max {:8.3f}'.format(
                NAMES[mfma], tm, flop / tm / 1e9, tb, tm + tv, max(tm, tv)), flush=True)
