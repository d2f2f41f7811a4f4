#!/usr/bin/env python3
"""GPU-box tool: one padded c-major convolution with the window test in the gather, as padding pass + test-free gather, and with the
Add folded into the pass: where do the outputs differ?  python scripts/repro_prepad.py N C H W K KS STRIDE PT PL PB PR [seed]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution
dev.init(0)
n, c, h, w, k, ks, st, pt, pl, pb_, pr = [int(v) for v in sys.argv[1:12]]
i = int(sys.argv[12]) if len(sys.argv) > 12 else 1
x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 20.0
wt = (synth.normal(i, 3, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks))
addc = synth.normal(i, 5, c).astype(np.float32).reshape((1, c, 1, 1)) * 50.0
b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
outs = []
for env, pre in (({'PVHIP_CONV_PREPAD': '0'}, False), ({'PVHIP_CONV_PREPAD': '1'}, False), ({'PVHIP_CONV_PREPAD': '1'}, True)):
    os.environ.update(env); dev.reload_settings()
    nd = {'_pre_add': dev.DeviceTensor.from_numpy(addc)} if pre else {}
    xin = dev.DeviceTensor.from_numpy((x + addc).astype(np.float32) if not pre else x)
    outs.append(np.asarray(Convolution.launch(nd, xin, dev.DeviceTensor.from_numpy(wt), (st, st), (pt, pl), (pb_, pr), 'explicit', bias=b, act=('relu',))))
    print(env, pre, 'route', nd.get('_hip_route'))
for j in (1, 2):
    d = outs[j].view(np.uint32) != outs[0].view(np.uint32)
    print('variant', j, 'differs in', int(d.sum()), 'of', d.size, 'max |diff|', float(np.abs(outs[j] - outs[0]).max()))
    if d.any():
        idx = np.argwhere(d)
        print('  first', idx[:5].tolist(), 'last', idx[-3:].tolist(), 'n range', idx[:, 0].min(), idx[:, 0].max(), 'k range', idx[:, 1].min(), idx[:, 1].max(),
              'rows', idx[:, 2].min(), idx[:, 2].max(), 'cols', idx[:, 3].min(), idx[:, 3].max())
