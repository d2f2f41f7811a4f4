#!/usr/bin/env python3
"""GPU-box check: ssd_mobilenet_v1_coco as an FP16 IR (synthetic weights) with fp16_as_fp32=False (f16 matrix cores; blocked fp16 tensors where the
plan finds them) against fp16_as_fp32=True (fp32 arithmetic on the same constants): no exception, finite outputs, the raw head tensors close."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyopenvino_amd import IECore, device, synth
device.init(0)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
blob = synth.synth_weights(xml, 1234)
ie = IECore()
x = synth.uniform_pixels(9, (2, 3, 300, 300))
outs = {}
for mode in (True, False):
    with tempfile.TemporaryDirectory() as tmp:
        xml16, blob16 = synth.fp16_ir(xml, blob, tmp)
        net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=mode)
    net.set_batch(2)
    ex = ie.load_network(net)
    res = ex.infer({net.inputs[0]['name']: x})
    kinds = {}
    for n in net.G.nodes:
        k = net.G.nodes[n].get('_hip_f16')
        if k:
            kinds[k] = kinds.get(k, 0) + 1
    head = {}
    for n in net.G.nodes:
        if net.G.nodes[n]['type'] in ('Sigmoid',) or net.G.nodes[n]['name'] in ('concat', 'concat_1'):
            head[net.G.nodes[n]['name']] = np.asarray(next(iter(net.G.nodes[n]['output'].values()))['data'])
    outs[mode] = head
    print('fp16_as_fp32', mode, 'f16_mfma', net.f16_mfma, 'blocked outputs', len(getattr(ex, '_c8_out', ())), 'blocked concats', len(getattr(ex, '_c8_concat', ())), kinds,
          {k: (v.shape, bool(np.isfinite(v).all())) for k, v in head.items()})
for k in outs[True]:
    a, b = outs[True][k], outs[False][k]
    print(k, 'max-norm difference', float(np.abs(a - b).max() / max(1e-30, np.abs(a).max())))
