#!/usr/bin/env python3
"""GPU-box tool: SSD-MobileNet backbone + heads (BASELINE config 5), batch 128, images/s and per-op device time."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import IECore, device, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device.init(0)
xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
ie = IECore()
net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
net.set_batch(B)
ex = ie.load_network(net)
heads = ['concat', 'concat_1', 'do_ExpandDims_conf/sigmoid']
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (B, 3, 300, 300)))
name = net.inputs[0]['name']
for _ in range(3):
    ex.infer_until({name: x}, heads)
device.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    ex.infer_until({name: x}, heads)
device.synchronize()
dt = (time.perf_counter() - t0) / K
print('ssd backbone batch {}: {:.2f} ms/step, {:.0f} images/s'.format(B, dt * 1e3, B / dt))
ex.device_timing = 'all'
ex.infer_until({name: x}, heads)
agg = {}
for nid, typ, nm, ms in ex.device_times_ms():
    agg[typ] = agg.get(typ, 0.0) + ms
for typ, ms in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
    print('  {:18s} {:.3f} ms'.format(typ, ms))
