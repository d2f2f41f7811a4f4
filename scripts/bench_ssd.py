#!/usr/bin/env python3
"""GPU-box tool: SSD-MobileNet (BASELINE config 5), batch 128: the whole IR through infer() (prior boxes constant
folded, DetectionOutput on the device, detections copied to the host) and the backbone + heads alone; images/s and
per-op device time."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import IECore, device, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device.init(0)
xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
ie = IECore()
net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
net.set_batch(B)
ex = ie.load_network(net)
heads = ['concat', 'concat_1', 'do_ExpandDims_conf/sigmoid']
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (B, 3, 300, 300)))
name, out_name = net.inputs[0]['name'], net.outputs[0]['name']


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    device.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    device.synchronize()
    return (time.perf_counter() - t0) / reps


dt = timed(lambda: ex.infer({name: x}))
det = ex.infer({name: x})[out_name]
print('ssd whole IR batch {}: {:.2f} ms/step, {:.0f} images/s; detections {} ({} records in image 0)'.format(
    B, dt * 1e3, B / dt, det.shape, int((det[0, 0, :100, 0] >= 0).sum())))
dt = timed(lambda: ex.infer_until({name: x}, heads))
print('ssd backbone + heads batch {}: {:.2f} ms/step, {:.0f} images/s'.format(B, dt * 1e3, B / dt))
ex.device_timing = 'all'
ex.compute_streams = 1
ex.infer({name: x})
agg = {}
for nid, typ, nm, ms in ex.device_times_ms():
    agg[typ] = agg.get(typ, 0.0) + ms
for typ, ms in sorted(agg.items(), key=lambda kv: -kv[1])[:9]:
    print('  {:18s} {:.3f} ms'.format(typ, ms))
