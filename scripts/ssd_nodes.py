import os, sys
sys.path.insert(0, '/root/repo')
from pyopenvino_amd import IECore, device, synth
device.init(0)
xml = '/root/repo/models/ssd_mobilenet_v1_coco.xml'
ie = IECore()
net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
net.set_batch(128)
ex = ie.load_network(net)
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (128, 3, 300, 300)))
feed = {net.inputs[0]['name']: x}
for _ in range(3): ex.infer(feed)
ex.device_timing, ex.compute_streams = 'all', 1
ex.infer(feed)
rows = [(t, typ, nm) for nid, typ, nm, t in ex.device_times_ms() if typ not in ('Const', 'Parameter', 'Reshape')]
tot = sum(r[0] for r in rows)
print('total', round(tot, 3))
for t, typ, nm in sorted(rows, reverse=True)[:40]:
    print('{:8.4f} {:18s} {}'.format(t, typ, nm[:70]))
