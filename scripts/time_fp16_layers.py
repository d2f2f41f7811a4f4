"""GPU-box tool: GoogLeNet as an FP16 IR (f16-MFMA kernels), batch 256: device time per Convolution launch, grouped by window."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyopenvino_amd import IECore, device, synth
device.init(0)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
xml = os.path.join(REPO, 'models', 'googlenet-v1.xml')
blob = synth.synth_weights(xml, 1234)
ie = IECore()
with tempfile.TemporaryDirectory() as tmp:
    xml16, blob16 = synth.fp16_ir(xml, blob, tmp)
    net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
net.set_batch(256)
ex = ie.load_network(net)
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1000, (256, 3, 224, 224)))
feed = {net.inputs[0]['name']: x}
os.environ['PVHIP_AUTO_GRAPH'] = '0'
for _ in range(3):
    ex.infer(feed)
ex.device_timing, ex.compute_streams = 'all', 1
acc = {}
for _ in range(3):
    ex.infer(feed)
    for nid, typ, name, ms in ex.device_times_ms():
        acc.setdefault((nid, typ, name), []).append(ms)
groups = {}
tot = 0.0
for (nid, typ, name), v in sorted(acc.items()):
    ms = min(v)
    if typ in ('Const', 'Parameter', 'Reshape'):
        continue
    key = typ
    if typ == 'Convolution':
        k = net.G.nodes[nid]['input'][1]['dims']
        key = 'Convolution {}x{}'.format(k[2], k[3])
        print('{:45s} {:14s} {:.4f} ms'.format(name[:45], key, ms))
    groups[key] = groups.get(key, 0.0) + ms
    tot += ms
print({k: round(v, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])}, 'total', round(tot, 3))
