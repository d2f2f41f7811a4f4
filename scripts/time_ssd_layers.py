#!/usr/bin/env python3
"""GPU-box tool: SSD-MobileNet (BASELINE config 5) at batch 128: device time per launch of the backbone's GroupConvolution (depthwise)
and Convolution nodes with the bytes each moves.  python scripts/time_ssd_layers.py [batch]"""
import os, sys
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
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (B, 3, 300, 300)))
name = net.inputs[0]['name']
for _ in range(2):
    ex.infer({name: x})
ex.device_timing = 'all'
ex.infer({name: x})
G = net.G
tot = {}
for nid, typ, nm, ms in ex.device_times_ms():
    node = G.nodes[nid]
    tot[typ] = tot.get(typ, 0.0) + ms
    if typ in ('GroupConvolution', 'Convolution'):
        ins = [node['input'][p]['dims'] for p in sorted(node['input'])]
        out = next(iter(node['output'].values()))['dims']
        mb = 4.0 * (np.prod(ins[0]) + np.prod(out)) / 1e6
        print('{:58s} {:17s} in {} w {} -> {:.4f} ms  {:6.0f} GB/s'.format(nm[:58], typ, tuple(ins[0]), tuple(ins[1]), ms, mb / ms))
print({k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}, 'total', round(sum(tot.values()), 3))
