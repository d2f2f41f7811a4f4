#!/usr/bin/env python3
"""GPU-box tool: per-node device time of GoogLeNet as an FP16 IR (fp16_as_fp32=False, batch 256; one hipEvent bracket per launch on one stream), with the
algorithmic TFLOP/s of the convolution launches and the kernel each ran on."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from pyopenvino_amd import device, synth
device.init(0)
blob = synth.synth_weights(os.path.join(REPO, 'models', bench.MODEL + '.xml'), bench.WEIGHT_SEED)
net, ex, feed = bench.fp16_network(blob)
for _ in range(3): ex.infer(feed)
ex.device_timing, ex.compute_streams = 'all', 1
ex.infer(feed)
work = bench.collect_work(net)
rows = [(t, typ, nm, nid) for nid, typ, nm, t in ex.device_times_ms() if typ not in ('Const', 'Parameter', 'Reshape')]
print('total {:.3f} ms'.format(sum(r[0] for r in rows)))
for t, typ, nm, nid in sorted(rows, reverse=True)[:48]:
    fl = work.get(nid, (0, 0))[0]
    members = [nid] + [m for m in net.G.nodes if net.G.nodes[m].get('_launched_by') == nid]
    print('{:8.4f} {:12s} {:52s} {:7.1f} TF  {}'.format(t, typ, nm[:52], fl / (t * 1e-3) / 1e12 if t > 0 else 0.0, str(net.G.nodes[nid].get('_hip_f16', ''))[:60]))
