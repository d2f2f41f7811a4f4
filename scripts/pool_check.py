#!/usr/bin/env python3
"""GPU-box tool: device memory the pool holds while eight whole-batch GoogLeNet requests replay their own recorded passes
(load_network(num_requests=8), batch 256): live and cached GB per round, and how many requests have a recording.
  python scripts/pool_check.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from pyopenvino_amd import IECore, synth, device
xml = os.path.join(REPO, 'models', 'googlenet-v1.xml')
ie = IECore(); net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234)); net.set_batch(256)
ex = ie.load_network(net, 'GPU', num_requests=8)
name = net.inputs[0]['name']
xs = [device.DeviceTensor.from_numpy(synth.uniform_pixels(100 + i, (256, 3, 224, 224))) for i in range(8)]
for rnd in range(5):
    for i in range(8): ex.start_async(i, {name: xs[i]})
    for i in range(8): ex.wait(i)
    print('round', rnd, 'pool (live, cached) GB:', [round(v / 2**30, 1) for v in device.pool_stats()], 'graphs', sum(1 for r in ex.requests if r.runner.__dict__.get('_graph')), flush=True)
