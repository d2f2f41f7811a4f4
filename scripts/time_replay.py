"""GPU-box tool: googlenet-v1 fp32 batch 256, one infer() at a time: eager dispatch and hipGraph replay for 1..4 compute streams."""
import os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pyopenvino_amd import IECore, device, synth
device.init(0)
ie = IECore()
xml = os.path.join(os.getcwd(), 'models', 'googlenet-v1.xml')
net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234)); net.set_batch(256)
ex = ie.load_network(net)
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1000, (256, 3, 224, 224)))
name = net.inputs[0]['name']
os.environ['PVHIP_AUTO_GRAPH'] = '0'
def med(fn, reps=15):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        device.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3
for n in (1, 2, 3, 4):
    ex.compute_streams = n
    plan = ex.plan_streams()
    used = sorted(set(plan[0].values())) if plan else [0]
    e = med(lambda: ex.infer({name: x}))
    ex.capture_graph({name: x}, streams='plan')
    g = med(lambda: ex.infer_graph())
    ex.release_graph()
    print('compute_streams {} (plan uses {}): eager {:.3f} ms = {:.0f} img/s, replay {:.3f} ms = {:.0f} img/s'.format(n, used, e, 256e3 / e, g, 256e3 / g), flush=True)
