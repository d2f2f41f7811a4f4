import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from pyopenvino_amd import IECore, device, synth
device.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
ie = IECore()
if which in ('all', 'mnist'):
    net = ie.read_network(os.path.join(bench.REPO, 'models', 'mnist.xml')); net.set_batch(64)
    ex = ie.load_network(net)
    x = device.DeviceTensor.from_numpy(np.concatenate([synth.uniform_pixels(50 + i, (1, 1, 28, 28)) for i in range(64)], 0))
    for i in range(8):
        r = ex.infer({net.inputs[0]['name']: x}); print('mnist', i, ex.__dict__.get('_auto_graph'), flush=True)
    del ex, net
if which in ('all', 'ssd'):
    xml = os.path.join(bench.REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
    net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234)); net.set_batch(128)
    ex = ie.load_network(net)
    x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (128, 3, 300, 300)))
    for i in range(8):
        r = ex.infer({net.inputs[0]['name']: x}); print('ssd', i, ex.__dict__.get('_auto_graph'), flush=True)
print('done')
