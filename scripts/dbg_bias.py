import sys, os, numpy as np, importlib
sys.path.insert(0, os.getcwd())
from pyopenvino_amd import device as dev, synth
dev.init(0)
conv = importlib.import_module('pyopenvino_amd.op_plugins.Convolution')
xs, ws = (1, 32, 6, 6), (64, 32, 1, 1)
x = np.ones(xs, np.float32) * 0
w = np.zeros(ws, np.float32)
b = (np.arange(64, dtype=np.float32) + 1).reshape(1, 64, 1, 1)
node = {'name': 'c', 'type': 'Convolution', 'data': {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'},
        'input': {0: {'precision': 'FP32', 'dims': xs}, 1: {'precision': 'FP32', 'dims': ws}}, 'output': {2: {'precision': 'FP32', 'dims': (1, 64, 6, 6)}}}
node['_fuse_bias'] = dev.DeviceTensor.from_numpy(b)
node['_fuse_relu'] = False
y = np.asarray(conv.compute(node, {0: x, 1: w})[2])
print(y[0, :, 0, 0])
print(y[0, :, 3, 2])
