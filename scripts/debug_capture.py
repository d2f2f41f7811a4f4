#!/usr/bin/env python3
"""GPU-box tool: capture a forward pass of mnist node by node and report the first node after which the capture is invalid."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from pyopenvino_amd import IECore, device, synth
device.init(0)
ie = IECore()
net = ie.read_network(os.path.join(REPO, 'models', 'mnist.xml'))
net.set_batch(8)
ex = ie.load_network(net)
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1, (8, 1, 28, 28)))
name = net.inputs[0]['name']
for _ in range(2):
    ex.infer({name: x})
reg = ie.plugins.plugins
def status():
    s = ctypes.c_int(0); device.call('pvhip_graph_capture_status', ctypes.byref(s)); return s.value
for typ, mod in list(reg.items()):
    orig = mod.compute
    def wrapped(node, inputs=None, kernel_type='hip', debug=False, _o=orig, _t=typ):
        before = status()
        r = _o(node, inputs, kernel_type=kernel_type, debug=debug)
        after = status()
        if before != after:
            print('capture status {} -> {} in {} ({})'.format(before, after, _t, node.get('name')), flush=True)
        return r
    mod.compute = wrapped
_call = device.call
state = {'last': 0, 'busy': False}
def traced(fname, *args):
    r = _call(fname, *args)
    if not state['busy'] and fname != 'pvhip_graph_capture_status':
        state['busy'] = True
        try:
            st = status()
        finally:
            state['busy'] = False
        if st != state['last'] or (state['last'] == 1 and state.setdefault("n", 0) < 400):
            state['n'] = state.get('n', 0) + 1
            print('status {} -> {} after {} {}'.format(state['last'], st, fname, [a.value if hasattr(a, 'value') else a for a in args][:3]), flush=True)
            state['last'] = st
    return r
device.call = traced
import pyopenvino_amd.inference_engine as eng
try:
    ex.capture_graph({name: x})
    print('captured ok; replay:', ex.infer_graph()[net.outputs[0]['name']].shape)
except Exception as e:
    print('FAILED', e)
