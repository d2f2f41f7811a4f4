# Clamp -- HIP plugin.  Replaces kernel_Clamp_numpy (reference op_plugins/Clamp.py:9-12).
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('Clamp')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    max_val = float(node['data']['max'])
    min_val = float(node['data']['min'])
    x = dev.as_device(inputs[0])
    y = dev.DeviceTensor.empty(x.shape)
    dev.call('pvhip_clamp_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), x.size, min_val, max_val)
    return {common_def.first_output_port(node): y}
