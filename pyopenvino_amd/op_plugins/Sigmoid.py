# Sigmoid -- HIP plugin.  Replaces kernel_Sigmoid_numpy (reference op_plugins/Sigmoid.py:10-13).
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('Sigmoid')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = dev.as_device(inputs[0])
    y = dev.DeviceTensor.empty(x.shape)
    dev.call('pvhip_sigmoid_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), x.size)
    return {common_def.first_output_port(node): y}
