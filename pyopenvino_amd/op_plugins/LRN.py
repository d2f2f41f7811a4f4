# LRN -- HIP plugin.  Replaces kernel_LRN_numpy (reference op_plugins/LRN.py:10-22): cross-channel
# window of `size`, alpha NOT divided by size; the axes input (port 1) is validated and unused.
import ctypes

from .. import common_def
from .. import device as dev


def name():
    print('LRN')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    attrs = node['data']
    alpha = float(attrs['alpha'])
    beta = float(attrs['beta'])
    bias = float(attrs['bias'])
    size = int(attrs['size'])
    x = dev.as_device(inputs[0])
    n, c, h, w = x.shape
    y = dev.DeviceTensor.empty(x.shape)
    dev.call('pvhip_lrn_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h * w, size, alpha, beta, bias)
    return {common_def.first_output_port(node): y}
