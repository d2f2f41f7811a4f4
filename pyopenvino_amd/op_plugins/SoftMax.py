# SoftMax -- HIP plugin.  Replaces kernel_SoftMax_numpy (reference op_plugins/SoftMax.py:10-14):
# exp(x)/sum(exp(x)), fp32, no max shift, 'axis' attribute unused.  The reference normalises over the
# whole tensor, which at its only supported batch (N=1) is one image; here every leading-axis slice is
# one independent image, so the normalisation runs per slice (identical at N=1).
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('SoftMax')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = dev.as_device(inputs[0])
    rows = x.shape[0] if x.ndim > 1 else 1
    cols = x.size // rows if rows else 0
    y = dev.DeviceTensor.empty(x.shape)
    dev.call('pvhip_softmax_rows_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), rows, cols)
    return {common_def.first_output_port(node): y}
