# Transpose -- HIP plugin.  Replaces kernel_Transpose_numpy (reference op_plugins/Transpose.py:9-13).
# The reference returns a strided view; here the permutation is materialised in HBM (consumers are
# kernels that want dense tensors).  The axes operand (port 1, I64) stays on the host.
import ctypes

import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Transpose')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = dev.as_device(inputs[0])
    axes = [int(a) for a in np.asarray(inputs[1]).ravel()]
    rank = x.ndim
    axes = [a + rank if a < 0 else a for a in axes]
    if sorted(axes) != list(range(rank)):
        raise ValueError("axes don't match array")
    if rank > dev.MAX_RANK:
        raise NotImplementedError('rank {} > {}'.format(rank, dev.MAX_RANK))
    y = dev.DeviceTensor.empty([x.shape[a] for a in axes])
    dev.call('pvhip_transpose_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), rank, dev.i64_array(x.shape),
             dev.i64_array(axes))
    return {common_def.first_output_port(node): y}
