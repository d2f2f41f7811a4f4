# Const -- HIP plugin.  Replaces reference op_plugins/Const.py:8-14.  fp32 constants are uploaded to
# HBM once and the same DeviceTensor is returned on every infer (the reference rebuilds an ndarray from
# a tuple each time); integer constants (shapes, axes, permutations) stay on the host as ndarrays.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Const')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    cached = node.get('_hip_const')
    if cached is not None:
        return {0: cached}
    shape = node['data']['shape']
    precision = common_def.type_convert_tbl[node['data']['element_type']]
    host = np.asarray(node['const']['data'], dtype=precision).reshape(shape)
    value = dev.DeviceTensor.from_numpy(host) if host.dtype == np.float32 else host
    node['_hip_const'] = value
    return {0: value}
