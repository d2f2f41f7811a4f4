# Parameter -- HIP plugin.  Replaces reference op_plugins/Parameter.py:8-14: the user's array is
# reshaped to the IR shape, cast to the IR element type and uploaded to HBM.  A DeviceTensor passed by
# the caller (input already resident on the GPU) is used as is.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Parameter')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    shape = node['data']['shape']
    precision = common_def.type_convert_tbl[node['data']['element_type']]
    param = node['param']
    if isinstance(param, dev.DeviceTensor):
        if param.dtype != np.dtype(precision):
            raise TypeError('device-resident input must already be {}'.format(np.dtype(precision).name))
        return {0: param.reshape(shape)}
    # no host-side copies unless the caller's array needs one (the reference copies twice, Parameter.py:11-13; at
    # (256,3,224,224) fp32 each copy costs more than the upload)
    host = np.asarray(param).reshape(shape).astype(precision, copy=False)
    if host.dtype != np.float32:
        return {0: host}
    return {0: dev.DeviceTensor.from_numpy(host)}
