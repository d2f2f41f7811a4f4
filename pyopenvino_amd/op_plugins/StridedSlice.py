# StridedSlice -- HIP plugin.  Replaces kernel_StridedSlice_naive (reference op_plugins/StridedSlice.py:9-26):
# x[b0:e0:s0, b1:e1:s1, ...] over the leading len(begin) axes; the five mask attributes are read and ignored there.
# In the shipped IRs it only ever slices ShapeOf vectors (host integers), which is what this plugin implements.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('StridedSlice')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = inputs[0]
    if isinstance(x, (dev.DeviceTensor, dev.ChannelSlice)):
        raise NotImplementedError('StridedSlice of a device-resident tensor ({}): only shape vectors are sliced in the '
                                  'IRs of this path'.format(node['name']))
    x = np.asarray(x)
    begin, end, stride = (np.asarray(inputs[p]).ravel() for p in (1, 2, 3))
    index = tuple(slice(int(b), int(e), int(s)) for b, e, s in zip(begin[:x.ndim], end[:x.ndim], stride[:x.ndim]))
    return {common_def.first_output_port(node): x[index]}
