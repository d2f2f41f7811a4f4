# Unsqueeze -- HIP plugin.  Replaces kernel_Unsqueeze_numpy (reference op_plugins/Unsqueeze.py:10-15):
# np.expand_dims(x, axes).  Metadata only: a device tensor keeps its block.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('Unsqueeze')


def expanded_shape(shape, axes):
    rank = len(shape) + len(axes)
    axes = sorted(int(a) + rank if int(a) < 0 else int(a) for a in axes)
    if len(set(axes)) != len(axes) or any(not 0 <= a < rank for a in axes):
        raise ValueError('bad axes {} for an input of rank {}'.format(list(axes), len(shape)))
    rest = iter(shape)
    return tuple(1 if i in axes else next(rest) for i in range(rank))


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = inputs[0]
    shape = expanded_shape(tuple(x.shape), np.asarray(inputs[1]).ravel())
    if isinstance(x, dev.DeviceTensor):
        return {common_def.first_output_port(node): x.reshape(shape)}
    return {common_def.first_output_port(node): np.asarray(x).reshape(shape)}
