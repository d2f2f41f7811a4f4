# Reshape -- HIP plugin.  Replaces kernel_Reshape_numpy (reference op_plugins/Reshape.py:14-44): target
# dims with 0 (copy the input dim, left aligned only) and one -1 (inferred); `special_zero` is not read,
# as in the reference.  Metadata only: the result shares the device block of its input.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Reshape')


def resolve_dims(in_shape, target):
    remaining = 1
    for d in in_shape:
        remaining *= int(d)
    dims, deferred, zeros_allowed = [], -1, True
    for idx, dim in enumerate(int(t) for t in target):
        if dim == 0:
            assert zeros_allowed            # zeros must be left aligned
            src = int(in_shape[idx])
            assert remaining % src == 0
            dims.append(src)
            remaining //= src
        else:
            zeros_allowed = False
            if dim == -1:
                assert deferred == -1       # at most one -1
                deferred = idx
                dims.append(-1)
            else:
                assert remaining % dim == 0
                dims.append(dim)
                remaining //= dim
    if deferred != -1:
        dims[deferred] = remaining
    return dims


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    x = dev.as_device(inputs[0])
    dims = resolve_dims(x.shape, np.asarray(inputs[1]).ravel())
    return {common_def.first_output_port(node): x.reshape(dims)}
