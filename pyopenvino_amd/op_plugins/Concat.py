# Concat -- HIP plugin.  Replaces kernel_Concat_numpy (reference op_plugins/Concat.py:9-13):
# np.concatenate(list(inputs.values()), axis) -- inputs are taken in dict (edge) order.
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Concat')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    axis = int(node['data']['axis'])
    assert len(inputs) > 1
    assert axis <= inputs[0].ndim
    parts = [dev.as_device(t) for t in inputs.values()]
    if len(parts) > dev.MAX_CONCAT:
        raise NotImplementedError('Concat of {} inputs (> {})'.format(len(parts), dev.MAX_CONCAT))
    first = parts[0].shape
    rank = len(first)
    if axis < 0:
        axis += rank
    if not 0 <= axis < rank:
        raise ValueError('axis {} is out of bounds for array of dimension {}'.format(axis, rank))
    for p in parts[1:]:
        if len(p.shape) != rank or any(p.shape[d] != first[d] for d in range(rank) if d != axis):
            raise ValueError('all the input array dimensions except for the concatenation axis must match exactly')
    outer = 1
    for d in first[:axis]:
        outer *= d
    tail = 1
    for d in first[axis + 1:]:
        tail *= d
    inner = [p.shape[axis] * tail for p in parts]
    out_shape = list(first)
    out_shape[axis] = sum(p.shape[axis] for p in parts)
    y = dev.DeviceTensor.empty(out_shape)
    srcs = (ctypes.c_void_p * len(parts))(*[p.ptr for p in parts])
    dev.call('pvhip_concat_f32', len(parts), srcs, dev.i64_array(inner), ctypes.c_void_p(y.ptr), outer)
    return {common_def.first_output_port(node): y}
