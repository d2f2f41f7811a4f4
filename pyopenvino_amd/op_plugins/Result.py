# Result -- HIP plugin.  Replaces reference op_plugins/Result.py:7-18: validates and stores its input
# as node['result'].  This is where the tensor leaves HBM (device-to-host copy, synchronises the
# stream).  When the batch is sharded over ranks (node['comm'] set by the engine) the shards' Result
# tensors are all-gathered over RCCL first, so every rank returns the whole batch.
import numpy as np

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Result')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    value = inputs[0]
    comm = node.get('comm')
    if comm is not None and comm.world > 1:
        value = comm.allgather_rows(value)
    if node.get('_async') and isinstance(value, dev.DeviceTensor):
        node['result'] = value          # an asynchronous request: InferRequest.wait() copies it to the host
    else:
        node['result'] = value.numpy() if isinstance(value, dev.DeviceTensor) else np.asarray(value)
    return []
