# Add -- HIP plugin.  Replaces kernel_Add_numpy (reference op_plugins/Add.py:9-14):
# input1 is broadcast to input0's shape (only that direction, as the reference does).
from .. import common_def
from .. import device as dev
from . import _broadcast


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Add')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    a = dev.as_device(inputs[0])
    b = dev.as_device(inputs[1])
    res = _broadcast.launch('pvhip_add_f32', a, b, a.shape)
    return {common_def.first_output_port(node): res}
