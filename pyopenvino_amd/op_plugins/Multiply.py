# Multiply -- HIP plugin.  Replaces kernel_Multiply_numpy (reference op_plugins/Multiply.py:9-17):
# the operand with fewer elements is broadcast to the other's shape.  (The reference's compute()
# always ends in the numpy kernel whatever kernel_type says, Multiply.py:46-62.)
from .. import common_def
from .. import device as dev
from . import _broadcast


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('Multiply')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    a = dev.as_device(inputs[0])
    b = dev.as_device(inputs[1])
    out_shape = a.shape if a.size > b.size else b.shape
    res = _broadcast.launch('pvhip_mul_f32', a, b, out_shape)
    return {common_def.first_output_port(node): res}
