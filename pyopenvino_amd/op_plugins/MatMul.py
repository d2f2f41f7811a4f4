# MatMul -- HIP plugin (fp32 MFMA).  Replaces kernel_MatMul_numpy (reference op_plugins/MatMul.py:9-17):
# 2-D operands, transpose flags are the strings 'true' / 'false'.
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('MatMul')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    data = node['data']
    a = dev.as_device(inputs[0])
    b = dev.as_device(inputs[1])
    if a.ndim != 2 or b.ndim != 2:
        raise NotImplementedError('MatMul operands must be 2-D, got {} and {}'.format(a.shape, b.shape))
    ta = data['transpose_a'] == 'true'
    tb = data['transpose_b'] == 'true'
    m, ka = (a.shape[1], a.shape[0]) if ta else a.shape
    kb, n = (b.shape[1], b.shape[0]) if tb else b.shape
    if ka != kb:
        raise ValueError('matmul: Input operand 1 has a mismatch in its core dimension 0 (size {} is different '
                         'from {})'.format(kb, ka))
    c = dev.DeviceTensor.empty((m, n))
    # an FP16 IR read with fp16_as_fp32=False: fp16 operands on the f16 matrix cores, fp32 accumulation (the engine's hint)
    dev.call('pvhip_matmul_f16' if node.get('_f16_mfma') else 'pvhip_matmul_f32', ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(c.ptr), m, n, ka,
             int(ta), int(tb))
    return {common_def.first_output_port(node): c}
