# AvgPool -- HIP plugin.  Replaces kernel_AvgPool_numpy (reference op_plugins/AvgPool.py:41-59): pads are
# never applied and the window is clipped at h-1 / w-1 (so GoogLeNet's 7x7 pool averages the top-left
# 6x6) -- kept, because results must equal the reference's.
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def name():
    print('AvgPool')


def calc_output_shape(input_dim, kernel_dim, strides, pads_begin, pads_end, rounding_type, auto_pad):
    return tuple(common_def.pooled_extent(input_dim[i], kernel_dim[i], strides[i], pads_begin[i], pads_end[i],
                                          rounding_type, auto_pad, same_means_input=True) for i in (0, 1))


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    attrs = node['data']
    strides = common_def.string_to_tuple(attrs['strides'])
    pads_begin = common_def.string_to_tuple(attrs['pads_begin'])
    pads_end = common_def.string_to_tuple(attrs['pads_end'])
    kernel = common_def.string_to_tuple(attrs['kernel'])
    blocked = inputs[0] if isinstance(inputs[0], dev.BlockedHalf) else None      # FP16 IRs: fp16 blocked by eight channels, averaged as it is
    x = blocked if blocked is not None else dev.as_device(inputs[0])
    n, c, h, w = x.shape
    oh, ow = calc_output_shape((h, w), kernel, strides, pads_begin, pads_end, attrs['rounding_type'], attrs['auto_pad'])
    y = dev.DeviceTensor.empty((n, c, oh, ow))
    if blocked is not None and oh > 0 and ow > 0:
        dev.call('pvhip_avgpool_c8', ctypes.c_void_p(blocked.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, ow, kernel[0], kernel[1], strides[0], strides[1])
        return {common_def.first_output_port(node): y}
    if blocked is not None:
        x = dev.as_device(blocked)
    dev.call('pvhip_avgpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, ow,
             kernel[0], kernel[1], strides[0], strides[1])
    return {common_def.first_output_port(node): y}
