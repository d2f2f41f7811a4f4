# MaxPool -- HIP plugin.  Replaces kernel_MaxPool_numpy (reference op_plugins/MaxPool.py:41-72);
# output extent rule of MaxPool.py:10-38 (same_* keeps the input extent, reference behaviour).
import ctypes

from .. import common_def
from .. import device as dev


def name():
    print('MaxPool')


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
    x = dev.as_device(inputs[0])
    n, c, h, w = x.shape
    oh, ow = calc_output_shape((h, w), kernel, strides, pads_begin, pads_end, attrs['rounding_type'], attrs['auto_pad'])
    hp, wp = h + pads_begin[0] + pads_end[0], w + pads_begin[1] + pads_end[1]
    if oh > 0 and ow > 0 and ((oh - 1) * strides[0] >= hp or (ow - 1) * strides[1] >= wp):
        # np.max over an empty patch (MaxPool.py:69-70)
        raise ValueError('zero-size array to reduction operation maximum which has no identity')
    y = dev.DeviceTensor.empty((n, c, oh, ow))
    dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, ow,
             kernel[0], kernel[1], strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1])
    return {common_def.first_output_port(node): y}
