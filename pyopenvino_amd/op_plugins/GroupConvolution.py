# GroupConvolution -- HIP plugin, depthwise case.  Replaces kernel_GroupConvolution_numpy (reference
# op_plugins/GroupConvolution.py:53-79), whose channel indexing (gp*ci+gp, :77) is only meaningful for
# one input and one output channel per group -- the only case the shipped IRs contain.  The reference
# computes batch element 0 only (:77-78); here every image of the batch is computed the same way.
import ctypes

from .. import common_def
from .. import device as dev


# node['_fuse_bias'] / node['_fuse_act'] (set by the engine's fusion peephole) are applied in the kernel epilogue.
SUPPORTS_FUSED_EPILOGUE = True


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('GroupConvolution')


def calc_output_shape_group_conv(input_dim, kernel_dim, strides, pads_begin, pads_end, rounding_type, auto_pad):
    return tuple(common_def.pooled_extent(input_dim[i], kernel_dim[i], strides[i], pads_begin[i], pads_end[i],
                                          rounding_type, auto_pad, same_means_input=False) for i in (0, 1))


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    attrs = node['data']
    strides = common_def.string_to_tuple(attrs['strides'])
    pads_begin = common_def.string_to_tuple(attrs['pads_begin'])
    pads_end = common_def.string_to_tuple(attrs['pads_end'])
    x = dev.as_device(inputs[0])
    w = dev.as_device(inputs[1])
    n, c, h, wd = x.shape
    grp, ch_o, ch_i, kh, kw = w.shape
    if ch_o != 1 or ch_i != 1 or grp != c:
        raise NotImplementedError('only depthwise GroupConvolution (weights [G,1,1,kh,kw], G == C) is supported, '
                                  'got weights {} for input {}'.format(w.shape, x.shape))
    oh, ow = calc_output_shape_group_conv((h, wd), (kh, kw), strides, pads_begin, pads_end, 'floor', attrs['auto_pad'])
    hp, wp = h + pads_begin[0] + pads_end[0], wd + pads_begin[1] + pads_end[1]
    if oh > 0 and ow > 0 and ((oh - 1) * strides[0] + kh > hp or (ow - 1) * strides[1] + kw > wp):
        raise ValueError('operands could not be broadcast together: window exceeds the padded input')
    bias = node.get('_fuse_bias')
    if bias is not None:
        bias = dev.as_device(bias)
        assert bias.size == grp
    act = node.get('_fuse_act')
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    y = dev.DeviceTensor.empty((n, grp, oh, ow))
    dev.call('pvhip_dwconv2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(w.ptr), ctypes.c_void_p(y.ptr),
             n, grp, h, wd, kh, kw, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1],
             ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, act_lo, act_hi)
    return {common_def.first_output_port(node): y}
