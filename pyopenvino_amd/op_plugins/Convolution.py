# Convolution -- HIP plugin (implicit GEMM on the fp32 matrix cores).
# Replaces im2col + kernel_Convolution_im2col, the 'special' kernel (reference
# op_plugins/Convolution.py:57-87): zero padding by pads_begin/pads_end, output extent by
# calc_output_shape (:21-49) with 'floor', dilation ignored exactly as :72-87 ignores it.
import ctypes

import numpy as np

from .. import common_def
from .. import device as dev


# The engine may hand over a Convolution -> Add(per-channel Const) -> ReLU chain as one call: node['_fuse_bias']
# (DeviceTensor of K values) and node['_fuse_act'] = ('relu',) | ('clamp', lo, hi) are then applied in the kernel epilogue; node['_out_into'] =
# (tensor, channel offset) makes the kernel write its channels straight into the output of the channel Concat
# that consumes it.
SUPPORTS_FUSED_EPILOGUE = True


def name():
    print('Convolution')


def calc_output_shape(input_dim, kernel_dim, strides, pads_begin, pads_end, rounding_type, auto_pad):
    return tuple(common_def.pooled_extent(input_dim[i], kernel_dim[i], strides[i], pads_begin[i], pads_end[i],
                                          rounding_type, auto_pad, same_means_input=False) for i in (0, 1))


def packed_weights(node: dict, w, h: int, wd: int) -> 'dev.DeviceTensor':
    """K-major weight panel + gather table for the kernel, built once per (weight tensor, input extent)
    and kept on the node (weights are Const outputs: the same device block arrives on every infer)."""
    cached = node.get('_hip_wpack')
    key = (w.shape, h, wd)
    if cached is not None and cached[0] is w._block and cached[1] == key:
        return cached[2]
    k, c, kh, kw = w.shape
    elems = dev.call('pvhip_conv2d_pack_elems', k, c, kh, kw)
    wpack = dev.DeviceTensor.empty((int(elems),))
    dev.call('pvhip_conv2d_pack_f32', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wpack.ptr), k, c, kh, kw, h, wd)
    node['_hip_wpack'] = (w._block, key, wpack)
    return wpack


def launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=None, act=None, into=None):
    n, c, h, wd = x.shape
    kn, kc, kh, kw = w.shape
    if kc != c:
        raise ValueError('shapes {} and {} not aligned: {} (dim 1) != {} (dim 1)'.format(x.shape, w.shape, c, kc))
    oh, ow = calc_output_shape((h, wd), (kh, kw), strides, pads_begin, pads_end, 'floor', auto_pad)
    hp, wp = h + pads_begin[0] + pads_end[0], wd + pads_begin[1] + pads_end[1]
    if oh > 0 and ow > 0 and ((oh - 1) * strides[0] + kh > hp or (ow - 1) * strides[1] + kw > wp):
        # the strided slice of the padded image is shorter than (oh, ow): numpy refuses the assignment (:68)
        raise ValueError('could not broadcast input array: window exceeds the padded input '
                         '({}x{} padded, kernel {}x{}, stride {}, output {}x{})'.format(hp, wp, kh, kw, strides, oh, ow))
    wpack = packed_weights(node, w, h, wd)
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    if into is None:
        y, target, coff, ctotal = None, dev.DeviceTensor.empty((n, kn, oh, ow)), 0, 0
        y = target
    else:
        target, coff = into                                  # the Concat's output tensor and our first channel in it
        ctotal = target.shape[1]
        assert target.shape[0] == n and tuple(target.shape[2:]) == (oh, ow) and coff + kn <= ctotal
        y = dev.ChannelSlice(target, coff, kn)
    dev.call('pvhip_conv2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr),
             n, c, h, wd, kn, kh, kw, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1],
             ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, int(coff), int(ctotal), act_lo, act_hi)
    return y


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    attrs = node['data']
    strides = common_def.string_to_tuple(attrs['strides'])
    dilation = common_def.string_to_tuple(attrs['dilations'])  # parsed, unused (as the 'special' kernel)
    pads_begin = common_def.string_to_tuple(attrs['pads_begin'])
    pads_end = common_def.string_to_tuple(attrs['pads_end'])
    auto_pad = attrs['auto_pad']
    x = dev.as_device(inputs[0])
    w = dev.as_device(inputs[1])
    bias = node.get('_fuse_bias')
    if bias is not None:
        bias = dev.as_device(bias)
        assert bias.size == w.shape[0]
    y = launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=bias, act=node.get('_fuse_act'),
               into=node.get('_out_into'))
    port = common_def.first_output_port(node)
    assert common_def.type_convert_tbl[node['output'][port]['precision']] == np.float32
    return {port: y}
