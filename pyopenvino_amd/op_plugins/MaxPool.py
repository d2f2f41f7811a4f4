# MaxPool -- HIP plugin.  Replaces kernel_MaxPool_numpy (reference op_plugins/MaxPool.py:41-72);
# output extent rule of MaxPool.py:10-38 (same_* keeps the input extent, reference behaviour).
import ctypes

from .. import common_def
from .. import device as dev


# The engine may hand over a 3x3 MaxPool whose only consumer is an LRN as one call: node['_fuse_lrn'] is then the LRN's node
# dict, the kernel normalises the pooled values while it walks the channels and the pooled tensor (written once and read once
# otherwise) never exists; what is returned is the LRN's output.  The engine asks lrn_fusable() first.
SUPPORTS_FUSED_LRN = True


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def _geometry(node: dict, h: int, w: int):
    attrs = node['data']
    strides = common_def.string_to_tuple(attrs['strides'])
    pads_begin = common_def.string_to_tuple(attrs['pads_begin'])
    pads_end = common_def.string_to_tuple(attrs['pads_end'])
    kernel = common_def.string_to_tuple(attrs['kernel'])
    oh, ow = calc_output_shape((h, w), kernel, strides, pads_begin, pads_end, attrs['rounding_type'], attrs['auto_pad'])
    return kernel, strides, pads_begin, pads_end, oh, ow


def lrn_fusable(node: dict, lrn_node: dict) -> bool:
    """True when libpvhip's fused MaxPool -> LRN kernel covers this pair (shapes from the IR ports; no device needed)."""
    try:
        dims = node['input'][0]['dims']
        if len(dims) != 4 or node['input'][0]['precision'] != 'FP32':
            return False
        n, c, h, w = (int(d) for d in dims)
        kernel, strides, pads_begin, pads_end, oh, ow = _geometry(node, h, w)
        if len(kernel) != 2 or tuple(lrn_node['output'][common_def.first_output_port(lrn_node)]['dims']) != (n, c, oh, ow):
            return False
        la = lrn_node['data']
        return bool(dev.call('pvhip_maxpool_lrn_supported', n, c, h, w, oh, ow, kernel[0], kernel[1], strides[0], strides[1],
                             pads_begin[0], pads_begin[1], pads_end[0], pads_end[1], int(la['size']), float(la['beta']), float(la['bias'])))
    except (KeyError, ValueError, AssertionError):
        return False


# ... and, behind that LRN, a 1x1 / stride 1 / unpadded convolution with its fused bias / activation as part of the SAME launch: node['_fuse_conv']
# = {'node': the Convolution's node dict, 'w': its weights, 'bias': its fused bias or None, 'act': its fused activation or None}; what is
# returned is then the convolution's output.  The engine asks lrn_conv_fusable() first.
SUPPORTS_FUSED_LRN_CONV = True


def lrn_conv_fusable(node: dict, lrn_node: dict, conv_node: dict, f16: bool = False) -> bool:
    """True when libpvhip's MaxPool -> LRN -> 1x1 convolution kernel covers this triple (IR attributes and port dims; no device needed).
    f16: the triple of an FP16 IR on blocked fp16 tensors (pvhip_maxpool3x3_lrn_conv1x1_c8)."""
    try:
        if f16:
            if not blocked_ok(node, lrn_node):
                return False
            dims, wd, ca = node['input'][0]['dims'], conv_node['input'][1]['dims'], conv_node['data']
            n, c, h, w = (int(d) for d in dims)
            kernel, strides, pads_begin, pads_end, oh, ow = _geometry(node, h, w)
            cs, cpb, cpe = (common_def.string_to_tuple(ca[k]) for k in ('strides', 'pads_begin', 'pads_end'))
            if tuple(wd[2:]) != (1, 1) or tuple(cs) != (1, 1) or tuple(cpb) != (0, 0) or tuple(cpe) != (0, 0) or ca['auto_pad'] not in ('explicit', 'valid'):
                return False
            if tuple(conv_node['input'][0]['dims']) != (n, c, oh, ow) or int(wd[1]) != c or n * 64 * oh * ow >= 2 ** 31:
                return False
            return bool(dev.call('pvhip_maxpool3x3_lrn_conv1x1_c8_supported', c, int(wd[0]), int(lrn_node['data']['size'])))
        if not lrn_fusable(node, lrn_node):
            return False
        dims = node['input'][0]['dims']
        n, c, h, w = (int(d) for d in dims)
        kernel, strides, pads_begin, pads_end, oh, ow = _geometry(node, h, w)
        ca, xd, wd = conv_node['data'], conv_node['input'][0]['dims'], conv_node['input'][1]['dims']
        cs, cpb, cpe = (common_def.string_to_tuple(ca[k]) for k in ('strides', 'pads_begin', 'pads_end'))
        if tuple(wd[2:]) != (1, 1) or tuple(cs) != (1, 1) or tuple(cpb) != (0, 0) or tuple(cpe) != (0, 0) or ca['auto_pad'] not in ('explicit', 'valid'):
            return False
        if tuple(xd) != (n, c, oh, ow) or int(wd[1]) != c or conv_node['input'][0]['precision'] != 'FP32':
            return False
        la = lrn_node['data']
        return bool(dev.call('pvhip_maxpool_lrn_conv1x1_supported', n, c, h, w, oh, ow, kernel[0], kernel[1], strides[0], strides[1],
                             pads_begin[0], pads_begin[1], pads_end[0], pads_end[1], int(la['size']), float(la['beta']), float(la['bias']), int(wd[0])))
    except (KeyError, ValueError, AssertionError, IndexError, TypeError):
        return False


def blocked_ok(node: dict, lrn_node: dict = None) -> bool:
    """True when compute() pools a dev.BlockedHalf input of this node as it is and returns a dev.BlockedHalf (FP16 IRs; IR attributes
    and port dims, no device needed): a 3x3 window with a non-empty output, and an LRN folded behind it only over five channels.
    The ONE predicate of the plan (Executable_Network.plan_c8_modules) and of compute(): what the plan calls blocked IS blocked."""
    try:
        attrs = node['data']
        if tuple(common_def.string_to_tuple(attrs['kernel'])) != (3, 3):
            return False
        if lrn_node is not None and int(lrn_node['data']['size']) != 5:
            return False
        dims = node['input'][0]['dims']
        if len(dims) != 4:
            return False
        _, _, _, _, oh, ow = _geometry(node, int(dims[2]), int(dims[3]))
        return oh > 0 and ow > 0
    except (KeyError, ValueError, AssertionError, IndexError, TypeError):
        return False


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
    lrn_in = node.get('_fuse_lrn')
    blocked = inputs[0] if isinstance(inputs[0], dev.BlockedHalf) and blocked_ok(node, lrn_in) else None
    if blocked is None and node.get('_fuse_conv') is not None and node['_fuse_conv'].get('c8'):
        # the plan folded the convolution of an FP16 IR behind this launch (blocked tensors) but a dense tensor arrived: convert it, go on
        blocked = dev.BlockedHalf.from_dense(dev.as_device(inputs[0]))
    x = blocked if blocked is not None else dev.as_device(inputs[0])
    n, c, h, w = x.shape
    oh, ow = calc_output_shape((h, w), kernel, strides, pads_begin, pads_end, attrs['rounding_type'], attrs['auto_pad'])
    hp, wp = h + pads_begin[0] + pads_end[0], w + pads_begin[1] + pads_end[1]
    if oh > 0 and ow > 0 and ((oh - 1) * strides[0] >= hp or (ow - 1) * strides[1] >= wp):
        # np.max over an empty patch (MaxPool.py:69-70)
        raise ValueError('zero-size array to reduction operation maximum which has no identity')
    if blocked is not None and oh > 0 and ow > 0:
        # FP16 IRs: the input is fp16 blocked by eight channels (dev.BlockedHalf, what the reference holds here is a float16 tensor):
        # pooled as it is, the output is blocked too
        conv = node.get('_fuse_conv')
        if lrn_in is not None and conv is not None:
            # MaxPool -> LRN -> 1x1 convolution (+ bias, ReLU) as one launch on the blocked tensor: the output is the convolution's, blocked too
            la = lrn_in['data']
            cw = dev.as_device(conv['w'])
            cb_ = dev.as_device(conv['bias']) if conv.get('bias') is not None else None
            k_out = cw.shape[0]
            assert tuple(cw.shape[1:]) == (c, 1, 1) and (cb_ is None or cb_.size == k_out) and (conv.get('act') is None or conv['act'][0] == 'relu')
            yc = dev.BlockedHalf((n, k_out, oh, ow))
            dev.call('pvhip_maxpool3x3_lrn_conv1x1_c8', ctypes.c_void_p(blocked.ptr), ctypes.c_void_p(cw.ptr), ctypes.c_void_p(yc.ptr), n, c, h, w, oh, ow,
                     strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1], int(la['size']), float(la['alpha']), float(la['beta']),
                     float(la['bias']), k_out, ctypes.c_void_p(cb_.ptr if cb_ is not None else 0), 1 if conv.get('act') is not None else 0)
            conv['node']['_hip_f16'] = 'inside MaxPool + LRN (blocked tensors)'
            return {common_def.first_output_port(node): yc}
        yb = dev.BlockedHalf((n, c, oh, ow))
        if lrn_in is not None:       # MaxPool + LRN as one launch, on the blocked tensor
            la = lrn_in['data']
            dev.call('pvhip_maxpool3x3_lrn_c8', ctypes.c_void_p(blocked.ptr), ctypes.c_void_p(yb.ptr), n, c, h, w, oh, ow, strides[0], strides[1],
                     pads_begin[0], pads_begin[1], pads_end[0], pads_end[1], int(la['size']), float(la['alpha']), float(la['beta']), float(la['bias']))
            return {common_def.first_output_port(node): yb}
        dev.call('pvhip_maxpool3x3_c8', ctypes.c_void_p(blocked.ptr), ctypes.c_void_p(yb.ptr), n, c, h, w, oh, ow, strides[0], strides[1],
                 pads_begin[0], pads_begin[1], pads_end[0], pads_end[1])
        return {common_def.first_output_port(node): yb}
    if blocked is not None:
        x = dev.as_device(blocked)
    lrn_node = node.get('_fuse_lrn')
    conv = node.get('_fuse_conv')
    if lrn_node is not None and conv is not None:
        # MaxPool -> LRN -> 1x1 convolution (+ bias, activation) as one launch: neither the pooled nor the normalised tensor exists
        la = lrn_node['data']
        cw = dev.as_device(conv['w'])
        cb = dev.as_device(conv['bias']) if conv.get('bias') is not None else None
        k_out = cw.shape[0]
        assert tuple(cw.shape[1:]) == (c, 1, 1) and (cb is None or cb.size == k_out)
        act, act_code, act_lo, act_hi = conv.get('act'), 0, 0.0, 0.0
        if act is not None:
            act_code = 1 if act[0] == 'relu' else 2
            if act_code == 2:
                act_lo, act_hi = float(act[1]), float(act[2])
        yc = dev.DeviceTensor.empty((n, k_out, oh, ow))
        dev.call('pvhip_maxpool_lrn_conv1x1_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(cw.ptr), ctypes.c_void_p(yc.ptr), n, c, h, w, oh, ow,
                 kernel[0], kernel[1], strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1],
                 int(la['size']), float(la['alpha']), float(la['beta']), float(la['bias']), k_out, ctypes.c_void_p(cb.ptr if cb is not None else 0),
                 act_code, act_lo, act_hi)
        return {common_def.first_output_port(node): yc}
    y = dev.DeviceTensor.empty((n, c, oh, ow))
    if lrn_node is not None:
        la = lrn_node['data']
        dev.call('pvhip_maxpool_lrn_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, ow,
                 kernel[0], kernel[1], strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1],
                 int(la['size']), float(la['alpha']), float(la['beta']), float(la['bias']))
        return {common_def.first_output_port(node): y}
    dev.call('pvhip_maxpool2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, oh, ow,
             kernel[0], kernel[1], strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1])
    return {common_def.first_output_port(node): y}
