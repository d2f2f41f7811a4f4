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
# Convolutions that read the same tensor (the 1x1 / 3x3_reduce / 5x5_reduce arms of an inception module) may be handed
# over as ONE call: node['_siblings'] = [{'node', 'inputs', 'bias', 'into'}, ...] lists the others (same attributes,
# same activation); the input is then read once, by one launch whose output-channel tiles store into the tensor of the
# convolution they belong to.  The siblings' outputs are left in node['_sibling_out'], in the same order.
SUPPORTS_SIBLINGS = True
# A 3x3 / stride 1 / pad 1 MaxPool whose only consumer is a 1x1 convolution (the pool -> pool_proj arm of an inception module) may
# be handed over with the convolution: node['_fuse_pool_in'] = the MaxPool's node dict, inputs[0] = the MaxPool's own input.  The
# kernel pools while it builds its input tile; the pooled tensor is never written.
SUPPORTS_POOLED_INPUT = True
# An Add of one fp32 constant per INPUT channel whose only consumer is a convolution that pads its input in a pass of its own (a layer
# with C % 16 != 0 and padding: GoogLeNet's data/mean -> conv1) may be handed over with the convolution: node['_pre_add'] = the constant
# (1, C, 1, 1), inputs[0] = the Add's own data input.  The padding pass adds it on the way (the same fp32 add: the same bits).
SUPPORTS_PRE_ADD = True
# FP16 IRs (node['_f16_mfma']): node['_out_c8'] (a sibling: its dict's 'c8') asks for the output as dev.BlockedHalf -- fp16, channels
# blocked by eight -- for a reader that c8_reader_ok() accepts; compute() takes such an input through pvhip_conv2d_f16_c8.
SUPPORTS_C8 = True
# ... and, second step: a blocked INPUT with blocked outputs (node['_out_into'] = (dev.BlockedHalf, channel offset): the module's blocked
# Concat buffer), several members, or a MaxPool in front runs as one launch of the module form (launch_c8_multi; c8_module_member_ok()).
SUPPORTS_C8_MODULES = True


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

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


def packed_weights_f16(node: dict, w, h: int, wd: int) -> 'dev.DeviceTensor':
    """fp16 weight panel + gather table of the f16-MFMA kernel (FP16 IRs), cached on the node like packed_weights."""
    cached = node.get('_hip_wpack16')
    key = (w.shape, h, wd)
    if cached is not None and cached[0] is w._block and cached[1] == key:
        return cached[2]
    k, c, kh, kw = w.shape
    wpack = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_f16_pack_elems', k, c, kh, kw)),))
    dev.call('pvhip_conv2d_f16_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wpack.ptr), k, c, kh, kw, h, wd)
    node['_hip_wpack16'] = (w._block, key, wpack)
    return wpack


def packed_weights_f16_span(node: dict, w) -> 'dev.DeviceTensor':
    """fp16 MFMA fragments of the span kernel (FP16 IRs, stride-1 "same" windows), cached on the node like packed_weights."""
    cached = node.get('_hip_wspan')
    if cached is not None and cached[0] is w._block and cached[1] == w.shape:
        return cached[2]
    k, c, kh, kw = w.shape
    wpack = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_f16_span_pack_elems', k, c, kh, kw)),))
    dev.call('pvhip_conv2d_f16_span_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wpack.ptr), k, c, kh, kw)
    node['_hip_wspan'] = (w._block, w.shape, wpack)
    return wpack


def packed_weights_f16_c8(node: dict, w) -> 'dev.DeviceTensor':
    """fp16 MFMA fragments of pvhip_conv2d_f16_c8 (input channels padded to whole 16-channel stages), cached on the node."""
    cached = node.get('_hip_wpack_c8')
    if cached is not None and cached[0] is w._block:
        return cached[1]
    k, c, kh, kw = w.shape
    wpack = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_f16_c8_pack_elems', k, c, kh, kw)),))
    dev.call('pvhip_conv2d_f16_c8_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wpack.ptr), k, c, kh, kw)
    node['_hip_wpack_c8'] = (w._block, wpack)
    return wpack


def c8_reader_ok(node: dict) -> bool:
    """True when pvhip_conv2d_f16_c8 covers this Convolution node (IR attributes and port dims; no device needed): it may then be handed
    its input as fp16 blocked by eight channels (dev.BlockedHalf) by the 1x1 convolution in front of it."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        strides, pb, pe = (common_def.string_to_tuple(attrs[k]) for k in ('strides', 'pads_begin', 'pads_end'))
        if len(xd) != 4 or len(wd) != 4 or wd[1] != xd[1] or tuple(pb) != tuple(pe):
            return False
        oh, ow = calc_output_shape(xd[2:], wd[2:], strides, pb, pe, 'floor', attrs['auto_pad'])
        if 2 * int(xd[0]) * (-(-int(xd[1]) // 16) * 16) * int(xd[2]) * int(xd[3]) >= 2 ** 31 or int(xd[0]) * int(wd[0]) * oh * ow >= 2 ** 31:
            return False       # (32-bit offsets in the kernel; the query does not know n)
        return bool(dev.call('pvhip_conv2d_f16_c8_supported', int(xd[1]), int(xd[2]), int(xd[3]), int(wd[2]), int(wd[3]), strides[0], strides[1],
                             pb[0], pb[1], oh, ow))
    except (KeyError, ValueError, AssertionError, IndexError):
        return False


def c8_writer_ok(node: dict) -> bool:
    """True when the f16 multi-destination launch (launch_siblings with one or more members) runs this Convolution node, i.e. when it
    can store its output as dev.BlockedHalf: 1x1, stride 1, unpadded, C % 16 == 0."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        st, pb, pe = (common_def.string_to_tuple(attrs[key]) for key in ('strides', 'pads_begin', 'pads_end'))
        if len(xd) != 4 or len(wd) != 4 or tuple(pe) != (0, 0) or wd[1] != xd[1] or attrs['auto_pad'] not in ('explicit', 'valid'):
            return False
        return bool(dev.call('pvhip_conv2d_multi_supported', int(xd[1]), int(wd[2]), int(wd[3]), st[0], st[1], pb[0], pb[1], 1))
    except (KeyError, ValueError, AssertionError, IndexError):
        return False


def launch_c8(node, xb, w, bias=None, act=None, into=None):
    """FP16 IRs: the convolution of a dev.BlockedHalf input (pvhip_conv2d_f16_c8); output fp32 NCHW as launch()."""
    n, c, h, wd = xb.shape
    kn, kc, kh, kw = w.shape
    if kc != c:
        raise ValueError('shapes {} and {} not aligned: {} (dim 1) != {} (dim 1)'.format(xb.shape, w.shape, c, kc))
    wpack = packed_weights_f16_c8(node, w)
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    if into is None:
        target, coff, ctotal = dev.DeviceTensor.empty((n, kn, h, wd)), 0, 0
        y = target
    else:
        target, coff = into
        ctotal = target.shape[1]
        assert target.shape[0] == n and tuple(target.shape[2:]) == (h, wd) and coff + kn <= ctotal
        y = dev.ChannelSlice(target, coff, kn)
    node['_hip_f16'] = 'c8'
    dev.call('pvhip_conv2d_f16_c8', ctypes.c_void_p(xb.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), n, c, h, wd, kn, kh, kw,
             ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, int(coff), int(ctotal), act_lo, act_hi)
    return y


def c8_multi_ok(x_shape, kh: int, kw: int, pool: bool, n_members: int) -> bool:
    """True when pvhip_conv2d_f16_c8_multi covers a launch of n_members convolutions with this window over a blocked input of this shape."""
    n, c, h, wd = x_shape
    if 2 * int(n) * (-(-int(c) // 16) * 16) * int(h) * int(wd) >= 2 ** 31:
        return False       # (the launch addresses its input with 32-bit byte offsets; the query does not know n)
    return bool(dev.call('pvhip_conv2d_f16_c8_multi_supported', int(c), int(h), int(wd), int(kh), int(kw), 1 if pool else 0, int(n_members)))


def c8_module_member_ok(node: dict, pool_node: dict = None) -> bool:
    """True when this Convolution node can run on pvhip_conv2d_f16_c8_multi with a blocked input AND a blocked output: a stride-1 "same"
    1x1 / 3x3 / 5x5 window (a 1x1 optionally behind a 3x3 / 1 / 1 MaxPool), output channels a multiple of 8."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        st, pb, pe = (common_def.string_to_tuple(attrs[key]) for key in ('strides', 'pads_begin', 'pads_end'))
        if len(xd) != 4 or len(wd) != 4 or wd[1] != xd[1] or wd[2] != wd[3] or wd[0] % 8 != 0 or tuple(st) != (1, 1):
            return False
        pad = (wd[2] - 1) // 2
        if tuple(pb) != (pad, pad) or tuple(pe) != (pad, pad) or attrs['auto_pad'] not in ('explicit', 'valid'):
            return False
        if pool_node is not None:
            pa = pool_node['data']
            pk, ps, ppb, ppe = (common_def.string_to_tuple(pa[k]) for k in ('kernel', 'strides', 'pads_begin', 'pads_end'))
            if wd[2] != 1 or tuple(pk) != (3, 3) or tuple(ps) != (1, 1) or tuple(ppb) != (1, 1) or tuple(ppe) != (1, 1) or pa['auto_pad'] != 'explicit':
                return False
            if tuple(pool_node['input'][0]['dims']) != tuple(xd):
                return False
        return c8_multi_ok(xd, wd[2], wd[3], pool_node is not None, 1)
    except (KeyError, ValueError, AssertionError, IndexError):
        return False


def c8_panel(node: dict, ws, biases):
    """(fp16 MFMA fragments of the members' weights laid one after the other, each padded to whole 32-channel tiles; fused bias or None),
    built once and kept on the leading node."""
    key = tuple(w._block for w in ws) + tuple(b._block if b is not None else None for b in biases)
    cached = node.get('_hip_c8panel')
    if cached is not None and len(cached[0]) == len(key) and all(a is b for a, b in zip(cached[0], key)):
        return cached[1]
    c, kh, kw = ws[0].shape[1:]
    pads = [-(-w.shape[0] // 32) * 32 for w in ws]
    host_w = np.zeros((sum(pads), c, kh, kw), dtype=np.float32)
    host_b = np.zeros((sum(pads),), dtype=np.float32)
    row = 0
    for w, b, kp in zip(ws, biases, pads):
        host_w[row:row + w.shape[0]] = w.numpy()
        if b is not None:
            host_b[row:row + w.shape[0]] = b.numpy().reshape(-1)
        row += kp
    wf = dev.DeviceTensor.from_numpy(host_w)
    wpack = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_f16_c8_pack_elems', host_w.shape[0], c, kh, kw)),))
    dev.call('pvhip_conv2d_f16_c8_pack', ctypes.c_void_p(wf.ptr), ctypes.c_void_p(wpack.ptr), host_w.shape[0], c, kh, kw)
    packed = (wpack, dev.DeviceTensor.from_numpy(host_b) if any(b is not None for b in biases) else None)
    node['_hip_c8panel'] = (key, packed)
    return packed


def launch_c8_multi(node, xb, members, pool=False, act=None):
    """FP16 IRs, the module form: members = [(weights, bias or None, into or None, blocked)] over the blocked input xb, ONE launch
    (pvhip_conv2d_f16_c8_multi).  into = (tensor, channel offset) with a dev.BlockedHalf (the module's blocked Concat buffer) or a
    DeviceTensor (an fp32 Concat buffer); without it `blocked` chooses a dev.BlockedHalf or an fp32 tensor of the member's own.
    pool: a 3x3 / 1 / 1 MaxPool of xb in front (one member).  -> list of outputs."""
    n, c, h, wd = xb.shape
    ws = [m[0] for m in members]
    kh, kw = ws[0].shape[2:]
    for w in ws:
        if w.shape[1] != c or tuple(w.shape[2:]) != (kh, kw):
            raise ValueError('the members of a launch are convolutions of the same {} channels with the same window, got {}'.format(c, w.shape))
    wpack, bias = c8_panel(node, ws, [m[1] for m in members])
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    dests = (dev.ConvDest * len(members))()
    outs, keep = [], []
    for i, (w, _, into, blocked) in enumerate(members):
        kn = w.shape[0]
        if into is not None:
            target, coff = into
            assert target.shape[0] == n and tuple(target.shape[2:]) == (h, wd) and coff + kn <= target.shape[1]
            if isinstance(target, dev.BlockedHalf):
                outs.append(dev.BlockedChannelSlice(target, coff, kn))
                layout = 1
            else:
                outs.append(dev.ChannelSlice(target, coff, kn))
                layout = 0
            ctotal = target.shape[1]
        else:
            target = dev.BlockedHalf((n, kn, h, wd)) if blocked else dev.DeviceTensor.empty((n, kn, h, wd))
            outs.append(target)
            coff, ctotal, layout = 0, 0, 1 if blocked else 0
        if layout == 1 and act_code == 2:
            raise ValueError('a blocked fp16 output takes no Clamp')
        keep.append(target)
        dests[i].y, dests[i].k, dests[i].channel_offset, dests[i].channels_total, dests[i].layout = target.ptr, kn, int(coff), int(ctotal), layout
    node['_hip_f16'] = 'c8 module' + (', MaxPool' if pool else '') + (', {} members'.format(len(members)) if len(members) > 1 else '')
    dev.call('pvhip_conv2d_f16_c8_multi', ctypes.c_void_p(xb.ptr), ctypes.c_void_p(wpack.ptr), n, c, h, wd, kh, kw, 1 if pool else 0,
             ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, act_lo, act_hi, len(members), ctypes.cast(dests, ctypes.c_void_p))
    return outs


def prepad_wanted(n, c, h, wd, kn, kh, kw, oh, ow, strides, pads_begin, pads_end, f16=False) -> bool:
    """True for a padded layer that libpvhip runs on the c-major form of the LDS-DMA kernel (C % 16 != 0, not a Winograd or pointwise
    layer): its gather tests every tap against the window unless no window leaves the tensor (PVHIP_CONV_PREPAD=0: never)."""
    if not dev.conv_prepad or c % 16 == 0 or not (any(pads_begin) or any(pads_end)) or oh <= 0 or ow <= 0 or kh * kw >= 64:
        return False
    if f16:         # FP16 IRs: the c-major f16 form of the LDS-DMA kernel (every such layer: there is no Winograd or pointwise form in front of it)
        return dev.conv_f16_dma and bool(dev.call('pvhip_conv2d_f16_dma_supported', c, kh, kw))
    # (kind 5, the row-span kernel, pads its input in a pass of its own too: the Add in front of the layer rides in that pass just the same)
    return int(dev.call('pvhip_conv2d_kernel_kind', n, c, h, wd, kn, kh, kw, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1])) in (0, 5, 6)


def pre_add_fusable(node: dict, add_node: dict, const_node: dict, f16: bool = False) -> bool:
    """True when this layer pads its input in a pass of its own (prepad_wanted) and the Add in front of it adds one fp32 constant per
    input channel: the padding pass then does the Add (IR attributes and port dims; no device needed)."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        strides, pb, pe = (common_def.string_to_tuple(attrs[k]) for k in ('strides', 'pads_begin', 'pads_end'))
        if len(xd) != 4 or attrs['auto_pad'] != 'explicit':
            return False
        if tuple(const_node['data']['shape']) != (1, int(xd[1]), 1, 1) or const_node['data']['element_type'] != 'f32':
            return False
        add_out = add_node['output'][common_def.first_output_port(add_node)]['dims']
        if tuple(add_out) != tuple(xd) or not all(tuple(p_['dims']) in (tuple(xd), (1, int(xd[1]), 1, 1)) for p_ in add_node['input'].values()):
            return False
        oh, ow = calc_output_shape(xd[2:], wd[2:], strides, pb, pe, 'floor', attrs['auto_pad'])
        return prepad_wanted(int(xd[0]), int(xd[1]), int(xd[2]), int(xd[3]), int(wd[0]), int(wd[2]), int(wd[3]), oh, ow, strides, pb, pe, f16)
    except (KeyError, ValueError, AssertionError, IndexError, TypeError):
        return False


def f16_route(c, h, wd, kh, kw, strides, pads_begin, oh, ow):
    """(span kernel, f16 form of the LDS-DMA kernel) -- which f16 kernel launch() runs a dense-input layer of an FP16 IR on; neither: the first
    (gather) f16 kernel.  Settings and libpvhip's _supported queries only (no device needed)."""
    span_ok = dev.conv_f16_span >= (2 if kh == 1 else 1) and bool(dev.call(
        'pvhip_conv2d_f16_span_supported', c, h, wd, kh, kw, strides[0], strides[1], pads_begin[0], pads_begin[1], oh, ow))
    return span_ok, (not span_ok) and dev.conv_f16_dma and bool(dev.call('pvhip_conv2d_f16_dma_supported', c, kh, kw))


def f16_stem_row(c, h, wd, kn, kh, kw, strides, pads_begin, pads_end, oh, ow) -> int:
    """Row length (floats) of the padded image pvhip_conv2d_f16_stem wants for this layer, 0 when the row-span kernel does not cover it."""
    if not dev.conv_f16_stem or tuple(pads_begin) != tuple(pads_end):
        return 0
    return int(dev.call('pvhip_conv2d_f16_stem_supported', c, h, wd, kn, kh, kw, strides[0], strides[1], pads_begin[0], pads_begin[1], oh, ow))


def c8_dma_writer_ok(node: dict) -> bool:
    """True when launch(..., out_c8=True) stores this Convolution node's output as dev.BlockedHalf: the row-span kernel (GoogLeNet's conv1:
    pvhip_conv2d_f16_stem) or the f16 form of the LDS-DMA kernel (pvhip_conv2d_f16_dma_c8) runs it.  The SAME route as launch() takes --
    f16_stem_row / f16_route -- so a layer the span kernel is preferred for (a dense fp32 output) is not planned blocked."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        if len(xd) != 4 or len(wd) != 4 or wd[1] != xd[1]:
            return False
        strides, pb, pe = (common_def.string_to_tuple(attrs[k]) for k in ('strides', 'pads_begin', 'pads_end'))
        oh, ow = calc_output_shape(xd[2:], wd[2:], strides, pb, pe, 'floor', attrs['auto_pad'])
        if oh <= 0 or ow <= 0:
            return False
        c, h, w_, kn, kh, kw = int(xd[1]), int(xd[2]), int(xd[3]), int(wd[0]), int(wd[2]), int(wd[3])
        if f16_stem_row(c, h, w_, kn, kh, kw, strides, pb, pe, oh, ow) > 0:
            return True
        return f16_route(c, h, w_, kh, kw, strides, pb, oh, ow)[1]
    except (KeyError, ValueError, AssertionError, IndexError, TypeError):
        return False


def launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=None, act=None, into=None, f16=False, out_c8=False):
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
    pre_add = node.get('_pre_add')            # plan_fusion: the per-channel Add in front of this layer rides in the padding pass
    if pre_add is not None:
        pre_add = dev.as_device(pre_add)
        assert pre_add.size == c
    if f16 and out_c8 and into is None and (act is None or act[0] == 'relu'):
        # FP16 IRs: a 7x7 / 2 convolution over three channels with a blocked fp16 output (GoogLeNet's conv1) from row spans of the padded image
        wps = f16_stem_row(c, h, wd, kn, kh, kw, strides, pads_begin, pads_end, oh, ow)
        if wps > 0 and (4 * n * c * hp * wps >= 2 ** 31 or n * 64 * oh * ow >= 2 ** 31):
            wps = 0        # (the kernel addresses its input with 32-bit byte offsets and the query does not know n: the LDS-DMA form takes it)
        if wps > 0:
            direct = dev.conv_stem_direct and bool(dev.call('pvhip_conv2d_f16_stem_direct_supported', c, h, wd, kn, kh, kw, strides[0], strides[1],
                                                            pads_begin[0], pads_begin[1], oh, ow))
            cached = node.get('_hip_wpack_stem')
            if cached is None or cached[0] is not w._block or cached[2] != direct:
                wf = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_f16_stem_pack_elems', kn)),))
                dev.call('pvhip_conv2d_f16_stem_direct_pack' if direct else 'pvhip_conv2d_f16_stem_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wf.ptr), kn)
                cached = node['_hip_wpack_stem'] = (w._block, wf, direct)
            yb = dev.BlockedHalf((n, kn, oh, ow))
            node['_hip_f16'] = 'row spans, blocked output'
            if direct:       # rows of a multiple of four pixels: straight from the image, the Add in front of the layer applied in LDS
                dev.call('pvhip_conv2d_f16_stem_direct', ctypes.c_void_p(x.ptr), ctypes.c_void_p(cached[1].ptr), ctypes.c_void_p(yb.ptr), n, h, wd, kn, oh, ow,
                         ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0), ctypes.c_void_p(bias.ptr if bias is not None else 0), 1 if act is not None else 0)
                return yb
            xp = dev.DeviceTensor.empty((n, c, hp, wps))
            dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, wd, pads_begin[0], pads_begin[1],
                     pads_end[0], wps - wd - pads_begin[1], ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0))
            dev.call('pvhip_conv2d_f16_stem', ctypes.c_void_p(xp.ptr), ctypes.c_void_p(cached[1].ptr), ctypes.c_void_p(yb.ptr), n, hp, wps, kn, oh, ow,
                     ctypes.c_void_p(bias.ptr if bias is not None else 0), 1 if act is not None else 0)
            return yb
    # which kernel forms the library offers for this geometry: asked once per node and settings (not on every launch)
    route_key = (dev.settings_serial, x.shape, w.shape, tuple(strides), tuple(pads_begin), tuple(pads_end), bool(f16))
    route = node.get('_hip_route')
    if route is None or route[0] != route_key:
        span_ok, dma_ok = f16_route(c, h, wd, kh, kw, strides, pads_begin, oh, ow) if f16 else (False, False)
        stem_wps = 0           # fp32: a 7x7 / 2 first convolution over three channels from row spans (pvhip_conv2d_stem_f32): floats per padded row
        stem_wino = False      # ... or as Winograd F(3x3,4x4) on the space-to-depth image (pvhip_conv2d_stem_wino_f32; kind 6)
        if not f16 and tuple(pads_begin) == tuple(pads_end) and oh > 0 and ow > 0:
            kind = int(dev.call('pvhip_conv2d_kernel_kind', n, c, h, wd, kn, kh, kw, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1]))
            if kind in (5, 6):
                stem_wps = int(dev.call('pvhip_conv2d_stem_f32_supported', c, h, wd, kn, kh, kw, strides[0], strides[1], pads_begin[0], pads_begin[1], oh, ow))
                stem_wino = kind == 6
        route = (route_key, prepad_wanted(n, c, h, wd, kn, kh, kw, oh, ow, strides, pads_begin, pads_end, f16), span_ok, dma_ok, stem_wps, stem_wino)
        node['_hip_route'] = route
    if route[4] > 0 and into is None:
        # the zero-padded image in rows of route[4] floats (the per-channel Add in front of the layer rides in the padding pass), then the
        # row-span kernel: weights resident in registers, no vector instruction in its reduction loop
        wps = route[4]
        if route[5]:
            # Winograd F(3x3,4x4) on the space-to-depth image: the 7x7 / 2 layer as a 4x4 / 1 one over 12 phase channels, 0.34 of the multiplies
            cached = node.get('_hip_wpack_stemw')
            if cached is None or cached[0] is not w._block:
                u = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_stem_wino_pack_elems')),))
                dev.call('pvhip_conv2d_stem_wino_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(u.ptr), kn)
                cached = node['_hip_wpack_stemw'] = (w._block, u)
            y = dev.DeviceTensor.empty((n, kn, oh, ow))
            act_code, act_lo, act_hi = 0, 0.0, 0.0
            if act is not None:
                act_code = 1 if act[0] == 'relu' else 2
                if act_code == 2:
                    act_lo, act_hi = float(act[1]), float(act[2])
            dev.call('pvhip_conv2d_stem_wino_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(cached[1].ptr), ctypes.c_void_p(y.ptr), n, h, wd, kn, oh, ow,
                     ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0), ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, act_lo, act_hi)
            return y
        cached = node.get('_hip_wpack_stem32')
        if cached is None or cached[0] is not w._block:
            wf = dev.DeviceTensor.empty((int(dev.call('pvhip_conv2d_stem_f32_pack_elems', kn)),))
            dev.call('pvhip_conv2d_stem_f32_pack', ctypes.c_void_p(w.ptr), ctypes.c_void_p(wf.ptr), kn)
            cached = node['_hip_wpack_stem32'] = (w._block, wf)
        y = dev.DeviceTensor.empty((n, kn, oh, ow))
        act_code, act_lo, act_hi = 0, 0.0, 0.0
        if act is not None:
            act_code = 1 if act[0] == 'relu' else 2
            if act_code == 2:
                act_lo, act_hi = float(act[1]), float(act[2])
        if dev.conv_stem_direct and dev.call('pvhip_conv2d_stem_direct_supported', c, h, wd, kn, kh, kw, strides[0], strides[1], pads_begin[0], pads_begin[1], oh, ow):
            # rows of a multiple of four pixels: straight from the image -- no padding pass, the Add in front of the layer happens in LDS
            dev.call('pvhip_conv2d_stem_direct_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(cached[1].ptr), ctypes.c_void_p(y.ptr), n, h, wd, kn, oh, ow,
                     ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0), ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, act_lo, act_hi)
            return y
        xp = dev.DeviceTensor.empty((n, c, hp, wps))
        dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, wd, pads_begin[0], pads_begin[1],
                 pads_end[0], wps - wd - pads_begin[1], ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0))
        dev.call('pvhip_conv2d_stem_f32', ctypes.c_void_p(xp.ptr), ctypes.c_void_p(cached[1].ptr), ctypes.c_void_p(y.ptr), n, hp, wps, kn, oh, ow,
                 ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, act_lo, act_hi)
        return y
    if route[1] or pre_add is not None:
        # The zero-padded image (Convolution.py:64-66) as a tensor of its own, convolved WITHOUT padding: the gather of a layer whose
        # channel count is not a multiple of 16 (conv1: C = 3) then needs no window test -- zero vector instructions per gathered row
        # instead of five, and vector instructions are matrix time lost.  The pass costs less than the tests did.
        xp = dev.DeviceTensor.empty((n, c, hp, wp))
        dev.call('pvhip_pad2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(xp.ptr), n, c, h, wd, pads_begin[0], pads_begin[1],
                 pads_end[0], pads_end[1], ctypes.c_void_p(pre_add.ptr if pre_add is not None else 0))
        x, h, wd, pads_begin, pads_end = xp, hp, wp, (0, 0), (0, 0)
    # FP16 IRs: layers with C % 16 == 0 run the f16 form of the LDS-DMA kernel on the fp32 panel (PVHIP_CONV_F16_DMA=0: the first f16 kernel)
    f16_span, f16_dma = route[2], route[3]
    if f16_span:
        wpack = packed_weights_f16_span(node, w)
    else:
        wpack = packed_weights_f16(node, w, h, wd) if (f16 and not f16_dma) else packed_weights(node, w, h, wd)
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
    tail = (n, c, h, wd, kn, kh, kw, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1],
            ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, int(coff), int(ctotal), act_lo, act_hi)
    if f16:
        node['_hip_f16'] = 'span' if f16_span else ('lds-dma' if f16_dma else 'gather')      # which f16 kernel ran (tests)
    if out_c8 and f16_dma and into is None and act_code in (0, 1):
        # FP16 IRs: the reader takes fp16 blocked by eight channels (a MaxPool (+ LRN) on blocked tensors in front of a convolution)
        yb = dev.BlockedHalf((n, kn, oh, ow))
        node['_hip_f16'] = 'lds-dma, blocked output'
        dev.call('pvhip_conv2d_f16_dma_c8', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(yb.ptr), n, c, h, wd, kn, kh, kw, oh, ow,
                 strides[0], strides[1], pads_begin[0], pads_begin[1], ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code)
        return yb
    if f16_span:
        dev.call('pvhip_conv2d_f16_span', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), *tail)
    elif f16_dma:
        dev.call('pvhip_conv2d_f16_dma', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), *tail)
    elif f16:       # FP16 IR: fp16 operands on the f16 matrix cores, fp32 accumulation
        dev.call('pvhip_conv2d_f16', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), *tail)
    else:
        dev.call('pvhip_conv2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), *tail)
    return y


# pvhip_conv2d_kernel_kind codes -> (family name, fraction of the algorithmic multiply-adds the matrix cores execute)
KERNEL_KINDS = {5: ('row spans (stem)', 148.0 / 147.0),          # four taps per MFMA step: 147 taps in 37 steps, one slot of zero weight
                6: ('Winograd F(3x3,4x4), space-to-depth (stem)', 36.0 * 4.0 / (9.0 * 49.0)),      # 36 points x 4 phases per 9 outputs x 49 taps
                0: ('implicit GEMM (LDS-DMA)', 1.0), 1: ('pointwise', 1.0), 2: ('Winograd F(2x2,3x3)', 16.0 / 36.0),
                3: ('Winograd F(4x4,3x3)', 36.0 / 144.0), 4: ('Winograd F(2x2,5x5)', 36.0 / 100.0)}


def kernel_kind(node: dict):
    """(family name, executed fraction) of the kernel libpvhip runs for this Convolution node (IR port dims; no device needed)."""
    attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
    strides, pb, pe = (common_def.string_to_tuple(attrs[k]) for k in ('strides', 'pads_begin', 'pads_end'))
    oh, ow = calc_output_shape(xd[2:], wd[2:], strides, pb, pe, 'floor', attrs['auto_pad'])
    code = dev.call('pvhip_conv2d_kernel_kind', int(xd[0]), int(xd[1]), int(xd[2]), int(xd[3]), int(wd[0]), int(wd[2]), int(wd[3]), oh, ow,
                    strides[0], strides[1], pb[0], pb[1])
    family, frac = KERNEL_KINDS[int(code)]
    m = {2: 2, 3: 4, 4: 2, 6: 3}.get(int(code))          # Winograd: whole m x m output patches are computed (14x14 as 16x16 under F(4x4))
    if m:
        frac *= (-(-oh // m) * m) * (-(-ow // m) * m) / float(oh * ow)
    return family, frac


def pooled_fusable(node: dict, pool_node: dict) -> bool:
    """True when libpvhip's MaxPool + 1x1 convolution kernel covers this pair (IR attributes and port dims; no device needed)."""
    try:
        attrs, xd, wd = node['data'], node['input'][0]['dims'], node['input'][1]['dims']
        strides, pb, pe = (common_def.string_to_tuple(attrs[k]) for k in ('strides', 'pads_begin', 'pads_end'))
        if tuple(wd[2:]) != (1, 1) or tuple(strides) != (1, 1) or tuple(pb) != (0, 0) or tuple(pe) != (0, 0):
            return False
        if attrs['auto_pad'] not in ('explicit', 'valid'):
            return False
        pa = pool_node['data']
        pk, ps, ppb, ppe = (common_def.string_to_tuple(pa[k]) for k in ('kernel', 'strides', 'pads_begin', 'pads_end'))
        if tuple(pk) != (3, 3) or tuple(ps) != (1, 1) or tuple(ppb) != (1, 1) or tuple(ppe) != (1, 1) or pa['auto_pad'] != 'explicit':
            return False
        pin = pool_node['input'][0]['dims']
        if len(pin) != 4 or tuple(pin) != tuple(xd) or tuple(pool_node['output'][common_def.first_output_port(pool_node)]['dims']) != tuple(xd):
            return False
        return bool(dev.call('pvhip_conv2d_pooled_supported', int(xd[0]), int(xd[1]), int(xd[2]), int(xd[3]), int(wd[0])))
    except (KeyError, ValueError, AssertionError, IndexError):
        return False


def launch_pooled(node, x, w, bias=None, act=None, into=None, f16=False):
    """conv1x1(maxpool3x3/s1/p1(x)) in one launch; arguments as launch()."""
    n, c, h, wd = x.shape
    kn = w.shape[0]
    if w.shape[1] != c or tuple(w.shape[2:]) != (1, 1):
        raise ValueError('the pooled-input launch is for 1x1 convolutions over the same {} channels, got {}'.format(c, w.shape))
    wpack = packed_weights(node, w, h, wd)
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    if into is None:
        target, coff, ctotal = dev.DeviceTensor.empty((n, kn, h, wd)), 0, 0
        y = target
    else:
        target, coff = into
        ctotal = target.shape[1]
        assert target.shape[0] == n and tuple(target.shape[2:]) == (h, wd) and coff + kn <= ctotal
        y = dev.ChannelSlice(target, coff, kn)
    if f16:
        node['_hip_f16'] = 'MaxPool + 1x1'
    dev.call('pvhip_conv2d_pooled_f16' if f16 else 'pvhip_conv2d_pooled_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), ctypes.c_void_p(target.ptr), n, c, h, wd, kn,
             ctypes.c_void_p(bias.ptr if bias is not None else 0), act_code, int(coff), int(ctotal), act_lo, act_hi)
    return y


def siblings_fusable(nodes) -> bool:
    """True when libpvhip's multi-destination launch covers these Convolution nodes (IR port dims; no device needed)."""
    try:
        if not 2 <= len(nodes) <= dev.MAX_CONV_DESTS:
            return False
        first = nodes[0]
        for node in nodes:
            attrs, wd, xd = node['data'], node['input'][1]['dims'], node['input'][0]['dims']
            if len(xd) != 4 or len(wd) != 4 or tuple(xd) != tuple(first['input'][0]['dims']):
                return False
            same = all(common_def.string_to_tuple(attrs[key]) == common_def.string_to_tuple(first['data'][key])
                       for key in ('strides', 'pads_begin', 'pads_end'))
            if not same or attrs['auto_pad'] != first['data']['auto_pad'] or attrs['auto_pad'] not in ('explicit', 'valid'):
                return False
            st, pb, pe = (common_def.string_to_tuple(attrs[key]) for key in ('strides', 'pads_begin', 'pads_end'))
            if tuple(pe) != (0, 0) or wd[1] != xd[1]:
                return False
            if not dev.call('pvhip_conv2d_multi_supported', int(xd[1]), int(wd[2]), int(wd[3]), st[0], st[1], pb[0], pb[1], len(nodes)):
                return False
        return True
    except (KeyError, ValueError, AssertionError):
        return False


def fused_panel(node: dict, ws, biases, h: int, wd: int):
    """(packed panel, fused bias or None, padded channel counts) of the siblings' weights laid one after the other, each
    padded to whole 32-channel tiles; built once and kept on the leading node."""
    key = tuple(w._block for w in ws) + tuple(b._block if b is not None else None for b in biases) + (h, wd)
    cached = node.get('_hip_sibpack')
    if cached is not None and len(cached[0]) == len(key) and all(a is b for a, b in zip(cached[0], key)):
        return cached[1]
    c = ws[0].shape[1]
    pads = [-(-w.shape[0] // 32) * 32 for w in ws]
    host_w = np.zeros((sum(pads), c, 1, 1), dtype=np.float32)
    host_b = np.zeros((sum(pads),), dtype=np.float32)
    row = 0
    for w, b, kp in zip(ws, biases, pads):
        host_w[row:row + w.shape[0]] = w.numpy()
        if b is not None:
            host_b[row:row + w.shape[0]] = b.numpy().reshape(-1)
        row += kp
    wf = dev.DeviceTensor.from_numpy(host_w)
    elems = dev.call('pvhip_conv2d_pack_elems', host_w.shape[0], c, 1, 1)
    wpack = dev.DeviceTensor.empty((int(elems),))
    dev.call('pvhip_conv2d_pack_f32', ctypes.c_void_p(wf.ptr), ctypes.c_void_p(wpack.ptr), host_w.shape[0], c, 1, 1, h, wd)
    bias = dev.DeviceTensor.from_numpy(host_b) if any(b is not None for b in biases) else None
    packed = (wpack, bias, pads)
    node['_hip_sibpack'] = (key, packed)
    return packed


def launch_siblings(node, x, members, strides, pads_begin, act, f16=False):
    """members: [(weights, bias or None, into or None)], the node's own convolution first.  -> list of outputs."""
    n, c, h, wd = x.shape
    ws = [m[0] for m in members]
    for w in ws:
        if w.shape[1] != c or tuple(w.shape[2:]) != (1, 1):
            raise ValueError('sibling convolutions must be 1x1 over the same {} channels, got {}'.format(c, w.shape))
    wpack, bias, _ = fused_panel(node, ws, [m[1] for m in members], h, wd)
    act_code, act_lo, act_hi = 0, 0.0, 0.0
    if act is not None:
        act_code = 1 if act[0] == 'relu' else 2
        if act_code == 2:
            act_lo, act_hi = float(act[1]), float(act[2])
    dests = (dev.ConvDest * len(members))()
    outs, keep = [], []
    for i, member in enumerate(members):
        w, into = member[0], member[2]
        kn = w.shape[0]
        if len(member) > 3 and member[3]:         # FP16 IRs: this member's only reader is pvhip_conv2d_f16_c8 -- fp16, channels blocked by eight
            assert f16 and into is None and act_code in (0, 1)
            target = dev.BlockedHalf((n, kn, h, wd))
            outs.append(target)
            keep.append(target)
            dests[i].y, dests[i].k, dests[i].channel_offset, dests[i].channels_total, dests[i].layout = target.ptr, kn, 0, 0, 1
            continue
        if into is None:
            target, coff, ctotal = dev.DeviceTensor.empty((n, kn, h, wd)), 0, 0
            outs.append(target)
        else:
            target, coff = into
            ctotal = target.shape[1]
            assert target.shape[0] == n and tuple(target.shape[2:]) == (h, wd) and coff + kn <= ctotal
            outs.append(dev.ChannelSlice(target, coff, kn))
        keep.append(target)
        dests[i].y, dests[i].k, dests[i].channel_offset, dests[i].channels_total, dests[i].layout = target.ptr, kn, int(coff), int(ctotal), 0
    if f16:
        node['_hip_f16'] = 'lds-dma, siblings'
        for m in node.get('_siblings', []):
            m['node']['_hip_f16'] = 'lds-dma, siblings'
    dev.call('pvhip_conv2d_multi_f16_dma' if f16 else 'pvhip_conv2d_multi_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wpack.ptr), n, c, h, wd, 1, 1, h, wd,
             strides[0], strides[1], pads_begin[0], pads_begin[1], ctypes.c_void_p(bias.ptr if bias is not None else 0),
             act_code, act_lo, act_hi, len(members), ctypes.cast(dests, ctypes.c_void_p))
    return outs


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
    w = dev.as_device(inputs[1])
    # FP16 IRs: an input that is fp16 blocked by eight channels (dev.BlockedHalf) goes to the module form (blocked outputs, several members,
    # a MaxPool in front) or to the reader kernel (fp32 output); any other geometry densifies it (the same fp16 values)
    module_path = reader_path = False
    if node.get('_f16_mfma'):
        kh_, kw_ = w.shape[2], w.shape[3]
        pad_ = (kh_ - 1) // 2
        same = kh_ == kw_ and tuple(strides) == (1, 1) and tuple(pads_begin) == (pad_, pad_) and tuple(pads_end) == (pad_, pad_)
        into_ = node.get('_out_into')
        pool_ = node.get('_fuse_pool_in') is not None
        sibs_ = node.get('_siblings') or ()
        blocked_into = (into_ is not None and isinstance(into_[0], dev.BlockedHalf)) or \
            any(sib.get('into') is not None and isinstance(sib['into'][0], dev.BlockedHalf) for sib in sibs_)
        if blocked_into and not isinstance(inputs[0], dev.BlockedHalf):
            # plan_c8_modules gave this launch the module's blocked Concat buffer, but a producer in front of it handed over a dense tensor
            # after all (a kernel refused its size at launch): convert it here -- the values the reference holds (float16) -- and go on
            inputs = dict(inputs)
            inputs[0] = dev.BlockedHalf.from_dense(dev.as_device(inputs[0]))
    if isinstance(inputs[0], dev.BlockedHalf) and node.get('_f16_mfma'):
        wants_module = bool(sibs_) or pool_ or bool(node.get('_out_c8')) or blocked_into
        module_path = same and wants_module and c8_multi_ok(inputs[0].shape, kh_, kw_, pool_, 1 + len(sibs_))
        reader_path = not module_path and not sibs_ and not pool_ and c8_reader_ok(node)
    x = inputs[0] if (module_path or reader_path) else dev.as_device(inputs[0])
    if x is not inputs[0]:
        inputs = dict(inputs)
        inputs[0] = x
    bias = node.get('_fuse_bias')
    if bias is not None:
        bias = dev.as_device(bias)
        assert bias.size == w.shape[0]
    siblings = node.get('_siblings')
    into = node.get('_out_into')
    pooled = node.get('_fuse_pool_in') is not None
    if module_path:
        # the module form: a blocked input, and blocked outputs / several members / a MaxPool in front
        members = [(w, bias, into, bool(node.get('_out_c8')))]
        for sib in siblings or ():
            common_def.validate_inputs(sib['node'], sib['inputs'])
            sb = sib.get('bias')
            members.append((dev.as_device(sib['inputs'][1]), dev.as_device(sb) if sb is not None else None, sib.get('into'), bool(sib.get('c8'))))
        outs = launch_c8_multi(node, inputs[0], members, pool=pooled, act=node.get('_fuse_act'))
        y, node['_sibling_out'] = outs[0], outs[1:]
    elif (into is not None and isinstance(into[0], dev.BlockedHalf)) or any(sib.get('into') is not None and isinstance(sib['into'][0], dev.BlockedHalf)
                                                                            for sib in siblings or ()):
        raise RuntimeError('{}: a blocked fp16 Concat buffer, but this launch cannot write it (plan_c8_modules promised a blocked input and '
                           'pvhip_conv2d_f16_c8_multi)'.format(node.get('name')))
    elif reader_path:
        y = launch_c8(node, inputs[0], w, bias=bias, act=node.get('_fuse_act'), into=into)
    elif node.get('_f16_mfma') and node.get('_out_c8') and not siblings and not c8_writer_ok(node) and dev.conv_f16_dma:
        y = launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=bias, act=node.get('_fuse_act'), into=node.get('_out_into'), f16=True, out_c8=True)
    elif node.get('_f16_mfma') and (siblings or node.get('_out_c8')) and dev.conv_f16_dma:
        members = [(w, bias, node.get('_out_into'), bool(node.get('_out_c8')))]
        for sib in siblings or ():
            common_def.validate_inputs(sib['node'], sib['inputs'])
            sb = sib.get('bias')
            members.append((dev.as_device(sib['inputs'][1]), dev.as_device(sb) if sb is not None else None, sib.get('into'), bool(sib.get('c8'))))
        outs = launch_siblings(node, x, members, strides, pads_begin, node.get('_fuse_act'), f16=True)
        y, node['_sibling_out'] = outs[0], outs[1:]
    elif node.get('_f16_mfma') and node.get('_fuse_pool_in') is not None:
        y = launch_pooled(node, x, w, bias=bias, act=node.get('_fuse_act'), into=node.get('_out_into'), f16=True)
    elif node.get('_f16_mfma'):
        y = launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=bias, act=node.get('_fuse_act'), into=node.get('_out_into'), f16=True)
    elif node.get('_fuse_pool_in') is not None:
        y = launch_pooled(node, x, w, bias=bias, act=node.get('_fuse_act'), into=node.get('_out_into'))
    elif siblings:
        members = [(w, bias, node.get('_out_into'))]
        for sib in siblings:
            common_def.validate_inputs(sib['node'], sib['inputs'])
            sb = sib.get('bias')
            members.append((dev.as_device(sib['inputs'][1]), dev.as_device(sb) if sb is not None else None, sib.get('into')))
        outs = launch_siblings(node, x, members, strides, pads_begin, node.get('_fuse_act'))
        y, node['_sibling_out'] = outs[0], outs[1:]
    else:
        y = launch(node, x, w, strides, pads_begin, pads_end, auto_pad, bias=bias, act=node.get('_fuse_act'), into=node.get('_out_into'))
    port = common_def.first_output_port(node)
    assert common_def.type_convert_tbl[node['output'][port]['precision']] == np.float32
    return {port: y}
