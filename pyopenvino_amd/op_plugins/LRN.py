# LRN -- HIP plugin.  Replaces kernel_LRN_numpy (reference op_plugins/LRN.py:10-22): cross-channel
# window of `size`, alpha NOT divided by size; the axes input (port 1) is validated and unused.
import ctypes

from .. import common_def
from .. import device as dev
from . import MaxPool

# The engine may hand over an LRN whose only consumer is a 3x3 MaxPool as one call: node['_fuse_pool'] is then the
# MaxPool's node dict, the kernel pools the normalised values straight out of LDS and the LRN tensor (written once and
# read once otherwise) never exists; what is returned is the MaxPool's output.  The engine asks pool_fusable() first.
SUPPORTS_FUSED_POOL = True


# A pass made of such nodes can be recorded into a hipGraph (Executable_Network.infer does so by itself for device-resident inputs):
# nothing in compute() synchronises with the host or reads a tensor back once the constants are cached.
GRAPH_CAPTURE_SAFE = True

def _pool_geometry(pool_node: dict, h: int, w: int):
    attrs = pool_node['data']
    strides = common_def.string_to_tuple(attrs['strides'])
    pads_begin = common_def.string_to_tuple(attrs['pads_begin'])
    pads_end = common_def.string_to_tuple(attrs['pads_end'])
    kernel = common_def.string_to_tuple(attrs['kernel'])
    oh, ow = MaxPool.calc_output_shape((h, w), kernel, strides, pads_begin, pads_end, attrs['rounding_type'], attrs['auto_pad'])
    return kernel, strides, pads_begin, pads_end, oh, ow


def pool_fusable(node: dict, pool_node: dict) -> bool:
    """True when libpvhip's fused LRN -> MaxPool kernel covers this pair (shapes from the IR ports; no device needed)."""
    try:
        dims = node['input'][0]['dims']
        if len(dims) != 4 or pool_node['input'][0]['precision'] != 'FP32':
            return False
        n, c, h, w = (int(d) for d in dims)
        attrs = node['data']
        kernel, strides, pads_begin, pads_end, oh, ow = _pool_geometry(pool_node, h, w)
        if len(kernel) != 2 or tuple(pool_node['output'][common_def.first_output_port(pool_node)]['dims']) != (n, c, oh, ow):
            return False
        return bool(dev.call('pvhip_lrn_maxpool_supported', n, c, h, w, int(attrs['size']), float(attrs['beta']),
                             float(attrs['bias']), oh, ow, kernel[0], kernel[1], strides[0], strides[1],
                             pads_begin[0], pads_begin[1], pads_end[0], pads_end[1]))
    except (KeyError, ValueError, AssertionError):
        return False


def blocked_ok(node: dict, pool_node: dict) -> bool:
    """True when compute() runs LRN + MaxPool (node['_fuse_pool'] = pool_node) on a dev.BlockedHalf input as it is and returns a
    dev.BlockedHalf (FP16 IRs; IR attributes and port dims, no device needed).  The ONE predicate of the plan
    (Executable_Network.plan_c8_modules) and of compute()."""
    try:
        if pool_node is None or int(node['data']['size']) != 5:
            return False
        dims = node['input'][0]['dims']
        if len(dims) != 4:
            return False
        h, w = int(dims[2]), int(dims[3])
        kernel, strides, pads_begin, pads_end, oh, ow = _pool_geometry(pool_node, h, w)
        return tuple(kernel) == (3, 3) and oh > 0 and ow > 0 and bool(dev.call(
            'pvhip_lrn_maxpool3x3_c8_supported', h, w, oh, ow, strides[0], strides[1], pads_begin[0], pads_begin[1], 5))
    except (KeyError, ValueError, AssertionError, IndexError, TypeError):
        return False


def name():
    print('LRN')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    attrs = node['data']
    alpha = float(attrs['alpha'])
    beta = float(attrs['beta'])
    bias = float(attrs['bias'])
    size = int(attrs['size'])
    pool_node = node.get('_fuse_pool')
    if isinstance(inputs[0], dev.BlockedHalf) and blocked_ok(node, pool_node):
        # FP16 IRs: the input is fp16 blocked by eight channels (what the reference holds here is a float16 tensor): LRN + MaxPool on it as it
        # is, the output is blocked too
        xb = inputs[0]
        n, c, h, w = xb.shape
        kernel, strides, pads_begin, pads_end, oh, ow = _pool_geometry(pool_node, h, w)
        yb = dev.BlockedHalf((n, c, oh, ow))
        dev.call('pvhip_lrn_maxpool3x3_c8', ctypes.c_void_p(xb.ptr), ctypes.c_void_p(yb.ptr), n, c, h, w, size, alpha, beta, bias, oh, ow,
                 strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1])
        return {common_def.first_output_port(node): yb}
    x = dev.as_device(inputs[0])
    n, c, h, w = x.shape
    if pool_node is not None:
        kernel, strides, pads_begin, pads_end, oh, ow = _pool_geometry(pool_node, h, w)
        y = dev.DeviceTensor.empty((n, c, oh, ow))
        dev.call('pvhip_lrn_maxpool_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h, w, size, alpha, beta, bias,
                 oh, ow, kernel[0], kernel[1], strides[0], strides[1], pads_begin[0], pads_begin[1], pads_end[0], pads_end[1])
        return {common_def.first_output_port(node): y}
    y = dev.DeviceTensor.empty(x.shape)
    dev.call('pvhip_lrn_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n, c, h * w, size, alpha, beta, bias)
    return {common_def.first_output_port(node): y}
