# ORACLE -- test infrastructure only.  GroupConvolution: CPU restatement of reference op_plugins/GroupConvolution.py:114-137.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('GroupConvolution')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    a = node['data']
    res = ops.group_convolution_depthwise(inputs[0], inputs[1], ints(a['strides']), ints(a['pads_begin']), ints(a['pads_end']), a['auto_pad'])
    return {out_port(node): res}
