# ORACLE -- test infrastructure only.  AvgPool: CPU restatement of reference op_plugins/AvgPool.py:94-118.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('AvgPool')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    a = node['data']
    res = ops.avgpool(inputs[0], ints(a['strides']), ints(a['pads_begin']), ints(a['pads_end']), ints(a['kernel']), a['rounding_type'], a['auto_pad'])
    return {out_port(node): res}
