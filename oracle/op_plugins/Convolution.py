# ORACLE -- test infrastructure only.  Convolution: CPU restatement of reference op_plugins/Convolution.py:149-176 (special branch :167-168).
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Convolution')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    a = node['data']
    res = ops.convolution_special(inputs[0], inputs[1], ints(a['strides']), ints(a['pads_begin']), ints(a['pads_end']), a['auto_pad'])
    res = res.astype(DTYPES[node['output'][out_port(node)]['precision']])          # Convolution.py:172-174
    return {out_port(node): res}
