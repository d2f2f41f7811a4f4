# ORACLE -- test infrastructure only.  Concat: CPU restatement of reference op_plugins/Concat.py:16-33.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Concat')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    axis = int(node['data']['axis'])
    assert axis <= inputs[0].ndim
    res = ops.concat(inputs.values(), axis)
    return {out_port(node): res}
