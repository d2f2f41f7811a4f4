# ORACLE -- test infrastructure only.  Add: CPU restatement of reference op_plugins/Add.py:57-74.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Add')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.add(inputs[0], inputs[1])
    return {out_port(node): res}
