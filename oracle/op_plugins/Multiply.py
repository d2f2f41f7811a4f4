# ORACLE -- test infrastructure only.  Multiply: CPU restatement of reference op_plugins/Multiply.py:46-62.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Multiply')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.multiply(inputs[0], inputs[1])
    return {out_port(node): res}
