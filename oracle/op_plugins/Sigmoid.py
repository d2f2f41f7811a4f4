# ORACLE -- test infrastructure only.  Sigmoid: CPU restatement of reference op_plugins/Sigmoid.py:24-40.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Sigmoid')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.sigmoid(inputs[0])
    return {out_port(node): res}
