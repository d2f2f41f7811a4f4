# ORACLE -- test infrastructure only.  SoftMax: CPU restatement of reference op_plugins/SoftMax.py:28-45.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('SoftMax')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.softmax_rows(inputs[0])
    return {out_port(node): res}
