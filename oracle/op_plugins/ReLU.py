# ORACLE -- test infrastructure only.  ReLU: CPU restatement of reference op_plugins/ReLU.py:23-40.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('ReLU')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.relu(inputs[0])
    return {out_port(node): res}
