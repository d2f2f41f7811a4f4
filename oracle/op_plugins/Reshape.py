# ORACLE -- test infrastructure only.  Reshape: CPU restatement of reference op_plugins/Reshape.py:47-61.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Reshape')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = inputs[0].reshape(ops.reshape_dims(inputs[0].shape, inputs[1]))
    return {out_port(node): res}
