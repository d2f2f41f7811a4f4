# ORACLE -- test infrastructure only.  Transpose: CPU restatement of reference op_plugins/Transpose.py:16-30.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Transpose')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = np.ascontiguousarray(inputs[0].transpose(inputs[1]))                     # Transpose.py:12 (view there)
    return {out_port(node): res}
