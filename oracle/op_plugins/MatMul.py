# ORACLE -- test infrastructure only.  MatMul: CPU restatement of reference op_plugins/MatMul.py:45-64.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('MatMul')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.matmul(inputs[0], inputs[1], node['data']['transpose_a'], node['data']['transpose_b'])
    return {out_port(node): res}
