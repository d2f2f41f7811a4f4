# ORACLE -- test infrastructure only.  Clamp: CPU restatement of reference op_plugins/Clamp.py:35-55.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('Clamp')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    res = ops.clamp(inputs[0], float(node['data']['min']), float(node['data']['max']))
    return {out_port(node): res}
