# ORACLE -- test infrastructure only.  LRN: CPU restatement of reference op_plugins/LRN.py:41-63.
import numpy as np

from .. import ops
from ._util import DTYPES, check, ints, out_port


def name():
    print('LRN')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    a = node['data']
    res = ops.lrn(inputs[0], float(a['alpha']), float(a['beta']), float(a['bias']), int(a['size']))
    return {out_port(node): res}
