# ORACLE -- test infrastructure only.  Unsqueeze: CPU restatement of reference op_plugins/Unsqueeze.py:10-31.
import numpy as np

from ._util import check, out_port


def name():
    print('Unsqueeze')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    return {out_port(node): np.expand_dims(inputs[0], [int(a) for a in inputs[1]])}
