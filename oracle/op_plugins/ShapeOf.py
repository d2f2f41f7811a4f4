# ORACLE -- test infrastructure only.  ShapeOf: CPU restatement of reference op_plugins/ShapeOf.py:10-25.
import numpy as np

from ._util import DTYPES, check, out_port


def name():
    print('ShapeOf')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    port = out_port(node)
    return {port: np.array(node['input'][next(iter(node['input']))]['dims'], dtype=DTYPES[node['output'][port]['precision']])}
