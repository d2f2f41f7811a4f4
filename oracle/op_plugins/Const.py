# ORACLE -- test infrastructure only.  Const: reference op_plugins/Const.py:8-14.
import numpy as np

from ._util import DTYPES


def name():
    print('Const')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    precision = DTYPES[node['data']['element_type']]
    return {0: np.array(node['const']['data'], dtype=precision).reshape(node['data']['shape'])}
