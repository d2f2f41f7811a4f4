# ORACLE -- test infrastructure only.  Parameter: reference op_plugins/Parameter.py:8-14.
import numpy as np

from ._util import DTYPES


def name():
    print('Parameter')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    precision = DTYPES[node['data']['element_type']]
    return {0: np.array(node['param']).reshape(node['data']['shape']).astype(precision)}
