# ORACLE -- test infrastructure only.  Result: reference op_plugins/Result.py:7-18.
from ._util import check


def name():
    print('Result')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    node['result'] = inputs[0]
    return []
