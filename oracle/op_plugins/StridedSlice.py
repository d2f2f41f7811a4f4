# ORACLE -- test infrastructure only.  StridedSlice: CPU restatement of reference op_plugins/StridedSlice.py:9-49.
from .. import ops
from ._util import check, out_port


def name():
    print('StridedSlice')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    return {out_port(node): ops.strided_slice(inputs[0], inputs[1], inputs[2], inputs[3])}
