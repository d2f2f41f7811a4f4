# ORACLE -- test infrastructure only.
import numpy as np

DTYPES = {'f32': np.float32, 'f16': np.float16, 'i64': np.int64, 'i32': np.int32,
          'FP32': np.float32, 'FP16': np.float16, 'I64': np.int64, 'I32': np.int32}


def ints(text):
    return tuple(int(t) for t in text.split(','))


def check(node, inputs):
    """The plugins' input validation (e.g. Convolution.py:153-157)."""
    for port, data in inputs.items():
        decl = node['input'][port]
        assert data.dtype == DTYPES[decl['precision']]
        assert data.shape == decl['dims']


def out_port(node):
    return next(iter(node['output']))
