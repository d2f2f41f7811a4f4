"""Shared helpers of the plugin boundary: IR precision tables and attribute-string parsers.

Mirrors the role of the reference's ``pyopenvino/common_def.py`` (dtype tables ``:13-17``, attribute
parsers ``:21-32``).  Only what the hot path needs is kept; the Windows-console and debug-dump helpers
of the reference are not part of the path.
"""
import math

import numpy as np

# IR precision / element_type string -> (struct code, bytes per element)   [reference common_def.py:13-14]
format_config = {
    'FP32': ('f', 4), 'F32': ('f', 4), 'FP16': ('e', 2), 'F16': ('e', 2),
    'I64': ('q', 8), 'I32': ('i', 4), 'I16': ('h', 2), 'I8': ('b', 1), 'U8': ('B', 1),
}

# IR precision / element_type string -> numpy dtype                         [reference common_def.py:16-17]
type_convert_tbl = {
    'f32': np.float32, 'f16': np.float16, 'i64': np.int64, 'i32': np.int32, 'i16': np.int16,
    'i8': np.int8, 'u8': np.uint8,
    'FP32': np.float32, 'FP16': np.float16, 'I64': np.int64, 'I32': np.int32,
}


def string_to_boolean(text: str) -> bool:
    """'true' / 'TRUE' / '1' -> True, anything else False (reference common_def.py:21-24)."""
    return text.strip().upper() in ('TRUE', '1')


def string_to_tuple(text: str) -> tuple:
    """'1, 2' -> (1, 2) (reference common_def.py:26-28)."""
    return tuple(int(tok) for tok in text.split(','))


def string_to_tuple_float(text: str) -> tuple:
    """'0.1, 0.2' -> (0.1, 0.2) (reference common_def.py:30-32)."""
    return tuple(float(tok) for tok in text.split(','))


def pooled_extent(size: int, kernel: int, stride: int, pad_begin: int, pad_end: int,
                  rounding_type: str, auto_pad: str, same_means_input: bool) -> int:
    """Output extent of one spatial axis for Convolution / pooling.

    Restates the shape rule shared by ``Convolution.calc_output_shape`` (Convolution.py:21-49),
    ``MaxPool.calc_output_shape`` (MaxPool.py:10-38) and ``AvgPool.calc_output_shape`` (AvgPool.py:10-38):

    * ``explicit``: round((size + pad_begin + pad_end - kernel) / stride) + 1
    * ``valid``   : round((size - kernel) / stride) + 1  (pads ignored for the extent)
    * ``same_*``  : ceil(size / stride) for convolutions; the pooling plugins return the input extent
      unchanged, ignoring the stride (``same_means_input=True``) -- a reference quirk kept on purpose.
    """
    if auto_pad not in ('explicit', 'valid', 'same_upper', 'same_lower'):
        raise AssertionError('unknown auto_pad {!r}'.format(auto_pad))
    if rounding_type not in ('floor', 'ceil'):
        raise AssertionError('unknown rounding_type {!r}'.format(rounding_type))
    rnd = math.floor if rounding_type == 'floor' else math.ceil
    if auto_pad == 'explicit':
        return rnd((size + pad_begin + pad_end - kernel) / stride) + 1
    if auto_pad == 'valid':
        return rnd((size - kernel) / stride) + 1
    return size if same_means_input else math.ceil(size / stride)


def validate_inputs(node: dict, inputs: dict) -> None:
    """The per-plugin input check of the reference (e.g. Convolution.py:153-157): dtype and dims of
    every input must equal what the IR port declares.  Works for ndarray and DeviceTensor alike
    (both expose ``.dtype`` and ``.shape``)."""
    for port, data in inputs.items():
        decl = node['input'][port]
        assert data.dtype == type_convert_tbl[decl['precision']], \
            '{}: port {} dtype {} != {}'.format(node.get('name'), port, data.dtype, decl['precision'])
        assert tuple(data.shape) == tuple(decl['dims']), \
            '{}: port {} shape {} != {}'.format(node.get('name'), port, tuple(data.shape), decl['dims'])


def first_output_port(node: dict) -> int:
    """Plugins return ``{first output port id: tensor}`` (e.g. ReLU.py:38-39)."""
    return next(iter(node['output']))
