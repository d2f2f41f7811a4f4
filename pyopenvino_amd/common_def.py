"""Shared helpers of the plugin boundary: IR precision tables and attribute-string parsers.

Mirrors the role of the reference's ``pyopenvino/common_def.py`` (dtype tables ``:13-17``, attribute
parsers ``:21-32``) and its debug helpers of the per-layer compare hook (``print_dict`` ``:60-67``,
``compare_results`` ``:71-105``, ``disp_result`` ``:109-116``, ``dump_graph`` ``:120-126``).  The Windows-console
switch (``enable_escape_sequence`` ``:35-56``) is a no-op here: the box is Linux.
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


# ---- debug helpers of the reference's per-layer compare hook ---------------------------------------------------------

def enable_escape_sequence() -> bool:
    """The reference turns on ANSI colours of the Windows console (common_def.py:35-56) and returns None elsewhere;
    a Linux terminal needs nothing."""
    return True


def print_dict(dic: dict, indent_level: int = 0, indent_step: int = 4) -> None:
    """One ``key : value`` line per entry, nested dicts indented one step deeper (common_def.py:60-67)."""
    pad = ' ' * (indent_step * indent_level)
    for key, val in dic.items():
        if isinstance(val, dict):
            print(pad, key, ': ')
            print_dict(val, indent_level + 1, indent_step)
        else:
            print(pad, key, ': ', val)


def golden_array(entry):
    """The tensor of one entry of an ``expected_result`` dict.  The reference's format is ``{node name: [precision, dims,
    ndarray]}`` and it reads ``GT[node_name][2]`` (common_def.py:76); a bare ndarray (what this build's tests pass) is taken too."""
    if isinstance(entry, (list, tuple)) and len(entry) == 3 and not np.isscalar(entry[2]) and isinstance(entry[0], str):
        return np.asarray(entry[2])
    return np.asarray(entry)


def compare_results(node_name: str, result, GT: dict, disp_results: bool = False, rtol: float = 1.0):
    """Per-layer compare of the hook ``Executable_Network.expected_result`` (common_def.py:71-105, called from
    inference_engine.py:284-287): prints the node name green when ``np.allclose(result, GT[node_name][2], rtol=1)`` holds --
    the reference's (very loose) rule, kept as the default so that its golden dicts print what they print there -- red otherwise,
    'Skipped' for a node without an entry.  Returns True / False / None (the reference returns nothing).  ``result`` may be a
    DeviceTensor: it is copied to the host here."""
    if node_name not in GT:
        print('{} : Skipped'.format(node_name))
        return None
    got = np.asarray(result)
    want = golden_array(GT[node_name]).astype(got.dtype)
    ok = got.shape == want.shape and bool(np.allclose(got, want, rtol=rtol))
    print('{}{} : {} / {}\x1b[37m'.format('\x1b[32m' if ok else '\x1b[31m', node_name, got.shape, want.shape))
    if not ok and disp_results:
        if got.shape == want.shape:
            print(int(np.count_nonzero(np.isclose(got, want))))
        print('* Result')
        print(got)
        print('* GT')
        print(want)
    return ok


def disp_result(data) -> None:
    """Image 0 of an NCHW tensor, plane by plane, ``%6.3f`` per value (common_def.py:109-116)."""
    data = np.asarray(data)
    _, C, H, W = data.shape
    for c in range(C):
        print('C=', c)
        for h in range(H):
            print(''.join('{:6.3f},'.format(data[0, c, h, w]) for w in range(W)))


def dump_graph(G) -> None:
    """Every node and edge of the ``nx.DiGraph`` with its attribute dict (common_def.py:120-126)."""
    for node_id, contents in G.nodes.items():
        print('node id=', node_id)
        print_dict(contents)
    for edge_id, contents in G.edges.items():
        print('edge_id=', edge_id)
        print(' ' * 2, contents)
