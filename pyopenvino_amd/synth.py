"""Deterministic synthetic ``.bin`` blobs and inputs for IRs whose weights are not shipped.

The reference repository ships ``models/mnist.bin`` only (``.MISSING_LARGE_BLOBS``): GoogLeNet, mnist_bn and
SSD-MobileNet come as ``.xml`` without weights.  Benchmarks and parity tests therefore run them on seeded
synthetic weights laid out from each Const layer's ``offset / size / shape / element_type``.  The generator
is a counter-based integer hash (splitmix64) followed by Box-Muller, written out here so that the build
container (where the golden outputs of the reference are recorded) and the GPU box regenerate bit-identical
blobs from the seed -- nothing depends on numpy's own Generator streams.

Value recipe (keeps activations O(1) and GoogLeNet logits well inside exp()'s fp32 range, so the
reference's un-shifted SoftMax stays finite):
  Convolution / MatMul weights  N(0, 2 / fan_in); the first Convolution fed (through Add/Multiply) by the
                                Parameter is divided by 64 more (raw 0..255 pixels);
  GroupConvolution weights      N(0, 2 / (kh*kw));
  Add constants                 N(0, 0.05^2); a (1,3,1,1) Add straight on the Parameter is the mean
                                (-104, -117, -123);
  Multiply constants            1 + N(0, 0.1^2) (folded BatchNorm scale); a scalar one on the Parameter is 1/127.5;
  I64 constants                 real values: LRN axes [1]; Reshape target [0, -1, out dims 2..]; Transpose
                                permutation NCHW -> NHWC; StridedSlice begin / end / stride and Unsqueeze axes
                                reconstructed from the port dims (SSD prior-box subgraph); others zero.
"""
import xml.etree.ElementTree as et

import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, count: int) -> np.ndarray:
    """`count` float64 values in (0, 1): value i = hash(seed, stream, i)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0x100000001B3) + np.uint64(stream))
        ctr = np.arange(count, dtype=np.uint64) + base
        bits = _splitmix64(ctr)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(seed: int, stream: int, count: int) -> np.ndarray:
    """Standard normal float64 values (Box-Muller over two hashed uniforms)."""
    u1 = uniform01(seed, 2 * stream, count)
    u2 = uniform01(seed, 2 * stream + 1, count)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def uniform_pixels(seed: int, shape) -> np.ndarray:
    """Integer pixel values 0..255 as float32 (what cv2.imread(...).astype(float32) hands the reference)."""
    n = int(np.prod(shape))
    return np.floor(uniform01(seed, 0x5EED, n) * 256.0).astype(np.float32).reshape(shape)


def _layers(xml_path):
    root = et.parse(xml_path).getroot()
    layers = {}
    for layer in root.iterfind('./layers/layer'):
        lid = int(layer.attrib['id'])
        ports = {}
        for tag in ('input', 'output'):
            sec = layer.find(tag)
            if sec is not None:
                for port in sec.findall('port'):
                    ports[(tag, int(port.attrib['id']))] = tuple(int(d.text) for d in port.findall('dim'))
        data = layer.find('data')
        layers[lid] = {'type': layer.attrib['type'], 'name': layer.attrib['name'],
                       'data': dict(data.attrib) if data is not None else {}, 'ports': ports}
    edges = [(int(e.attrib['from-layer']), int(e.attrib['from-port']), int(e.attrib['to-layer']), int(e.attrib['to-port']))
             for e in root.iterfind('./edges/edge')]
    return layers, edges


def synth_weights(xml_path: str, seed: int = 1234) -> bytes:
    """Build the whole ``.bin`` blob for `xml_path` from `seed`."""
    layers, edges = _layers(xml_path)
    consumers = {}
    producer = {}
    for src, sp, dst, dp in edges:
        consumers.setdefault(src, []).append((dst, dp))
        producer[(dst, dp)] = src
    params = {lid for lid, l in layers.items() if l['type'] == 'Parameter'}

    def fed_by_parameter(lid, depth=0):
        """Is layer `lid`'s data input the Parameter, possibly through Add / Multiply preprocessing?"""
        src = producer.get((lid, 0))
        if src is None or depth > 4:
            return False
        if src in params:
            return True
        if layers[src]['type'] in ('Add', 'Multiply'):
            return fed_by_parameter(src, depth + 1) or (producer.get((src, 1)) in params)
        return False

    size_total = 0
    for l in layers.values():
        if l['type'] == 'Const':
            size_total = max(size_total, int(l['data']['offset']) + int(l['data']['size']))
    blob = bytearray(size_total)

    for lid in sorted(layers):
        l = layers[lid]
        if l['type'] != 'Const':
            continue
        d = l['data']
        offset, size = int(d['offset']), int(d['size'])
        shape = tuple(int(t) for t in d['shape'].split(',')) if d['shape'].strip() else ()
        etype = d['element_type'].lower()
        count = int(np.prod(shape)) if len(shape) else 1
        uses = consumers.get(lid, [])
        dst, dport = uses[0] if uses else (None, None)
        dtype_name = layers[dst]['type'] if dst is not None else None
        stream = offset + 1  # one PRNG stream per blob region: shared offsets get identical content
        if etype in ('i64', 'i32'):
            width = np.int64 if etype == 'i64' else np.int32
            vals = np.zeros(count, dtype=width)
            if dtype_name == 'LRN':
                vals[:] = 1
            elif dtype_name == 'Reshape':
                out_dims = next(v for (tag, _), v in layers[dst]['ports'].items() if tag == 'output')
                # batch-agnostic target, and identical for layers that share one constant in the blob (the SSD
                # heads do): copy the batch axis, infer the first remaining axis, keep the others
                tgt = [0] + list(out_dims[1:])
                if len(tgt) > 1:
                    tgt[1] = -1
                vals[:] = np.array(tgt[:count], dtype=width)
            elif dtype_name == 'StridedSlice':
                # the SSD head slices [H, W] out of a ShapeOf vector: begin = len(in) - len(out), end = len(in), stride 1
                n_in = layers[dst]['ports'][('input', 0)][0]
                n_out = next(v for (tag, _), v in layers[dst]['ports'].items() if tag == 'output')[0]
                vals[:] = {1: n_in - n_out, 2: n_in, 3: 1}[dport]
            elif dtype_name == 'Unsqueeze':
                in_dims = layers[dst]['ports'][('input', 0)]
                out_dims = next(v for (tag, _), v in layers[dst]['ports'].items() if tag == 'output')
                import itertools
                axes = next(c for c in itertools.combinations(range(len(out_dims)), len(out_dims) - len(in_dims))
                            if all(out_dims[i] == 1 for i in c) and
                            tuple(d for i, d in enumerate(out_dims) if i not in c) == tuple(in_dims))
                vals[:] = np.array(axes[:count], dtype=width)
            elif dtype_name == 'Transpose':
                in_dims = layers[dst]['ports'][('input', 0)]
                perm = [0, 2, 3, 1] if len(in_dims) == 4 else list(range(len(in_dims)))[::-1]
                vals[:] = np.array(perm[:count], dtype=width)
            blob[offset:offset + size] = vals.astype('<' + np.dtype(width).str[1:]).tobytes()[:size]
            continue
        if etype not in ('f32',):
            raise NotImplementedError('synthetic constant of type {}'.format(etype))
        z = normal(seed, stream, count)
        if dtype_name in ('Convolution',) and dport == 1:
            fan_in = int(np.prod(shape[1:]))
            vals = z * np.sqrt(2.0 / fan_in)
            if fed_by_parameter(dst):
                vals = vals / 64.0
        elif dtype_name == 'GroupConvolution' and dport == 1:
            vals = z * np.sqrt(2.0 / float(shape[-1] * shape[-2]))
        elif dtype_name == 'MatMul':
            fan_in = shape[-1] if layers[dst]['data'].get('transpose_b', 'false') == 'true' else shape[0]
            vals = z * np.sqrt(2.0 / fan_in)
        elif dtype_name == 'Add':
            other = producer.get((dst, 0))
            if other in params and count == 3:
                vals = np.array([-104.0, -117.0, -123.0])
            else:
                vals = z * 0.05
        elif dtype_name == 'Multiply':
            other = producer.get((dst, 1 - dport))
            if count == 1 and other in params:
                vals = np.array([1.0 / 127.5])
            else:
                vals = 1.0 + 0.1 * z
        else:
            vals = z * 0.05
        blob[offset:offset + size] = np.asarray(vals, dtype='<f4').tobytes()[:size]
    return bytes(blob)


def fp16_ir(xml_path, blob, out_dir):
    """Write the FP16 twin of an FP32 IR (what Model Optimizer's --data_type FP16 produces): every port FP16, every f32
    constant stored as f16 in a new blob, the Parameter f16.  Returns (path of the new .xml, the new blob)."""
    import xml.etree.ElementTree as et
    tree = et.parse(xml_path)
    root = tree.getroot()
    src = memoryview(blob)
    out = bytearray()
    moved = {}                                   # constants that share a blob region keep sharing it
    for layer in root.iterfind('./layers/layer'):
        for port in layer.iter('port'):
            if port.attrib.get('precision') == 'FP32':
                port.attrib['precision'] = 'FP16'
        data = layer.find('data')
        if data is None:
            continue
        if layer.attrib['type'] == 'Parameter' and data.attrib.get('element_type') == 'f32':
            data.attrib['element_type'] = 'f16'
        if layer.attrib['type'] != 'Const':
            continue
        offset, size = int(data.attrib['offset']), int(data.attrib['size'])
        key = (offset, size, data.attrib['element_type'])
        if key not in moved:
            raw = bytes(src[offset:offset + size])
            if data.attrib['element_type'] == 'f32':
                raw = np.frombuffer(raw, dtype='<f4').astype('<f2').tobytes()
            while len(out) % 8:
                out.append(0)
            moved[key] = (len(out), len(raw))
            out += raw
        data.attrib['offset'], data.attrib['size'] = str(moved[key][0]), str(moved[key][1])
        if data.attrib['element_type'] == 'f32':
            data.attrib['element_type'] = 'f16'
    path = os.path.join(out_dir, os.path.basename(xml_path)[:-4] + '_fp16.xml')
    tree.write(path)
    return path, bytes(out)
