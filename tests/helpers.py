"""Shared test helpers: golden-fixture loading, tolerance checks, model builders."""
import glob
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
MODELS = os.path.join(REPO, 'models')

# Ops whose HIP result must equal the reference bit for bit (pure selection / copy / one IEEE op)
BIT_EXACT = {'ReLU', 'MaxPool', 'Add', 'Multiply', 'Concat', 'Transpose', 'Reshape', 'Clamp', 'ShapeOf', 'StridedSlice', 'Unsqueeze',
             'PriorBoxClustered'}
# Stated tolerance of the path (BASELINE.json north_star): 1e-4 relative, fp32
REL_TOL = 1e-4


def setenv(monkeypatch, name, value):
    """Set (value None: unset) a PVHIP_* variable for this test and make libpvhip read its variables again: the library parses
    them once, never on a launch path.  conftest.py re-reads them after every test (when monkeypatch has restored them)."""
    from pyopenvino_amd import device
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, value)
    device.reload_settings()


def op_case_files():
    return sorted(glob.glob(os.path.join(GOLDEN, 'ops', '*.npz'))) + [os.path.join(GOLDEN, 'conv_node6_crop.npz')]


def load_case(path):
    """-> (node dict as the engine would build it, {port: ndarray}, expected output)."""
    z = np.load(path, allow_pickle=False)
    raw = json.loads(str(z['node']))
    node = {k: v for k, v in raw.items() if k not in ('input', 'output')}
    for tag in ('input', 'output'):
        node[tag] = {int(p): {'precision': d['precision'], 'dims': tuple(d['dims'])} for p, d in raw[tag].items()}
    inputs = {p: z['in{}'.format(p)] for p in node['input']}
    return node, inputs, z['out']


def rel_err(got, want):
    """max |got - want| / max |want| over finite entries; non-finite entries must match exactly."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, 'shape {} != {}'.format(got.shape, want.shape)
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN pattern differs'
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), 'inf pattern differs'
    if not fin.any():
        return 0.0
    scale = max(np.abs(want[fin]).max(), 1e-30)
    return float(np.abs(got[fin] - want[fin]).max() / scale)


def elementwise_excess(got, want, tol=REL_TOL):
    """max over elements of |got - want| / (tol * |want| + tol * rms(want)): <= 1 means every element is within `tol` RELATIVE to
    its own magnitude, with a floor of `tol` x the tensor's rms for elements near zero (cancellation in a sum leaves an absolute
    error of that order).  Much tighter than rel_err's max-norm for tensors with a large dynamic range (SoftMax rows, conv
    outputs near zero)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, 'shape {} != {}'.format(got.shape, want.shape)
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN pattern differs'
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), 'inf pattern differs'
    if not fin.any():
        return 0.0
    w, g = want[fin], got[fin]
    rms = float(np.sqrt(np.mean(w * w)))
    return float((np.abs(g - w) / (tol * np.abs(w) + tol * max(rms, 1e-30))).max())


def assert_close(got, want, tol=REL_TOL, what='', elementwise=True):
    """The stated tolerance of the path, `tol` relative, in BOTH norms: max-norm (rel_err) and element by element
    (elementwise_excess, against REL_TOL unless `tol` is looser).  Returns the max-norm error."""
    err = rel_err(got, want)
    assert err <= tol, '{}: relative error {:.3e} > {:.1e}'.format(what, err, tol)
    if elementwise:
        ex = elementwise_excess(got, want, max(tol, REL_TOL))
        assert ex <= 1.0, '{}: an element is {:.2f} x outside |d| <= {:.0e} |want| + {:.0e} rms(want)'.format(what, ex, max(tol, REL_TOL), max(tol, REL_TOL))
    return err


def assert_bit_exact(got, want, what=''):
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, '{}: {} {} vs {} {}'.format(what, got.shape, got.dtype, want.shape, want.dtype)
    # compare bit patterns, but let +0.0 == -0.0 only where the reference itself is ambiguous: it is not -- exact bits
    same = got.view(np.uint32) == want.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(want)
    assert bool(np.all(same | both_nan)), '{}: {} of {} elements differ bitwise'.format(what, int((~(same | both_nan)).sum()), got.size)


def first_out(res):
    return np.asarray(next(iter(res.values())))


def build_network(plugin_package, model, weights=None, batch=1, fuse=True):
    """fuse=False dispatches every node on its own (needed when per-layer outputs are compared)."""
    from pyopenvino_amd import IECore
    ie = IECore(plugin_package=plugin_package)
    net = ie.read_network(os.path.join(MODELS, model + '.xml'), weights)
    if batch != 1:
        net.set_batch(batch)
    ex = ie.load_network(net)
    if not fuse:
        ex.fuse_epilogues = False
        ex.plan_fusion()
    return ie, net, ex


def infer_one(ex, net, x):
    res = ex.infer({net.inputs[0]['name']: x})
    return np.asarray(res[net.outputs[0]['name']])


def layer_sums(net):
    """{node id: float64 sum of its fp32 output} after an infer (per-layer checksum)."""
    sums = {}
    for nid in net.G.nodes:
        node = net.G.nodes[nid]
        if node['type'] in ('Const', 'Result') or 'output' not in node:
            continue
        for p in node['output'].values():
            if 'data' in p and p['data'].dtype == np.float32:
                sums[int(nid)] = float(np.asarray(p['data'], dtype=np.float64).sum())
    return sums


def fp16_ir(xml_path, blob, out_dir):
    """The FP16 twin of an FP32 IR (pyopenvino_amd.synth.fp16_ir)."""
    from pyopenvino_amd import synth
    return synth.fp16_ir(xml_path, blob, out_dir)
