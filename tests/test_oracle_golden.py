"""The oracle (CPU restatement, oracle/) against the outputs of the REAL reference recorded in
tests/golden/ by make_golden.py -- this is what pins the oracle.  CPU only."""
import os

import numpy as np
import pytest

import helpers
from helpers import GOLDEN, assert_close, build_network, infer_one, layer_sums, load_case

ORACLE_TOL = 2e-6  # oracle vs reference: same numpy underneath, only summation order may differ


@pytest.mark.parametrize('path', helpers.op_case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_op_matches_reference(path):
    import importlib
    node, inputs, want = load_case(path)
    plugin = importlib.import_module('oracle.op_plugins.' + node['type'])
    got = helpers.first_out(plugin.compute(node, inputs, kernel_type='special', debug=False))
    assert got.shape == want.shape
    if want.dtype.kind == 'i':
        assert got.dtype == want.dtype and np.array_equal(got, want), node['name']
    elif node['type'] in helpers.BIT_EXACT:
        helpers.assert_bit_exact(got.astype(np.float32), want, node['name'])
    else:
        assert_close(got, want, ORACLE_TOL, node['name'])


def test_oracle_mnist_end_to_end_and_layer_sums():
    z = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))
    _, net, ex = build_network('oracle.op_plugins', 'mnist')
    ex.kernel_type = 'special'
    for i in range(z['images'].shape[0]):
        got = infer_one(ex, net, z['images'][i:i + 1])
        assert_close(got, z['out'][i:i + 1], ORACLE_TOL, 'mnist image {}'.format(i))
        if i == 0:
            sums = layer_sums(net)
            for nid, want in zip(z['layer_ids'], z['layer_sums']):
                assert abs(sums[int(nid)] - want) <= 1e-5 * max(1.0, abs(want)), 'layer {}'.format(nid)
    # the reference's own integrity assertions (integrity_test.py:57) and README.md:69-72 values
    top = np.argsort(z['out'][0])[::-1]
    assert list(top[:3]) == [2, 0, 1]
    assert abs(z['out'][0][2] - 9.9999917e-01) < 1e-6 and abs(z['out'][0][0] - 7.8985e-07) < 1e-10
    assert int(np.argmax(z['out'][1])) == 7


def test_oracle_mnist_batch_equals_stacked_single_images():
    z = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))
    _, net, ex = build_network('oracle.op_plugins', 'mnist', batch=8)
    got = infer_one(ex, net, z['images'])
    assert_close(got, z['out'], ORACLE_TOL, 'mnist batch 8')


@pytest.mark.parametrize('model,fname,shape', [('googlenet-v1', 'googlenet_e2e.npz', (1, 3, 224, 224)),
                                               ('mnist_bn', 'mnist_bn_e2e.npz', (1, 1, 28, 28))])
def test_oracle_synthetic_models(model, fname, shape):
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, fname))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, model + '.xml'), int(z['weight_seed']))
    _, net, ex = build_network('oracle.op_plugins', model, weights=blob)
    for i, seed in enumerate(z['image_seeds']):
        x = synth.uniform_pixels(int(seed), shape)
        got = infer_one(ex, net, x)
        assert_close(got, z['out'][i:i + 1], 1e-5, '{} image {}'.format(model, i))
        if i == 0:
            sums = layer_sums(net)
            for nid, want in zip(z['layer_ids'], z['layer_sums']):
                assert abs(sums[int(nid)] - want) <= 2e-5 * max(1.0, abs(want)), 'layer {}'.format(nid)


SSD_HEADS = ['concat', 'concat_1', 'do_ExpandDims_conf/sigmoid']


def ssd_backbone(plugin_package, batch, x, fuse=True):
    from pyopenvino_amd import synth
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml'), 1234)
    _, net, ex = build_network(plugin_package, 'ssd_mobilenet_v1_coco', weights=blob, batch=batch, fuse=fuse)
    out = ex.infer_until({net.inputs[0]['name']: x}, SSD_HEADS)
    return {k: np.asarray(v) for k, v in out.items()}


def check_ssd_against_fixture(out, z, tol):
    assert_close(out['concat'][0:1], z['concat'], tol, 'ssd concat (box encodings)')
    assert_close(out['concat_1'][0:1][:, ::3, ::5], z['concat_1_sub'], tol, 'ssd concat_1 (class logits)')
    assert_close(out['do_ExpandDims_conf/sigmoid'][0:1][:, :, ::3, ::5], z['sigmoid_sub'], tol, 'ssd sigmoid')
    s = float(out['concat_1'][0:1].astype(np.float64).sum())
    assert abs(s - float(z['concat_1_sum'])) <= 10 * tol * max(1.0, abs(float(z['concat_1_sum'])))


def test_oracle_ssd_backbone_vs_reference():
    """BASELINE config 5 (backbone + heads; PriorBox / DetectionOutput are host glue outside the path)."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'ssd_backbone_e2e.npz'))
    x = synth.uniform_pixels(int(z['image_seed']), (1, 3, 300, 300))
    check_ssd_against_fixture(ssd_backbone('oracle.op_plugins', 1, x), z, 2e-5)


def test_oracle_ssd_whole_ir_vs_reference():
    """the WHOLE SSD IR (prior-box subgraph and DetectionOutput included) through infer(), against the
    reference's own infer() on the same synthetic weights and image; and the batch rule at N=2."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'ssd_full_e2e.npz'))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml'), int(z['weight_seed']))
    x = synth.uniform_pixels(int(z['image_seed']), (1, 3, 300, 300))
    _, net, ex = build_network('oracle.op_plugins', 'ssd_mobilenet_v1_coco', weights=blob)
    ex.kernel_type = 'special'
    out = helpers.infer_one(ex, net, x)
    assert out.shape == z['out'].shape and np.array_equal(out[..., :2], z['out'][..., :2])
    assert_close(out, z['out'], 1e-6, 'ssd detections')
    by_name = {net.G.nodes[n]['name']: n for n in net.G.nodes}
    priors = next(iter(net.G.nodes[by_name['ConcatPriorBoxesClustered']]['output'].values()))['data']
    helpers.assert_bit_exact(np.asarray(priors), z['priors'], 'prior boxes')
    _, net2, ex2 = build_network('oracle.op_plugins', 'ssd_mobilenet_v1_coco', weights=blob, batch=2)
    ex2.kernel_type = 'special'
    out2 = helpers.infer_one(ex2, net2, np.concatenate([synth.uniform_pixels(9, (1, 3, 300, 300)), x], 0))
    assert out2.shape == (1, 1, 200, 7)
    assert_close(out2[:, :, 100:], z['out'], 1e-6, 'image 1 of a batch of 2')


def check_fp16_ir(plugin_package, tmp_path):
    """models/mnist rewritten as an FP16 IR, computed with fp32 tensors (IENetwork.promote_fp16), against what the
    reference makes of the same IR in numpy float16: equal within float16 resolution where the reference is finite;
    where its float16 exp() overflowed (NaN at the winning class of the two real digits) fp32 gives the probability."""
    z = np.load(os.path.join(GOLDEN, 'mnist_fp16_e2e.npz'))
    images = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))['images'][:int(z['n_images'])]
    blob = open(os.path.join(helpers.MODELS, 'mnist.bin'), 'rb').read()
    xml16, blob16 = helpers.fp16_ir(os.path.join(helpers.MODELS, 'mnist.xml'), blob, str(tmp_path))
    from pyopenvino_amd import IECore
    ie = IECore(plugin_package=plugin_package)
    net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=True)
    assert net.ir_precision == 'FP16' and all(p['precision'] != 'FP16' for n in net.G.nodes for p in net.G.nodes[n].get('output', {}).values())
    net.set_batch(len(images))
    ex = ie.load_network(net)
    ex.kernel_type = 'special'
    got = helpers.infer_one(ex, net, images)
    want = z['out']
    assert got.dtype == np.float32 and got.shape == want.shape and np.isfinite(got).all()
    finite = np.isfinite(want)
    assert np.abs(got[finite] - want[finite]).max() <= 2e-3
    for row in range(len(want)):
        overflowed = np.isnan(want[row])
        if overflowed.any():
            assert overflowed.sum() == 1 and got[row, overflowed][0] > 0.99 and got[row].argmax() == int(np.argmax(overflowed))
    ref32 = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))['out'][:len(images)]     # the FP32 IR: f16 weights barely move it
    assert_close(got, ref32, 2e-3, 'FP16 IR vs the FP32 IR')


def test_oracle_fp16_ir_as_fp32(tmp_path):
    check_fp16_ir('oracle.op_plugins', tmp_path)


def test_oracle_on_the_reference_fp16_node_fixture():
    """conv_node6_fp16.npz (the reference's float16 run of its own node fixture): the oracle's fp32 sum of the same fp16 operands
    is within fp16 tolerance of it -- what pins the fixture the f16-MFMA kernel is tested against on the GPU."""
    import importlib
    import json
    z = np.load(os.path.join(GOLDEN, 'conv_node6_fp16.npz'))
    x, w, ref = z['x'].astype(np.float32), z['w'].astype(np.float32), z['out'].astype(np.float32)
    node = {'name': 'n6', 'type': 'Convolution', 'data': json.loads(str(z['data'])),
            'input': {0: {'precision': 'FP32', 'dims': x.shape}, 1: {'precision': 'FP32', 'dims': w.shape}},
            'output': {2: {'precision': 'FP32', 'dims': ref.shape}}}
    got = next(iter(importlib.import_module('oracle.op_plugins.Convolution').compute(node, {0: x, 1: w}, kernel_type='special').values()))
    assert got.shape == ref.shape and helpers.rel_err(got, ref) <= 2e-2


def test_oracle_fp32_arithmetic_on_the_googlenet_fp16_ir_vs_reference_float16(tmp_path):
    """googlenet_fp16_rows2.npz (the reference's numpy-float16 run of GoogLeNet as an FP16 IR, 2 images): the oracle's fp32 arithmetic on
    the same f16 constants lands within float16 tolerance of its float16 logits and picks the same class -- what pins the fixture the
    f16-MFMA pass is held against on the GPU (bench.py's FP16 entry)."""
    from pyopenvino_amd import IECore, synth
    z = np.load(os.path.join(GOLDEN, 'googlenet_fp16_rows2.npz'))
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    xml16, blob16 = synth.fp16_ir(xml, synth.synth_weights(xml, int(z['weight_seed'])), str(tmp_path))
    ie = IECore(plugin_package='oracle.op_plugins')
    net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=True)
    ex = ie.load_network(net)
    ex.kernel_type = 'special'
    soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
    pre = next(iter(net.G.pred[soft]))
    for i, seed in enumerate(z['image_seeds'][:1]):
        helpers.infer_one(ex, net, synth.uniform_pixels(int(seed), (1, 3, 224, 224)))
        logits = np.asarray(next(iter(net.G.nodes[pre]['output'].values()))['data'])
        assert helpers.rel_err(logits, z['logits'][i:i + 1]) <= 3e-3
        assert logits.argmax() == z['logits'][i].argmax()
