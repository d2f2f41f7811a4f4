#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported from where it lies (read-only tree, cwd = /root/reference because its plugin
loader globs 'pyopenvino/op_plugins' relative to the cwd -- inference_engine.py:40-43,51) with an empty
stand-in module for cv2 (imported but never used inside pyopenvino/, DetectionOutput.py:36).  Everything
written here is DATA: seeded inputs, node attribute dicts and the reference's outputs.

Fixtures
  ops/<case>.npz           per-op known answers: in<port> arrays, 'out', and 'node' (JSON of the node dict)
                           computed by the reference plugin's compute(node, inputs, kernel_type='special')
  mnist_e2e.npz            models/mnist (real weights): 8 images (mnist2, mnist7, 6 seeded noise images), the
                           reference's (1,10) output per image stacked, and per-layer float64 sums for mnist2
  googlenet_e2e.npz        models/googlenet-v1 on synthetic weights (pyopenvino_amd.synth, seed 1234): 2 seeded
                           images, (2,1000) outputs stacked from two N=1 runs, per-layer sums of image 0
  mnist_bn_e2e.npz         models/mnist_bn on synthetic weights, 2 images
  ssd_backbone_e2e.npz     models/ssd_mobilenet_v1_coco backbone + heads (up to 'concat', 'concat_1', the Sigmoid) on
                           synthetic weights, 1 image: 'concat' in full, the two class tensors subsampled + float64 sums
  ssd_full_e2e.npz         the whole SSD IR (prior boxes + DetectionOutput included) through the reference's infer(),
                           synthetic weights, 1 image: the (1,1,100,7) detections and the (1,2,7668) prior tensor
  mnist_fp16_e2e.npz       models/mnist rewritten as an FP16 IR (pyopenvino_amd.synth.fp16_ir) through the reference (numpy float16), 3 images
  conv_node6_fp16.npz      the same node fixture in float16 as the reference computes it (x, w, float16 output), cropped to 64x64
  googlenet_fp16_rows2.npz models/googlenet-v1 rewritten as an FP16 IR (synthetic weights seed 1234 rounded to f16) through the reference, which
                           computes it in numpy float16: float16 logits (the MatMul + Add in front of the SoftMax) of 2 seeded images
  googlenet_rows8.npz      the reference's N=1 answers for 8 seeded images on the synthetic GoogLeNet weights
  conv_node6_crop.npz      the reference's own single-node fixture resources/node_args_6.pickle (SSD Conv2d_0,
                           3x3 stride 2 same_upper pads (0,0)/(1,1)), input cropped to 64x64 and cast to fp32
"""
import json
import os
import pickle
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from pyopenvino_amd import synth  # noqa: E402  (the build's own seeded generator)


def import_reference():
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    os.chdir(REF)
    sys.path.insert(0, REF)
    from pyopenvino.inference_engine import IECore  # noqa
    return IECore


def rnd(seed, shape, scale=1.0, shift=0.0):
    n = int(np.prod(shape))
    return (synth.normal(seed, 77, n) * scale + shift).astype(np.float32).reshape(shape)


def port(prec, dims):
    return {'precision': prec, 'dims': tuple(int(d) for d in dims)}


def make_node(type_, ins, out_dims, data=None, out_prec='FP32', name=None):
    node = {'name': name or type_.lower() + '_case', 'type': type_, 'version': 'opset1'}
    if data is not None:
        node['data'] = dict(data)
    node['input'] = {i: port('I64' if a.dtype == np.int64 else 'FP32', a.shape) for i, a in enumerate(ins)}
    node['output'] = {len(ins): port(out_prec, out_dims)}
    return node


def save_case(plugins, name, type_, ins, data=None, per_image=False, out_prec='FP32'):
    """Run the reference plugin; per_image=True runs one N=1 call per leading-axis slice and stacks."""
    plugin = plugins[type_]

    def run(arrs):
        node = make_node(type_, arrs, (1,), data, out_prec=out_prec)
        res = plugin.compute(node, {i: a for i, a in enumerate(arrs)}, kernel_type='special', debug=False)
        return np.ascontiguousarray(next(iter(res.values())))

    if per_image:
        outs = [run([ins[0][i:i + 1]] + list(ins[1:])) for i in range(ins[0].shape[0])]
        out = np.concatenate(outs, axis=0)
    else:
        out = run(list(ins))
    node = make_node(type_, ins, out.shape, data, out_prec=out_prec, name=name)
    blob = {'in{}'.format(i): a for i, a in enumerate(ins)}
    blob['out'] = out
    blob['node'] = np.array(json.dumps(node))
    os.makedirs(os.path.join(HERE, 'ops'), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, 'ops', name + '.npz'), **blob)
    print('  {:34s} {} -> {}'.format(name, [a.shape for a in ins], out.shape))


def conv_data(strides, pb, pe, auto_pad='explicit'):
    return {'strides': '{}, {}'.format(*strides), 'dilations': '1, 1', 'pads_begin': '{}, {}'.format(*pb),
            'pads_end': '{}, {}'.format(*pe), 'auto_pad': auto_pad}


def pool_data(kernel, strides, pb, pe, rounding, auto_pad='explicit'):
    return {'kernel': '{}, {}'.format(*kernel), 'strides': '{}, {}'.format(*strides), 'pads_begin': '{}, {}'.format(*pb),
            'pads_end': '{}, {}'.format(*pe), 'rounding_type': rounding, 'auto_pad': auto_pad, 'exclude-pad': 'false'}


def op_cases(plugins):
    print('per-op cases')
    # ---- Convolution ('special' = im2col, batched)
    save_case(plugins, 'conv_3x3_valid', 'Convolution', [rnd(1, (2, 8, 13, 13)), rnd(2, (16, 8, 3, 3), 0.2)],
              conv_data((1, 1), (0, 0), (0, 0), 'valid'))
    save_case(plugins, 'conv_1x1', 'Convolution', [rnd(3, (2, 48, 7, 7)), rnd(4, (24, 48, 1, 1), 0.2)],
              conv_data((1, 1), (0, 0), (0, 0)))
    save_case(plugins, 'conv_3x3_p1', 'Convolution', [rnd(5, (1, 16, 14, 14)), rnd(6, (40, 16, 3, 3), 0.1)],
              conv_data((1, 1), (1, 1), (1, 1)))
    save_case(plugins, 'conv_5x5_p2', 'Convolution', [rnd(7, (1, 8, 14, 14)), rnd(8, (12, 8, 5, 5), 0.1)],
              conv_data((1, 1), (2, 2), (2, 2)))
    save_case(plugins, 'conv_7x7_s2_p3', 'Convolution', [rnd(9, (1, 3, 32, 32), 50.0), rnd(10, (16, 3, 7, 7), 0.05)],
              conv_data((2, 2), (3, 3), (3, 3)))
    save_case(plugins, 'conv_3x3_s2_same_upper', 'Convolution', [rnd(11, (1, 3, 16, 16)), rnd(12, (8, 3, 3, 3), 0.3)],
              conv_data((2, 2), (0, 0), (1, 1), 'same_upper'))
    save_case(plugins, 'conv_k130_b3', 'Convolution', [rnd(13, (3, 5, 9, 11)), rnd(14, (130, 5, 3, 3), 0.2)],
              conv_data((1, 1), (1, 1), (1, 1)))
    save_case(plugins, 'conv_mnist_first', 'Convolution', [rnd(15, (2, 1, 28, 28), 80.0, 100.0), rnd(16, (32, 1, 3, 3), 0.01)],
              conv_data((1, 1), (0, 0), (0, 0), 'valid'))
    # ---- MatMul, the four transpose combinations
    for ta in ('false', 'true'):
        for tb in ('false', 'true'):
            a = rnd(20, (37, 5) if ta == 'true' else (5, 37))
            b = rnd(21, (11, 37) if tb == 'true' else (37, 11))
            save_case(plugins, 'matmul_ta{}_tb{}'.format(ta[0], tb[0]), 'MatMul', [a, b], {'transpose_a': ta, 'transpose_b': tb})
    save_case(plugins, 'matmul_fc_70x130', 'MatMul', [rnd(22, (70, 129)), rnd(23, (130, 129), 0.1)],
              {'transpose_a': 'false', 'transpose_b': 'true'})
    # ---- MaxPool (zero padding takes part: inputs are mostly negative in one case)
    save_case(plugins, 'maxpool_2x2_s2_floor', 'MaxPool', [rnd(30, (2, 4, 26, 26))], pool_data((2, 2), (2, 2), (0, 0), (0, 0), 'floor', 'valid'))
    save_case(plugins, 'maxpool_3x3_s2_ceil', 'MaxPool', [rnd(31, (2, 5, 15, 15), 1.0, -2.0)], pool_data((3, 3), (2, 2), (0, 0), (0, 0), 'ceil'))
    save_case(plugins, 'maxpool_3x3_s1_p1_ceil', 'MaxPool', [rnd(32, (2, 6, 7, 7), 1.0, -1.5)], pool_data((3, 3), (1, 1), (1, 1), (1, 1), 'ceil'))
    save_case(plugins, 'maxpool_3x3_s2_ceil_even', 'MaxPool', [rnd(33, (1, 3, 14, 14), 1.0, -3.0)], pool_data((3, 3), (2, 2), (0, 0), (0, 0), 'ceil'))
    save_case(plugins, 'maxpool_same_upper_quirk', 'MaxPool', [rnd(34, (1, 2, 6, 6))], pool_data((3, 3), (1, 1), (1, 1), (1, 1), 'floor', 'same_upper'))
    # ---- AvgPool (h-1 / w-1 clipping)
    save_case(plugins, 'avgpool_7x7_global', 'AvgPool', [rnd(40, (2, 16, 7, 7))], pool_data((7, 7), (1, 1), (0, 0), (0, 0), 'ceil'))
    save_case(plugins, 'avgpool_3x3_s2', 'AvgPool', [rnd(41, (1, 3, 9, 9))], pool_data((3, 3), (2, 2), (0, 0), (0, 0), 'floor'))
    # ---- Add / Multiply broadcasting
    bc = {'auto_broadcast': 'numpy'}
    save_case(plugins, 'add_bias_nchw', 'Add', [rnd(50, (2, 6, 5, 7)), rnd(51, (1, 6, 1, 1))], bc)
    save_case(plugins, 'add_bias_fc', 'Add', [rnd(52, (3, 10)), rnd(53, (1, 10))], bc)
    save_case(plugins, 'add_same_shape', 'Add', [rnd(54, (2, 3, 4, 5)), rnd(55, (2, 3, 4, 5))], bc)
    save_case(plugins, 'add_scalar', 'Add', [rnd(56, (2, 3, 4, 5)), rnd(57, (1, 1, 1, 1))], bc)
    save_case(plugins, 'add_row_col', 'Add', [rnd(58, (2, 3, 4, 5)), rnd(59, (1, 3, 4, 1))], bc)
    save_case(plugins, 'mul_scale_nchw', 'Multiply', [rnd(60, (2, 8, 7, 7)), rnd(61, (1, 8, 1, 1))], bc)
    save_case(plugins, 'mul_scalar_first', 'Multiply', [rnd(62, (1, 1, 1, 1)), rnd(63, (2, 3, 9, 9))], bc)
    # ---- unary
    x = rnd(70, (2, 3, 5, 7))
    x.ravel()[:6] = [-0.0, 0.0, np.nan, np.inf, -np.inf, -1e-30]
    save_case(plugins, 'relu_special_values', 'ReLU', [x])
    save_case(plugins, 'relu_odd_size', 'ReLU', [rnd(71, (1, 3, 11, 13))])
    save_case(plugins, 'clamp_relu6', 'Clamp', [rnd(72, (2, 4, 6, 6), 4.0)], {'min': '0', 'max': '6'})
    save_case(plugins, 'sigmoid', 'Sigmoid', [rnd(73, (1, 1, 37, 11), 4.0)])
    # ---- SoftMax: the reference normalises the whole tensor -> one N=1 call per row
    save_case(plugins, 'softmax_10', 'SoftMax', [rnd(80, (4, 10), 5.0)], {'axis': '1'}, per_image=True)
    save_case(plugins, 'softmax_1000', 'SoftMax', [rnd(81, (3, 1000), 6.0)], {'axis': '1'}, per_image=True)
    # ---- LRN
    lrn = {'alpha': '9.9999997473787516e-05', 'beta': '0.75', 'bias': '1', 'size': '5'}
    save_case(plugins, 'lrn_c8', 'LRN', [rnd(90, (2, 8, 6, 6), 30.0), np.array([1], dtype=np.int64)], lrn)
    save_case(plugins, 'lrn_c3_odd_hw', 'LRN', [rnd(91, (1, 3, 5, 7), 30.0), np.array([1], dtype=np.int64)], lrn)
    save_case(plugins, 'lrn_size3_beta05', 'LRN', [rnd(92, (1, 6, 4, 4), 10.0), np.array([1], dtype=np.int64)],
              {'alpha': '0.001', 'beta': '0.5', 'bias': '2', 'size': '3'})
    # ---- Concat
    save_case(plugins, 'concat_ch4', 'Concat', [rnd(100, (2, 3, 4, 4)), rnd(101, (2, 5, 4, 4)), rnd(102, (2, 1, 4, 4)), rnd(103, (2, 7, 4, 4))], {'axis': '1'})
    save_case(plugins, 'concat_axis2_odd', 'Concat', [rnd(104, (1, 2, 3)), rnd(105, (1, 2, 5))], {'axis': '2'})
    # ---- GroupConvolution (depthwise; the reference computes image 0 only -> per image)
    save_case(plugins, 'dwconv_3x3_s1_p1', 'GroupConvolution', [rnd(110, (2, 6, 9, 9)), rnd(111, (6, 1, 1, 3, 3), 0.3)],
              conv_data((1, 1), (1, 1), (1, 1), 'same_upper'), per_image=True)
    save_case(plugins, 'dwconv_3x3_s2_pe1', 'GroupConvolution', [rnd(112, (1, 4, 10, 10)), rnd(113, (4, 1, 1, 3, 3), 0.3)],
              conv_data((2, 2), (0, 0), (1, 1), 'same_upper'), per_image=True)
    # ---- Transpose / Reshape
    save_case(plugins, 'transpose_nchw_nhwc', 'Transpose', [rnd(120, (2, 5, 3, 4)), np.array([0, 2, 3, 1], dtype=np.int64)])
    save_case(plugins, 'reshape_flatten', 'Reshape', [rnd(121, (2, 3, 3, 4)), np.array([-1, 36], dtype=np.int64)], {'special_zero': 'false'})
    save_case(plugins, 'reshape_zero_copy', 'Reshape', [rnd(122, (2, 6, 1, 1)), np.array([0, -1], dtype=np.int64)], {'special_zero': 'true'})


def i64(*values):
    return np.array(values, dtype=np.int64)


DETECTION = {'background_label_id': '0', 'clip_after_nms': 'true', 'clip_before_nms': 'false',
             'code_type': 'caffe.PriorBoxParameter.CENTER_SIZE', 'confidence_threshold': '0.30000001192092896',
             'decrease_label_id': 'false', 'input_height': '1', 'input_width': '1', 'keep_top_k': '100',
             'nms_threshold': '0.60000002384185791', 'normalized': 'true', 'num_classes': '21', 'objectness_score': '0',
             'share_location': 'true', 'top_k': '100', 'variance_encoded_in_target': 'false'}


def head_cases(plugins):
    """SSD head glue: ShapeOf / StridedSlice / Unsqueeze / PriorBoxClustered / DetectionOutput."""
    save_case(plugins, 'shapeof_nchw', 'ShapeOf', [rnd(130, (1, 12, 19, 19))], {'output_type': 'i64'}, out_prec='I64')
    ss = {'begin_mask': '0', 'ellipsis_mask': '0', 'end_mask': '1', 'new_axis_mask': '0', 'shrink_axis_mask': '0'}
    save_case(plugins, 'stridedslice_hw_of_shape', 'StridedSlice', [i64(1, 12, 19, 19), i64(2), i64(4), i64(1)], ss, out_prec='I64')
    save_case(plugins, 'stridedslice_step2', 'StridedSlice', [i64(3, 1, 4, 1, 5, 9, 2, 6), i64(1), i64(8), i64(2)], ss, out_prec='I64')
    save_case(plugins, 'unsqueeze_front', 'Unsqueeze', [rnd(131, (2, 24)), i64(0)])
    save_case(plugins, 'unsqueeze_two_axes', 'Unsqueeze', [rnd(132, (3, 3)), i64(0, 3)])
    pb = {'clip': 'false', 'height': '30, 42.4264, 84.8528', 'offset': '0.5', 'step': '0', 'step_h': '0', 'step_w': '0',
          'variance': '0.1, 0.1, 0.2, 0.2', 'width': '30, 84.8528, 42.4264'}
    save_case(plugins, 'priorbox_19x19_3', 'PriorBoxClustered', [i64(19, 19), i64(300, 300)], pb)
    pb2 = dict(pb, height='105, 74.2462, 148.492, 60.6218, 181.874, 125.499', width='105, 148.492, 74.2462, 181.865, 60.6187, 125.499',
               step='16', offset='0.25')
    save_case(plugins, 'priorbox_3x5_6_step16', 'PriorBoxClustered', [i64(3, 5), i64(120, 200)], pb2)
    # DetectionOutput: priors from the reference's own PriorBoxClustered (10x10 grid, 3 boxes per cell = 300 priors),
    # box deltas ~ N(0, 0.5^2), class scores = a seeded permutation of an even grid on (0, 1) (no ties), 21 classes
    node = make_node('PriorBoxClustered', [i64(10, 10), i64(300, 300)], (1,), pb)
    priors = np.ascontiguousarray(next(iter(plugins['PriorBoxClustered'].compute(node, {0: i64(10, 10), 1: i64(300, 300)}, kernel_type='special').values())))[None]
    P = priors.shape[2] // 4
    loc = rnd(140, (1, P * 4), 0.5)
    order = np.argsort(synth.uniform01(141, 7, P * 21), kind='stable')          # a seeded permutation: all scores distinct
    conf = ((order + 0.5) / float(P * 21)).astype(np.float32).reshape(1, P * 21)
    assert len(np.unique(conf)) == conf.size
    save_case(plugins, 'detout_center_size_300x21', 'DetectionOutput', [loc, conf, priors], DETECTION)
    save_case(plugins, 'detout_few_boxes_terminator', 'DetectionOutput', [loc, conf, priors],
              dict(DETECTION, confidence_threshold='0.995', keep_top_k='20'))
    save_case(plugins, 'detout_corner_encoded_clip_before', 'DetectionOutput', [rnd(142, (1, P * 4), 0.05), conf, priors],
              dict(DETECTION, code_type='caffe.PriorBoxParameter.CORNER', variance_encoded_in_target='true', clip_before_nms='true',
                   clip_after_nms='false', keep_top_k='30', nms_threshold='0.45', confidence_threshold='0.9'))
    save_case(plugins, 'detout_corner_variance', 'DetectionOutput', [rnd(143, (1, P * 4), 0.3), conf, priors],
              dict(DETECTION, code_type='caffe.PriorBoxParameter.CORNER', keep_top_k='-1', top_k='3', confidence_threshold='0.98'))


def ssd_full_case(IECore):
    """The WHOLE SSD-MobileNet IR (backbone, prior boxes, DetectionOutput) through the reference's own infer() on
    synthetic weights, one image."""
    print('ssd_mobilenet_v1_coco end to end on synthetic weights (seed 1234), 1 image -- slow (python loops, O(n^2) NMS)')
    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    xml = os.path.join(REF, 'models', 'ssd_mobilenet_v1_coco.xml')
    stem = os.path.join(tmp, 'ssd_mobilenet_v1_coco')
    with open(stem + '.bin', 'wb') as f:
        f.write(synth.synth_weights(xml, 1234))
    if not os.path.exists(stem + '.xml'):
        os.symlink(xml, stem + '.xml')
    x = synth.uniform_pixels(700, (1, 3, 300, 300))
    ie = IECore()
    net = ie.read_network(stem + '.xml', stem + '.bin')
    ex = ie.load_network(net, 'CPU')
    ex.kernel_type = 'special'
    res = ex.infer({net.inputs[0]['name']: x})
    out = np.ascontiguousarray(res[net.outputs[0]['name']])
    by_name = {net.G.nodes[n]['name']: n for n in net.G.nodes}
    priors = np.ascontiguousarray(next(iter(net.G.nodes[by_name['ConcatPriorBoxesClustered']]['output'].values()))['data'])
    np.savez_compressed(os.path.join(HERE, 'ssd_full_e2e.npz'), image_seed=np.array(700), weight_seed=np.array(1234), out=out, priors=priors)
    print('  detections', out.shape, 'records', int((out[0, 0, :, 0] >= 0).sum()), 'first', out[0, 0, 0])


def fp16_case(IECore):
    """models/mnist as an FP16 IR (pyopenvino_amd.synth.fp16_ir: every port FP16, constants stored as f16)
    through the reference, which computes it in numpy float16; three of the mnist_e2e images."""
    helpers = synth
    print('mnist as an FP16 IR through the reference (float16 numpy)')
    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    blob = open(os.path.join(REF, 'models', 'mnist.bin'), 'rb').read()
    xml16, blob16 = helpers.fp16_ir(os.path.join(REF, 'models', 'mnist.xml'), blob, tmp)
    stem = xml16[:-4]
    with open(stem + '.bin', 'wb') as f:
        f.write(blob16)
    images = np.load(os.path.join(HERE, 'mnist_e2e.npz'))['images'][:3]
    outs, logits = [], []
    for i in range(len(images)):
        ie = IECore()
        net = ie.read_network(stem + '.xml', stem + '.bin')
        ex = ie.load_network(net, 'CPU')
        ex.kernel_type = 'special'
        o = np.ascontiguousarray(ex.infer({net.inputs[0]['name']: images[i:i + 1]})[net.outputs[0]['name']])
        assert o.dtype == np.float16, o.dtype
        outs.append(o.astype(np.float32))
        soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
        pre = next(iter(net.G.pred[soft]))                       # the Add in front of the SoftMax: the float16 logits
        logits.append(np.asarray(next(iter(net.G.nodes[pre]['output'].values()))['data']).astype(np.float32))
    out = np.concatenate(outs, 0)
    np.savez_compressed(os.path.join(HERE, 'mnist_fp16_e2e.npz'), out=out, n_images=np.array(len(images)), logits=np.concatenate(logits, 0))
    print('  out', out.shape, 'argmax', out.argmax(axis=1), 'logits', logits[0])


def googlenet_fp16_case(IECore, nimg=2):
    """googlenet_fp16_rows2.npz: the benchmarked FP16 entry's parity anchor.  models/googlenet-v1 as an FP16 IR (synth.fp16_ir of the
    synthetic seed-1234 blob: every constant stored as f16, every port FP16) through the reference, which then computes every node in
    numpy float16 (`pyopenvino/common_def.py:13-17`); the float16 logits -- the tensor in front of the SoftMax, whose exp() overflows
    float16 -- and the SoftMax output as it comes, for images 500 and 501 of googlenet_rows8.npz."""
    print('googlenet-v1 as an FP16 IR through the reference (float16 numpy; slow: no BLAS for float16)')
    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    xml = os.path.join(REF, 'models', 'googlenet-v1.xml')
    xml16, blob16 = synth.fp16_ir(xml, synth.synth_weights(xml, 1234), tmp)
    stem = xml16[:-4]
    with open(stem + '.bin', 'wb') as f:
        f.write(blob16)
    outs, logits = [], []
    for i in range(nimg):
        ie = IECore()
        net = ie.read_network(stem + '.xml', stem + '.bin')
        ex = ie.load_network(net, 'CPU')
        ex.kernel_type = 'special'
        o = np.ascontiguousarray(ex.infer({net.inputs[0]['name']: synth.uniform_pixels(500 + i, (1, 3, 224, 224))})[net.outputs[0]['name']])
        assert o.dtype == np.float16, o.dtype
        outs.append(o.astype(np.float32))
        soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
        pre = next(iter(net.G.pred[soft]))
        lg = np.asarray(next(iter(net.G.nodes[pre]['output'].values()))['data'])
        assert lg.dtype == np.float16 and np.isfinite(lg).all()
        logits.append(lg.astype(np.float32))
        print('  image', 500 + i, 'logits max', float(np.abs(lg).max()), 'argmax', int(lg.argmax()))
    np.savez_compressed(os.path.join(HERE, 'googlenet_fp16_rows2.npz'), logits=np.concatenate(logits, 0), out=np.concatenate(outs, 0),
                        image_seeds=np.array([500 + i for i in range(nimg)]), weight_seed=np.array(1234))


def run_model(IECore, model, x, input_name=None, capture_layers=False):
    """Reference, kernel_type='special', N=1.  Returns the Result array and {node id: float64 sum}."""
    ie = IECore()
    net = ie.read_network(model + '.xml', model + '.bin')
    ex = ie.load_network(net, 'CPU')
    ex.kernel_type = 'special'
    res = ex.infer({net.inputs[0]['name']: x})
    out = np.ascontiguousarray(res[net.outputs[0]['name']])
    sums = {}
    if capture_layers:
        for nid in net.G.nodes:
            node = net.G.nodes[nid]
            if node['type'] in ('Const', 'Result') or 'output' not in node:
                continue
            for p in node['output'].values():
                if 'data' in p and np.asarray(p['data']).dtype == np.float32:
                    sums[int(nid)] = float(np.asarray(p['data'], dtype=np.float64).sum())
    return out, sums


def model_cases(IECore):
    from PIL import Image
    print('mnist end to end (real weights)')
    imgs = [np.array(Image.open(os.path.join(REF, 'resources', f)).convert('L')).astype(np.float32).reshape(1, 1, 28, 28)
            for f in ('mnist2.png', 'mnist7.png')]
    imgs += [synth.uniform_pixels(100 + i, (1, 1, 28, 28)) for i in range(6)]
    outs, sums0 = [], None
    for i, im in enumerate(imgs):
        o, s = run_model(IECore, 'models/mnist', im, capture_layers=(i == 0))
        outs.append(o)
        if i == 0:
            sums0 = s
    np.savez_compressed(os.path.join(HERE, 'mnist_e2e.npz'), images=np.concatenate(imgs, 0), out=np.concatenate(outs, 0),
                        layer_ids=np.array(sorted(sums0), dtype=np.int64), layer_sums=np.array([sums0[k] for k in sorted(sums0)]))
    print('  top-3 mnist2', np.argsort(outs[0][0])[::-1][:3], ' mnist7', np.argsort(outs[1][0])[::-1][:3])

    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    for model, shape, nimg, fname in (('googlenet-v1', (1, 3, 224, 224), 2, 'googlenet_e2e.npz'),
                                      ('mnist_bn', (1, 1, 28, 28), 2, 'mnist_bn_e2e.npz')):
        print(model, 'on synthetic weights (seed 1234)')
        xml = os.path.join(REF, 'models', model + '.xml')
        blob = synth.synth_weights(xml, 1234)
        stem = os.path.join(tmp, model)
        with open(stem + '.bin', 'wb') as f:
            f.write(blob)
        if not os.path.exists(stem + '.xml'):
            os.symlink(xml, stem + '.xml')
        outs, sums0 = [], None
        for i in range(nimg):
            x = synth.uniform_pixels(500 + i, shape)
            o, s = run_model(IECore, stem, x, capture_layers=(i == 0))
            outs.append(o)
            if i == 0:
                sums0 = s
        out = np.concatenate(outs, 0)
        np.savez_compressed(os.path.join(HERE, fname), out=out, image_seeds=np.array([500 + i for i in range(nimg)]),
                            weight_seed=np.array(1234), layer_ids=np.array(sorted(sums0), dtype=np.int64),
                            layer_sums=np.array([sums0[k] for k in sorted(sums0)]))
        print('  out', out.shape, 'row sums', out.sum(axis=1), 'argmax', out.argmax(axis=1))


def googlenet_rows_case(IECore, nimg=8):
    """googlenet_rows8.npz: the reference's N=1 answers for 8 seeded images on the synthetic weights (the first two are the rows
    of googlenet_e2e.npz): what rows 0-7 of the batch-256 run on the GPU are held against."""
    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    xml = os.path.join(REF, 'models', 'googlenet-v1.xml')
    stem = os.path.join(tmp, 'googlenet-v1')
    with open(stem + '.bin', 'wb') as f:
        f.write(synth.synth_weights(xml, 1234))
    if not os.path.exists(stem + '.xml'):
        os.symlink(xml, stem + '.xml')
    outs = [run_model(IECore, stem, synth.uniform_pixels(500 + i, (1, 3, 224, 224)))[0] for i in range(nimg)]
    out = np.concatenate(outs, 0)
    np.savez_compressed(os.path.join(HERE, 'googlenet_rows8.npz'), out=out, image_seeds=np.array([500 + i for i in range(nimg)]),
                        weight_seed=np.array(1234))
    print('googlenet rows', out.shape, 'argmax', out.argmax(axis=1))


def ssd_backbone_case(IECore):
    """SSD-MobileNet backbone + box/class heads on synthetic weights, ONE image, through the reference's own
    scheduler loop restricted to the ancestors of 'concat' / 'concat_1' / the Sigmoid (its PriorBox /
    DetectionOutput tail is host-side glue outside the path).  GroupConvolution 'special' == its numpy loops."""
    import networkx as nx
    print('ssd_mobilenet_v1_coco backbone on synthetic weights (seed 1234), 1 image -- slow (python loops)')
    tmp = '/tmp/pv_golden_models'
    os.makedirs(tmp, exist_ok=True)
    xml = os.path.join(REF, 'models', 'ssd_mobilenet_v1_coco.xml')
    stem = os.path.join(tmp, 'ssd_mobilenet_v1_coco')
    with open(stem + '.bin', 'wb') as f:
        f.write(synth.synth_weights(xml, 1234))
    if not os.path.exists(stem + '.xml'):
        os.symlink(xml, stem + '.xml')
    ie = IECore()
    net = ie.read_network(stem + '.xml', stem + '.bin')
    ex = ie.load_network(net, 'CPU')
    G = net.G
    names = ['concat', 'concat_1', 'do_ExpandDims_conf/sigmoid']
    by_name = {G.nodes[n]['name']: n for n in G.nodes}
    needed = set()
    for nm in names:
        needed.add(by_name[nm])
        needed.update(nx.ancestors(G, by_name[nm]))
    x = synth.uniform_pixels(700, (1, 3, 300, 300))
    G.nodes[by_name[net.inputs[0]['name']]]['param'] = x
    plugins = ie.plugins.plugins
    for task in ex.task_list:
        if task not in needed:
            continue
        node = G.nodes[task]
        inputs = ex.prepare_inputs_for_task(task) if 'input' in node else {}
        res = plugins[node['type']].compute(node, inputs, kernel_type='special', debug=False)
        if len(res) > 0:
            for port_id, data in res.items():
                node['output'][port_id]['data'] = data
    outs = {nm: np.ascontiguousarray(next(iter(G.nodes[by_name[nm]]['output'].values()))['data']) for nm in names}
    np.savez_compressed(os.path.join(HERE, 'ssd_backbone_e2e.npz'), image_seed=np.array(700), weight_seed=np.array(1234),
                        concat=outs['concat'], concat_1_sub=outs['concat_1'][:, ::3, ::5],
                        concat_1_sum=np.array(outs['concat_1'].astype(np.float64).sum()),
                        sigmoid_sub=outs['do_ExpandDims_conf/sigmoid'][:, :, ::3, ::5],
                        sigmoid_sum=np.array(outs['do_ExpandDims_conf/sigmoid'].astype(np.float64).sum()))
    print('  concat', outs['concat'].shape, 'concat_1', outs['concat_1'].shape, 'sum', float(outs['concat_1'].sum()))


def node6_case(plugins):
    print('reference single-node fixture resources/node_args_6.pickle (cropped, fp32)')
    with open(os.path.join(REF, 'resources', 'node_args_6.pickle'), 'rb') as f:
        node, inputs = pickle.load(f)
    x = np.ascontiguousarray(inputs[0][:, :, :64, :64]).astype(np.float32)
    w = np.ascontiguousarray(inputs[1]).astype(np.float32)
    save_case(plugins, 'conv_node6_crop', 'Convolution', [x, w], dict(node['data']))
    os.replace(os.path.join(HERE, 'ops', 'conv_node6_crop.npz'), os.path.join(HERE, 'conv_node6_crop.npz'))


def node6_fp16_case(plugins):
    """conv_node6_fp16.npz: the reference's own FP16 node fixture (resources/node_args_6.pickle: SSD Conv2d_0, 3x3 stride 2
    same_upper, replayed as test_node_sample.py:1-16 does) in float16 as it stands, input cropped to 64x64: x, w and the
    'special' kernel's float16 output."""
    print('reference single-node fixture resources/node_args_6.pickle (cropped, float16 as the reference computes it)')
    with open(os.path.join(REF, 'resources', 'node_args_6.pickle'), 'rb') as f:
        node, inputs = pickle.load(f)
    x = np.ascontiguousarray(inputs[0][:, :, :64, :64])
    w = np.ascontiguousarray(inputs[1])
    assert x.dtype == np.float16 and w.dtype == np.float16
    node = dict(node)
    node['input'] = {0: port('FP16', x.shape), 1: port('FP16', w.shape)}
    oh = (64 + 0 + 1 - 3) // 2 + 1
    node['output'] = {2: port('FP16', (1, w.shape[0], oh, oh))}
    res = plugins['Convolution'].compute(node, {0: x, 1: w}, kernel_type='special', debug=False)
    out = np.ascontiguousarray(next(iter(res.values())))
    assert out.dtype == np.float16, out.dtype
    np.savez_compressed(os.path.join(HERE, 'conv_node6_fp16.npz'), x=x, w=w, out=out, data=json.dumps(dict(node['data'])))
    print('  out', out.shape, out.dtype, 'min', out.min(), 'max', out.max())


def main():
    IECore = import_reference()
    plugins = IECore().plugins.plugins
    if 'ssd' in sys.argv[1:]:            # only (re)generate the slow SSD fixtures
        ssd_backbone_case(IECore)
        ssd_full_case(IECore)
        return
    if 'rows' in sys.argv[1:]:           # only the 8-row GoogLeNet fixture
        googlenet_rows_case(IECore)
        return
    if 'fp16' in sys.argv[1:]:           # only the FP16 fixtures
        fp16_case(IECore)
        node6_fp16_case(plugins)
        return
    if 'googlenet_fp16' in sys.argv[1:]:
        googlenet_fp16_case(IECore)
        return
    if 'head' in sys.argv[1:]:           # only the SSD head per-op fixtures and the end-to-end SSD fixture
        head_cases(plugins)
        ssd_full_case(IECore)
        return
    op_cases(plugins)
    head_cases(plugins)
    node6_case(plugins)
    node6_fp16_case(plugins)
    model_cases(IECore)
    googlenet_rows_case(IECore)
    fp16_case(IECore)
    googlenet_fp16_case(IECore)
    ssd_backbone_case(IECore)
    ssd_full_case(IECore)
    print('done')


if __name__ == '__main__':
    main()
