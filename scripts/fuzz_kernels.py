#!/usr/bin/env python3
"""GPU-box tool: random shapes through the kernels that have a slower twin -- the six-point Winograd kernel (3x3, 5x5; forced) against the
direct kernel, MaxPool + 1x1 convolution / MaxPool + LRN / LRN + MaxPool as one launch against two, the padding pass + test-free
gather of the c-major kernel (with and without the Add folded in) against the window test in the gather, and the f16 kernels of an
FP16 IR (span kernel, LDS-DMA form, the first gather kernel) against each other; the pipelined depthwise 3x3 kernel (lanes storing / through an output stage) against the one-shot one, and the
pipelined 3x3 MaxPool against the one-shot one with NaNs and infinities strewn in -- bit for bit.  python scripts/fuzz_kernels.py [cases] [seed]"""
import os, sys, random
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from pyopenvino_amd import device as dev, synth
from pyopenvino_amd.op_plugins import Convolution, MaxPool, LRN, GroupConvolution
dev.init(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def node(type_, ins, data):
    return {'name': type_, 'type': type_, 'version': 'opset1', 'data': dict(data),
            'input': {i: {'precision': 'I64' if a.dtype == np.int64 else 'FP32', 'dims': tuple(a.shape)} for i, a in enumerate(ins)},
            'output': {len(ins): {'precision': 'FP32', 'dims': ()}}}
def setenv(env):
    for k_ in ('PVHIP_CONV_WINOGRAD4', 'PVHIP_CONV_WINOGRAD5', 'PVHIP_CONV_WINOGRAD', 'PVHIP_CONV_PREPAD', 'PVHIP_CONV_F16_SPAN', 'PVHIP_CONV_F16_DMA', 'PVHIP_DWCONV_COLS', 'PVHIP_POOL3', 'PVHIP_CONV_STEM', 'PVHIP_CONV_STEM_DIRECT'): os.environ.pop(k_, None)
    os.environ.update(env); dev.reload_settings()
bad = 0
compared = {}
for i in range(cases):
    kind = rng.choice(['w3', 'w5', 'poolconv', 'poollrn', 'lrnpool', 'prepad', 'f16', 'dw', 'pool', 'stem', 'stemconv'])
    n, h, w = rng.randint(1, 9), rng.randint(1, 30), rng.randint(1, 30)
    if kind in ('w3', 'w5'):
        ks = 3 if kind == 'w3' else 5
        c, k = 4 * rng.randint(1, 12), rng.randint(1, 100)
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w))
        wt = (synth.normal(i, 3, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks))
        b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
        outs = []
        for env in ({'PVHIP_CONV_WINOGRAD4': 'force', 'PVHIP_CONV_WINOGRAD5': 'force'}, {'PVHIP_CONV_WINOGRAD': '0', 'PVHIP_CONV_WINOGRAD5': '0'}):
            setenv(env)
            outs.append(np.asarray(Convolution.launch({}, dev.DeviceTensor.from_numpy(x), dev.DeviceTensor.from_numpy(wt), (1, 1), (ks // 2,) * 2, (ks // 2,) * 2, 'explicit', bias=b, act=('relu',))))
        setenv({})
        err = float(np.abs(outs[0] - outs[1]).max() / max(1e-20, np.abs(outs[1]).max()))
        ok = err < 5e-5
        what = '{}x{} conv x{} k{}: {:.1e}'.format(ks, ks, (n, c, h, w), k, err)
    elif kind == 'stem':              # 7x7 / 2 / pad 3 over three channels (round 5): the row-span kernel on a padded copy and straight from the image == the general kernel, bit for bit
        k = rng.randint(1, 64)
        h, w = rng.randint(7, 60), 4 * rng.randint(2, 56)          # (output rows of a multiple of four pixels; odd ones stay on the general kernel anyway)
        if rng.random() < 0.3:
            w = 2 * rng.randint(4, 110)
        x = synth.normal(i, 2, n * 3 * h * w).astype(np.float32).reshape((n, 3, h, w)) * 60.0
        wt = (synth.normal(i, 3, k * 147) * (2.0 / 147) ** 0.5).astype(np.float32).reshape((k, 3, 7, 7))
        addc = synth.normal(i, 5, 3).astype(np.float32).reshape((1, 3, 1, 1)) * 50.0
        b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
        act = rng.choice([None, ('relu',), ('clamp', -0.5, 0.7)])
        pre = rng.random() < 0.5
        outs = []
        for env in ({'PVHIP_CONV_STEM': '0'}, {'PVHIP_CONV_STEM': '1', 'PVHIP_CONV_STEM_DIRECT': '0'}, {'PVHIP_CONV_STEM': '1', 'PVHIP_CONV_STEM_DIRECT': '1'}):
            setenv(env)
            nd = {'_pre_add': dev.DeviceTensor.from_numpy(addc)} if pre else {}
            outs.append(np.asarray(Convolution.launch(nd, dev.DeviceTensor.from_numpy(x), dev.DeviceTensor.from_numpy(wt), (2, 2), (3, 3), (3, 3), 'explicit', bias=b, act=act)))
        setenv({})
        ok = all(bool((o.view(np.uint32) == outs[0].view(np.uint32)).all()) for o in outs[1:])
        what = 'stem conv x{} k{} act {} pre_add {}'.format(x.shape, k, act, pre)
    elif kind == 'stemconv':          # MaxPool -> LRN -> 1x1 convolution as one launch (round 5) == MaxPool + LRN then the pointwise launch, bit for bit
        c, k = 8 * rng.randint(1, 8), rng.randint(1, 64)
        st = rng.choice([1, 2])
        h, w = rng.randint(3, 40), 4 * rng.randint(1, 28)
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 30.0
        pads = rng.choice([((0, 0), (0, 0)), ((1, 1), (1, 1))])
        pn = node('MaxPool', [x], {'kernel': '3, 3', 'strides': '{0}, {0}'.format(st), 'pads_begin': '{}, {}'.format(*pads[0]), 'pads_end': '{}, {}'.format(*pads[1]),
                                   'rounding_type': rng.choice(['ceil', 'floor']), 'auto_pad': 'explicit'})
        try:
            oh, ow = MaxPool.calc_output_shape((h, w), (3, 3), (st, st), pads[0], pads[1], pn['data']['rounding_type'], 'explicit')
            if oh <= 0 or ow <= 0:
                continue
            axes = np.array([1], dtype=np.int64)
            ln = node('LRN', [np.zeros((n, c, oh, ow), np.float32), axes], {'alpha': '9.9999997473787516e-05', 'beta': '0.75', 'bias': '1', 'size': '5'})
            ln['output'][2]['dims'] = (n, c, oh, ow)
            wt = (synth.normal(i, 3, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1))
            cn = node('Convolution', [np.zeros((n, c, oh, ow), np.float32), wt], {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'})
            if not MaxPool.lrn_conv_fusable(pn, ln, cn):
                continue
            b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
            act = rng.choice([None, ('relu',), ('clamp', -0.5, 0.7)])
            two_p = dict(pn); two_p['_fuse_lrn'] = ln
            two_c = dict(cn); two_c['_fuse_bias'], two_c['_fuse_act'] = b, act
            two = np.asarray(Convolution.compute(two_c, {0: MaxPool.compute(two_p, {0: x})[1], 1: wt})[2])
            one_p = dict(two_p); one_p['_fuse_conv'] = {'node': cn, 'w': wt, 'bias': b, 'act': act}
            one = np.asarray(MaxPool.compute(one_p, {0: x})[1])
        except (ValueError, dev.PvhipError):
            continue
        ok = bool((one.view(np.uint32) == two.view(np.uint32)).all()); what = 'MaxPool + LRN + 1x1 x{} k{} stride {} pads {} act {}'.format(x.shape, k, st, pads, act)
    elif kind == 'prepad':            # c-major layers (C % 16 != 0) with padding: padding pass + test-free gather == window test in the gather, bit for bit
        c, k, ks, st = rng.choice([1, 3, 5, 7, 24]), rng.randint(1, 80), rng.choice([3, 5, 7]), rng.choice([1, 2])
        pb, pe = (rng.randint(0, ks // 2), rng.randint(0, ks // 2)), (rng.randint(0, ks // 2), rng.randint(0, ks // 2))
        h, w = h + ks, w + ks
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 20.0
        wt = (synth.normal(i, 3, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks))
        addc = synth.normal(i, 5, c).astype(np.float32).reshape((1, c, 1, 1)) * 50.0
        b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
        outs = []
        for env, pre in (({'PVHIP_CONV_PREPAD': '0'}, False), ({'PVHIP_CONV_PREPAD': '1'}, False), ({'PVHIP_CONV_PREPAD': '1'}, True)):
            setenv(env)
            nd = {'_pre_add': dev.DeviceTensor.from_numpy(addc)} if pre else {}
            xin = dev.DeviceTensor.from_numpy((x + addc).astype(np.float32) if not pre else x)
            outs.append(np.asarray(Convolution.launch(nd, xin, dev.DeviceTensor.from_numpy(wt), (st, st), pb, pe, 'explicit', bias=b, act=('relu',))))
        setenv({})
        # bit for bit where the plugin pads by itself (a c-major layer on the LDS-DMA kernel); a layer it would run on a Winograd kernel
        # (C % 4 == 0, 3x3 / pad 1) is only handed a folded Add by this tool, never by plan_fusion: then 1e-5 of another kernel's sum
        oh_, ow_ = outs[0].shape[2:]
        if Convolution.prepad_wanted(n, c, h, w, k, ks, ks, oh_, ow_, (st, st), pb, pe):
            ok = all(bool((o.view(np.uint32) == outs[0].view(np.uint32)).all()) for o in outs[1:])
        else:
            ok = bool((outs[1].view(np.uint32) == outs[0].view(np.uint32)).all()) and \
                float(np.abs(outs[2] - outs[0]).max()) <= 1e-5 * max(1e-20, float(np.abs(outs[0]).max()))
        what = 'padding pass {}x{} conv x{} k{} stride {} pads {} {}'.format(ks, ks, (n, c, h, w), k, st, pb, pe)
    elif kind == 'f16':               # FP16 IRs: the three f16 kernels agree to 1e-5 (same operands, another summation order)
        ks = rng.choice([1, 3, 5])
        c, k = 16 * rng.randint(1, 10), rng.randint(1, 300)
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w))
        wt = (synth.normal(i, 3, k * c * ks * ks) * (2.0 / (c * ks * ks)) ** 0.5).astype(np.float32).reshape((k, c, ks, ks))
        b = dev.DeviceTensor.from_numpy(synth.normal(i, 4, k).astype(np.float32).reshape((1, k, 1, 1)))
        outs = []
        for env in ({'PVHIP_CONV_F16_SPAN': '2'}, {'PVHIP_CONV_F16_SPAN': '0'}, {'PVHIP_CONV_F16_SPAN': '0', 'PVHIP_CONV_F16_DMA': '0'}):
            setenv(env)
            outs.append(np.asarray(Convolution.launch({}, dev.DeviceTensor.from_numpy(x), dev.DeviceTensor.from_numpy(wt), (1, 1), (ks // 2,) * 2, (ks // 2,) * 2,
                                                      'explicit', bias=b, act=('relu',), f16=True)))
        setenv({})
        err = max(float(np.abs(o - outs[2]).max() / max(1e-20, np.abs(outs[2]).max())) for o in outs[:2])
        ok = err < 1e-5
        what = 'f16 {}x{} conv x{} k{}: {:.1e}'.format(ks, ks, (n, c, h, w), k, err)
    elif kind == 'dw':                # depthwise 3x3, stride 1 / 2: the pipelined kernel in both store forms == the one-shot kernel
        c, st = rng.randint(1, 40), rng.choice([1, 2])
        n, h, w = rng.randint(1, 6), rng.randint(1, 160), rng.randint(1, 160)
        pb, pe = (rng.randint(0, 1), rng.randint(0, 1)), (rng.randint(0, 1), rng.randint(0, 1))
        if h + pb[0] + pe[0] < 3 or w + pb[1] + pe[1] < 3:
            continue
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 3.0
        x.reshape(-1)[rng.randrange(x.size)] = np.inf
        x.reshape(-1)[rng.randrange(x.size)] = -np.inf
        wt = (synth.normal(i, 3, c * 9) * 0.4).astype(np.float32).reshape((c, 1, 1, 3, 3))
        gn = node('GroupConvolution', [x, wt], {'strides': '{0}, {0}'.format(st), 'dilations': '1, 1', 'pads_begin': '{}, {}'.format(*pb), 'pads_end': '{}, {}'.format(*pe), 'auto_pad': 'explicit'})
        act = rng.choice([None, ('relu',), ('clamp', 0.0, 6.0)])
        if act is not None:
            gn['_fuse_bias'], gn['_fuse_act'] = dev.DeviceTensor.from_numpy(synth.normal(i, 4, c).astype(np.float32).reshape((1, c, 1, 1))), act
        outs = []
        for mode in ('0', '1', '2'):
            setenv({'PVHIP_DWCONV_COLS': mode})
            outs.append(np.asarray(GroupConvolution.compute(dict(gn), {0: dev.DeviceTensor.from_numpy(x), 1: dev.DeviceTensor.from_numpy(wt)})[2]))
        setenv({})
        ok = all(bool(((o.view(np.uint32) == outs[0].view(np.uint32)) | (np.isnan(o) & np.isnan(outs[0]))).all()) for o in outs[1:])
        what = 'depthwise x{} stride {} pads {} {} act {}'.format(x.shape, st, pb, pe, act)
    elif kind == 'pool':              # MaxPool 3x3, stride 1 / 2: the pipelined kernel == the one-shot kernel, NaNs and infinities included
        c, st = rng.randint(1, 48), rng.choice([1, 2])
        n, h, w = rng.randint(1, 6), rng.randint(1, 120), rng.randint(1, 120)
        pb, pe = rng.choice([0, 1]), rng.choice([0, 1])
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 3.0
        for bad_v in (np.nan, -np.nan, np.inf, -np.inf):
            x.reshape(-1)[rng.randrange(x.size)] = bad_v
        pn = node('MaxPool', [x], {'kernel': '3, 3', 'strides': '{0}, {0}'.format(st), 'pads_begin': '{0}, {0}'.format(pb), 'pads_end': '{0}, {0}'.format(pe), 'rounding_type': rng.choice(['ceil', 'floor']), 'auto_pad': 'explicit'})
        outs = []
        try:
            for mode in ('0', '1'):
                setenv({'PVHIP_POOL3': mode})
                outs.append(np.asarray(MaxPool.compute(dict(pn), {0: dev.DeviceTensor.from_numpy(x)})[1]))
        except (ValueError, dev.PvhipError):
            setenv({})
            continue
        setenv({})
        ok = bool(((outs[1].view(np.uint32) == outs[0].view(np.uint32)) | (np.isnan(outs[1]) & np.isnan(outs[0]))).all())
        what = 'MaxPool 3x3 x{} stride {} pads {} {}'.format(x.shape, st, pb, pe)
    elif kind == 'poolconv':
        c, k = 16 * rng.randint(1, 6), rng.randint(1, 128)
        w = 2 * rng.randint(1, 15)
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w))
        wt = (synth.normal(i, 3, k * c) * (2.0 / c) ** 0.5).astype(np.float32).reshape((k, c, 1, 1))
        pn = node('MaxPool', [x], {'kernel': '3, 3', 'strides': '1, 1', 'pads_begin': '1, 1', 'pads_end': '1, 1', 'rounding_type': 'ceil', 'auto_pad': 'explicit'}); pn['output'][1]['dims'] = x.shape
        cn = node('Convolution', [x, wt], {'strides': '1, 1', 'dilations': '1, 1', 'pads_begin': '0, 0', 'pads_end': '0, 0', 'auto_pad': 'explicit'})
        if not Convolution.pooled_fusable(cn, pn):
            continue
        xd, wd = dev.DeviceTensor.from_numpy(x), dev.DeviceTensor.from_numpy(wt)
        two = np.asarray(Convolution.compute(dict(cn), {0: MaxPool.compute(dict(pn), {0: xd})[1], 1: wd})[2])
        one_n = dict(cn); one_n['_fuse_pool_in'] = pn
        one = np.asarray(Convolution.compute(one_n, {0: xd, 1: wd})[2])
        ok = bool((one.view(np.uint32) == two.view(np.uint32)).all()); what = 'MaxPool + 1x1 x{} k{}'.format(x.shape, k)
    else:
        c = 8 * rng.randint(1, 8)
        st = rng.choice([1, 2]); pb = rng.choice([0, 1]); pe = rng.choice([0, 1])
        x = synth.normal(i, 2, n * c * h * w).astype(np.float32).reshape((n, c, h, w)) * 30.0
        axes = np.array([1], dtype=np.int64)
        pdata = {'kernel': '3, 3', 'strides': '{0}, {0}'.format(st), 'pads_begin': '{0}, {0}'.format(pb), 'pads_end': '{0}, {0}'.format(pe), 'rounding_type': rng.choice(['ceil', 'floor']), 'auto_pad': 'explicit'}
        ldata = {'alpha': '9.9999997473787516e-05', 'beta': '0.75', 'bias': '1', 'size': '5'}
        xd = dev.DeviceTensor.from_numpy(x)
        try:
            if kind == 'poollrn':
                pn = node('MaxPool', [x], pdata)
                p = MaxPool.compute(dict(pn), {0: xd})[1]
                ln = node('LRN', [np.zeros(p.shape, np.float32), axes], ldata); ln['output'][2]['dims'] = tuple(p.shape)
                if not MaxPool.lrn_fusable(pn, ln):
                    continue
                two = np.asarray(LRN.compute(dict(ln), {0: p, 1: axes})[2])
                fn = dict(pn); fn['_fuse_lrn'] = ln
                one = np.asarray(MaxPool.compute(fn, {0: xd})[1])
            else:
                ln = node('LRN', [x, axes], ldata)
                l = LRN.compute(dict(ln), {0: xd, 1: axes})[2]
                pn = node('MaxPool', [x], pdata)
                two_t = MaxPool.compute(dict(pn), {0: l})[1]
                pn['output'][1]['dims'] = tuple(two_t.shape)
                if not LRN.pool_fusable(ln, pn):
                    continue
                two = np.asarray(two_t)
                fn = dict(ln); fn['_fuse_pool'] = pn
                one = np.asarray(LRN.compute(fn, {0: xd, 1: axes})[2])
        except (ValueError, dev.PvhipError):      # a window that does not fit the input: the library (like the reference) refuses
            continue
        ok = bool((one.view(np.uint32) == two.view(np.uint32)).all()); what = '{} x{} stride {} pads {} {}'.format(kind, x.shape, st, pb, pe)
    compared[kind] = compared.get(kind, 0) + 1
    if not ok:
        bad += 1
        print('MISMATCH', what, flush=True)
print('{} cases, compared {}, {} mismatches'.format(cases, compared, bad))
sys.exit(1 if bad else 0)
