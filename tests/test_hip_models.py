"""Whole models through the HIP plugins: against the reference's recorded outputs, against the oracle at
small batch, and through size-independent properties at the BASELINE batch sizes.  GPU only."""
import ctypes
import os

import numpy as np
import pytest

import helpers
from helpers import GOLDEN, assert_close, build_network, infer_one, layer_sums

pytestmark = pytest.mark.gpu

HIP = 'pyopenvino_amd.op_plugins'
ORACLE = 'oracle.op_plugins'


def test_mnist_real_weights_vs_reference(hip):
    """BASELINE config 1: models/mnist, mnist2.png -> the README.md:69-72 / integrity_test.py:57 answer."""
    z = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))
    _, net, ex = build_network(HIP, 'mnist', fuse=False)
    got = infer_one(ex, net, z['images'][0:1])
    assert_close(got, z['out'][0:1], helpers.REL_TOL, 'mnist2')
    assert list(np.argsort(got[0])[::-1][:3]) == [2, 0, 1]
    sums = layer_sums(net)
    for nid, want in zip(z['layer_ids'], z['layer_sums']):
        assert abs(sums[int(nid)] - want) <= 1e-4 * max(1.0, abs(want)), 'layer {}'.format(nid)
    _, net8, ex8 = build_network(HIP, 'mnist', batch=8)
    assert_close(infer_one(ex8, net8, z['images']), z['out'], helpers.REL_TOL, 'mnist batch 8')


def test_mnist_batch64_vs_oracle(hip):
    """BASELINE config 2: mnist batch 64 (mnist2 / mnist7 alternating + seeded noise images)."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))
    imgs = [z['images'][i % 2:i % 2 + 1] if i < 16 else synth.uniform_pixels(900 + i, (1, 1, 28, 28)) for i in range(64)]
    x = np.concatenate(imgs, 0)
    _, net, ex = build_network(HIP, 'mnist', batch=64)
    got = infer_one(ex, net, x)
    _, onet, oex = build_network(ORACLE, 'mnist', batch=64)
    want = infer_one(oex, onet, x)
    assert_close(got, want, helpers.REL_TOL, 'mnist batch 64')
    assert_close(got[0:1], z['out'][0:1], helpers.REL_TOL, 'row 0 == reference mnist2')
    assert np.allclose(got.sum(axis=1), 1.0, atol=1e-5)
    # second infer on the same executable network (buffers recycled through the pool): same answer
    assert np.array_equal(infer_one(ex, net, x), got)


@pytest.mark.parametrize('model,fname,shape', [('googlenet-v1', 'googlenet_e2e.npz', (3, 224, 224)),
                                               ('mnist_bn', 'mnist_bn_e2e.npz', (1, 28, 28))])
def test_synthetic_models_vs_reference(hip, model, fname, shape):
    """GoogLeNet / mnist_bn on seeded synthetic weights: batch 2 on the GPU vs two N=1 runs of the reference."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, fname))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, model + '.xml'), int(z['weight_seed']))
    x = np.concatenate([synth.uniform_pixels(int(s), (1,) + shape) for s in z['image_seeds']], 0)
    _, net, ex = build_network(HIP, model, weights=blob, batch=x.shape[0])
    got = infer_one(ex, net, x)
    assert_close(got, z['out'], helpers.REL_TOL, model)
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(z['out'], axis=1))
    # per-layer checksums of image 0 (batch 1 run)
    _, net1, ex1 = build_network(HIP, model, weights=blob, fuse=False)
    infer_one(ex1, net1, x[0:1])
    sums = layer_sums(net1)
    worst = 0.0
    for nid, want in zip(z['layer_ids'], z['layer_sums']):
        worst = max(worst, abs(sums[int(nid)] - want) / max(1.0, abs(want)))
    assert worst <= 1e-4, 'per-layer checksum drift {:.2e}'.format(worst)


def test_googlenet_layerwise_vs_oracle(hip):
    """Every node of GoogLeNet at batch 3: HIP output vs the oracle's output of the SAME node."""
    from pyopenvino_amd import synth
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), 1234)
    x = np.concatenate([synth.uniform_pixels(40 + i, (1, 3, 224, 224)) for i in range(3)], 0)
    _, net, ex = build_network(HIP, 'googlenet-v1', weights=blob, batch=3, fuse=False)
    _, onet, oex = build_network(ORACLE, 'googlenet-v1', weights=blob, batch=3)
    infer_one(ex, net, x)
    infer_one(oex, onet, x)
    worst, worst_el = ('', 0.0), 0.0
    for nid in net.G.nodes:
        node = net.G.nodes[nid]
        if node['type'] in ('Const', 'Parameter', 'Result'):
            continue
        for port, p in node['output'].items():
            got = np.asarray(p['data'])
            want = np.asarray(onet.G.nodes[nid]['output'][port]['data'])
            err = helpers.rel_err(got, want)
            if err > worst[1]:
                worst = ('{} {}'.format(nid, node['name']), err)
            assert err <= helpers.REL_TOL, 'node {} ({}): {:.2e}'.format(nid, node['name'], err)
            ex_ = helpers.elementwise_excess(got, want)
            worst_el = max(worst_el, ex_)
            assert ex_ <= 1.0, 'node {} ({}): an element is {:.2f} x outside 1e-4 |want| + 1e-4 rms'.format(nid, node['name'], ex_)
    print('worst layer', worst, 'worst element-wise excess', worst_el)


def test_googlenet_winograd_f4x4_end_to_end(hip, monkeypatch):
    """The F(4x4, 3x3) kernel forced onto every layer it covers at a small batch (at batch 256 it is the default for
    conv2/3x3 and the 28x28 inception layers): the network output stays inside the stated tolerance of the reference's
    recorded output, with margin."""
    from pyopenvino_amd import synth
    helpers.setenv(monkeypatch, 'PVHIP_CONV_WINOGRAD4', 'force')
    z = np.load(os.path.join(GOLDEN, 'googlenet_e2e.npz'))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), int(z['weight_seed']))
    x = np.concatenate([synth.uniform_pixels(int(s), (1, 3, 224, 224)) for s in z['image_seeds']], 0)
    _, net, ex = build_network(HIP, 'googlenet-v1', weights=blob, batch=x.shape[0])
    err = assert_close(infer_one(ex, net, x), z['out'], helpers.REL_TOL, 'GoogLeNet with F(4x4,3x3)')
    print('GoogLeNet end to end with F(4x4,3x3): {:.2e}'.format(err))
    assert err <= 3e-5, err


def test_googlenet_batch256_properties(hip):
    """BASELINE config 3 at full size (no CPU oracle run at this size): rows sum to 1; images 0-7 of the
    256-batch equal the reference's N=1 answers (googlenet_rows8.npz, max-norm and element by element); a permuted batch
    gives permuted rows (independence)."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'googlenet_rows8.npz'))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), int(z['weight_seed']))
    B = 256
    x = synth.uniform_pixels(4242, (B, 3, 224, 224))
    for i, s in enumerate(z['image_seeds']):
        x[i] = synth.uniform_pixels(int(s), (1, 3, 224, 224))[0]
    _, net, ex = build_network(HIP, 'googlenet-v1', weights=blob, batch=B)
    got = infer_one(ex, net, x)
    assert got.shape == (B, 1000) and np.isfinite(got).all()
    assert np.allclose(got.sum(axis=1), 1.0, atol=2e-5)
    assert_close(got[:8], z['out'], helpers.REL_TOL, 'rows 0-7 vs reference')
    assert np.array_equal(np.argsort(got[:8], axis=1)[:, -5:], np.argsort(z['out'], axis=1)[:, -5:])      # integrity_test.py:104-108 ranks
    perm = np.roll(np.arange(B), 37)
    got_p = infer_one(ex, net, np.ascontiguousarray(x[perm]))
    assert_close(got_p, got[perm], 1e-5, 'permuted batch')


def test_device_resident_input_and_result_gather_world1(hip):
    """A DeviceTensor handed to infer() is used in place (bench path); world==1 comm is the identity."""
    from pyopenvino_amd import shard
    z = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))
    _, net, ex = build_network(HIP, 'mnist', batch=8)
    ex.comm = shard.BatchShardComm(shard.SingleGroup())
    xd = hip.DeviceTensor.from_numpy(z['images'])
    got = infer_one(ex, net, xd)
    assert_close(got, z['out'], helpers.REL_TOL, 'device-resident input')


def test_fused_epilogue_is_bit_identical_and_aliases(hip):
    """Convolution -> Add(bias) -> ReLU run as one launch (the default) gives exactly the bits of the three
    separate launches; the fused-away nodes' output ports alias the fused tensor."""
    from pyopenvino_amd import synth
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), 1234)
    x = np.concatenate([synth.uniform_pixels(70 + i, (1, 3, 224, 224)) for i in range(2)], 0)
    _, net_f, ex_f = build_network(HIP, 'googlenet-v1', weights=blob, batch=2)
    _, net_u, ex_u = build_network(HIP, 'googlenet-v1', weights=blob, batch=2, fuse=False)
    assert len(ex_f._fusion) == 57 and len(ex_f._fused_away) == 114 + 9 + 2 + 18 + 7 + 1 + 1 and len(ex_f._concat_direct) == 9 and not ex_u._fusion
    assert len(ex_f._stem_conv) == 1 and not ex_u._stem_conv          # conv2/3x3_reduce inside the pool1/3x3_s2 + pool1/norm1 launch (its Add / ReLU were folded away already)
    assert len(ex_f._pre_add) == 1 and not ex_u._pre_add              # data/mean folded into conv1's padding pass
    assert len(ex_f._pool_conv) == 7 and not ex_u._pool_conv           # pool + pool_proj of the 28x28 and 14x14 modules (3a .. 4e) as one launch (the 7x7 ones only with PVHIP_FUSE_POOLCONV=1: slower)
    assert len(ex_f._siblings) == 9 and not ex_u._siblings            # 1x1 + 3x3_reduce + 5x5_reduce of a module as one launch
    assert len(ex_f._lrn_pool) == 2 and not ex_u._lrn_pool          # conv2/norm2 -> pool2/3x3_s2 and pool1/3x3_s2 -> pool1/norm1 as one launch each
    out_f, out_u = infer_one(ex_f, net_f, x), infer_one(ex_u, net_u, x)
    assert np.array_equal(out_f, out_u)
    for cid, f in ex_f._fusion.items():
        fused = next(iter(net_f.G.nodes[cid]['output'].values()))['data']
        relu_u = next(iter(net_u.G.nodes[f['relu']]['output'].values()))['data']
        assert next(iter(net_f.G.nodes[f['relu']]['output'].values()))['data'] is fused
        if cid in (4, 293) or cid in ex_f._stem_conv.values():   # first and one late layer (the latter written in place into its Concat), and the one inside the MaxPool + LRN launch: bit for bit
            helpers.assert_bit_exact(np.asarray(fused), np.asarray(relu_u), 'fused conv {}'.format(cid))
    for lid, pid in ex_f._lrn_pool.items():   # LRN + MaxPool in one launch == the two launches; the MaxPool's port carries the tensor
        if lid in ex_f._stem_conv:            # (the 1x1 convolution behind this pair rides in the launch: neither tensor exists, the chain's output was checked above)
            continue
        port = next(iter(net_f.G.nodes[pid]['output']))
        assert net_f.G.nodes[pid]['output'][port]['data'] is next(iter(net_f.G.nodes[lid]['output'].values()))['data']
        helpers.assert_bit_exact(np.asarray(net_f.G.nodes[pid]['output'][port]['data']),
                                 np.asarray(net_u.G.nodes[pid]['output'][port]['data']), 'lrn+pool {}'.format(pid))
    for cat in ex_f._concat_direct:   # the inception outputs assembled by their producers == the Concat kernel's result
        port = next(iter(net_f.G.nodes[cat]['output']))
        helpers.assert_bit_exact(np.asarray(net_f.G.nodes[cat]['output'][port]['data']),
                                 np.asarray(net_u.G.nodes[cat]['output'][port]['data']), 'concat {}'.format(cat))


def test_forked_streams_give_the_single_stream_bits(hip):
    """The inception arms forked onto 4 compute streams (the default) give exactly the bits of the serial single
    stream run, for every layer, on repeated passes (a missing event would show as a stale or half-written
    tensor); the pool does not grow from pass to pass (blocks freed during a forked pass come back after its
    closing synchronisation)."""
    from pyopenvino_amd import synth
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), 1234)
    B = 48    # large enough that kernels of different arms really overlap
    x = synth.uniform_pixels(555, (B, 3, 224, 224))
    _, net_s, ex_s = build_network(HIP, 'googlenet-v1', weights=blob, batch=B)
    ex_s.compute_streams = 1
    want = infer_one(ex_s, net_s, x)
    want_layers = helpers.layer_sums(net_s)
    for fuse in (True, False):
        _, net_m, ex_m = build_network(HIP, 'googlenet-v1', weights=blob, batch=B, fuse=fuse)
        ex_m.compute_streams = 4
        assert ex_m.plan_streams() is not None
        sizes = []
        for it in range(4):
            got = infer_one(ex_m, net_m, x)
            assert np.array_equal(got, want), 'pass {} fuse {}'.format(it, fuse)
            in_use, cached = ctypes.c_size_t(0), ctypes.c_size_t(0)
            hip.call('pvhip_pool_stats', ctypes.byref(in_use), ctypes.byref(cached))
            sizes.append(in_use.value + cached.value)
        assert sizes[-1] == sizes[-2] == sizes[-3], sizes
        if fuse:
            got_layers = helpers.layer_sums(net_m)
            assert got_layers == want_layers
    del net_s, ex_s


def test_requests_in_flight_give_the_synchronous_bits(hip):
    """Three GoogLeNet requests in flight on disjoint stream sets (load_network(num_requests=3)) return exactly what
    the synchronous infer() returns for the same inputs, over several rounds with the requests restarted in a
    different order; the pool stops growing after the first rounds (workspaces of a pass are parked until it has
    finished, previous outputs are recycled at once)."""
    from pyopenvino_amd import IECore, synth
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    blob = synth.synth_weights(xml, 1234)
    B, R = 32, 3
    ie = IECore(plugin_package=HIP)
    net = ie.read_network(xml, weights=blob)
    net.set_batch(B)
    ex = ie.load_network(net, 'GPU', num_requests=R)
    assert [r.runner.stream_base for r in ex.requests] == [0, 1, 2] and ex.requests[2].runner.compute_streams == 1
    name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    xs = [synth.uniform_pixels(900 + i, (B, 3, 224, 224)) for i in range(R)]
    want = [ex.infer({name: x})[out_name].copy() for x in xs]
    for r in ex.requests:                    # two streams per request: forks inside a request AND requests side by side
        r.runner.stream_base, r.runner.compute_streams = 2 * r.index, 2
    sizes = []
    for rnd_ in range(5):
        order = [(i + rnd_) % R for i in range(R)]
        for i in order:
            ex.start_async(i, {name: xs[(i + rnd_) % R]})
        for i in reversed(order):
            got = ex.wait(i)[out_name]
            assert np.array_equal(got, want[(i + rnd_) % R]), 'round {} request {}'.format(rnd_, i)
        in_use, cached = ctypes.c_size_t(0), ctypes.c_size_t(0)
        hip.call('pvhip_pool_stats', ctypes.byref(in_use), ctypes.byref(cached))
        sizes.append(in_use.value + cached.value)
    assert sizes[-1] == sizes[-2] == sizes[-3], sizes
    # the synchronous call still works on the same network afterwards
    assert np.array_equal(ex.infer({name: xs[1]})[out_name], want[1])


def test_requests_with_device_resident_inputs_replay_their_own_recordings(hip, monkeypatch):
    """start_async() with the SAME device tensors call after call: after AUTO_GRAPH_AFTER eager passes each request records its pass
    into its own hipGraph (own stream, own tensors) and replays it with one call; three requests side by side, started in changing
    orders, return the bits of the synchronous eager pass every time; a request given another tensor drops its recording, computes
    eagerly and records again; PVHIP_AUTO_GRAPH=0 keeps every request eager.  (The temporaries of a recorded pass -- padded images,
    workspaces -- stay with ITS graph: were they handed to another request, two replays would share scratch memory.)"""
    from pyopenvino_amd import IECore, synth, device
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    blob = synth.synth_weights(xml, 1234)
    B, R = 16, 3
    ie = IECore(plugin_package=HIP)
    net = ie.read_network(xml, weights=blob)
    net.set_batch(B)
    ex = ie.load_network(net, 'GPU', num_requests=R)
    name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    xs_host = [synth.uniform_pixels(700 + i, (B, 3, 224, 224)) for i in range(R + 1)]
    monkeypatch.setenv('PVHIP_AUTO_GRAPH', '0')
    want = [ex.infer({name: x})[out_name].copy() for x in xs_host]
    xs = [device.DeviceTensor.from_numpy(x) for x in xs_host]
    for rnd_ in range(3):                                   # switched off: eager, and nothing is recorded
        for i in range(R):
            ex.start_async(i, {name: xs[i]})
        for i in range(R):
            assert np.array_equal(ex.wait(i)[out_name], want[i])
    assert all(r.runner.__dict__.get('_graph') is None for r in ex.requests)
    monkeypatch.delenv('PVHIP_AUTO_GRAPH')
    for rnd_ in range(7):
        order = [(i + rnd_) % R for i in range(R)]
        for i in order:
            ex.start_async(i, {name: xs[i]})
        for i in reversed(order):
            assert np.array_equal(ex.wait(i)[out_name], want[i]), 'round {} request {}'.format(rnd_, i)
        if rnd_ == 1:
            assert all(r.runner.__dict__.get('_graph') is None for r in ex.requests)
        if rnd_ >= 2:
            assert all(r.runner.__dict__.get('_graph') is not None for r in ex.requests), rnd_
    handles = [r.runner._graph['handle'] for r in ex.requests]
    assert len(set(handles)) == R
    for rnd_ in range(4):                                   # request 1 moves to another tensor while 0 and 2 keep replaying
        ex.start_async(0, {name: xs[0]})
        ex.start_async(1, {name: xs[R]})
        ex.start_async(2, {name: xs[2]})
        assert np.array_equal(ex.wait(2)[out_name], want[2])
        assert np.array_equal(ex.wait(1)[out_name], want[R]), rnd_
        assert np.array_equal(ex.wait(0)[out_name], want[0])
        assert (ex.requests[1].runner.__dict__.get('_graph') is not None) == (rnd_ >= 2)
    assert [r.runner._graph['handle'] for r in (ex.requests[0], ex.requests[2])] == [handles[0], handles[2]]
    for x_dev, x_host in zip(xs, xs_host):                  # no recording ever wrote into a caller's tensor
        assert np.array_equal(np.asarray(x_dev), x_host)
    assert np.array_equal(ex.infer({name: xs_host[1]})[out_name], want[1])      # the synchronous call on the same network afterwards


def test_eight_requests_of_batch_256_replayed_side_by_side_match_the_reference_and_the_eager_bits(hip):
    """The exact shape bench.py times: 8 requests x batch 256, each replaying its own hipGraph with the seven others in flight
    (lesson 24: wrong images showed ONLY at batch 256 next to other streams).  Rows 0-7 of request 0 are the reference's own
    N=1 answers (googlenet_rows8.npz); every request's replayed Result equals, bit for bit, one eager synchronous infer() of the
    same tensor -- in three rounds with changing start orders."""
    from pyopenvino_amd import IECore, synth, device
    z = np.load(os.path.join(GOLDEN, 'googlenet_rows8.npz'))
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    blob = synth.synth_weights(xml, int(z['weight_seed']))
    B, R = 256, 8
    ie = IECore(plugin_package=HIP)
    net = ie.read_network(xml, weights=blob)
    net.set_batch(B)
    ex = ie.load_network(net, 'GPU', num_requests=R)
    name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    xs = []
    for r in range(R):
        x = synth.uniform_pixels(9000 + r, (B, 3, 224, 224))
        if r == 0:
            for i, sd in enumerate(z['image_seeds']):
                x[i] = synth.uniform_pixels(int(sd), (1, 3, 224, 224))[0]
        xs.append(device.DeviceTensor.from_numpy(x))
        del x
    for rnd_ in range(4):                               # two eager passes, the third records, then replays
        for r in range(R):
            ex.start_async(r, {name: xs[r]})
        outs = [np.array(ex.wait(r)[out_name], copy=True) for r in range(R)]
    assert all(req.runner.__dict__.get('_graph') is not None for req in ex.requests)
    for rnd_ in range(3):
        order = [(r * 3 + rnd_) % R for r in range(R)]
        for r in order:
            ex.start_async(r, {name: xs[r]})
        for r in reversed(order):
            got = ex.wait(r)[out_name]
            assert np.array_equal(got, outs[r]), 'round {} request {}'.format(rnd_, r)
    assert_close(outs[0][:8], z['out'], helpers.REL_TOL, 'request 0 rows 0-7 vs the reference')
    os.environ['PVHIP_AUTO_GRAPH'] = '0'
    try:
        for r in range(R):
            eager = ex.requests[r].infer({name: xs[r]})[out_name]
            assert np.array_equal(np.asarray(eager), outs[r]), 'request {}: replay differs from the eager pass'.format(r)
            assert np.isfinite(outs[r]).all() and np.allclose(outs[r].sum(axis=1), 1.0, atol=2e-5)
    finally:
        os.environ.pop('PVHIP_AUTO_GRAPH')


def test_rccl_binding_single_rank(hip):
    """The RCCL path of the C ABI (dlopen, unique id, communicator, all-gather, destroy) with one rank:
    exercises every call the multi-GPU Result gather makes; with world == 1 the gather is a device copy."""
    import ctypes
    buf = ctypes.create_string_buffer(hip.UNIQUE_ID_BYTES)
    hip.call('pvhip_comm_unique_id', buf)
    assert any(b != 0 for b in buf.raw)
    hip.call('pvhip_comm_init', ctypes.c_char_p(buf.raw), 0, 1)
    try:
        x = np.arange(256 * 1000, dtype=np.float32).reshape(256, 1000) * 0.5
        send = hip.DeviceTensor.from_numpy(x)
        recv = hip.DeviceTensor.empty(x.shape)
        hip.call('pvhip_comm_allgather_f32', ctypes.c_void_p(send.ptr), ctypes.c_void_p(recv.ptr), x.size)
        helpers.assert_bit_exact(np.asarray(recv), x, 'single-rank all-gather')
        ranks = ctypes.c_int(0)
        hip.call('pvhip_comm_ranks', ctypes.byref(ranks))          # ncclCommCount: what bench.py reports as rccl_ranks
        assert ranks.value == 1
    finally:
        hip.call('pvhip_comm_destroy')


def test_ssd_backbone_vs_reference_and_oracle(hip):
    """BASELINE config 5: SSD-MobileNet backbone + box/class heads (Convolution, depthwise GroupConvolution,
    Clamp, Add, Multiply, Transpose, Reshape, Concat, Sigmoid) on the GPU; image 0 against the reference's
    recorded output, a second image against the oracle."""
    from pyopenvino_amd import synth
    from test_oracle_golden import check_ssd_against_fixture, ssd_backbone
    z = np.load(os.path.join(GOLDEN, 'ssd_backbone_e2e.npz'))
    x = np.concatenate([synth.uniform_pixels(int(z['image_seed']), (1, 3, 300, 300)), synth.uniform_pixels(701, (1, 3, 300, 300))], 0)
    got = ssd_backbone(HIP, 2, x)
    check_ssd_against_fixture(got, z, helpers.REL_TOL)
    want = ssd_backbone(ORACLE, 2, x)
    for k in got:
        assert_close(got[k], want[k], helpers.REL_TOL, 'ssd ' + k)


def test_ssd_whole_ir_vs_reference_and_batch(hip):
    """the whole SSD IR on the device (prior boxes constant-folded and uploaded once, DetectionOutput
    kernel) against the reference's own infer() at N=1, and at batch 6 against the oracle (images independent)."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'ssd_full_e2e.npz'))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml'), int(z['weight_seed']))
    x = synth.uniform_pixels(int(z['image_seed']), (1, 3, 300, 300))
    _, net, ex = build_network(HIP, 'ssd_mobilenet_v1_coco', weights=blob)
    out = infer_one(ex, net, x)
    assert out.shape == z['out'].shape
    assert np.array_equal(out[..., :2], z['out'][..., :2]), 'record order / classes differ from the reference'
    assert_close(out, z['out'], helpers.REL_TOL, 'ssd detections vs reference')
    by_name = {net.G.nodes[n]['name']: n for n in net.G.nodes}
    priors = next(iter(net.G.nodes[by_name['ConcatPriorBoxesClustered']]['output'].values()))['data']
    helpers.assert_bit_exact(np.asarray(priors), z['priors'], 'prior boxes')
    B = 6
    xb = np.concatenate([synth.uniform_pixels(800 + i, (1, 3, 300, 300)) for i in range(B - 1)] + [x], 0)
    _, net_b, ex_b = build_network(HIP, 'ssd_mobilenet_v1_coco', weights=blob, batch=B)
    got = infer_one(ex_b, net_b, xb)
    assert got.shape == (1, 1, B * 100, 7)
    assert np.array_equal(got[0, 0, (B - 1) * 100:, :2], z['out'][0, 0, :, :2])
    assert_close(got[:, :, (B - 1) * 100:], z['out'], helpers.REL_TOL, 'last image of the batch vs reference')
    _, net_o, ex_o = build_network(ORACLE, 'ssd_mobilenet_v1_coco', weights=blob, batch=2)
    ex_o.kernel_type = 'special'
    want2 = infer_one(ex_o, net_o, xb[:2])
    # scores within 1e-4 of each other may legitimately swap ranks between fp32 implementations: compare as sets of rows
    for b in range(2):
        g, w = got[0, 0, b * 100:(b + 1) * 100], want2[0, 0, b * 100:(b + 1) * 100]
        assert np.array_equal(np.sort(g[:, 1]), np.sort(w[:, 1])), 'image {}: detected classes differ'.format(b)
        assert_close(g[np.argsort(-g[:, 2], kind='stable'), 2:], w[np.argsort(-w[:, 2], kind='stable'), 2:], helpers.REL_TOL, 'image {} of the batch'.format(b))


def test_ssd_batch128_properties(hip):
    """BASELINE config 5 at full size (no CPU oracle run at this size): the whole SSD IR at batch 128 returns finite
    records; the image that is the reference fixture's (put LAST in the batch) reproduces the reference's detections; a
    permuted batch gives the permuted blocks of records (images are independent through DetectionOutput too)."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'ssd_full_e2e.npz'))
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml'), int(z['weight_seed']))
    B = 128
    x = synth.uniform_pixels(777, (B, 3, 300, 300))
    x[B - 1] = synth.uniform_pixels(int(z['image_seed']), (1, 3, 300, 300))[0]
    _, net, ex = build_network(HIP, 'ssd_mobilenet_v1_coco', weights=blob, batch=B)
    got = infer_one(ex, net, x)
    assert got.shape == (1, 1, B * 100, 7) and np.isfinite(got).all()
    last = got[:, :, (B - 1) * 100:]
    assert np.array_equal(last[0, 0, :, :2], z['out'][0, 0, :, :2]), 'record order / classes differ from the reference'
    assert_close(last, z['out'], helpers.REL_TOL, 'image 127 of the batch vs reference')
    perm = np.roll(np.arange(B), 11)
    got_p = infer_one(ex, net, np.ascontiguousarray(x[perm]))
    blocks, blocks_p = got.reshape(B, 100, 7), got_p.reshape(B, 100, 7)
    assert np.array_equal(blocks_p, blocks[perm]), 'records of a permuted batch are not the permuted records'


def test_infer_without_explicit_device_init():
    """A fresh process that never calls device.init(): the first infer() binds the GPU itself (the scheduler's stream /
    pool calls come before the first tensor is created)."""
    import subprocess
    import sys
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r); "
            "from pyopenvino_amd import IECore; ie = IECore(); "
            "net = ie.read_network(os.path.join(%r, 'mnist.xml')); ex = ie.load_network(net); "
            "out = ex.infer({net.inputs[0]['name']: np.zeros((1, 1, 28, 28), np.float32)}); "
            "print(next(iter(out.values())).shape)") % (helpers.REPO, helpers.MODELS)
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and '(1, 10)' in res.stdout, res.stderr[-2000:]


def test_fp16_ir_computed_in_fp32(hip, tmp_path):
    """SURVEY 8(f)-4, first step: an FP16 IR loads (constants upcast once, ports declared FP32) and runs on the fp32 kernels."""
    from test_oracle_golden import check_fp16_ir
    check_fp16_ir(HIP, tmp_path)


def test_fp16_ir_on_the_f16_matrix_cores(hip, tmp_path):
    """models/mnist as an FP16 IR read with fp16_as_fp32=False: Convolution and MatMul run the f16-MFMA kernels
    (fp16 operands, fp32 accumulation; every other node in fp32).  Against the reference's own numpy-float16 run of the same
    IR: the logits within fp16 tolerance (2e-2 of their maximum: the reference rounds every tensor and every partial sum to
    float16), the same classes; against this build's fp32 arithmetic on the same IR: 5e-3."""
    from pyopenvino_amd import IECore, synth
    z = np.load(os.path.join(GOLDEN, 'mnist_fp16_e2e.npz'))
    images = np.load(os.path.join(GOLDEN, 'mnist_e2e.npz'))['images'][:int(z['n_images'])]
    blob = open(os.path.join(helpers.MODELS, 'mnist.bin'), 'rb').read()
    xml16, blob16 = synth.fp16_ir(os.path.join(helpers.MODELS, 'mnist.xml'), blob, str(tmp_path))
    ie = IECore(plugin_package=HIP)
    outs = {}
    for mode in (False, True):
        net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=mode)
        assert net.ir_precision == 'FP16' and net.f16_mfma == (not mode)
        net.set_batch(len(images))
        ex = ie.load_network(net)
        prob = helpers.infer_one(ex, net, images)
        soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
        logits = np.asarray(next(iter(net.G.nodes[next(iter(net.G.pred[soft]))]['output'].values()))['data'])
        outs[mode] = (prob, logits)
        ran_f16 = ['_hip_f16' in net.G.nodes[n] for n in net.G.nodes if net.G.nodes[n]['type'] == 'Convolution']
        assert all(ran_f16) if not mode else not any(ran_f16)
    prob, logits = outs[False]
    assert np.isfinite(prob).all() and np.array_equal(prob.argmax(axis=1), z['logits'].argmax(axis=1))
    err = helpers.rel_err(logits, z['logits'])
    print('FP16 IR on f16 MFMA: logits {:.2e} from the reference float16 run, {:.2e} from fp32 arithmetic'.format(
        err, helpers.rel_err(logits, outs[True][1])))
    assert err <= 2e-2, err
    assert_close(logits, outs[True][1], 5e-3, 'f16 MFMA vs fp32 arithmetic on the FP16 IR', elementwise=False)
    assert not np.array_equal(logits, outs[True][1])
    # an FP32 IR is never switched to f16 arithmetic
    net32 = ie.read_network(os.path.join(helpers.MODELS, 'mnist.xml'), fp16_as_fp32=False)
    assert net32.ir_precision == 'FP32' and not net32.f16_mfma


@pytest.mark.parametrize('model,shape,streams', [('mnist', (1, 28, 28), 1), ('googlenet-v1', (3, 224, 224), 4), ('googlenet-v1', (3, 224, 224), 1)])
def test_hipgraph_replay_of_a_forward_pass_is_bit_identical(hip, model, shape, streams):
    """Executable_Network.capture_graph records one pass (all streams of the stream plan) into a hipGraph; infer_graph replays it
    with one call: the bits of the eager infer(), also for NEW inputs (copied into the captured input tensor) and after eager
    passes in between (the graph keeps its tensors alive; blocks freed during the capture stay pinned)."""
    from pyopenvino_amd import synth
    B = 8
    blob = None if model == 'mnist' else synth.synth_weights(os.path.join(helpers.MODELS, model + '.xml'), 1234)
    _, net, ex = build_network(HIP, model, weights=blob, batch=B)
    ex.compute_streams = streams
    name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    x1, x2 = (synth.uniform_pixels(s, (B,) + shape) for s in (31, 32))
    d1, d2 = hip.DeviceTensor.from_numpy(x1), hip.DeviceTensor.from_numpy(x2)
    want1 = infer_one(ex, net, d1)
    want2 = infer_one(ex, net, d2)
    assert not np.array_equal(want1, want2)
    ex.capture_graph({name: hip.DeviceTensor.from_numpy(x1)}, streams='plan' if streams > 1 else 1)      # the graph's own input tensor: new inputs are copied into it
    helpers.assert_bit_exact(ex.infer_graph()[out_name], want1, 'replay')
    helpers.assert_bit_exact(ex.infer_graph({name: d2})[out_name], want2, 'replay with new input')
    helpers.assert_bit_exact(infer_one(ex, net, d1), want1, 'eager pass after a capture')
    helpers.assert_bit_exact(ex.infer_graph({name: x1})[out_name], want1, 'replay after an eager pass, host input')
    ex.release_graph()
    with pytest.raises(RuntimeError):
        ex.infer_graph()
    helpers.assert_bit_exact(infer_one(ex, net, d2), want2, 'eager pass after the graph is gone')


@pytest.mark.parametrize('kind,streams', [('unfused', 3), ('ssd', 4), ('fp16', 4), ('unfused', 4)])
def test_forked_recordings_that_used_to_kill_the_process_replay_the_eager_bits(hip, kind, streams):
    """capture_graph(streams='plan') on the plans whose recording ended in a stack overflow inside hipStreamEndCapture (ROCm 7.2:
    its recursive walk over the per-stream lists of parallel capture streams meets a ring when non-origin streams wait for each
    other in both directions over time; LESSONS.md lesson 30 rewritten, profiles/r04_capture.md): the dispatcher relays the
    ring-closing waits through the origin stream while it records, the recording is made, and its replay gives the bits of the
    eager pass.  The waits issued are the ones Executable_Network.recorded_waits() predicts from the plan."""
    import tempfile
    from pyopenvino_amd import IECore, synth
    ie = IECore(plugin_package=HIP)
    B = 8
    if kind == 'ssd':
        xml = os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml')
        net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
        shape = (B, 3, 300, 300)
    else:
        xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
        blob = synth.synth_weights(xml, 1234)
        shape = (B, 3, 224, 224)
        if kind == 'fp16':
            with tempfile.TemporaryDirectory() as tmp:
                xml16, blob16 = synth.fp16_ir(xml, blob, tmp)
                net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
        else:
            net = ie.read_network(xml, weights=blob)
    net.set_batch(B)
    ex = ie.load_network(net)
    if kind == 'unfused':
        ex.fuse_epilogues = False
        ex.plan_fusion()
    ex.compute_streams = streams
    x = hip.DeviceTensor.from_numpy(synth.uniform_pixels(5, shape))
    name = net.inputs[0]['name']
    want = {k: np.array(np.asarray(v), copy=True) for k, v in ex.infer({name: x}).items()}
    predicted, _ = ex.recorded_waits()
    assert (kind == 'fp16') == (not any(how == 'relay' for how, *_ in predicted))       # the FP16 plan needs no relay, the others do
    ex._stream_ops = []
    try:
        ex.capture_graph({name: x}, streams='plan')
        issued = list(ex._stream_ops)
    finally:
        del ex._stream_ops
    assert issued[-len(predicted):] == predicted        # (capture_graph's warm-up passes come first: eager, every wait plain)
    assert all(how == 'plain' for how, *_ in issued[:-len(predicted)])
    for _ in range(2):
        got = ex.infer_graph()
        for k in want:
            helpers.assert_bit_exact(np.asarray(got[k]), want[k], 'replay of the forked recording: {}'.format(k))
    ex.release_graph()


def test_pickle_node_args_from_a_fused_device_pass_replays_through_the_oracle(hip, tmp_path):
    """The reference's node-replay hook on the product path: a Convolution that runs fused (bias + ReLU in its epilogue, device
    tensors in and out) is dumped as the plain IR node with host ndarrays, and replaying the file through the oracle's plugin
    (`test_node_sample.py:1-16` style) gives the convolution the device computed before its epilogue."""
    import importlib
    import pickle
    from pyopenvino_amd import synth
    _, net, ex = build_network(HIP, 'mnist', batch=4)
    conv = [n for n in net.G.nodes if net.G.nodes[n]['type'] == 'Convolution'][1]
    ex.pickle_node_args, ex.pickle_dir = [conv], str(tmp_path)
    x = np.concatenate([synth.uniform_pixels(70 + i, (1, 1, 28, 28)) for i in range(4)], 0)
    infer_one(ex, net, x)
    with open(os.path.join(str(tmp_path), 'node_args_{}.pickle'.format(conv)), 'rb') as f:
        node, inputs = pickle.load(file=f)
    assert all(type(v) is np.ndarray for v in inputs.values()) and '_fuse_bias' not in node
    want = next(iter(importlib.import_module(ORACLE + '.Convolution').compute(node, inputs, kernel_type='special').values()))
    _, net1, ex1 = build_network(HIP, 'mnist', batch=4, fuse=False)
    infer_one(ex1, net1, x)
    got = np.asarray(next(iter(net1.G.nodes[conv]['output'].values()))['data'])
    assert_close(got, want, helpers.REL_TOL, 'replayed node {}'.format(node['name']))


def _googlenet_fp16_logits(plugin_package, fp16_as_fp32, images, tmp_path):
    from pyopenvino_amd import IECore, synth
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    xml16, blob16 = synth.fp16_ir(xml, synth.synth_weights(xml, 1234), str(tmp_path))
    ie = IECore(plugin_package=plugin_package)
    net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=fp16_as_fp32)
    net.set_batch(len(images))
    ex = ie.load_network(net)
    prob = helpers.infer_one(ex, net, images)
    soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
    return prob, np.asarray(next(iter(net.G.nodes[next(iter(net.G.pred[soft]))]['output'].values()))['data']), net


def test_googlenet_fp16_ir_on_the_f16_matrix_cores_vs_reference_float16(hip, tmp_path, monkeypatch):
    """The FP16 entry bench.py times (GoogLeNet as an FP16 IR, Convolution / MatMul on v_mfma_f32_32x32x16_f16 with fp32 accumulation)
    against the REFERENCE's numpy-float16 run of the same IR (googlenet_fp16_rows2.npz: float16 logits of images 500 / 501; its
    float16 SoftMax overflows): logits within 1e-2 of their maximum (the reference rounds every tensor and partial sum to float16; fp32
    arithmetic on the same constants is 9e-4 away from it), the same class; and within 5e-3 of this build's fp32 pass on the same IR."""
    from pyopenvino_amd import synth
    z = np.load(os.path.join(GOLDEN, 'googlenet_fp16_rows2.npz'))
    images = np.concatenate([synth.uniform_pixels(int(s), (1, 3, 224, 224)) for s in z['image_seeds']], 0)
    helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', '1')         # the first step only: the tensors between a 1x1 convolution and the 3x3 / 5x5 behind it
    try:
        prob16, logits16, net16 = _googlenet_fp16_logits(HIP, False, images, tmp_path)
    finally:
        helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', None)
    assert net16.f16_mfma and all('_hip_f16' in net16.G.nodes[n] for n in net16.G.nodes if net16.G.nodes[n]['type'] == 'Convolution')
    # the 3x3_reduce / 5x5_reduce tensors are fp16 in HBM (blocked by eight channels) and every 3x3 / 5x5 convolution reads them so
    spatial = [n for n in net16.G.nodes if net16.G.nodes[n]['type'] == 'Convolution' and net16.G.nodes[n]['input'][1]['dims'][2] in (3, 5)]
    assert len(spatial) == 19 and all(net16.G.nodes[n]['_hip_f16'] == 'c8' for n in spatial)
    helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', '0')
    try:
        _, logits_dense, net_dense = _googlenet_fp16_logits(HIP, False, images, tmp_path)
    finally:
        helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', None)
    assert not any(net_dense.G.nodes[n]['_hip_f16'] == 'c8' for n in spatial)
    assert_close(logits16, logits_dense, 1e-3, 'fp16 tensors between the convolutions vs fp32 tensors rounded at the reader', elementwise=False)      # the same values, other summation orders, and fp16 roundings that flip behind them
    prob32, logits32, net32 = _googlenet_fp16_logits(HIP, True, images, tmp_path)
    assert not net32.f16_mfma
    err_ref, err_32 = helpers.rel_err(logits16, z['logits']), helpers.rel_err(logits16, logits32)
    print('GoogLeNet FP16 IR on f16 MFMA: logits {:.2e} from the reference float16 run, {:.2e} from fp32 arithmetic; fp32 arithmetic {:.2e} '
          'from the reference'.format(err_ref, err_32, helpers.rel_err(logits32, z['logits'])))
    assert np.isfinite(prob16).all() and np.array_equal(logits16.argmax(axis=1), z['logits'].argmax(axis=1))
    assert err_ref <= 1e-2, err_ref
    # ... and element by element (1e-2 |want| + 1e-2 rms(want)): the max-norm alone would not notice a wrong channel block of small values
    ex_ref = helpers.elementwise_excess(logits16, z['logits'].astype(np.float32), 1e-2)
    print('  worst element at {:.2f} of its 1e-2 allowance'.format(ex_ref))
    assert ex_ref <= 1.0, ex_ref
    assert helpers.rel_err(logits32, z['logits']) <= 3e-3
    assert_close(logits16, logits32, 5e-3, 'f16 MFMA vs fp32 arithmetic, GoogLeNet FP16 IR', elementwise=False)
    assert np.abs(prob16.sum(axis=1) - 1).max() <= 1e-4


def test_googlenet_fp16_ir_whole_modules_on_blocked_fp16_tensors(hip, tmp_path, monkeypatch):
    """PVHIP_CONV_F16_C8=2: the inception modules of the FP16 IR from the module input to the channel Concat on fp16 tensors blocked by eight
    channels (pvhip_conv2d_f16_c8_multi, pvhip_maxpool3x3_c8): every Concat buffer is blocked, the tensor module 3a reads is converted once.
    Against the reference's float16 run of the same IR (1e-2 of the logits' maximum, the same classes) and against the default mode
    (fp32 Concat buffers; 2e-3: the same arithmetic, one more fp16 rounding per module output -- which the reference does too)."""
    from pyopenvino_amd import device, synth
    z = np.load(os.path.join(GOLDEN, 'googlenet_fp16_rows2.npz'))
    images = np.concatenate([synth.uniform_pixels(int(s), (1, 3, 224, 224)) for s in z['image_seeds']], 0)
    helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', '1')
    _, logits1, _ = _googlenet_fp16_logits(HIP, False, images, tmp_path)
    helpers.setenv(monkeypatch, 'PVHIP_CONV_F16_C8', None)         # the default
    from pyopenvino_amd import IECore
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    xml16, blob16 = synth.fp16_ir(xml, synth.synth_weights(xml, 1234), str(tmp_path))
    ie = IECore(plugin_package=HIP)
    net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
    net.set_batch(len(images))
    ex = ie.load_network(net)
    prob = helpers.infer_one(ex, net, images)
    assert len(ex._c8_concat) == 9 and len(ex._c8_entry) == 0          # (no conversion: the stem hands module 3a a blocked tensor)
    assert len(ex._stem_conv) == 1                                       # conv2/3x3_reduce rides in the blocked MaxPool + LRN launch (round 5)
    reduce_ = next(iter(ex._stem_conv.values()))
    assert net.G.nodes[reduce_]['_hip_f16'] == 'inside MaxPool + LRN (blocked tensors)'
    conv1 = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'Convolution' and net.G.nodes[n]['input'][1]['dims'][2] == 7)
    assert net.G.nodes[conv1]['_hip_f16'] == 'row spans, blocked output'        # the stem: conv1 -> (MaxPool + LRN on the blocked tensor) -> conv2/3x3_reduce
    pool1 = next(iter(net.G.successors(ex._fusion[conv1]['relu'])))
    assert isinstance(next(iter(net.G.nodes[pool1]['output'].values()))['data'], device.BlockedHalf)
    pool2 = next(n for n in net.G.nodes if net.G.nodes[n]['name'] == 'pool2/3x3_s2')
    assert isinstance(next(iter(net.G.nodes[pool2]['output'].values()))['data'], device.BlockedHalf)      # conv2/3x3 -> (LRN + MaxPool on the blocked tensor) -> module 3a
    kinds = [net.G.nodes[n].get('_hip_f16', '') for n in net.G.nodes if net.G.nodes[n]['type'] == 'Convolution' and n not in ex._fused_away]
    assert sum('c8 module' in k for k in kinds) >= 9 * 4, kinds
    cat = next(n for n in ex._c8_concat)
    assert isinstance(next(iter(net.G.nodes[cat]['output'].values()))['data'], device.BlockedHalf)
    soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
    logits2 = np.asarray(next(iter(net.G.nodes[next(iter(net.G.pred[soft]))]['output'].values()))['data'])
    err_ref, err_1 = helpers.rel_err(logits2, z['logits']), helpers.rel_err(logits2, logits1)
    print('GoogLeNet FP16 IR, blocked modules: logits {:.2e} from the reference float16 run, {:.2e} from the default mode'.format(err_ref, err_1))
    assert np.isfinite(prob).all() and np.array_equal(logits2.argmax(axis=1), z['logits'].argmax(axis=1))
    assert err_ref <= 1e-2, err_ref
    ex_ref = helpers.elementwise_excess(logits2, z['logits'].astype(np.float32), 1e-2)      # element by element: 1e-2 |want| + 1e-2 rms(want)
    print('  worst element at {:.2f} of its 1e-2 allowance'.format(ex_ref))
    assert ex_ref <= 1.0, ex_ref
    assert err_1 <= 2e-3, err_1
    # a second pass (fresh Concat buffers) gives the same bits
    helpers.assert_bit_exact(helpers.infer_one(ex, net, images), prob, 'second pass')


def test_fp16_ir_the_blocked_kernels_do_not_cover_runs_anyway(hip, tmp_path, monkeypatch):
    """An FP16 IR the blocked-fp16 stem does not cover (GoogLeNet with LRN windows of three channels: neither MaxPool + LRN nor LRN + MaxPool
    has a blocked form): plan and plugins decide with the same predicates, so the stem stays on fp32 tensors, module 3a's input is
    converted once and the pass runs -- against the fp32 arithmetic of the same IR (5e-3: one fp16 rounding per tensor).  And the
    recovery branch of Convolution.compute: a member of a blocked Concat that is handed a DENSE input converts it and goes on."""
    from pyopenvino_amd import IECore, device, synth
    import test_host_logic
    images = np.concatenate([synth.uniform_pixels(500 + i, (1, 3, 224, 224)) for i in range(2)], 0)
    xml16, blob16 = test_host_logic._googlenet_fp16_ir_with_lrn_size(str(tmp_path), 3)
    logits = {}
    for as32 in (False, True):
        ie = IECore(plugin_package=HIP)
        net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=as32)
        net.set_batch(len(images))
        ex = ie.load_network(net)
        prob = helpers.infer_one(ex, net, images)
        assert np.isfinite(prob).all()
        soft = next(n for n in net.G.nodes if net.G.nodes[n]['type'] == 'SoftMax')
        logits[as32] = np.asarray(next(iter(net.G.nodes[next(iter(net.G.pred[soft]))]['output'].values()))['data'])
        if not as32:
            assert len(ex._c8_concat) == 9 and len(ex._c8_entry) == 1
            cat = next(iter(ex._c8_concat))
            assert isinstance(next(iter(net.G.nodes[cat]['output'].values()))['data'], device.BlockedHalf)
            # the recovery branch: the tensor module 3a reads, handed over dense although the plan says blocked
            entry = next(iter(ex._c8_entry))
            ex._c8_entry.clear()
            prob2 = helpers.infer_one(ex, net, images)
            helpers.assert_bit_exact(prob2, prob, 'dense input of a blocked module converted by the convolution itself')
            ex._c8_entry.add(entry)
    assert_close(logits[False], logits[True], 5e-3, 'f16 MFMA vs fp32 arithmetic, GoogLeNet FP16 IR with LRN windows of three', elementwise=False)


def test_googlenet_stem_convolution_rides_in_the_maxpool_lrn_launch(hip, monkeypatch):
    """pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce (+ bias + ReLU) as ONE launch (plan_fusion: `_stem_conv`; PVHIP_FUSE_STEM_CONV=0: the
    MaxPool + LRN launch followed by the pointwise convolution): the same bits end to end, one dispatched task fewer, and the folded
    chain's ports carry the launch's output."""
    from pyopenvino_amd import synth
    blob = synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), 1234)
    x = np.concatenate([synth.uniform_pixels(500 + i, (1, 3, 224, 224)) for i in range(3)], 0)
    out, tasks = {}, {}
    for mode in ('0', '1'):
        helpers.setenv(monkeypatch, 'PVHIP_FUSE_STEM_CONV', mode)
        _, net, ex = build_network(HIP, 'googlenet-v1', weights=blob, batch=3)
        out[mode] = infer_one(ex, net, x)
        tasks[mode] = len([t for t in ex.task_list if t not in ex._fused_away])
        if mode == '1':
            assert len(ex._stem_conv) == 1
            (pid, cid), = ex._stem_conv.items()
            G = net.G
            assert G.nodes[pid]['name'].startswith('pool1/3x3_s2') and G.nodes[cid]['name'].startswith('conv2/3x3_reduce')
            f = ex._fusion[cid]
            got = next(iter(G.nodes[f['relu']]['output'].values()))['data']
            assert tuple(got.shape) == (3, 64, 56, 56) and got is next(iter(G.nodes[pid]['output'].values()))['data']
        else:
            assert not ex._stem_conv
    assert tasks['1'] == tasks['0'] - 1
    helpers.assert_bit_exact(out['1'], out['0'], 'GoogLeNet with conv2/3x3_reduce inside the MaxPool + LRN launch')
    helpers.setenv(monkeypatch, 'PVHIP_FUSE_STEM_CONV', None)


def test_infer_replays_a_hipgraph_by_itself_for_device_resident_inputs(hip, monkeypatch):
    """infer() with inputs that live on the device dispatches eagerly AUTO_GRAPH_AFTER times, then records the pass into a hipGraph and
    replays it (one call instead of ~100 dispatches): the same bits every time; a host array, another stream plan or PVHIP_AUTO_GRAPH=0
    go (back) to eager dispatch (the recording reads the inputs where they lie and never writes into a caller's tensor), and node hooks (device_timing) are honoured by running eagerly."""
    from pyopenvino_amd import device, synth
    _, net, ex = build_network(HIP, 'googlenet-v1', weights=synth.synth_weights(os.path.join(helpers.MODELS, 'googlenet-v1.xml'), 1234), batch=4)
    x_host = np.concatenate([synth.uniform_pixels(500 + i, (1, 3, 224, 224)) for i in range(4)], 0)
    x = device.DeviceTensor.from_numpy(x_host)
    name, out = net.inputs[0]['name'], net.outputs[0]['name']
    first = np.asarray(ex.infer({name: x})[out])
    assert ex.__dict__.get('_graph') is None
    got = [np.asarray(ex.infer({name: x})[out]) for _ in range(4)]
    assert ex.__dict__.get('_graph') is not None and ex._auto_graph['captured']
    for g in got:
        helpers.assert_bit_exact(g, first, 'replayed pass')
    x2 = device.DeviceTensor.from_numpy(x_host[::-1].copy())                # another tensor: another recording, the first tensor is not touched
    helpers.assert_bit_exact(np.asarray(ex.infer({name: x2})[out]), first[::-1].copy(), 'another input tensor')
    assert np.array_equal(np.asarray(x), x_host) and not ex._auto_graph['captured']
    helpers.assert_bit_exact(np.asarray(ex.infer({name: x_host})[out]), first, 'host input: eager')
    ex.device_timing = {'Convolution'}
    helpers.assert_bit_exact(np.asarray(ex.infer({name: x})[out]), first, 'node hooks: eager')
    assert len(ex.device_times_ms()) > 0
    ex.device_timing = None
    ex.compute_streams = 1                                                 # another stream plan: the recording is dropped and made again
    for _ in range(4):
        helpers.assert_bit_exact(np.asarray(ex.infer({name: x})[out]), first, 'after a change of plan')
    assert ex._auto_graph['captured']
    monkeypatch.setenv('PVHIP_AUTO_GRAPH', '0')
    ex.release_graph()
    for _ in range(4):
        helpers.assert_bit_exact(np.asarray(ex.infer({name: x})[out]), first, 'PVHIP_AUTO_GRAPH=0')
    assert ex.__dict__.get('_graph') is None
    assert_close(first[:2], np.load(os.path.join(GOLDEN, 'googlenet_rows8.npz'))['out'][:2], helpers.REL_TOL, 'rows vs the reference')


def test_the_ssd_ir_and_the_fp16_googlenet_replay_too(hip, tmp_path):
    """infer() records on ONE stream (hipStreamEndCapture crashed for the forked plans of these two networks): the whole SSD IR --
    DetectionOutput with its workspace, the folded prior boxes -- and GoogLeNet as an FP16 IR on the f16 kernels replay with the bits
    of their eager passes."""
    from pyopenvino_amd import IECore, device, synth
    xml = os.path.join(helpers.MODELS, 'ssd_mobilenet_v1_coco.xml')
    _, net, ex = build_network(HIP, 'ssd_mobilenet_v1_coco', weights=synth.synth_weights(xml, 1234), batch=2)
    x = device.DeviceTensor.from_numpy(synth.uniform_pixels(700, (2, 3, 300, 300)))
    name, out = net.inputs[0]['name'], net.outputs[0]['name']
    first = np.asarray(ex.infer({name: x})[out])
    for _ in range(4):
        assert np.array_equal(np.asarray(ex.infer({name: x})[out]), first)
    assert ex.__dict__.get('_graph') is not None and ex._auto_graph['captured']
    gx = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    xml16, blob16 = synth.fp16_ir(gx, synth.synth_weights(gx, 1234), str(tmp_path))
    ie = IECore(plugin_package=HIP)
    net16 = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
    net16.set_batch(4)
    ex16 = ie.load_network(net16)
    x16 = device.DeviceTensor.from_numpy(synth.uniform_pixels(31, (4, 3, 224, 224)))
    n16, o16 = net16.inputs[0]['name'], net16.outputs[0]['name']
    first16 = np.asarray(ex16.infer({n16: x16})[o16])
    for _ in range(4):
        helpers.assert_bit_exact(np.asarray(ex16.infer({n16: x16})[o16]), first16, 'FP16 GoogLeNet replay')
    assert ex16._auto_graph['captured']
