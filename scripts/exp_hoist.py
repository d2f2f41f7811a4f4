#!/usr/bin/env python3
"""GPU-box experiment: does MaxPool + pool_proj get cheaper when it is launched RIGHT BEHIND its module's sibling 1x1 launch (the module
input, 38-205 MB, may still sit in L2 / the 256 MB Infinity Cache) instead of behind the 3x3 and 5x5 arms?  Same kernels, another legal
order of the list schedule.  Per-launch device times on one stream and the replayed single-request pass, both orders, alternating."""
import os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from pyopenvino_amd import IECore, device, synth

device.init(0)
xml = os.path.join(REPO, 'models', 'googlenet-v1.xml')
ie = IECore()
net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
net.set_batch(256)
ex = ie.load_network(net)
G = net.G
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1000, (256, 3, 224, 224)))
feed = {net.inputs[0]['name']: x}
base_order = list(ex.task_list)


def hoisted():
    order = list(base_order)
    for conv_id, (pool_id, src_id) in ex._pool_conv.items():
        # the sibling lead that reads the same tensor
        lead = next((l for l in ex._siblings if any(G.nodes[p]['type'] != 'Const' and (p == src_id or src_id in (ex._fusion.get(q, {}).get('relu'),))
                                                    for p in G.pred[l] for q in [p])), None)
        if lead is None:
            lead = next(l for l in ex._siblings if src_id in G.pred[l])
        # move the pool_proj chain (its Consts first: they are scheduled just before it) right behind the lead
        chain = [t for t in order if t == conv_id or (t in G.pred[conv_id] and G.nodes[t]['type'] == 'Const')]
        f = ex._fusion.get(conv_id)
        if f:
            chain += [t for t in order if t in (f['bias'],)]
        chain = [t for t in order if t in set(chain)]
        for t in chain:
            order.remove(t)
        at = order.index(lead) + 1
        order[at:at] = chain
    return order


def per_launch(order):
    ex.task_list = order
    ex._stream_plans = {}
    ex.release_graph()
    os.environ['PVHIP_AUTO_GRAPH'] = '0'
    ex.compute_streams = 1
    for _ in range(2):
        ex.infer(feed)
    ex.device_timing = {'Convolution'}
    acc = {}
    for _ in range(5):
        ex.infer(feed)
        for nid, typ, name, ms in ex.device_times_ms():
            acc.setdefault(name, []).append(ms)
    ex.device_timing = None
    os.environ.pop('PVHIP_AUTO_GRAPH')
    pool = sum(statistics.median(v) for k, v in acc.items() if '/pool_proj/' in k and k.startswith(('inception_3', 'inception_4')))
    total = sum(statistics.median(v) for v in acc.values())
    times = []
    for _ in range(4):
        ex.infer(feed)
    for _ in range(15):
        device.synchronize()
        t0 = time.perf_counter()
        ex.infer(feed)
        times.append(time.perf_counter() - t0)
    return pool, total, statistics.median(times) * 1e3, {k: round(statistics.median(v), 4) for k, v in acc.items() if '/pool_proj/' in k}


def swapped(order):
    """the 5x5 arm in front of the 3x3 arm (the largest arm of the module's output written last)"""
    order = list(order)
    by_name = {G.nodes[n]['name']: n for n in G.nodes}
    for mod in ('3a', '3b', '4a', '4b', '4c', '4d', '4e', '5a', '5b'):
        c3, c5 = by_name['inception_{}/3x3/WithoutBiases'.format(mod)], by_name['inception_{}/5x5/WithoutBiases'.format(mod)]
        chain5 = [t for t in order if t == c5 or (t in G.pred[c5] and G.nodes[t]['type'] == 'Const') or t == ex._fusion[c5]['bias']]
        for t in chain5:
            order.remove(t)
        first3 = min(order.index(t) for t in order if t == c3 or (t in G.pred[c3] and G.nodes[t]['type'] == 'Const') or t == ex._fusion[c3]['bias'])
        order[first3:first3] = chain5
    return order


want = np.array(ex.infer(feed)[net.outputs[0]['name']], copy=True)
ho = hoisted()
hs = swapped(ho)
assert sorted(ho) == sorted(base_order) == sorted(hs)
for rnd in range(3):
    for tag, order in (('list order', base_order), ('pool_proj hoisted', ho), ('hoisted + 5x5 first', hs)):
        pool, total, ms, detail = per_launch(order)
        got = np.asarray(ex.infer(feed)[net.outputs[0]['name']])
        print('{:18s} pool_proj 3a-4e {:.4f} ms  all convolutions {:.4f} ms  replayed infer() {:.3f} ms  same bits {}'.format(
            tag, pool, total, ms, bool(np.array_equal(got, want))), flush=True)
        if rnd == 0:
            print('   ', detail, flush=True)
