#!/usr/bin/env python3
"""GPU-box tool: what does each kernel family COST the pipelined step?  bench.py's headline runs 8 whole-batch requests in flight, where
memory-bound launches hide behind matrix-core-bound ones of other requests: a launch's own duration (per-layer pass, one stream) says
little about what removing or shrinking it would buy.  This script measures it directly: the same pipelined loop as bench.py with the
launches of a family SKIPPED (the node returns its output tensor of the warm-up pass: wrong on purpose, harness only -- the plugins are
wrapped here, nothing in the product changes), and reports ms/step against the full pass.

    python scripts/marginal_cost.py [--steps 40] [--requests 8]
"""
import argparse, os, re, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import IECore, device, synth  # noqa: E402

FAMILIES = [
    ('nothing skipped', None, None),
    ('conv1', 'Convolution', r'^conv1/'),
    ('conv2/3x3 (Winograd F(4x4,3x3), 0.59 ms alone)', 'Convolution', r'^conv2/3x3/'),
    ('inception 3x3 layers (Winograd F(4x4,3x3))', 'Convolution', r'^inception_.*/3x3/'),
    ('inception 5x5 layers (Winograd F(2x2,5x5))', 'Convolution', r'^inception_.*/5x5/'),
    ('sibling 1x1 launches', 'Convolution', r'^inception_.*/1x1/'),
    ('MaxPool + pool_proj launches (3a..4e)', 'Convolution', r'^inception_[34].*/pool_proj/'),
    ('pool_proj 5a/5b + conv2/3x3_reduce (pointwise singles)', 'Convolution', r'^(inception_5.*/pool_proj|conv2/3x3_reduce)/'),
    ('stand-alone MaxPool launches', 'MaxPool', r'^(pool[34]/|inception_5./pool$)'),
    ('MaxPool+LRN, LRN+MaxPool, data/mean', None, r'^(pool1/3x3_s2|conv2/norm2|data/mean)'),
]


class Shim:
    """A plugin module stand-in: compute() of nodes whose name matches is skipped (their previous output is handed on)."""
    def __init__(self, module):
        self.module, self.pattern = module, None
        self.__package__ = module.__package__
        self.__name__ = module.__name__
        for k in dir(module):
            if not k.startswith('__') and k != 'compute':
                setattr(self, k, getattr(module, k))

    def compute(self, node, inputs=None, kernel_type='hip', debug=False):
        if self.pattern is not None and self.pattern.search(node['name']):
            port = next(iter(node['output']))
            prev = node['output'][port].get('data')
            if prev is not None:
                if node.get('_siblings'):                     # the members of a sibling launch keep theirs too
                    node['_sibling_out'] = [next(iter(sib['node']['output'].values()))['data'] for sib in node['_siblings']]
                return {port: prev}
        return self.module.compute(node, inputs, kernel_type=kernel_type, debug=debug)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--requests', type=int, default=8)
    ap.add_argument('--batch', type=int, default=256)
    args = ap.parse_args()
    device.init(0)
    xml = os.path.join(REPO, 'models', 'googlenet-v1.xml')
    ie = IECore()
    shims = {}
    for typ in ('Convolution', 'MaxPool', 'LRN', 'Add'):
        shims[typ] = ie.plugins.plugins[typ] = Shim(ie.plugins.plugins[typ])
    net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234))
    net.set_batch(args.batch)
    ex = ie.load_network(net, 'GPU', num_requests=args.requests)
    xs = [device.DeviceTensor.from_numpy(synth.uniform_pixels(1000 + 100 * r, (args.batch, 3, 224, 224))) for r in range(args.requests)]
    in_name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    for req in ex.requests:
        for _ in range(3):
            req.infer({in_name: xs[req.index]})

    def run(steps):
        in_flight = []
        for step in range(steps):
            r = step % args.requests
            if r in in_flight:
                in_flight.remove(r)
                ex.wait(r)
            ex.start_async(r, {in_name: xs[r]})
            in_flight.append(r)
        while in_flight:
            ex.wait(in_flight.pop(0))

    base = None
    for name, typ, pat in FAMILIES:
        for s in shims.values():
            s.pattern = None
        if pat is not None:
            for t, s in shims.items():
                if typ is None or t == typ:
                    s.pattern = re.compile(pat)
        run(args.requests)
        best = 1e9
        for _ in range(3):
            device.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            device.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        if base is None:
            base = best
        print('{:62s} {:7.3f} ms/step  {:+7.3f} ms  ({:+5.1f} %)'.format(name, best, best - base, 100.0 * (best - base) / base), flush=True)


if __name__ == '__main__':
    main()
