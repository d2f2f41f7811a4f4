import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import helpers
z = np.load('tests/golden/mnist_e2e.npz')
for fuse in (False, True):
    _, net, ex = helpers.build_network('pyopenvino_amd.op_plugins', 'mnist', fuse=fuse)
    _, onet, oex = helpers.build_network('oracle.op_plugins', 'mnist')
    x = z['images'][0:1]
    got = helpers.infer_one(ex, net, x); want = helpers.infer_one(oex, onet, x)
    print('fuse', fuse, 'final err', helpers.rel_err(got, want))
    for nid in net.G.nodes:
        node = net.G.nodes[nid]
        if node['type'] in ('Const', 'Parameter', 'Result'): continue
        for port, p in node['output'].items():
            if 'data' not in p: continue
            g = np.asarray(p['data']); w = np.asarray(onet.G.nodes[nid]['output'][port]['data'])
            if g.shape != w.shape: print(nid, node['type'], 'shape', g.shape, w.shape); continue
            print(nid, node['type'], g.shape, '%.3e' % helpers.rel_err(g, w))
