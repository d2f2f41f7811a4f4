"""N > 1 path on CPU: two processes, torch.distributed (gloo), each running the unchanged scheduler on its
batch shard with the product Result plugin gathering the shards (host path of BatchShardComm; on GPUs the same
call goes through RCCL).  The per-node numerics here are the oracle plugins -- this test is about the shard /
gather logic, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank)})
    sys.path.insert(0, helpers.REPO)
    import importlib
    from pyopenvino_amd import IECore, shard, synth
    group = shard.TorchGroup('gloo')
    comm = shard.BatchShardComm(group, use_rccl=False)
    lo, hi = comm.shard(total)
    ie = IECore(plugin_package='oracle.op_plugins')
    ie.plugins.plugins['Result'] = importlib.import_module('pyopenvino_amd.op_plugins.Result')   # the product's gather
    net = ie.read_network(os.path.join(helpers.MODELS, 'mnist.xml'))
    net.set_batch(hi - lo)
    ex = ie.load_network(net, 'CPU', num_requests=2)
    ex.comm = comm
    x = np.concatenate([synth.uniform_pixels(300 + i, (1, 1, 28, 28)) for i in range(total)], 0)
    out = ex.infer({net.inputs[0]['name']: x[lo:hi]})[net.outputs[0]['name']]
    # the same through two requests in flight: the shards are gathered in wait(), in the same order on every rank
    for r in (0, 1):
        ex.start_async(r, {net.inputs[0]['name']: x[lo:hi]})
    for r in (1, 0):
        assert np.array_equal(ex.wait(r)[net.outputs[0]['name']], out)
    t = group.allreduce_max(float(rank + 1))
    group.barrier()
    np.save(os.path.join(out_dir, 'rank{}.npy'.format(rank)), out)
    assert t == float(world)
    group.close()


@pytest.mark.parametrize('total', [6, 5])
def test_two_rank_batch_shard_and_gather(tmp_path, total):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    from pyopenvino_amd import synth
    x = np.concatenate([synth.uniform_pixels(300 + i, (1, 1, 28, 28)) for i in range(total)], 0)
    _, net, ex = helpers.build_network('oracle.op_plugins', 'mnist', batch=total)
    want = helpers.infer_one(ex, net, x)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'rank{}.npy'.format(r)))
        assert got.shape == (total, 10)
        helpers.assert_close(got, want, 1e-6, 'rank {} gathered batch'.format(r))
