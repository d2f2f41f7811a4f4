"""N > 1 path on CPU: two processes, each running the unchanged scheduler on its batch shard with the product Result
plugin gathering the shards (host path of BatchShardComm; on GPUs the same call goes through RCCL).  The host group is
the product's own `shard.TcpGroup` (plain sockets, no torch in the package) and, beside it, torch.distributed's gloo
backend behind the same HostGroup protocol (`GlooGroup` below: test-only, what rounds 1-3 shipped as TorchGroup).  The
per-node numerics here are the oracle plugins -- this test is about the shard / gather logic, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers


class GlooGroup:
    """The HostGroup protocol over torch.distributed (gloo): rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        dist.init_process_group(backend='gloo')
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def broadcast_bytes(self, data, src=0):
        box = [data if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def allgather_array(self, arr):
        parts = [None] * self.world
        self._dist.all_gather_object(parts, np.ascontiguousarray(arr))
        return parts

    def allreduce_max(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self._dist.destroy_process_group()


def _group(kind):
    from pyopenvino_amd import shard
    return GlooGroup() if kind == 'gloo' else shard.TcpGroup(timeout=120)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir, kind):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank)})
    sys.path.insert(0, helpers.REPO)
    import importlib
    from pyopenvino_amd import IECore, shard, synth
    group = _group(kind)
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
    assert group.broadcast_bytes(b'id' * 64 if rank == 0 else None, src=0) == b'id' * 64      # (how the RCCL unique id travels)
    assert [int(a[0]) for a in group.allgather_array(np.array([rank * 7]))] == [0, 7]
    group.barrier()
    np.save(os.path.join(out_dir, 'rank{}.npy'.format(rank)), out)
    assert t == float(world)
    group.close()


@pytest.mark.parametrize('kind,total', [('tcp', 6), ('tcp', 5), ('gloo', 6), ('gloo', 5)])
def test_two_rank_batch_shard_and_gather(tmp_path, kind, total):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path), kind), nprocs=world, join=True)
    from pyopenvino_amd import synth
    x = np.concatenate([synth.uniform_pixels(300 + i, (1, 1, 28, 28)) for i in range(total)], 0)
    _, net, ex = helpers.build_network('oracle.op_plugins', 'mnist', batch=total)
    want = helpers.infer_one(ex, net, x)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'rank{}.npy'.format(r)))
        assert got.shape == (total, 10)
        helpers.assert_close(got, want, 1e-6, 'rank {} gathered batch'.format(r))


def _parity_worker(rank, world, port, out_dir, kind):
    """One rank of bench.py's `parity_of_timed_path` branch on a stub device (the oracle plugins on the CPU, the product's Result plugin and
    gather): GoogLeNet on the seeded weights, this rank's shard = the first two golden images, the gathered Result checked by bench.py's own
    rows_vs_reference() -- the golden rows of BOTH ranks, at rank * batch in the gathered tensor."""
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank)})
    sys.path.insert(0, helpers.REPO)
    import importlib
    import json
    import bench
    from pyopenvino_amd import IECore, shard, synth
    group = _group(kind)
    comm = shard.BatchShardComm(group, use_rccl=False)
    batch = 2
    lo, hi = comm.shard(batch * world)
    assert hi - lo == batch
    golden = np.load(os.path.join(helpers.GOLDEN, 'googlenet_rows8.npz'))
    ie = IECore(plugin_package='oracle.op_plugins')
    ie.plugins.plugins['Result'] = importlib.import_module('pyopenvino_amd.op_plugins.Result')   # the product's gather
    xml = os.path.join(helpers.MODELS, 'googlenet-v1.xml')
    net = ie.read_network(xml, weights=synth.synth_weights(xml, int(golden['weight_seed'])))
    net.set_batch(batch)
    ex = ie.load_network(net, 'CPU')
    ex.comm = comm
    x = np.concatenate([synth.uniform_pixels(int(s_), (1, 3, 224, 224)) for s_ in golden['image_seeds'][:batch]], 0)
    if rank == 1:
        x = x.copy()                     # (the same golden images on every rank, as bench.py feeds them)
    gathered = ex.infer({net.inputs[0]['name']: x})[net.outputs[0]['name']]
    parity = {'checked_requests': 1, 'rows_vs_reference': 0, 'max_norm_error_vs_reference': None, 'worst_element_of_1e-4_allowance': None}
    bench.rows_vs_reference(gathered, golden['out'], batch, world, batch, parity)
    # ... and a gathered tensor in which ONE rank's rows are wrong must show: rank 1's shard swapped
    broken = np.asarray(gathered).copy()
    broken[batch:] = broken[batch:][::-1]
    bad = {'rows_vs_reference': 0, 'max_norm_error_vs_reference': None, 'worst_element_of_1e-4_allowance': None}
    bench.rows_vs_reference(broken, golden['out'], batch, world, batch, bad)
    group.barrier()
    with open(os.path.join(out_dir, 'parity{}.json'.format(rank)), 'w') as f:
        json.dump({'parity': parity, 'bad': bad, 'shape': list(np.asarray(gathered).shape)}, f)
    group.close()


@pytest.mark.parametrize('kind', ['tcp', 'gloo'])
def test_two_rank_parity_of_the_timed_path_sees_both_ranks_golden_rows(tmp_path, kind):
    """bench.py's check of the gathered Result against the reference's recorded answers, with two ranks (CPU stub device): the golden rows of
    BOTH shards are compared (rows_vs_reference = ranks x rows), they pass at 1e-4 in both norms, and a wrong shard is noticed."""
    import json
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_parity_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    for r in range(world):
        rec = json.load(open(os.path.join(str(tmp_path), 'parity{}.json'.format(r))))
        assert rec['shape'] == [4, 1000]
        p = rec['parity']
        assert p['rows_vs_reference'] == 4 and p['max_norm_error_vs_reference'] <= 1e-4 and p['worst_element_of_1e-4_allowance'] <= 1.0, p
        assert rec['bad']['rows_vs_reference'] == 4 and rec['bad']['worst_element_of_1e-4_allowance'] > 1.0, rec['bad']


def _fallback_worker(rank, world, port, out_dir, kind):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank)})
    os.environ.pop('PVHIP_NO_RCCL', None)
    sys.path.insert(0, helpers.REPO)
    from pyopenvino_amd import shard
    group = _group(kind)
    comm = shard.BatchShardComm(group)              # use_rccl: the communicator cannot be made here (no GPU)
    assert comm.use_rccl
    path, err = comm.agree_on_gather()
    part = np.full((2 + rank, 3), float(rank), dtype=np.float32)            # uneven shards through the path agreed on
    got = comm.allgather_rows(part)
    with open(os.path.join(out_dir, 'rank{}.txt'.format(rank)), 'w') as f:
        f.write('{}\n{}\n{}\n{}\n'.format(path, int(comm.use_rccl), got.shape[0], comm.rccl_ranks()))
    group.barrier()
    group.close()


@pytest.mark.parametrize('kind', ['tcp', 'gloo'])
def test_every_rank_agrees_on_the_host_gather_when_rccl_is_unavailable(tmp_path, kind):
    """bench.py's N > 1 branch without GPUs: no rank can create the communicator, all of them learn it through the host
    group, switch to the host gather together and say so; the gather then still returns every rank's rows."""
    from pyopenvino_amd import device
    if device.device_count() > 0:
        pytest.skip('a GPU is visible here: the communicator can be created')
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_fallback_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    for r in range(world):
        path, use_rccl, rows, ranks = open(os.path.join(str(tmp_path), 'rank{}.txt'.format(r))).read().split('\n')[:4]
        assert path.startswith('host group (RCCL communicator unavailable') and use_rccl == '0' and rows == '5' and ranks == '0'


def test_uneven_shards_are_padded_for_ncclallgather(monkeypatch):
    """The RCCL branch of allgather_rows with device calls faked on the host: ncclAllGather is given the SAME count by every
    rank (the longest shard), and the padding rows are cut out of the result, in rank order."""
    from pyopenvino_amd import device as dev, shard

    class Group:
        rank, world = 1, 3
        def allgather_array(self, arr):
            return [np.array([3]), np.array([2]), np.array([2])]      # rows of ranks 0, 1, 2 (7 images on 3 ranks)

    heap, calls = {}, []

    class FakeTensor:
        _next = [0x1000]
        def __init__(self, shape):
            self.shape = tuple(shape)
            self.ptr = FakeTensor._next[0]
            FakeTensor._next[0] += 0x100000
            heap[self.ptr] = np.zeros(int(np.prod(self.shape)), dtype=np.float32)
        size = property(lambda self: int(np.prod(self.shape)))
        nbytes = property(lambda self: self.size * 4)

    def find(ptr):
        base = max(b for b in heap if b <= ptr)
        return heap[base], (ptr - base) // 4

    def fake_call(name, *args):
        val = [a.value if hasattr(a, 'value') else a for a in args]
        calls.append((name, val))
        if name == 'pvhip_memset':
            buf, off = find(val[0]); buf[off:off + val[2] // 4] = 0
        elif name == 'pvhip_memcpy_d2d':
            dst, do = find(val[0]); src, so = find(val[1]); n = val[2] // 4
            dst[do:do + n] = src[so:so + n]
        elif name == 'pvhip_comm_allgather_f32':
            src, so = find(val[0]); dst, do = find(val[1]); n = val[2]
            assert n == 3 * 4                                          # the longest shard: 3 rows of 4
            for r in range(3):                                         # what the three ranks would send
                rows = [3, 2, 2][r]
                block = np.zeros(n, dtype=np.float32)
                block[:rows * 4] = 100 * r + np.arange(rows * 4)
                if r == 1:
                    assert np.array_equal(src[so:so + n], block)        # this rank's padded shard
                dst[do + r * n:do + (r + 1) * n] = block
        return 0

    monkeypatch.setattr(dev, 'call', fake_call)
    monkeypatch.setattr(dev, 'ensure_init', lambda: None)
    monkeypatch.setattr(dev.DeviceTensor, 'empty', classmethod(lambda cls, shape, dtype=np.float32: FakeTensor(shape)))
    comm = shard.BatchShardComm(Group())
    comm._rccl_ready = True
    mine = FakeTensor((2, 4))
    heap[mine.ptr][:] = 100 + np.arange(8)
    monkeypatch.setattr(shard.dev, 'DeviceTensor', type('DT', (), {'empty': staticmethod(lambda shape, dtype=np.float32: FakeTensor(shape))}))
    # isinstance(value, dev.DeviceTensor) must hold for the fake
    shard.dev.DeviceTensor = FakeTensor
    FakeTensor.empty = staticmethod(lambda shape, dtype=np.float32: FakeTensor(shape))
    out = comm.allgather_rows(mine)
    assert out.shape == (7, 4)
    want = np.concatenate([100 * r + np.arange([3, 2, 2][r] * 4) for r in range(3)]).astype(np.float32)
    assert np.array_equal(heap[out.ptr][:28], want)
    assert [c[0] for c in calls].count('pvhip_comm_allgather_f32') == 1


def test_row_counts_never_depend_on_a_rank_local_cache():
    """World 2, a batch of 7 and then of 8 on the same communicator: rank 0 keeps 4 rows while rank 1 goes from 3 to 4.  The
    counts come from the total every rank passed to shard() -- no rank enters a host collective the other one skips -- and a
    communicator that was never told the total exchanges them every single time."""
    from pyopenvino_amd import shard

    class Group:
        def __init__(self, rank):
            self.rank, self.world, self.exchanges = rank, 2, 0
        def allgather_array(self, arr):
            self.exchanges += 1
            return [np.array([4]), np.array([int(np.asarray(arr).reshape(-1)[0])])]

    for rank in (0, 1):
        g = Group(rank)
        comm = shard.BatchShardComm(g, use_rccl=False)
        for total, want in ((7, [4, 3]), (8, [4, 4]), (7, [4, 3])):
            lo, hi = comm.shard(total)
            assert comm.row_counts(hi - lo) == want
        assert g.exchanges == 0
        with pytest.raises(ValueError):
            comm.row_counts(hi - lo + 1)          # a tensor that is not this rank's shard: loud, not a hang
    g = Group(1)
    comm = shard.BatchShardComm(g, use_rccl=False)
    assert comm.row_counts(3) == [4, 3] and comm.row_counts(3) == [4, 3] and comm.row_counts(4) == [4, 4]
    assert g.exchanges == 3


def test_tcp_group_survives_a_stale_rendezvous_file(tmp_path):
    """A rendezvous file left behind by an earlier run (dead port, other token) does not mislead rank 1: it keeps reading the file
    until this run's rank 0 has published itself.  Three ranks as threads of one process."""
    import threading
    from pyopenvino_amd import shard
    rdv = str(tmp_path / 'rdv')
    with open(rdv, 'w') as f:
        f.write('{} {}\n'.format(_free_port(), 'a' * 32))
    out, errs = {}, []

    def run(rank, delay):
        try:
            import time
            time.sleep(delay)
            g = shard.TcpGroup(rank=rank, world=3, timeout=60, rdv_file=rdv)
            out[rank] = (g.allreduce_max(float(rank)), [int(a[0]) for a in g.allgather_array(np.array([10 + rank]))],
                         g.broadcast_bytes(b'x' if rank == 2 else None, src=2))
            g.barrier()
            g.close()
        except Exception as exc:      # noqa: BLE001
            errs.append((rank, exc))

    threads = [threading.Thread(target=run, args=(r, 0.3 if r == 0 else 0.0)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(90)
    assert not errs, errs
    assert all(out[r] == (2.0, [10, 11, 12], b'x') for r in range(3)), out
    assert not os.path.exists(rdv)


def test_host_group_frames_carry_no_pickle_and_refuse_what_they_do_not_know():
    """The wire format of TcpGroup (a peer's bytes must never be unpickled): every kind the collectives exchange survives
    a round trip; object arrays are refused on both sides; malformed, truncated or foreign frames raise ValueError and nothing else."""
    import pickle
    from pyopenvino_amd import shard
    src = open(shard.__file__.replace('.pyc', '.py')).read()
    assert 'import pickle' not in src and 'pickle.loads' not in src
    cases = [None, b'\x00\x01raw' * 40, 'error: text \u00e9', 3.5, np.arange(12, dtype=np.int64).reshape(3, 4),
             np.zeros((0, 1000), dtype=np.float32), [np.float32(1.5) * np.ones((2, 3), dtype=np.float32), None, b'id', 2.0]]
    for obj in cases:
        got, end = shard._decode(memoryview(shard._encode(obj)))
        assert end == len(shard._encode(obj))
        if isinstance(obj, np.ndarray):
            assert got.dtype == obj.dtype and got.shape == obj.shape and np.array_equal(got, obj)
        elif isinstance(obj, list):
            assert len(got) == len(obj) and np.array_equal(got[0], obj[0]) and got[1] is None and got[2] == b'id' and got[3] == 2.0
        else:
            assert got == obj and type(got) is type(obj)
    with pytest.raises(TypeError):
        shard._encode(np.array([object()], dtype=object))
    with pytest.raises(TypeError):
        shard._encode({'rank': 1})
    hostile = [pickle.dumps({'rank': 1}), b'PVHG' + bytes([9]) + (0).to_bytes(8, 'little'), shard._encode(b'abc')[:-1],
               shard._HEADER.pack(b'PVHG', shard._K_ARRAY, 40) + (30).to_bytes(4, 'little') + b'{"dtype": "|O", "shape": [1]}    ' + b'\x00' * 6,
               shard._HEADER.pack(b'PVHG', shard._K_LIST, 4) + (5).to_bytes(4, 'little')]
    for blob in hostile:
        with pytest.raises(ValueError):
            shard._decode(memoryview(blob))


def test_rendezvous_file_is_private_and_never_followed_through_a_link(tmp_path):
    """rank 0 creates its rendezvous file O_EXCL | O_NOFOLLOW with mode 0600 inside a 0700 directory of its own; a rank refuses a file
    that is a symbolic link (what another user could plant under a predictable name)."""
    import stat
    from pyopenvino_amd import shard
    d = shard._rendezvous_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o077)
    victim = tmp_path / 'victim'
    victim.write_text('untouched')
    link = tmp_path / 'rdv'
    os.symlink(str(victim), str(link))
    shard._write_private(str(link), '1234 ' + 'b' * 32)            # the link is replaced, not followed
    assert victim.read_text() == 'untouched' and not os.path.islink(str(link))
    assert stat.S_IMODE(os.lstat(str(link)).st_mode) == 0o600 and shard._read_private(str(link)).split()[0] == '1234'
    os.unlink(str(link))
    os.symlink(str(victim), str(link))
    with pytest.raises(OSError):
        shard._read_private(str(link))


def test_tcp_group_ignores_a_stranger_on_its_port(tmp_path):
    """A client that connects to rank 0's port without this run's token (or with garbage) is dropped; the ranks of the run still meet."""
    import socket
    import threading
    import time
    from pyopenvino_amd import shard
    rdv = str(tmp_path / 'rdv')
    out, errs = {}, []

    def run(rank, delay):
        try:
            time.sleep(delay)
            g = shard.TcpGroup(rank=rank, world=2, timeout=60, rdv_file=rdv)
            out[rank] = g.allreduce_max(float(rank))
            g.barrier()
            g.close()
        except Exception as exc:      # noqa: BLE001
            errs.append((rank, exc))

    t0 = threading.Thread(target=run, args=(0, 0.0))
    t0.start()
    deadline = time.time() + 30
    while not os.path.exists(rdv) and time.time() < deadline:
        time.sleep(0.02)
    port = int(open(rdv).read().split()[0])
    for junk in (b'', b'\x80\x04\x95' + b'x' * 60, shard._HELLO.pack(b'PVHG', 1, 2, b'z' * 32)):
        s = socket.create_connection(('127.0.0.1', port), timeout=5)
        s.sendall(junk)
        s.close()
    t1 = threading.Thread(target=run, args=(1, 0.0))
    t1.start()
    for t in (t0, t1):
        t.join(90)
    assert not errs, errs
    assert out == {0: 1.0, 1: 1.0}


def test_launch_ranks_exports_the_launcher_contract_and_reports_failures(tmp_path):
    """shard.launch_ranks (what `bench.py --gpus N` uses when it was not started by torch.distributed.run): every child sees RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*; a failing rank makes the launch fail and the others are ended."""
    from pyopenvino_amd import shard
    script = tmp_path / 'child.py'
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, {!r})\n"
        "from pyopenvino_amd import shard\n"
        "g = shard.TcpGroup(timeout=60)\n"
        "assert g.world == int(os.environ['WORLD_SIZE']) == 2 and g.rank == int(os.environ['LOCAL_RANK'])\n"
        "assert g.allreduce_max(float(g.rank)) == 1.0\n"
        "open(os.path.join({!r}, 'ok%d' % g.rank), 'w').close()\n"
        "g.barrier(); g.close()\n"
        "if len(sys.argv) > 1 and g.rank == 1: sys.exit(3)\n"
        "if len(sys.argv) > 1: time.sleep(30)\n".format(helpers.REPO, str(tmp_path)))
    assert shard.launch_ranks([sys.executable, str(script)], 2) == 0
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('ok')) == ['ok0', 'ok1']
    import time
    t0 = time.time()
    assert shard.launch_ranks([sys.executable, str(script), 'fail'], 2) == 3
    assert time.time() - t0 < 25            # rank 0 did not sleep its 30 s out
