"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective except
an all-gather of the Result tensors (RCCL over xGMI, through the C ABI's pvhip_comm_*).

The reference has no counterpart (it is single-process, N=1); images of a batch are independent through
the whole graph, so each rank runs the unchanged scheduler on a contiguous slice of the batch.

Host-side coordination (rendezvous, barrier, exchanging RCCL's unique id, max-over-ranks timing) goes
through a small HostGroup protocol; TcpGroup implements it over plain sockets from the environment the
launcher exports (``python -m torch.distributed.run``, or ``bench.py --gpus N`` by itself through
``launch_ranks``).  Nothing here imports torch.
"""
import ctypes
import json
import os
import secrets
import socket
import stat
import struct
import tempfile
import time

import numpy as np

from . import device as dev


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `total` rows owned by `rank`; the first total % world ranks get one
    extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad rank/world {}/{}'.format(rank, world))
    base, extra = divmod(int(total), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class SingleGroup:
    """world == 1: every collective is the identity."""
    rank, world = 0, 1

    def barrier(self):
        pass

    def broadcast_bytes(self, data, src=0):
        return data

    def allgather_array(self, arr):
        return [np.asarray(arr)]

    def allreduce_max(self, value: float) -> float:
        return float(value)


# ---- wire format of the host group: NO pickle (a peer's bytes are never executed).  A frame is a fixed 13-byte header
# (magic, kind, payload length) and a payload of one of six kinds; arrays travel as a JSON header (dtype string of a plain
# numeric type, shape) followed by the raw bytes.  What the collectives below exchange is None, bytes (RCCL's unique id), str
# (an error text), float (timings), numeric ndarrays (row counts, host-gathered Result tensors) and flat lists of those.
_MAGIC = b'PVHG'
_HEADER = struct.Struct('<4sBQ')
_K_NONE, _K_BYTES, _K_STR, _K_FLOAT, _K_ARRAY, _K_LIST = range(6)
_MAX_FRAME = 1 << 34
_HELLO = struct.Struct('<4sII32s')          # magic, rank, world, token: fixed size, checked BEFORE anything else is parsed
_HELLO_REPLY = struct.Struct('<4sBI')       # magic, ok, world


def _encode(obj) -> bytes:
    if obj is None:
        return _HEADER.pack(_MAGIC, _K_NONE, 0)
    if isinstance(obj, (bytes, bytearray, memoryview)):
        raw = bytes(obj)
        return _HEADER.pack(_MAGIC, _K_BYTES, len(raw)) + raw
    if isinstance(obj, str):
        raw = obj.encode('utf-8')
        return _HEADER.pack(_MAGIC, _K_STR, len(raw)) + raw
    if isinstance(obj, (bool, int, float, np.floating, np.integer)):
        return _HEADER.pack(_MAGIC, _K_FLOAT, 8) + struct.pack('<d', float(obj))
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind not in 'biuf' or obj.dtype.hasobject:
            raise TypeError('host group: only plain numeric arrays travel, not dtype {}'.format(obj.dtype))
        arr = np.ascontiguousarray(obj)
        head = json.dumps({'dtype': arr.dtype.str, 'shape': list(arr.shape)}).encode('ascii')
        raw = arr.tobytes()
        return _HEADER.pack(_MAGIC, _K_ARRAY, 4 + len(head) + len(raw)) + struct.pack('<I', len(head)) + head + raw
    if isinstance(obj, (list, tuple)):
        body = struct.pack('<I', len(obj)) + b''.join(_encode(item) for item in obj)
        return _HEADER.pack(_MAGIC, _K_LIST, len(body)) + body
    raise TypeError('host group: cannot send a {}'.format(type(obj).__name__))


def _decode(buf, at=0):
    """-> (object, offset behind it); ValueError on anything that is not a well-formed frame."""
    if len(buf) - at < _HEADER.size:
        raise ValueError('host group: truncated frame')
    magic, kind, n = _HEADER.unpack_from(buf, at)
    at += _HEADER.size
    if magic != _MAGIC or n > len(buf) - at:
        raise ValueError('host group: malformed frame')
    body, end = buf[at:at + n], at + n
    if kind == _K_NONE and n == 0:
        return None, end
    if kind == _K_BYTES:
        return bytes(body), end
    if kind == _K_STR:
        return bytes(body).decode('utf-8'), end
    if kind == _K_FLOAT and n == 8:
        return struct.unpack('<d', body)[0], end
    if kind == _K_ARRAY and n >= 4:
        (hn,) = struct.unpack_from('<I', body, 0)
        if hn > n - 4:
            raise ValueError('host group: malformed array header')
        head = json.loads(bytes(body[4:4 + hn]).decode('ascii'))
        dtype = np.dtype(str(head['dtype']))
        shape = tuple(int(d) for d in head['shape'])
        if dtype.kind not in 'biuf' or dtype.hasobject or any(d < 0 for d in shape):
            raise ValueError('host group: refused array of dtype {}'.format(dtype))
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if count * dtype.itemsize != n - 4 - hn:
            raise ValueError('host group: array payload does not match its header')
        return np.frombuffer(bytes(body[4 + hn:]), dtype=dtype).reshape(shape).copy(), end
    if kind == _K_LIST and n >= 4:
        (count,) = struct.unpack_from('<I', body, 0)
        items, pos = [], 4
        for _ in range(count):
            item, pos = _decode(body, pos)
            items.append(item)
        if pos != n:
            raise ValueError('host group: malformed list')
        return items, end
    raise ValueError('host group: unknown frame kind {}'.format(kind))


def _send_msg(sock, obj):
    sock.sendall(_encode(obj))


def _recv_exact(sock, n):
    parts, got = [], 0
    while got < n:
        piece = sock.recv(min(n - got, 1 << 20))
        if not piece:
            raise ConnectionError('host group: peer closed the connection')
        parts.append(piece)
        got += len(piece)
    return b''.join(parts)


def _recv_msg(sock):
    head = _recv_exact(sock, _HEADER.size)
    magic, _, n = _HEADER.unpack(head)
    if magic != _MAGIC or n > _MAX_FRAME:
        raise ValueError('host group: malformed frame')
    obj, _ = _decode(memoryview(head + _recv_exact(sock, n)))
    return obj


def _rendezvous_dir():
    """A directory only this user can enter (0700, owned by us, not a symlink): the rendezvous file of a run lives in it, so no other
    user can plant a file or a link under the name the ranks will read."""
    path = os.path.join(tempfile.gettempdir(), 'pvhip_rdv_{}'.format(os.getuid()))
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError('host group: {} is not a private directory of uid {}'.format(path, os.getuid()))
    return path


def _write_private(path, text):
    """Create `path` anew (never through a link, never over a file somebody else made), mode 0600."""
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    with os.fdopen(fd, 'w') as f:
        f.write(text)


def _read_private(path):
    fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
    with os.fdopen(fd) as f:
        st = os.fstat(f.fileno())
        if st.st_uid != os.getuid() or not stat.S_ISREG(st.st_mode):
            raise PermissionError('host group: {} was not written by uid {}'.format(path, os.getuid()))
        return f.read(4096)


class TcpGroup:
    """HostGroup over plain TCP sockets on one node (a star: rank 0 serves, ranks 1.. connect): rendezvous, barrier, the RCCL unique
    id, row counts and max-over-ranks timing -- a few hundred bytes per call, never tensors of the data path on a GPU run.  No torch,
    no pickle: frames of a fixed header and raw bytes (`_encode` / `_decode`), so nothing a peer sends is ever executed.

    Reads RANK / WORLD_SIZE / MASTER_PORT from the environment, which is what `python -m torch.distributed.run` and `bench.py --gpus N`
    (its own launcher) both export.  The design is single-node: rank 0 ALWAYS listens on 127.0.0.1 (whatever MASTER_ADDR says), on a
    port the system picks -- MASTER_PORT itself belongs to the launcher (torchrun keeps its store there) -- and publishes `port token`
    in a rendezvous file named after MASTER_PORT and the launcher's pid inside a directory only this user can enter (created 0700,
    the file O_EXCL | O_NOFOLLOW, 0600; PVHIP_RDV_FILE overrides the name).  A connecting rank sends a fixed-size hello whose token
    is compared before anything else is parsed.  A stale file of an earlier run is harmless: a rank that cannot connect, or whose hello
    is not answered with this run's world size, reads the file again."""

    def __init__(self, rank=None, world=None, timeout=600.0, rdv_file=None):
        self.rank = int(os.environ['RANK'] if rank is None else rank)
        self.world = int(os.environ['WORLD_SIZE'] if world is None else world)
        if not 0 <= self.rank < self.world:
            raise ValueError('bad rank/world {}/{}'.format(self.rank, self.world))
        self.timeout = float(timeout)
        port = os.environ.get('MASTER_PORT', '29500')
        self._peers, self._sock, self._server = {}, None, None
        if self.world == 1:
            self._rdv = None
            return
        self._rdv = rdv_file or os.environ.get('PVHIP_RDV_FILE') or os.path.join(
            _rendezvous_dir(), 'rdv_{}_{}'.format(port, os.getppid()))
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.world)
            srv.settimeout(self.timeout)
            self._server = srv
            token = secrets.token_hex(16).encode('ascii')          # 32 bytes
            tmp = '{}.{}'.format(self._rdv, os.getpid())
            _write_private(tmp, '{} {}\n'.format(srv.getsockname()[1], token.decode('ascii')))
            os.replace(tmp, self._rdv)
            while len(self._peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    magic, rank_, world_, token_ = _HELLO.unpack(_recv_exact(conn, _HELLO.size))
                except (ConnectionError, OSError, struct.error):
                    conn.close()
                    continue
                if magic != _MAGIC or not secrets.compare_digest(token_, token):
                    conn.close()                   # not one of ours: nothing else of what it sent is looked at
                    continue
                ok = world_ == self.world and 0 < rank_ < self.world and rank_ not in self._peers
                conn.sendall(_HELLO_REPLY.pack(_MAGIC, 1 if ok else 0, self.world))
                if ok:
                    self._peers[rank_] = conn
                else:
                    conn.close()
        else:
            deadline = time.monotonic() + self.timeout
            while True:
                sock = None
                try:
                    rport, token = _read_private(self._rdv).split()
                    token = token.encode('ascii')
                    if len(token) != 32:
                        raise ValueError('host group: malformed rendezvous file')
                    sock = socket.create_connection(('127.0.0.1', int(rport)), timeout=5.0)
                    sock.settimeout(self.timeout)
                    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    sock.sendall(_HELLO.pack(_MAGIC, self.rank, self.world, token))
                    magic, ok, world_ = _HELLO_REPLY.unpack(_recv_exact(sock, _HELLO_REPLY.size))
                    if magic == _MAGIC and ok == 1 and world_ == self.world:
                        self._sock = sock
                        break
                    sock.close()
                except PermissionError:
                    raise                          # a rendezvous file or directory that is not ours: never retried, never read
                except (OSError, ValueError, ConnectionError, struct.error):
                    if sock is not None:
                        sock.close()
                if time.monotonic() > deadline:
                    raise TimeoutError('host group: rank {} found no rank 0 behind {} in {:.0f} s'.format(self.rank, self._rdv, self.timeout))
                time.sleep(0.05)

    # every collective: ranks 1.. send their part to rank 0, which answers each of them with the combined result
    def _exchange(self, part, combine):
        if self.world == 1:
            return combine([part])
        if self.rank == 0:
            parts = [part] + [None] * (self.world - 1)
            for r, conn in self._peers.items():
                parts[r] = _recv_msg(conn)
            result = combine(parts)
            for conn in self._peers.values():
                _send_msg(conn, result)
            return result
        _send_msg(self._sock, part)
        return _recv_msg(self._sock)

    def barrier(self):
        self._exchange(None, lambda parts: None)

    def broadcast_bytes(self, data, src=0):
        return self._exchange(data if self.rank == src else None, lambda parts: parts[src])

    def allgather_array(self, arr):
        return self._exchange(np.ascontiguousarray(arr), list)

    def allreduce_max(self, value: float) -> float:
        return float(self._exchange(float(value), max))

    def close(self):
        for conn in self._peers.values():
            conn.close()
        self._peers = {}
        if self._sock is not None:
            self._sock.close()
            self._sock = None
        if self._server is not None:
            self._server.close()
            self._server = None
            try:
                os.remove(self._rdv)
            except (OSError, TypeError):
                pass


def launch_ranks(argv, world, master_port=None, env=None):
    """Start `world` fresh processes of `argv` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exported, as
    `python -m torch.distributed.run` would) and wait for them.  The caller must not have touched the GPU.  Returns the first
    non-zero exit code (the remaining ranks are then terminated), or 0."""
    import subprocess
    if master_port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            master_port = s.getsockname()[1]
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                  'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(master_port)})
        # RCCL across processes needs dmabuf IPC on this pool's driver (the image exports this already).  UNVERIFIED on hardware: RCCL has
        # never run with more than one rank (no multi-GPU node has been available); BatchShardComm sets the same default for ranks that
        # another launcher (torchrun) started, so both launchers behave alike.
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen(list(argv), env=e))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


class BatchShardComm:
    """What the Result plugin uses to turn per-rank Result tensors into the whole-batch tensor."""

    def __init__(self, group, use_rccl: bool = True):
        self.group = group
        self.rank = group.rank
        self.world = group.world
        # PVHIP_NO_RCCL=1 forces the host (gloo) gather: for rehearsing the multi-rank path on one GPU
        self.use_rccl = bool(use_rccl) and self.world > 1 and os.environ.get('PVHIP_NO_RCCL') != '1'
        self._rccl_ready = False
        if self.use_rccl and not dev.is_initialised():
            # the same default launch_ranks() exports, for ranks torchrun started (read by the HSA runtime when the process first touches
            # the GPU, so only useful before device.init(); see launch_ranks: unverified with more than one rank)
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    def shard(self, total: int):
        """This rank's slice of a batch of `total` rows.  Every rank calls it with the SAME total (it is how the batch gets
        split in the first place), so the total is also what allgather_rows() derives every rank's row count from."""
        self.total = int(total)
        return shard_bounds(total, self.rank, self.world)

    def init_device(self):
        """Create the RCCL communicator (rank 0 makes the unique id, the host group distributes it)."""
        if not self.use_rccl or self._rccl_ready:
            return
        dev.ensure_init()
        uid = None
        if self.rank == 0:
            try:
                buf = ctypes.create_string_buffer(dev.UNIQUE_ID_BYTES)
                dev.call('pvhip_comm_unique_id', buf)
                uid = buf.raw
            except dev.PvhipError as exc:      # the other ranks are waiting in the broadcast: tell them
                uid = 'error: {}'.format(exc)
        uid = self.group.broadcast_bytes(uid, src=0)
        if isinstance(uid, str):
            raise dev.PvhipError('rank 0 could not create the RCCL unique id ({})'.format(uid))
        dev.call('pvhip_comm_init', ctypes.c_char_p(uid), self.rank, self.world)
        self._rccl_ready = True

    def agree_on_gather(self):
        """Create the RCCL communicator on every rank, or agree -- all ranks together, through the host group -- to gather
        through the host group instead when ANY rank cannot (no librccl, no GPU, the unique id could not be made).  Never
        silently: returns (gather path for the record, error text of this rank or '')."""
        if self.world == 1:
            return 'none (one rank)', ''
        if not self.use_rccl:
            return 'host group (PVHIP_NO_RCCL=1)', ''
        error = ''
        try:
            self.init_device()
        except Exception as exc:       # noqa: BLE001 -- whatever it is, the other ranks must hear about it
            error = '{}: {}'.format(type(exc).__name__, exc)
        if self.group.allreduce_max(1.0 if error else 0.0) > 0.0:
            if self._rccl_ready:       # this rank has a communicator the others do not: drop it
                try:
                    self.close()
                except Exception:      # noqa: BLE001
                    pass
            self.use_rccl = False
            return 'host group (RCCL communicator unavailable{})'.format(': ' + error if error else ' on another rank'), error
        return 'rccl', ''

    def rccl_ranks(self) -> int:
        """ncclCommCount of the communicator (0 when there is none): what RCCL itself believes the world is."""
        if not self._rccl_ready:
            return 0
        n = ctypes.c_int(0)
        dev.call('pvhip_comm_ranks', ctypes.byref(n))
        return n.value

    def row_counts(self, rows: int):
        """Rows of every rank's shard, in rank order.  Derived from the global total every rank passed to shard() --
        shard_bounds() is a pure function of (total, rank, world), so all ranks compute the same list without talking to
        each other.  A communicator that was never told the total exchanges the counts through the host group EVERY time:
        a cache keyed on this rank's own row count would let the ranks disagree on whether to enter that collective (world 2,
        total 7 then 8: rank 0 stays at 4 rows, rank 1 goes from 3 to 4)."""
        total = getattr(self, 'total', None)
        if total is not None:
            counts = []
            for r in range(self.world):
                lo, hi = shard_bounds(total, r, self.world)
                counts.append(hi - lo)
            if counts[self.rank] != int(rows):
                raise ValueError('rank {} holds {} rows, its shard of a batch of {} on {} ranks has {}'.format(
                    self.rank, rows, total, self.world, counts[self.rank]))
            return counts
        return [int(np.asarray(p).reshape(-1)[0]) for p in self.group.allgather_array(np.array([rows], dtype=np.int64))]

    def allgather_rows(self, value):
        """Concatenate every rank's tensor along axis 0, in rank order.  ncclAllGather moves the SAME count from every
        rank, while shard() hands the first total % world ranks one more row: uneven shards are gathered padded to the
        longest one and the padding rows are cut out of the result."""
        if self.world == 1:
            return value
        if isinstance(value, dev.DeviceTensor) and self.use_rccl:
            self.init_device()
            counts = self.row_counts(value.shape[0])
            most, inner = max(counts), tuple(value.shape[1:])
            row = int(np.prod(inner, dtype=np.int64)) if inner else 1
            send = value
            if value.shape[0] != most:                                 # pad this rank's shard (zeros, cut out below)
                send = dev.DeviceTensor.empty((most,) + inner)
                dev.call('pvhip_memset', ctypes.c_void_p(send.ptr), 0, send.nbytes)
                if value.nbytes:
                    dev.call('pvhip_memcpy_d2d', ctypes.c_void_p(send.ptr), ctypes.c_void_p(value.ptr), value.nbytes)
            out = dev.DeviceTensor.empty((most * self.world,) + inner)
            dev.call('pvhip_comm_allgather_f32', ctypes.c_void_p(send.ptr), ctypes.c_void_p(out.ptr), most * row)
            if min(counts) == most:
                return out
            dense = dev.DeviceTensor.empty((sum(counts),) + inner)
            at = 0
            for r, n_rows in enumerate(counts):                        # rank r's real rows, packed
                if n_rows:
                    dev.call('pvhip_memcpy_d2d', ctypes.c_void_p(dense.ptr + at * row * 4), ctypes.c_void_p(out.ptr + r * most * row * 4),
                             n_rows * row * 4)
                at += n_rows
            return dense
        parts = self.group.allgather_array(np.asarray(value))
        return np.concatenate(parts, axis=0)

    def close(self):
        if self._rccl_ready:
            dev.call('pvhip_comm_destroy')
            self._rccl_ready = False
