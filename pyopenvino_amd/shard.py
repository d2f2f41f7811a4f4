"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective except
an all-gather of the Result tensors (RCCL over xGMI, through the C ABI's pvhip_comm_*).

The reference has no counterpart (it is single-process, N=1); images of a batch are independent through
the whole graph, so each rank runs the unchanged scheduler on a contiguous slice of the batch.

Host-side coordination (rendezvous, barrier, exchanging RCCL's unique id, max-over-ranks timing) goes
through a small HostGroup protocol; TorchGroup adapts ``torch.distributed`` (backend ``gloo``) to it,
which is what the launcher (``python -m torch.distributed.run``) sets up the environment for.  The
data path never touches torch.
"""
import ctypes
import os

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


class TorchGroup:
    """HostGroup over torch.distributed (gloo).  Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    from the environment when the default process group is not initialised yet."""

    def __init__(self, backend: str = 'gloo'):
        import torch.distributed as dist
        self._dist = dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(backend=backend)
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

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

    def allreduce_max(self, value: float) -> float:
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


class BatchShardComm:
    """What the Result plugin uses to turn per-rank Result tensors into the whole-batch tensor."""

    def __init__(self, group, use_rccl: bool = True):
        self.group = group
        self.rank = group.rank
        self.world = group.world
        # PVHIP_NO_RCCL=1 forces the host (gloo) gather: for rehearsing the multi-rank path on one GPU
        self.use_rccl = bool(use_rccl) and self.world > 1 and os.environ.get('PVHIP_NO_RCCL') != '1'
        self._rccl_ready = False

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
