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

    def allgather_rows(self, value):
        """Concatenate every rank's tensor along axis 0, in rank order."""
        if self.world == 1:
            return value
        if isinstance(value, dev.DeviceTensor) and self.use_rccl:
            self.init_device()
            out = dev.DeviceTensor.empty((value.shape[0] * self.world,) + tuple(value.shape[1:]))
            dev.call('pvhip_comm_allgather_f32', ctypes.c_void_p(value.ptr), ctypes.c_void_p(out.ptr), value.size)
            return out
        parts = self.group.allgather_array(np.asarray(value))
        return np.concatenate(parts, axis=0)

    def close(self):
        if self._rccl_ready:
            dev.call('pvhip_comm_destroy')
            self._rccl_ready = False
