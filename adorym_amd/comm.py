"""
Communication seam of the data-parallel (DP) mode.  The reference does
``gradient.arr = comm.allreduce(gradient.arr)`` with mpi4py pickles (adorym/ptychography.py:1113-1114)
and then applies the identical optimiser step on every rank.  Here, one process per GPU:

    reduce_scatter(sum) of the object gradient  ->  fused Adam on the owned shard (moments are
    sharded, ZeRO-1 style)  ->  all_gather of the updated object shards

over RCCL/xGMI through ``torch.distributed`` (backend "nccl" is RCCL on ROCm).  torch is plumbing
only: it owns the process group and the buffers that the collectives touch; every kernel is libadm's
and runs on the same HIP stream (the context is created on torch's current stream).

Backends: LocalComm (1 rank, no torch import), TorchComm('nccl') on GPUs, TorchComm('gloo') on host
buffers for CPU tests of the sharding logic.
"""
import os
import numpy as np


class LocalComm(object):
    """Single rank: the identity collectives of adorym/pseudo.py:27-60."""
    rank = 0
    size = 1
    backend = 'local'

    def barrier(self):
        pass

    def shard_range(self, n):
        return 0, n

    def max_over_ranks(self, value):
        return value

    def sum_over_ranks(self, value):
        return value

    def bcast_object(self, obj, root=0):
        return obj


def shard_bounds(n, size, rank, align=2):
    """Contiguous shard [lo, hi) of a flat array of n elements; boundaries are multiples of `align`
    (2 keeps a voxel's (delta, beta) pair on one rank)."""
    per = -(-n // size)
    per = -(-per // align) * align
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi


class TorchComm(object):
    """torch.distributed process group (env:// rendezvous: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""

    def __init__(self, backend='nccl', device_index=None, init=True):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.backend = backend
        if init and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            kw = {}
            if backend == 'nccl':
                if device_index is None:
                    device_index = int(os.environ.get('LOCAL_RANK', '0'))
                torch.cuda.set_device(device_index)
                try:
                    kw['device_id'] = torch.device('cuda', device_index)
                except Exception:
                    pass
            dist.init_process_group(backend=backend, **kw)
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.device_index = device_index
        self.device = torch.device('cuda', device_index) if backend == 'nccl' else torch.device('cpu')
        self.stream = None
        if backend == 'nccl':
            # a dedicated, explicit stream shared by libadm and torch: collectives are ordered against the
            # CURRENT torch stream, and the legacy null stream would not order against a non-blocking one
            self.stream = torch.cuda.Stream(device=self.device)
            torch.cuda.set_stream(self.stream)

    # ---- buffers the collectives touch -------------------------------------------------
    def alloc(self, n, dtype=None):
        """A flat fp32 torch tensor on the communication device (zero-filled)."""
        return self.torch.zeros(int(n), dtype=dtype or self.torch.float32, device=self.device)

    def stream_handle(self):
        """hipStream_t of torch's current stream, for adm_ctx_create(): libadm kernels and the
        collectives are then ordered on one stream."""
        return int(self.stream.cuda_stream) if self.backend == 'nccl' else None

    # ---- collectives ---------------------------------------------------------------------
    def barrier(self):
        self.dist.barrier()

    def shard_range(self, n):
        return shard_bounds(n, self.size, self.rank)

    def reduce_scatter_sum(self, full, shard_out):
        """shard_out[:] = sum over ranks of full[lo:hi] (lo, hi = this rank's shard).  Requires
        n == size * len(shard_out)."""
        if self.backend == 'nccl':
            self.dist.reduce_scatter_tensor(shard_out, full, op=self.dist.ReduceOp.SUM)
        else:   # gloo has no reduce_scatter_tensor: all_reduce then slice (CPU tests only)
            tmp = full.clone()
            self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM)
            lo = self.rank * shard_out.numel()
            shard_out.copy_(tmp[lo:lo + shard_out.numel()])

    def all_gather(self, full_out, shard_in):
        if self.backend == 'nccl':
            self.dist.all_gather_into_tensor(full_out, shard_in)
        else:
            parts = [self.torch.empty_like(shard_in) for _ in range(self.size)]
            self.dist.all_gather(parts, shard_in)
            full_out.copy_(self.torch.cat(parts))

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def max_over_ranks(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def bcast_object(self, obj, root=0):
        lst = [obj]
        self.dist.broadcast_object_list(lst, src=root)
        return lst[0]

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def from_env():
    """LocalComm unless launched under torch.distributed.run with WORLD_SIZE > 1."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        return TorchComm('nccl')
    return LocalComm()
