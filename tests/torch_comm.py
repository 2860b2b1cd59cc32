"""torch.distributed communicator with the interface of adorym_amd.comm's backends -- TEST INFRASTRUCTURE, not product code.
('gloo'): host buffers, for CPU tests of the sharding logic of adorym_amd/dp.py with a NumPy stand-in for the kernels
(tests/test_dp_gloo.py).  The product's transports (RcclComm, P2PComm, HostStagedComm) never import torch."""
import os

from adorym_amd.comm import shard_bounds


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
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
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

    def reduce_tensor(self, t, root):
        """t (a tensor or a view of one) of rank ``root`` = sum over ranks of their ``t``, in place."""
        self.dist.reduce(t, dst=int(root), op=self.dist.ReduceOp.SUM)
        return t

    def all_reduce_device(self, dev):
        """In-place sum over ranks of a libadm device array, through a torch tensor (host bounce: this backend has no
        view of libadm's memory; RcclComm reduces in place on the device)."""
        g = self.torch.from_numpy(dev.get()).to(self.device)
        self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
        dev.set(g.cpu().numpy())
        return dev

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
